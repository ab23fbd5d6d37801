#!/usr/bin/env python3
"""Up-sampling-aware Winograd forward (df_wino_upconv_fwd) vs the parity-class direct kernel (df_upconv_fwd)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402


def run(B, Dc, Hc, Wc, C, N, iters=5):
    torch.manual_seed(0)
    s = _stream()
    xc = torch.rand((B, Dc, Hc, Wc, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, N), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    bias = torch.rand(N, device="cuda") * 0.1
    wd = torch.empty(query("df_upconv_packed_elems", C, N, 3, 0), device="cuda")
    call("df_upconv_pack_weights", _ptr(w), _ptr(wd), C, N, 3, 0, s)
    ww = torch.empty(query("df_wino_packed_elems", C, N, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(ww), C, N, 0, s)
    y0 = torch.empty((B, 2 * Dc, 2 * Hc, 2 * Wc, N), device="cuda"); y1 = torch.full_like(y0, float("nan"))
    f0 = lambda: call("df_upconv_fwd", _ptr(xc), _ptr(wd), _ptr(bias), _ptr(y0), B, Dc, Hc, Wc, C, N, 3, 9, 0.2, s)
    f1 = lambda: call("df_wino_upconv_fwd", _ptr(xc), _ptr(ww), _ptr(bias), _ptr(y1), B, Dc, Hc, Wc, C, N, 9, 0.2, s)
    f0(); f1()
    torch.cuda.synchronize()
    err = (y0 - y1).abs().max().item() / y0.abs().max().item()
    l1 = (y0 - y1).abs().sum().item() / y0.abs().sum().item()
    t0 = timeit(f0, iters, 2)
    t1 = timeit(f1, iters, 2)
    g = torch.rand((B, 2 * Dc, 2 * Hc, 2 * Wc, N), device="cuda") * 2 - 1
    wdd = torch.empty(query("df_upconv_packed_elems", C, N, 3, 1), device="cuda")
    call("df_upconv_pack_weights", _ptr(w), _ptr(wdd), C, N, 3, 1, s)
    wwd = torch.empty(query("df_wino_packed_elems", C, N, 1), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wwd), C, N, 1, s)
    a0 = torch.zeros_like(xc); a1 = torch.zeros_like(xc)
    d0 = lambda: call("df_upconv_dgrad", _ptr(g), _ptr(wdd), _ptr(a0), B, Dc, Hc, Wc, C, N, 3, s)
    d1 = lambda: call("df_wino_upconv_dgrad", _ptr(g), _ptr(wwd), _ptr(a1), B, Dc, Hc, Wc, C, N, s)
    d0(); d1()
    torch.cuda.synchronize()
    derr = (a0 - a1).abs().max().item() / a0.abs().max().item()
    td0 = timeit(d0, iters, 2)
    td1 = timeit(d1, iters, 2)
    print("   dgrad: rel-linf %.2e | direct %.3f ms  wino-pool %.3f ms" % (derr, td0 * 1e3, td1 * 1e3), flush=True)
    print("B%d coarse %dx%dx%d C%d N%d: rel-linf %.2e rel-l1 %.2e | direct %.3f ms  wino-up %.3f ms" % (
        B, Dc, Hc, Wc, C, N, err, l1, t0 * 1e3, t1 * 1e3), flush=True)


if __name__ == "__main__":
    run(1, 3, 5, 6, 32, 32)
    run(2, 8, 12, 8, 128, 128)
    run(4, 32, 48, 32, 128, 128, iters=3)
    run(16, 32, 48, 32, 128, 128, iters=3)
    run(16, 16, 24, 16, 128, 128, iters=3)
