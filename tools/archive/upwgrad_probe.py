#!/usr/bin/env python3
"""Up-sampling-aware weight gradient: parity-class kernel (wgrad_up2) vs the 27-point Winograd-(x,y,z) form (gpurun tuning aid)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query, lib  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402


def run(B, Dc, Hc, Wc, C, iters=3):
    torch.manual_seed(0)
    s = _stream()
    xc = torch.rand((B, Dc, Hc, Wc, C), device="cuda") * 2 - 1
    g = torch.rand((B, 2 * Dc, 2 * Hc, 2 * Wc, C), device="cuda") * 2 - 1
    nb = query("df_upconv_wgrad_workspace_bytes", B, Dc, Hc, Wc, C, C, 3)
    ws = torch.empty((nb + 3) // 4, device="cuda")
    out = []
    for algo in (3, 4):
        gw = torch.empty((27, C, C), device="cuda"); gb = torch.empty(C, device="cuda")
        f = lambda: call("df_upconv_wgrad_algo", _ptr(xc), _ptr(g), _ptr(gw), _ptr(gb), B, Dc, Hc, Wc, C, C, 3, _ptr(ws), nb, algo, s)
        f(); torch.cuda.synchronize()
        t = timeit(f, iters, 1)
        out.append((gw.clone(), gb.clone(), t))
    (w0, b0, t0), (w1, b1, t1) = out
    print("B%d coarse %dx%dx%d C%d: parity-class %.3f ms | winograd-xyz 27-point %.3f ms  gw %.1e gb %.1e" % (
        B, Dc, Hc, Wc, C, t0 * 1e3, t1 * 1e3, ((w0 - w1).abs().max() / w0.abs().max()).item(), ((b0 - b1).abs().max() / b0.abs().max()).item()), flush=True)


RANGES = int(sys.argv[1]) if len(sys.argv) > 1 else 0

if __name__ == "__main__":
    run(16, 16, 24, 16, 128)
    run(16, 32, 48, 32, 128)
