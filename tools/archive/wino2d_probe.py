#!/usr/bin/env python3
"""2-D Winograd conv kernel vs the direct MFMA kernel: difference and timing at cfg2 shapes (gpurun tuning aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402


def run(B, H, W, C, N, iters=5):
    torch.manual_seed(0)
    s = _stream()
    w = (torch.rand((3, 3, C, N), device="cuda") * 2 - 1) * (2.0 / (9 * C)) ** 0.5
    for mode in (0, 1):
        K, NN = (C, N) if mode == 0 else (N, C)
        xin = torch.rand((B, H, W, K), device="cuda") * 2 - 1
        wd = torch.empty(query("df_conv_packed_elems", 9, C, N, mode), device="cuda")
        call("df_conv_pack_weights", _ptr(w), _ptr(wd), 9, C, N, mode, s)
        ww = torch.empty(query("df_wino2d_packed_elems", C, N, mode), device="cuda")
        call("df_wino2d_pack_weights", _ptr(w), _ptr(ww), C, N, mode, s)
        b_ = torch.rand(NN, device="cuda") * 0.1
        r_ = torch.rand((B, H, W, NN), device="cuda")
        fl = (8 | 1) if mode == 0 else 4
        y0 = torch.empty((B, H, W, NN), device="cuda"); y1 = torch.full_like(y0, float("nan"))
        call("df_conv_fwd", _ptr(xin), _ptr(wd), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y0), B, 1, H, W, K, NN, 1, fl, 0.2, s)
        call("df_wino2d_conv_fwd", _ptr(xin), _ptr(ww), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y1), B, H, W, K, NN, fl, 0.2, s)
        torch.cuda.synchronize()
        err = (y0 - y1).abs().max().item() / y0.abs().max().item()
        flops = 2.0 * B * H * W * 9 * C * N
        t0 = timeit(lambda: call("df_conv_fwd", _ptr(xin), _ptr(wd), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y0), B, 1, H, W, K, NN, 1, fl, 0.2, s), iters, 2)
        t1 = timeit(lambda: call("df_wino2d_conv_fwd", _ptr(xin), _ptr(ww), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y1), B, H, W, K, NN, fl, 0.2, s), iters, 2)
        print("B%d %dx%d C%d N%d mode%d: rel-linf %.2e | direct %.3f ms (%.0f TF)  wino2d %.3f ms (%.0f TF-eq, %.0f TF mfma)" % (
            B, H, W, C, N, mode, err, t0 * 1e3, flops / t0 / 1e12, t1 * 1e3, flops / t1 / 1e12, flops * 4 / 9 / t1 / 1e12), flush=True)


if __name__ == "__main__":
    run(1, 10, 12, 32, 32)
    run(2, 16, 32, 64, 32)
    run(3, 33, 47, 32, 64)
    run(64, 64, 48, 128, 128)
    run(64, 128, 96, 128, 128)
