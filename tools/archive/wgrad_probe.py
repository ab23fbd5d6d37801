#!/usr/bin/env python3
"""Direct vs Winograd-in-x weight gradient: difference and timing (gpurun tuning aid)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query, lib  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402


def run(B, D, H, W, C, N, kz=3, iters=3):
    torch.manual_seed(0)
    s = _stream()
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    g = torch.rand((B, D, H, W, N), device="cuda") * 2 - 1
    taps = 27 if kz == 3 else 9
    nb = query("df_conv_wgrad_workspace_bytes", B, D, H, W, C, N, kz)
    if RANGES:
        nb = max(nb, RANGES * 64 * 128 * 128 * 4 * 2 + (1 << 20))      # room for a caller-chosen number of partial ranges
    ws = torch.empty((nb + 3) // 4, device="cuda")
    out = []
    for algo in (1, 2, 3, 4, 0):
        gw = torch.empty((taps, C, N), device="cuda"); gb = torch.empty(N, device="cuda")
        f = lambda: call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, N, kz, _ptr(ws), nb,
                         algo | (RANGES << 3), s)
        f(); torch.cuda.synchronize()
        t = timeit(f, iters, 1)
        out.append((gw.clone(), gb.clone(), t))
    w0, b0, t0 = out[0]
    fl = 2.0 * taps * C * N * B * D * H * W
    msg = "B%d %dx%dx%d C%d N%d: direct %.3f ms (%.0f TF)" % (B, D, H, W, C, N, t0 * 1e3, fl / t0 / 1e12)
    for name, (w1, b1, t1) in zip(("x", "xy", "xyz", "default"), out[1:]):
        msg += " | %s %.3f ms (%.0f TF-eq) gw %.1e gb %.1e" % (name, t1 * 1e3, fl / t1 / 1e12, ((w0 - w1).abs().max() / w0.abs().max()).item(),
                                                             ((b0 - b1).abs().max() / b0.abs().max()).item())
    print(msg, flush=True)


RANGES = 0
if __name__ == "__main__":
    if len(sys.argv) > 1:
        RANGES = int(sys.argv[1])
    run(16, 8, 12, 8, 128, 128)
    run(16, 16, 24, 16, 128, 128)
    run(16, 32, 48, 32, 128, 128)
    if not os.environ.get("WGRAD_SMALL"):
        run(4, 64, 96, 64, 128, 128)
        run(16, 64, 96, 64, 128, 128, iters=2)
