#!/usr/bin/env python3
"""Winograd-(x,y,z) weight gradient at the top level (B = 16, 64x96x64, 128 -> 128) vs the number of partial voxel ranges (gpurun tuning aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, D, H, W, C = 16, 64, 96, 64, 128
s = _stream()
x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
g = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
nb = max(query("df_conv_wgrad_workspace_bytes", B, D, H, W, C, C, 3), max(R, 1) * 64 * 128 * 128 * 4 * 2 + (1 << 20))
ws = torch.empty((nb + 3) // 4, device="cuda")
gw = torch.empty((27, C, C), device="cuda"); gb = torch.empty(C, device="cuda")
f = lambda: call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, 4 | (R << 3), s)
print("xyz top level, %d ranges: %.3f ms" % (R, timeit(f, 5, 2) * 1e3), flush=True)
