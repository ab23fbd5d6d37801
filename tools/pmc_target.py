#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: the three roofline kernels at BASELINE cfg3 shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream, _pack  # noqa: E402
B, Z, Y, X, F = 16, 64, 96, 64, 128
s = _stream()
x = torch.rand((B, Z, Y, X, 3), device="cuda"); j = torch.empty((B, Z, Y, X, 9), device="cuda"); c = torch.empty((B, Z, Y, X, 3), device="cuda")
for _ in range(3):
    call("df_jacobian3d_fwd", _ptr(x), _ptr(j), _ptr(c), B, Z, Y, X, s)
xin = torch.rand((B, Z, Y, X, F), device="cuda") - 0.5
wt = (torch.rand((3, 3, 3, F, F), device="cuda") - 0.5) * 0.05
bias = torch.zeros(F, device="cuda"); y = torch.empty_like(xin)
wp = _pack(wt, 27, F, F, 0)
for _ in range(2):
    call("df_conv_fwd", _ptr(xin), _ptr(wp), _ptr(bias), None, None, _ptr(y), B, Z, Y, X, F, F, 3, 9, 0.2, s)
ww = torch.empty(query("df_wino_packed_elems", F, F, 0), device="cuda")
call("df_wino_pack_weights", _ptr(wt), _ptr(ww), F, F, 0, s)
for _ in range(2):
    call("df_wino_conv_fwd", _ptr(xin), _ptr(ww), _ptr(bias), None, None, _ptr(y), B, Z, Y, X, F, F, 9, 0.2, s)
gw = torch.empty_like(wt); gb = torch.empty(F, device="cuda")
nb = query("df_conv_wgrad_workspace_bytes", B, Z, Y, X, F, F, 3)
ws = torch.empty(nb // 4 + 1, device="cuda")
for _ in range(2):
    call("df_conv_wgrad", _ptr(xin), _ptr(y), _ptr(gw), _ptr(gb), B, Z, Y, X, F, F, 3, _ptr(ws), nb, s)
torch.cuda.synchronize()
