#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: the roofline kernels at BASELINE cfg3 shapes (B = 16, 64x96x64, F = 128)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream, _pack  # noqa: E402
B, Z, Y, X, F = 16, 64, 96, 64, 128
s = _stream()
x = torch.rand((B, Z, Y, X, 3), device="cuda"); j = torch.rand((B, Z, Y, X, 9), device="cuda"); c = torch.rand((B, Z, Y, X, 3), device="cuda")
x2 = torch.rand((B, Z, Y, X, 3), device="cuda"); g = torch.empty((B, Z, Y, X, 3), device="cuda")
for _ in range(3):
    call("df_jacobian3d_fwd", _ptr(x), _ptr(j), _ptr(c), B, Z, Y, X, s)
    call("df_jacobian3d_fwd", _ptr(x), None, _ptr(c), B, Z, Y, X, s)
    call("df_jacobian3d_bwd", _ptr(j), None, _ptr(g), B, Z, Y, X, s)
    call("df_jacobian3d_bwd", None, _ptr(c), _ptr(g), B, Z, Y, X, s)
nbv = query("df_velocity_loss3d_workspace_bytes", B, Z, Y, X)
wsv = torch.empty((nbv + 7) // 8, dtype=torch.float64, device="cuda")
l1 = torch.ones((), device="cuda"); jl1 = torch.ones((), device="cuda")
for _ in range(3):
    call("df_velocity_loss3d_fwd", _ptr(x), _ptr(x2), _ptr(c), _ptr(l1), _ptr(jl1), B, Z, Y, X, _ptr(wsv), nbv, s)
    l1.fill_(1.0); jl1.fill_(1.0)
    call("df_velocity_loss3d_bwd", _ptr(c), _ptr(x2), _ptr(l1), _ptr(jl1), _ptr(g), B, Z, Y, X, _ptr(wsv), nbv, s)
xin = torch.rand((B, Z, Y, X, F), device="cuda") - 0.5
wt = (torch.rand((3, 3, 3, F, F), device="cuda") - 0.5) * 0.05
bias = torch.zeros(F, device="cuda"); y = torch.empty_like(xin)
ww = torch.empty(query("df_wino_packed_elems", F, F, 0), device="cuda")
call("df_wino_pack_weights", _ptr(wt), _ptr(ww), F, F, 0, s)
for _ in range(2):
    call("df_wino_conv_fwd", _ptr(xin), _ptr(ww), _ptr(bias), None, None, _ptr(y), B, Z, Y, X, F, F, 9, 0.2, s)
for _ in range(2):
    call("df_wino_conv_fwd", _ptr(xin), _ptr(ww), None, None, _ptr(xin), _ptr(y), B, Z, Y, X, F, F, 4, 0.2, s)
# [r6] the F(2,3) x F(2,3) x F(4,3) family (conv_wino43.hip): plain forward and the dgrad with an fp32 lrelu mask, same shapes
w43 = torch.empty(query("df_wino43_packed_elems", F, F, 0), device="cuda")
call("df_wino43_pack_weights", _ptr(wt), _ptr(w43), F, F, 0, s)
for _ in range(2):
    call("df_wino43_conv", _ptr(xin), _ptr(w43), _ptr(bias), None, None, None, _ptr(y), None, None, B, Z, Y, X, F, F, 9, 0.2, s)
for _ in range(2):
    call("df_wino43_conv", _ptr(xin), _ptr(w43), None, None, _ptr(xin), None, _ptr(y), None, None, B, Z, Y, X, F, F, 4, 0.2, s)
# [r5] the 27-point forms at the same level: up-sampling-aware forward (coarse 32x48x32 -> 64x96x64, wino3d_kernel MODE 3) and its pooled adjoint (MODE 2)
xc = torch.rand((B, Z // 2, Y // 2, X // 2, F), device="cuda") - 0.5
wwd = torch.empty(query("df_wino_packed_elems", F, F, 1), device="cuda")
call("df_wino_pack_weights", _ptr(wt), _ptr(wwd), F, F, 1, s)
acc = torch.zeros_like(xc)
for _ in range(2):
    call("df_wino_upconv_fwd", _ptr(xc), _ptr(ww), _ptr(bias), _ptr(y), B, Z // 2, Y // 2, X // 2, F, F, 9, 0.2, s)
for _ in range(2):
    call("df_wino_upconv_dgrad", _ptr(xin), _ptr(wwd), _ptr(acc), B, Z // 2, Y // 2, X // 2, F, F, s)
gw = torch.empty_like(wt); gb = torch.empty(F, device="cuda")
nb = query("df_conv_wgrad_workspace_bytes", B, Z, Y, X, F, F, 3)
ws = torch.empty(nb // 4 + 1, device="cuda")
for _ in range(2):
    call("df_conv_wgrad", _ptr(xin), _ptr(y), _ptr(gw), _ptr(gb), B, Z, Y, X, F, F, 3, _ptr(ws), nb, s)
torch.cuda.synchronize()
