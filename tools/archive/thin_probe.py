#!/usr/bin/env python3
"""Timing of the thin last-layer kernels at cfg3 (gpurun tuning aid)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream, _pack  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402
B, Z, Y, X, F = 16, 64, 96, 64, 128
s = _stream()
x = torch.rand((B, Z, Y, X, F), device="cuda") - 0.5
w = (torch.rand((3, 3, 3, F, 3), device="cuda") - 0.5) * 0.05
b = torch.zeros(3, device="cuda")
y = torch.empty((B, Z, Y, X, 3), device="cuda")
wp = _pack(w, 27, F, 3, 0)
t = timeit(lambda: call("df_conv_fwd", _ptr(x), _ptr(wp), _ptr(b), None, None, _ptr(y), B, Z, Y, X, F, 3, 3, 8, 0.0, s), 5, 2)
print("conv 128->3 fwd   : %.3f ms" % (t * 1e3))
g = torch.rand((B, Z, Y, X, 3), device="cuda") - 0.5
wpd = _pack(w, 27, F, 3, 1)
dx = torch.empty_like(x)
t = timeit(lambda: call("df_conv_fwd", _ptr(g), _ptr(wpd), None, None, _ptr(x), _ptr(dx), B, Z, Y, X, 3, F, 3, 4, 0.2, s), 5, 2)
print("conv 3->128 dgrad : %.3f ms" % (t * 1e3))
gw = torch.empty_like(w); gb = torch.empty(3, device="cuda")
nb = query("df_conv_wgrad_workspace_bytes", B, Z, Y, X, F, 3, 3)
ws = torch.empty(nb // 4 + 1, device="cuda")
t = timeit(lambda: call("df_conv_wgrad", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, Z, Y, X, F, 3, 3, _ptr(ws), nb, s), 5, 2)
print("wgrad 128x3       : %.3f ms" % (t * 1e3))
import ctypes
from deep_fluids_amd._lib import lib
gw1 = torch.empty_like(w); gb1 = torch.empty(3, device="cuda")
t = timeit(lambda: call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw1), _ptr(gb1), B, Z, Y, X, F, 3, 3, _ptr(ws), nb, 1, s), 5, 2)
print("wgrad 128x3 (VALU): %.3f ms   mfma-vs-valu rel diff gw %.2e gb %.2e" % (
    t * 1e3, ((gw - gw1).abs().max() / gw1.abs().max()).item(), ((gb - gb1).abs().max() / gb1.abs().max()).item()))
dx1 = torch.empty_like(x)
t = timeit(lambda: call("df_conv_fwd", _ptr(g), _ptr(wpd), None, None, _ptr(x), _ptr(dx1), B, Z, Y, X, 3, F, 3, 4 | 32, 0.2, s), 5, 2)
print("conv 3->128 dgrad (VALU): %.3f ms   mfma-vs-valu rel diff %.2e" % (t * 1e3, ((dx - dx1).abs().max() / dx1.abs().max()).item()))
y1 = torch.empty_like(y)
t = timeit(lambda: call("df_conv_fwd", _ptr(x), _ptr(wp), _ptr(b), None, None, _ptr(y1), B, Z, Y, X, F, 3, 3, 8 | 32, 0.0, s), 5, 2)
call("df_conv_fwd", _ptr(x), _ptr(wp), _ptr(b), None, None, _ptr(y), B, Z, Y, X, F, 3, 3, 8, 0.0, s)
print("conv 128->3 fwd (VALU): %.3f ms   mfma-vs-valu rel diff %.2e" % (t * 1e3, ((y - y1).abs().max() / y1.abs().max()).item()))
