#!/usr/bin/env python3
"""Workload for a rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES pass (tuning library): effective shader clock (GRBM_GUI_ACTIVE / duration)
and wave cycles of wino3d_kernel (production), its x-blocked DMA-staging variant, the no-staging timing variant, the direct MFMA conv
and the (x,y,z) weight gradient at the top-level shape (B = 16, 64x96x64, F = 128), on random data."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import _lib as _libmod  # noqa: E402
_libmod.use_tuning_library()
from deep_fluids_amd._lib import call, query, lib  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from wino_xblk_probe import xblk  # noqa: E402

B, D, H, W, C = int(os.environ.get("B", "16")), 64, 96, 64, 128
s = _stream()
torch.manual_seed(0)
x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
ww = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda")
call("df_wino_pack_weights", _ptr(w), _ptr(ww), C, C, 0, s)
wd = torch.empty(query("df_conv_packed_elems", 27, C, C, 0), device="cuda")
call("df_conv_pack_weights", _ptr(w), _ptr(wd), 27, C, C, 0, s)
bias = torch.rand(C, device="cuda") * 0.1
y = torch.empty_like(x)
h = lib()
h.df_debug_wino_xblk_elems.restype = ctypes.c_int64
h.df_debug_wino_xblk_elems.argtypes = [ctypes.c_int64] * 5
xg = torch.empty(h.df_debug_wino_xblk_elems(B, D, H, W, C), device="cuda")
N = 3
for _ in range(N):
    call("df_wino_conv_fwd", _ptr(x), _ptr(ww), _ptr(bias), None, None, _ptr(y), B, D, H, W, C, C, 9, 0.2, s)
xblk(x, xg, ww, bias, y, (B, D, H, W), C, C, 9, s)
for v in (9, 1, 67):
    for _ in range(N):
        xblk(x, xg, ww, bias, y, (B, D, H, W), C, C, v | 256, s)
for _ in range(N):
    call("df_conv_fwd", _ptr(x), _ptr(wd), _ptr(bias), None, None, _ptr(y), B, D, H, W, C, C, 3, 9, 0.2, s)
gw = torch.empty_like(w); gb = torch.empty(C, device="cuda")
nb = query("df_conv_wgrad_workspace_bytes", B, D, H, W, C, C, 3)
ws = torch.empty(nb // 4 + 1, device="cuda")
for _ in range(N):
    call("df_conv_wgrad", _ptr(x), _ptr(y), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, s)
torch.cuda.synchronize()
