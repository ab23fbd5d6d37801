#!/usr/bin/env python3
"""Fused-tail kernel variants (tuning library): timing at cfg3 and agreement of the results."""
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
from tools.gpu_probe import timeit  # noqa: E402

B, Z, Y, X = 16, 64, 96, 64
s = _stream()
torch.manual_seed(0)
psis = [torch.rand((B, Z, Y, X, 3), device="cuda") * 2 - 1 for _ in range(4)]
xs = [torch.rand((B, Z, Y, X, 3), device="cuda") * 2 - 1 for _ in range(4)]
us = [torch.empty((B, Z, Y, X, 3), device="cuda") for _ in range(3)]
gp = torch.empty((B, Z, Y, X, 3), device="cuda")
nb = query("df_velocity_loss3d_workspace_bytes", B, Z, Y, X)
ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device="cuda")
l1 = torch.empty((), device="cuda"); jl1 = torch.empty((), device="cuda")
one = torch.ones((), device="cuda")
ref = None
for var in (1, 0, 8):
    lib().df_debug_set_tail(ctypes.c_int(var))
    k = [0]
    def fwd():
        k[0] += 1
        call("df_velocity_loss3d_fwd", _ptr(psis[k[0] % 4]), _ptr(xs[k[0] % 4]), _ptr(us[k[0] % 3]), _ptr(l1), _ptr(jl1), B, Z, Y, X, _ptr(ws), nb, s)
    def bwd():
        k[0] += 1
        call("df_velocity_loss3d_bwd", _ptr(psis[k[0] % 4]), _ptr(xs[k[0] % 4]), _ptr(one), _ptr(one), _ptr(gp), B, Z, Y, X, _ptr(ws), nb, s)
    tf = timeit(fwd, 20, 3); tb = timeit(bwd, 20, 3)
    call("df_velocity_loss3d_fwd", _ptr(psis[0]), _ptr(xs[0]), _ptr(us[0]), _ptr(l1), _ptr(jl1), B, Z, Y, X, _ptr(ws), nb, s)
    call("df_velocity_loss3d_bwd", _ptr(us[0]), _ptr(xs[0]), _ptr(one), _ptr(one), _ptr(gp), B, Z, Y, X, _ptr(ws), nb, s)
    res = (float(l1), float(jl1), gp.clone(), us[0].clone())
    if ref is None:
        ref = res
    print("variant %d (%s): fwd %.1f us  bwd %.1f us   l1 %.8f jl1 %.8f  dpsi max diff vs variant 1: %.2e  u identical: %s" % (
        var, {0: "default: persistent LDS-tiled forward", 8: "z-marching columns, LDS-DMA rings", 1: "curl3 + 16-byte quad reduction", 2: "record-per-lane kernels", 3: "tiled, one workgroup per CU", 4: "tiled, 2 x 4-row tiles", 5: "tiled, non-temporal u stores"}[var], tf * 1e6, tb * 1e6,
        res[0], res[1], (res[2] - ref[2]).abs().max().item(), torch.equal(res[3], ref[3])), flush=True)
lib().df_debug_set_tail(ctypes.c_int(0))
