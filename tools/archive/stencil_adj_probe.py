#!/usr/bin/env python3
"""Tuning probe for the 3-D stencil adjoints (GPU box): LDS-staged vs register-only kernels, rotating (cold) buffers,
bit-comparison of the two."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import _lib as _libmod  # noqa: E402
_libmod.use_tuning_library()      # needs `make -C deep_fluids_amd/csrc tuning`
from deep_fluids_amd._lib import call, lib  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402


def timeit(fn, iters=60, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


s = _stream()
h = lib()
for (B, Z, Y, X) in ((16, 64, 96, 64), (4, 112, 160, 112), (3, 10, 12, 20)):
    nvox = B * Z * Y * X
    nrot = 3 if nvox > 1000000 else 1
    js = [torch.randn((B, Z, Y, X, 9), device="cuda") for _ in range(nrot)]
    cs = [torch.randn((B, Z, Y, X, 3), device="cuda") for _ in range(nrot + 5 if nrot > 1 else 1)]
    gxs = [torch.empty((B, Z, Y, X, 3), device="cuda") for _ in range(3)]
    res = {}
    for lds in (0, 1):
        h.df_debug_set_stencil_lds(lds)
        k = [0]
        def fj():
            k[0] += 1
            call("df_jacobian3d_bwd", _ptr(js[k[0] % len(js)]), None, _ptr(gxs[k[0] % 3]), B, Z, Y, X, s)
        def fc():
            k[0] += 1
            call("df_jacobian3d_bwd", None, _ptr(cs[k[0] % len(cs)]), _ptr(gxs[k[0] % 3]), B, Z, Y, X, s)
        t = timeit(fj); print("%s lds=%d bwd<j> cold: %6.1f us %7.1f GB/s" % ((B, Z, Y, X), lds, t * 1e6, nvox * 48 / t / 1e9))
        t = timeit(fc); print("%s lds=%d bwd<c> cold: %6.1f us %7.1f GB/s" % ((B, Z, Y, X), lds, t * 1e6, nvox * 24 / t / 1e9))
        a = torch.empty_like(gxs[0]); b = torch.empty_like(gxs[0])
        call("df_jacobian3d_bwd", _ptr(js[0]), None, _ptr(a), B, Z, Y, X, s)
        call("df_jacobian3d_bwd", None, _ptr(cs[0]), _ptr(b), B, Z, Y, X, s)
        res[lds] = (a, b)
    print("  lds vs register-only: max |diff| j %.3g  c %.3g" % (float((res[0][0] - res[1][0]).abs().max()),
                                                                 float((res[0][1] - res[1][1]).abs().max())))
    del js, cs, gxs
