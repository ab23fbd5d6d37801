#!/usr/bin/env python3
"""Tuning probe for the 3-D stencil kernels (GPU box): variants x cold/warm input."""
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


B, Z, Y, X = 16, 64, 96, 64
nvox = B * Z * Y * X
xs = [torch.rand((B, Z, Y, X, 3), device="cuda") for _ in range(8)]      # 8 x 75 MB: rotate to defeat the 256 MB MALL
js = [torch.empty((B, Z, Y, X, 9), device="cuda") for _ in range(3)]
cs = [torch.empty((B, Z, Y, X, 3), device="cuda") for _ in range(3)]
s = _stream()
h = lib()
for nt in (0, 6, 12, 24, 48, 96, 384):
    h.df_debug_set_stencil_group(nt)
    k = [0]
    def warm_fn():
        call("df_jacobian3d_fwd", _ptr(xs[0]), _ptr(js[0]), _ptr(cs[0]), B, Z, Y, X, s)
    def cold_fn():
        k[0] += 1
        call("df_jacobian3d_fwd", _ptr(xs[k[0] % 8]), _ptr(js[k[0] % 3]), _ptr(cs[k[0] % 3]), B, Z, Y, X, s)
    t = timeit(warm_fn); print("nt=%d j+c same buffers : %6.1f us %7.1f GB/s" % (nt, t * 1e6, nvox * 60 / t / 1e9))
    t = timeit(cold_fn); print("nt=%d j+c rotating     : %6.1f us %7.1f GB/s" % (nt, t * 1e6, nvox * 60 / t / 1e9))
    t = timeit(lambda: call("df_jacobian3d_fwd", _ptr(xs[0]), None, _ptr(cs[0]), B, Z, Y, X, s))
    print("nt=%d curl3 same buffers: %6.1f us %7.1f GB/s" % (nt, t * 1e6, nvox * 24 / t / 1e9))
    t = timeit(lambda: call("df_jacobian3d_fwd", _ptr(xs[0]), _ptr(js[0]), None, B, Z, Y, X, s))
    print("nt=%d j only            : %6.1f us %7.1f GB/s" % (nt, t * 1e6, nvox * 48 / t / 1e9))
big = torch.empty((B, Z, Y, X, 9), device="cuda")
t = timeit(lambda: big.copy_(js[0])); print("torch copy 226->226 MB : %6.1f us %7.1f GB/s" % (t * 1e6, nvox * 72 / t / 1e9))
t = timeit(lambda: js[1].fill_(1.0)); print("torch fill 226 MB      : %6.1f us %7.1f GB/s" % (t * 1e6, nvox * 36 / t / 1e9))
gx = torch.empty_like(xs[0])
t = timeit(lambda: call("df_jacobian3d_bwd", _ptr(js[0]), None, _ptr(gx), B, Z, Y, X, s)); print("bwd gj : %6.1f us %7.1f GB/s" % (t * 1e6, nvox * 48 / t / 1e9))
t = timeit(lambda: call("df_jacobian3d_bwd", None, _ptr(cs[0]), _ptr(gx), B, Z, Y, X, s)); print("bwd gc : %6.1f us %7.1f GB/s" % (t * 1e6, nvox * 24 / t / 1e9))
