#!/usr/bin/env python3
"""GB/s of the element-wise kernels at the cfg3 top-level size (GPU box)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402


def timeit(fn, iters=20, warm=3):
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
B, D, H, W, C = 16, 64, 96, 64, 128
n = B * D * H * W * C
a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda"); y = torch.empty(n, device="cuda")
t = timeit(lambda: call("df_lrelu_bwd", _ptr(a), _ptr(b), _ptr(y), 0.2, n, s)); print("lrelu_bwd  %7.1f us %7.1f GB/s" % (t * 1e6, n * 12 / t / 1e9))
t = timeit(lambda: call("df_add", _ptr(a), _ptr(b), _ptr(y), n, s)); print("add        %7.1f us %7.1f GB/s" % (t * 1e6, n * 12 / t / 1e9))
t = timeit(lambda: call("df_lrelu_fwd", _ptr(a), _ptr(y), 0.2, n, s)); print("lrelu_fwd  %7.1f us %7.1f GB/s" % (t * 1e6, n * 8 / t / 1e9))
t = timeit(lambda: y.copy_(a)); print("torch copy %7.1f us %7.1f GB/s" % (t * 1e6, n * 8 / t / 1e9))
t = timeit(lambda: torch.add(a, b, out=y)); print("torch add  %7.1f us %7.1f GB/s" % (t * 1e6, n * 12 / t / 1e9))
gx = torch.empty(n // 8, device="cuda")
t = timeit(lambda: call("df_upsample2x_bwd", _ptr(a), _ptr(gx), B, D // 2, H // 2, W // 2, C, 1, s)); print("upsample_bwd %7.1f us %7.1f GB/s" % (t * 1e6, n * 4.5 / t / 1e9))
# the backward tail of an up-sampling block: separate passes / one pass on the fp32 activation / one pass on its sign bits
from deep_fluids_amd._lib import query  # noqa: E402
gp = torch.empty(n // 8, device="cuda")
t = timeit(lambda: call("df_lrelu_bwd_pool2x", _ptr(a), _ptr(b), _ptr(y), _ptr(gp), 0.2, B, D // 2, H // 2, W // 2, C, 1, s))
print("lrelu_bwd_pool2x       %7.1f us  (%.1f GB moved)" % (t * 1e6, n * 12.5 / 1e9))
bits = torch.zeros(query("df_wino_signbits_bytes", B, D, H, W, C) // 8, dtype=torch.int64, device="cuda")
t = timeit(lambda: call("df_lrelu_bits_bwd_pool2x", _ptr(a), _ptr(bits), _ptr(y), _ptr(gp), 0.2, B, D // 2, H // 2, W // 2, C, s))
print("lrelu_bits_bwd_pool2x  %7.1f us  (%.1f GB moved)" % (t * 1e6, n * 8.6 / 1e9))
