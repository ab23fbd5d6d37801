#!/usr/bin/env python3
"""Masked dgrad at cfg3: mask from the fp32 activation (DF_CONV_MASK + mask_src) vs from sign-bit words; forward with / without bit output."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402
B, D, H, W, C = 16, 64, 96, 64, 128
s = _stream()
torch.manual_seed(0)
x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
g = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
bias = torch.rand(C, device="cuda") * 0.1 - 0.05
ww = [torch.empty(query("df_wino_packed_elems", C, C, m), device="cuda") for m in (0, 1)]
for m in (0, 1):
    call("df_wino_pack_weights", _ptr(w), _ptr(ww[m]), C, C, m, s)
y0 = torch.empty_like(x); y1 = torch.empty_like(x)
bits = torch.empty(query("df_wino_signbits_bytes", B, D, H, W, C) // 8, dtype=torch.int64, device="cuda")
f0 = lambda: call("df_wino_conv_fwd", _ptr(x), _ptr(ww[0]), _ptr(bias), None, None, _ptr(y0), B, D, H, W, C, C, 9, 0.2, s)
f1 = lambda: call("df_wino_conv_fwd_bits", _ptr(x), _ptr(ww[0]), _ptr(bias), None, _ptr(y1), _ptr(bits), B, D, H, W, C, C, 9, 0.2, s)
for rep in range(2):
    print("forward            : %.3f ms   + sign bits: %.3f ms" % (timeit(f0, 4, 1) * 1e3, timeit(f1, 4, 1) * 1e3), flush=True)
assert torch.equal(y0, y1)
d0 = torch.empty_like(x); d1 = torch.empty_like(x)
g0 = lambda: call("df_wino_conv_fwd", _ptr(g), _ptr(ww[1]), None, None, _ptr(y0), _ptr(d0), B, D, H, W, C, C, 4, 0.2, s)
g1 = lambda: call("df_wino_conv_fwd_bits", _ptr(g), _ptr(ww[1]), None, _ptr(bits), _ptr(d1), None, B, D, H, W, C, C, 4, 0.2, s)
gp = lambda: call("df_wino_conv_fwd", _ptr(g), _ptr(ww[1]), None, None, None, _ptr(d0), B, D, H, W, C, C, 0, 0.2, s)
for rep in range(2):
    print("masked dgrad (fp32): %.3f ms   (bit words): %.3f ms   unmasked dgrad: %.3f ms" % (timeit(g0, 4, 1) * 1e3, timeit(g1, 4, 1) * 1e3, timeit(gp, 4, 1) * 1e3), flush=True)
g0(); g1(); torch.cuda.synchronize()
print("identical:", torch.equal(d0, d1))
