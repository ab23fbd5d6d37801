#!/usr/bin/env python3
"""Where the time of the direct bf16x3 convolution (conv_bf16.hip) goes: tuning-library variants whose results are wrong by construction."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import _lib as _libmod  # noqa: E402
_libmod.use_tuning_library()
from deep_fluids_amd._lib import call, query, lib, DF_CONV_BIAS, DF_CONV_LRELU  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402

s = _stream()
B, D, H, W, cin, cout = 4, 64, 96, 64, 128, 128
x = torch.rand((B, D, H, W, cin), device="cuda") * 2 - 1
w = (torch.rand((3, 3, 3, cin, cout), device="cuda") * 2 - 1) * 0.02
bias = torch.rand(cout, device="cuda") - 0.5
y = torch.empty((B, D, H, W, cout), device="cuda")
wp = torch.empty(query("df_conv_packed_elems_bf16x3", 27, cin, cout, 0), device="cuda")
call("df_conv_pack_weights_bf16x3", _ptr(w), _ptr(wp), 27, cin, cout, 0, s)
f = lambda: call("df_conv_fwd_bf16x3", _ptr(x), _ptr(wp), _ptr(bias), None, None, _ptr(y), B, D, H, W, cin, cout, 3, DF_CONV_BIAS | DF_CONV_LRELU, 0.2, s)
flops = 2.0 * 27 * cin * cout * B * D * H * W * 3
names = {0: "production", 1: "staging loads -> zeros", 8: "no staging", 2: "no weight loads", 4: "no LDS operand reads", 3: "zeros staged, no weights",
         6: "no weights, no LDS reads", 10: "no staging, no weights", 14: "MFMA + epilogue only"}
for dbg in (0, 1, 8, 2, 4, 3, 6, 10, 14):
    lib().df_debug_set_conv_bf16(ctypes.c_int(dbg))
    f(); torch.cuda.synchronize()
    t = timeit(f, 5, 2)
    print("variant %2d (%s): %.3f ms  (%.0f TFLOP/s of bf16 MFMA work, %.2f of 2500)" % (dbg, names[dbg], t * 1e3, flops / t / 1e12, flops / t / 2.5e15), flush=True)
lib().df_debug_set_conv_bf16(ctypes.c_int(0))
