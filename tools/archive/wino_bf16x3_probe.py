#!/usr/bin/env python3
"""The "bf16x3 in the Winograd domain" experiment (conv_wino.hip, wino3d_kernel PREC = 1; tuning library only): agreement with the fp32
Winograd kernel, timing against it and against the direct bf16x3 kernel, and where its time goes (variants whose results are wrong by
construction).  Result: correct at the first run (rel-Linf 9e-6 vs fp32), compute parts overlap (MFMA + transforms 1.7 ms
at 64x96x64x4) but the weight (26 GB / launch L2 -> L1) and staging streams do not hide behind k-steps that short: 3.1 ms = the direct
bf16x3 kernel."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import _lib as _libmod  # noqa: E402
_libmod.use_tuning_library()
from deep_fluids_amd._lib import call, query, lib, DF_CONV_BIAS, DF_CONV_LRELU, DF_CONV_MASK  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402

L = lib()
P, I64, I32, F32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
L.df_debug_wino_pack_weights_bf16x3.argtypes = [P, P, I64, I64, I32, P]
L.df_debug_wino_conv_fwd_bf16x3.argtypes = [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, F32, P]
s = _stream()
torch.manual_seed(0)


def pv(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def conv(kind, x, w, bias, mask, dims, cin, cout, flags):
    B, D, H, W = dims
    y = torch.empty((B, D, H, W, cout), device="cuda")
    if kind == "wino-bf16x3":
        wp = torch.empty(query("df_wino_packed_elems", cin, cout, 0), device="cuda")
        assert L.df_debug_wino_pack_weights_bf16x3(pv(w), pv(wp), cin, cout, 0, s) == 0
        f = lambda: L.df_debug_wino_conv_fwd_bf16x3(pv(x), pv(wp), pv(bias), None, pv(mask), pv(y), B, D, H, W, cin, cout, flags, 0.2, s)
    elif kind == "wino-fp32":
        wp = torch.empty(query("df_wino_packed_elems", cin, cout, 0), device="cuda")
        call("df_wino_pack_weights", _ptr(w), _ptr(wp), cin, cout, 0, s)
        f = lambda: call("df_wino_conv_fwd", _ptr(x), _ptr(wp), _ptr(bias), None, _ptr(mask), _ptr(y), B, D, H, W, cin, cout, flags, 0.2, s)
    else:
        wp = torch.empty(query("df_conv_packed_elems_bf16x3", 27, cin, cout, 0), device="cuda")
        call("df_conv_pack_weights_bf16x3", _ptr(w), _ptr(wp), 27, cin, cout, 0, s)
        f = lambda: call("df_conv_fwd_bf16x3", _ptr(x), _ptr(wp), _ptr(bias), None, _ptr(mask), _ptr(y), B, D, H, W, cin, cout, 3, flags, 0.2, s)
    assert f() in (0, None)
    torch.cuda.synchronize()
    return y, f


for dims, cin, cout in (((1, 8, 8, 8), 32, 32), ((2, 9, 17, 11), 64, 32), ((1, 16, 24, 16), 128, 128), ((4, 64, 96, 64), 128, 128)):
    x = torch.rand(dims + (cin,), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, cin, cout), device="cuda") * 2 - 1) * (1.0 / (27 * cin) ** 0.5)
    bias = torch.rand(cout, device="cuda") - 0.5
    mask = torch.rand(dims + (cout,), device="cuda") - 0.5
    for flags in (DF_CONV_BIAS | DF_CONV_LRELU, DF_CONV_MASK, 0):
        y0, f0 = conv("wino-fp32", x, w, bias, mask, dims, cin, cout, flags)
        y1, f1 = conv("wino-bf16x3", x, w, bias, mask, dims, cin, cout, flags)
        y2, f2 = conv("direct-bf16x3", x, w, bias, mask, dims, cin, cout, flags)
        sc = y0.abs().max().item()
        msg = "%s %d->%d flags %d: rel-Linf vs fp32 Winograd: Winograd-domain bf16x3 %.2e, direct bf16x3 %.2e" % (
            "x".join(map(str, dims)), cin, cout, flags, (y1 - y0).abs().max().item() / sc, (y2 - y0).abs().max().item() / sc)
        if dims[1] >= 64:
            msg += "   fp32 Winograd %.3f ms | Winograd-domain bf16x3 %.3f ms | direct bf16x3 %.3f ms" % tuple(timeit(f, 5, 2) * 1e3 for f in (f0, f1, f2))
        print(msg, flush=True)

dims, cin, cout = (4, 64, 96, 64), 128, 128
x = torch.rand(dims + (cin,), device="cuda") * 2 - 1
w = (torch.rand((3, 3, 3, cin, cout), device="cuda") * 2 - 1) * 0.02
bias = torch.rand(cout, device="cuda") - 0.5
names = {0: "production", 32: "xi_x-major MFMA order", 16: "staged chunk stored one k-step later", 64: "staging loads ahead of the MFMAs",
         1: "no transform", 2: "no LDS operand reads", 4: "no staging", 8: "no weight loads", 12: "no staging, no weight loads (all compute)",
         7: "MFMA + weights only", 11: "MFMA + staging only", 15: "MFMA + epilogue only"}
for dbg in (0, 32, 16, 64, 1, 2, 4, 8, 12, 7, 11, 15):
    L.df_debug_set_wino(ctypes.c_int(dbg))
    _, f = conv("wino-bf16x3", x, w, bias, None, dims, cin, cout, DF_CONV_BIAS | DF_CONV_LRELU)
    print("variant %2d (%s): %.3f ms" % (dbg, names[dbg], timeit(f, 5, 2) * 1e3), flush=True)
L.df_debug_set_wino(ctypes.c_int(0))
