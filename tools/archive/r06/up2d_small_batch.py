"""The 2-D 9-point forms (up-sampling-aware forward, pooled adjoint) and the plain F(2,3)xF(4,3) conv standalone at B = 8 vs B = 64 (128x96 fine, C = 128):
is the small-batch loss in the kernels or in the step's concurrency?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream
from tools.gpu_probe import timeit
s = _stream(); C = 128
w = (torch.rand((3, 3, C, C), device="cuda") * 2 - 1) * 0.03
bias = torch.rand(C, device="cuda") * 0.1
w0 = torch.empty(query("df_wino2d_packed_elems", C, C, 0), device="cuda"); call("df_wino2d_pack_weights", _ptr(w), _ptr(w0), C, C, 0, s)
w1 = torch.empty(query("df_wino2d_packed_elems", C, C, 1), device="cuda"); call("df_wino2d_pack_weights", _ptr(w), _ptr(w1), C, C, 1, s)
w4 = torch.empty(query("df_wino2d43_packed_elems", C, C, 0), device="cuda"); call("df_wino2d43_pack_weights", _ptr(w), _ptr(w4), C, C, 0, s)
for (Hc, Wc) in ((64, 48), (32, 24)):
    for B in (8, 64):
        xc = torch.rand((B, Hc, Wc, C), device="cuda") - 0.5
        xf = torch.rand((B, 2 * Hc, 2 * Wc, C), device="cuda") - 0.5
        y = torch.empty_like(xf); acc = torch.zeros_like(xc)
        tu = timeit(lambda: call("df_wino2d_upconv_fwd", _ptr(xc), _ptr(w0), _ptr(bias), _ptr(y), B, Hc, Wc, C, C, 9, 0.2, s), 10, 3)
        tp = timeit(lambda: call("df_wino2d_upconv_dgrad", _ptr(xf), _ptr(w1), _ptr(acc), B, Hc, Wc, C, C, s), 10, 3)
        tf = timeit(lambda: call("df_wino2d43_conv", _ptr(xf), _ptr(w4), _ptr(bias), None, None, _ptr(y), B, 2 * Hc, 2 * Wc, C, C, 9, 0.2, s), 10, 3)
        print("fine %dx%d B%d: up fwd %.1f us  pooled adjoint %.1f us  plain f24 fwd %.1f us" % (2 * Hc, 2 * Wc, B, tu * 1e6, tp * 1e6, tf * 1e6), flush=True)
