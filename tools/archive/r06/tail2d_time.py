"""The 2-D backward tail: df_lrelu_words2d_bwd_pool2x (sign words) vs df_lrelu_bwd_pool2x (fp32 activation), cfg2's top level."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream, _new_bits
from tools.gpu_probe import timeit
s = _stream(); B, H, W, C = 64, 128, 96, 128
gy = torch.rand((B, H, W, C), device="cuda") - 0.5; act = torch.rand((B, H, W, C), device="cuda") - 0.5
gx = torch.empty_like(gy); p = torch.empty((B, H // 2, W // 2, C), device="cuda")
bits = _new_bits((B, 1, H, W), C, gy, 2); bits.random_()
for rep in range(2):
    t1 = timeit(lambda: call("df_lrelu_bwd_pool2x", _ptr(gy), _ptr(act), _ptr(gx), _ptr(p), 0.2, B, 1, H // 2, W // 2, C, 0, s), 10, 3)
    t2 = timeit(lambda: call("df_lrelu_words2d_bwd_pool2x", _ptr(gy), _ptr(bits), _ptr(gx), _ptr(p), 0.2, B, H // 2, W // 2, C, s), 10, 3)
    print("fp32-mask tail %.1f us (%.2f TB/s of 1.31 GB)   sign-word tail %.1f us (%.2f TB/s of 0.92 GB)" % (t1 * 1e6, 1.309e9 / t1 / 1e12, t2 * 1e6, 0.919e9 / t2 / 1e12), flush=True)
