#!/usr/bin/env python3
"""A/B of build-time variants of conv_wino43.hip's main loop (libw43_probe_*.so: hipcc -DDF_W43_PROBE -DW43_NBQ=.. -DW43_STORE_KS=.. -DW43_EARLY=..
-shared conv_wino43.hip core.hip): sha-1 of the output against the first library's, and the top-level launch time at cfg3's shape."""
import ctypes
import glob
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402

libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "deep_fluids_amd", "csrc", "libw43_probe_*.so")))
P, I64 = ctypes.c_void_p, ctypes.c_int64
C = 128
s = _stream()
torch.manual_seed(0)
w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
bias = torch.rand(C, device="cuda") * 0.1
wb = torch.empty(96 * C * C, device="cuda")
xs = {B: torch.rand((B, 64, 96, 64, C), device="cuda") * 2 - 1 for B in (1, 16)}
xr = torch.rand((2, 7, 13, 10, C), device="cuda") * 2 - 1
ref = {}
for so in libs:
    h = ctypes.CDLL(so)
    h.df_wino43_probe.argtypes = [P, P, P, P, I64, I64, I64, I64, I64, I64, ctypes.c_float, ctypes.c_int, P]
    h.df_wino43_pack_weights.argtypes = [P, P, I64, I64, ctypes.c_int, P]
    h.df_wino43_pack_weights(_ptr(w), _ptr(wb), C, C, 0, s)
    out = [os.path.basename(so)]
    for name, x in (("ragged", xr), ("b1", xs[1])):
        y = torch.full_like(x, float("nan"))
        assert h.df_wino43_probe(_ptr(x), _ptr(wb), _ptr(bias), _ptr(y), x.shape[0], x.shape[1], x.shape[2], x.shape[3], C, C, 0.2, 0, s) == 0
        torch.cuda.synchronize()
        d = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:10]
        ref.setdefault(name, d)
        out.append("%s %s" % (name, "same" if d == ref[name] else "DIFFERENT"))
    x = xs[16]
    y = torch.empty_like(x)
    for rep in range(2):
        t = timeit(lambda: h.df_wino43_probe(_ptr(x), _ptr(wb), _ptr(bias), _ptr(y), 16, 64, 96, 64, C, C, 0.2, 0, s), 4, 2)
        out.append("B16 %.3f ms" % (t * 1e3))
    print("  ".join(out), flush=True)
