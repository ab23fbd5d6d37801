"""A/B of wino2d_kernel at cfg2's top level (B = 64, 128x96, 128 -> 128): time + sha-1 of the output (run under DF_HIP_LIBRARY=<other build> to compare)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream
from tools.gpu_probe import timeit
s = _stream(); torch.manual_seed(3)
B, H, W, C = 64, 128, 96, 128
x = torch.rand((B, H, W, C), device="cuda") * 2 - 1
w = (torch.rand((3, 3, C, C), device="cuda") * 2 - 1) * 0.05
bias = torch.rand(C, device="cuda") * 0.1
ww = torch.empty(query("df_wino2d_packed_elems", C, C, 0), device="cuda")
call("df_wino2d_pack_weights", _ptr(w), _ptr(ww), C, C, 0, s)
y = torch.empty_like(x)
f = lambda fl, m: call("df_wino2d_conv_fwd", _ptr(x), _ptr(ww), _ptr(bias), None, _ptr(m) if m is not None else None, _ptr(y), B, H, W, C, C, fl, 0.2, s)
for fl, m, name in ((9, None, "fwd bias+lrelu"), (4, x, "dgrad masked")):
    f(fl, m); torch.cuda.synchronize()
    h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
    t = timeit(lambda: f(fl, m), 6, 3)
    print("wino2d %-16s %.3f ms  executed %.3f  sha %s" % (name, t * 1e3, 2.0 * 9 * C * C * B * H * W * (4.0 / 9.0) / t / 157.3e12, h), flush=True)
w4 = torch.empty(query("df_wino2d43_packed_elems", C, C, 0), device="cuda")
call("df_wino2d43_pack_weights", _ptr(w), _ptr(w4), C, C, 0, s)
y2 = torch.empty_like(x)
f4 = lambda fl, m: call("df_wino2d43_conv", _ptr(x), _ptr(w4), _ptr(bias), None, _ptr(m) if m is not None else None, _ptr(y2), B, H, W, C, C, fl, 0.2, s)
for fl, m, name in ((9, None, "fwd bias+lrelu"), (4, x, "dgrad masked")):
    f(fl, m); f4(fl, m); torch.cuda.synchronize()
    err = float((y2 - y).abs().max() / y.abs().max())
    t = timeit(lambda: f4(fl, m), 6, 3)
    print("wino2d43 %-14s %.3f ms  executed %.3f  max rel diff vs F(2,3)^2 %.2e" % (name, t * 1e3, 2.0 * 9 * C * C * B * H * W * (3.0 / 9.0) / t / 157.3e12, err), flush=True)
