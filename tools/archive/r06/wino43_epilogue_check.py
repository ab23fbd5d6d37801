"""conv_wino43.hip epilogue A/B: sha-1 of every epilogue variant's outputs (primary, add-up, sign words) on full / ragged blocks + time at cfg3's top level.
Run once per build (DF_HIP_LIBRARY=<other build>) and diff the lines."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query, DF_CONV_BIAS, DF_CONV_LRELU, DF_CONV_RESIDUAL, DF_CONV_MASK, DF_CONV_ADDUP
from deep_fluids_amd.ops import _ptr, _stream, _new_bits
from tools.gpu_probe import timeit
s = _stream()
tag = os.path.basename(os.environ.get("DF_HIP_LIBRARY", "release"))
sha = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:10]
def run(B, D, H, W, C, name, time_it=False):
    torch.manual_seed(B + D * 3 + H * 5 + W * 7 + C)
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    bias = torch.rand(C, device="cuda") * 0.1 - 0.05
    aux = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    wb = torch.empty(query("df_wino43_packed_elems", C, C, 0), device="cuda")
    call("df_wino43_pack_weights", _ptr(w), _ptr(wb), C, C, 0, s)
    y = torch.full_like(x, float("nan")); y2 = torch.full_like(x, float("nan"))
    bits = _new_bits((B, D, H, W), C, x); bits.zero_()
    FW = DF_CONV_BIAS | DF_CONV_LRELU
    even = D % 2 == 0 and H % 2 == 0 and W % 2 == 0
    xc = torch.rand((B, D // 2, H // 2, W // 2, C), device="cuda") if even else None
    outs = []
    def go(label, fn, *tensors):
        fn(); torch.cuda.synchronize()
        t = timeit(fn, 4, 2) if time_it else 0.0
        outs.append("%s %s%s" % (label, "/".join(sha(q) for q in tensors), (" %.3f ms" % (t * 1e3)) if time_it else ""))
    go("fwd", lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), _ptr(bias), None, None, None, _ptr(y), None, None, B, D, H, W, C, C, FW, 0.2, s), y)
    go("fwd+bits", lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), _ptr(bias), None, None, None, _ptr(y), None, _ptr(bits), B, D, H, W, C, C, FW, 0.2, s), y, bits)
    go("mask<-bits", lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), None, None, None, _ptr(bits), _ptr(y), None, None, B, D, H, W, C, C, DF_CONV_MASK, 0.2, s), y)
    go("mask<-fp32", lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), None, None, _ptr(aux), None, _ptr(y), None, None, B, D, H, W, C, C, DF_CONV_MASK, 0.2, s), y)
    go("residual", lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), None, _ptr(aux), None, None, _ptr(y), None, None, B, D, H, W, C, C, DF_CONV_RESIDUAL, 0.2, s), y)
    go("plain", lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), None, None, None, None, _ptr(y), None, None, B, D, H, W, C, C, 0, 0.2, s), y)
    if even:
        go("addup", lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), _ptr(bias), _ptr(xc), None, None, _ptr(y), _ptr(y2), None, B, D, H, W, C, C, FW | DF_CONV_ADDUP, 0.2, s), y, y2)
        go("addup+bits,noY", lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), _ptr(bias), _ptr(xc), None, None, None, _ptr(y2), _ptr(bits), B, D, H, W, C, C, FW | DF_CONV_ADDUP, 0.2, s), y2, bits)
    print("%s  B%d %dx%dx%d C%d:  %s" % (tag, B, D, H, W, C, "  ".join(outs)), flush=True)
for shp in ((1, 8, 16, 8, 32), (2, 6, 10, 12, 64), (1, 7, 9, 13, 32), (1, 4, 8, 16, 128), (2, 12, 24, 16, 64)):
    run(*shp, "small")
run(int(os.environ.get("B", "16")), 64, 96, 64, 128, "top", True)
