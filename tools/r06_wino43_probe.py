#!/usr/bin/env python3
"""Round 6: the F(2,3) x F(2,3) x F(4,3) Winograd family (conv_wino43.hip) against the F(2,3)^3 family (conv_wino.hip) and the fp64 direct sum:
correctness on ragged shapes and all fused epilogues, precision, and the top-level launch time at cfg3's shape."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402


def ref64(x, w, bias, leak):
    import torch.nn.functional as F
    xx = x.double().permute(0, 4, 1, 2, 3)
    ww = w.double().permute(4, 3, 0, 1, 2)
    y = F.conv3d(xx.cpu(), ww.cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 4, 1)
    return torch.maximum(y, leak * y)


def check(B, D, H, W, C):
    s = _stream()
    torch.manual_seed(1)
    x = torch.randn((B, D, H, W, C), device="cuda") * 0.3
    x = torch.maximum(x, 0.2 * x)
    w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (6.0 / (54 * C)) ** 0.5
    bias = torch.rand(C, device="cuda") * 0.1 - 0.05
    wa = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wa), C, C, 0, s)
    wb = torch.empty(query("df_wino43_packed_elems", C, C, 0), device="cuda")
    call("df_wino43_pack_weights", _ptr(w), _ptr(wb), C, C, 0, s)
    ya = torch.empty_like(x); yb = torch.full_like(x, float("nan"))
    call("df_wino_conv_fwd", _ptr(x), _ptr(wa), _ptr(bias), None, None, _ptr(ya), B, D, H, W, C, C, 9, 0.2, s)
    call("df_wino43_conv", _ptr(x), _ptr(wb), _ptr(bias), None, None, None, _ptr(yb), None, None, B, D, H, W, C, C, 9, 0.2, s)
    torch.cuda.synchronize()
    r = ref64(x, w, bias, 0.2)
    ea = float((ya.double().cpu() - r).abs().sum() / r.abs().sum())
    eb = float((yb.double().cpu() - r).abs().sum() / r.abs().sum())
    mb = float((yb.double().cpu() - r).abs().max() / r.abs().max())
    print("shape %dx%dx%dx%d C%d: rel-L1 F(2,3)^3 %.3e   F(2,2,4) %.3e (max %.3e)   %s" % (B, D, H, W, C, ea, eb, mb, "OK" if eb < 5e-6 else "MISMATCH"), flush=True)
    return eb < 5e-6


def main():
    ok = True
    for shp in ((1, 4, 8, 8, 32), (2, 8, 16, 8, 32), (1, 6, 10, 12, 64), (1, 7, 9, 13, 32), (1, 16, 24, 16, 128)):
        ok &= check(*shp)
    if not ok or "--check" in sys.argv:
        return
    s = _stream()
    B, D, H, W, C = int(os.environ.get("B", "16")), 64, 96, 64, 128
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    bias = torch.rand(C, device="cuda") * 0.1
    wa = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wa), C, C, 0, s)
    wb = torch.empty(query("df_wino43_packed_elems", C, C, 0), device="cuda")
    call("df_wino43_pack_weights", _ptr(w), _ptr(wb), C, C, 0, s)
    y = torch.empty_like(x)
    ta = timeit(lambda: call("df_wino_conv_fwd", _ptr(x), _ptr(wa), _ptr(bias), None, None, _ptr(y), B, D, H, W, C, C, 9, 0.2, s), 4, 2)
    tb = timeit(lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), _ptr(bias), None, None, None, _ptr(y), None, None, B, D, H, W, C, C, 9, 0.2, s), 4, 2)
    # the 27-point forms of the F(2,3)^3 family at the same level (coarse 32x48x32)
    xc = torch.rand((B, D // 2, H // 2, W // 2, C), device="cuda") * 2 - 1
    wad = torch.empty(query("df_wino_packed_elems", C, C, 1), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wad), C, C, 1, s)
    acc = torch.zeros_like(xc)
    tu = timeit(lambda: call("df_wino_upconv_fwd", _ptr(xc), _ptr(wa), _ptr(bias), _ptr(y), B, D // 2, H // 2, W // 2, C, C, 9, 0.2, s), 4, 2)
    tp = timeit(lambda: call("df_wino_upconv_dgrad", _ptr(x), _ptr(wad), _ptr(acc), B, D // 2, H // 2, W // 2, C, C, s), 4, 2)
    f27 = 2.0 * C * C * B * D * H * W * 27.0 / 8.0
    print("27-point forms B%d: up-sampling-aware forward %.3f ms (executed %.3f)   pooled adjoint %.3f ms (executed %.3f)" % (
        B, tu * 1e3, f27 / tu / 157.3e12, tp * 1e3, f27 / tp / 157.3e12), flush=True)
    # the block tail: add-up + sign words without the primary output (FL 345), and the masked dgrad from sign words (FL 132)
    from deep_fluids_amd.ops import _new_bits
    from deep_fluids_amd._lib import DF_CONV_ADDUP
    tbits = _new_bits((B, D, H, W), C, x)
    y2 = torch.empty_like(x)
    tt = timeit(lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), _ptr(bias), _ptr(xc), None, None, None, _ptr(y2), _ptr(tbits), B, D, H, W, C, C, 9 | DF_CONV_ADDUP, 0.2, s), 4, 2)
    tm = timeit(lambda: call("df_wino43_conv", _ptr(x), _ptr(wb), None, None, None, _ptr(tbits), _ptr(y), None, None, B, D, H, W, C, C, 4, 0.2, s), 4, 2)
    print("F(2,2,4) epilogue variants B%d: add-up + sign words, no primary %.3f ms   mask from sign words %.3f ms" % (B, tt * 1e3, tm * 1e3), flush=True)
    fl = 2.0 * 27 * C * C * B * D * H * W
    print("top level 64x96x64 C128 B%d: F(2,3)^3 %.3f ms (executed %.3f)   F(2,2,4) %.3f ms (executed %.3f)   ratio %.3f" % (
        B, ta * 1e3, fl * 8 / 27 / ta / 157.3e12, tb * 1e3, fl * 6 / 27 / tb / 157.3e12, tb / ta), flush=True)


def breakdown():
    """Time with one part of the main loop removed (probe build libw43_probe.so: hipcc -DDF_W43_PROBE -shared conv_wino43.hip core.hip)."""
    import ctypes
    so = os.path.join(ROOT, "deep_fluids_amd", "csrc", os.environ.get("W43_PROBE_LIB", "libw43_probe.so"))
    if not os.path.exists(so):
        print("no probe library (%s)" % so)
        return
    h = ctypes.CDLL(so)
    P, I64 = ctypes.c_void_p, ctypes.c_int64
    h.df_wino43_probe.argtypes = [P, P, P, P, I64, I64, I64, I64, I64, I64, ctypes.c_float, ctypes.c_int, P]
    h.df_wino43_pack_weights.argtypes = [P, P, I64, I64, ctypes.c_int, P]
    s = _stream()
    B, D, H, W, C = int(os.environ.get("B", "4")), 64, 96, 64, 128
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    bias = torch.rand(C, device="cuda") * 0.1
    wb = torch.empty(96 * C * C, device="cuda")
    h.df_wino43_pack_weights(_ptr(w), _ptr(wb), C, C, 0, s)
    y = torch.empty_like(x)
    names = {0: "production", 1: "no input transform", 2: "no LDS operand reads", 3: "no transform, no LDS reads", 4: "no staging", 8: "no weight reloads",
             7: "MFMA + weights + epilogue", 64: "staging loads kept, no LDS writes", 128: "staged zeros (LDS writes only)", 12: "no staging, no weights", 15: "MFMA + epilogue only"}
    base = None
    fl = 2.0 * C * C * B * D * H * W * 6.0
    for v in [int(t) for t in os.environ.get("W43_VARIANTS", "0,1,2,3,4,8,12,7,15").split(",")]:
        def f():
            rc = h.df_wino43_probe(_ptr(x), _ptr(wb), _ptr(bias), _ptr(y), B, D, H, W, C, C, 0.2, v, s)
            assert rc == 0, rc
        t = timeit(f, 4, 2)
        base = base or t
        print("F(2,2,4) B%d variant %2d %-28s %7.3f ms (%.3f of production)  executed %.3f" % (B, v, names[v], t * 1e3, t / base, fl / t / 157.3e12), flush=True)


if __name__ == "__main__":
    if "--breakdown" in sys.argv:
        breakdown()
    else:
        main()
