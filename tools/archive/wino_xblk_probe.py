#!/usr/bin/env python3
"""Round-4 probe: wino3d_kernel with its input staged by LDS-DMA from an x-blocked copy (tuning library, df_debug_wino_conv_fwd_xblk;
DESIGN.md 4.1b).  Checks every variant against the production kernel (bit-identical expected: same operands, same order) on ragged
small shapes and on the top-level shape, then times them.  usage: wino_xblk_probe.py [B ...]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import _lib as _libmod  # noqa: E402
_libmod.use_tuning_library()
from deep_fluids_amd._lib import call, query, lib  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402

I64, F32, P = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
NAMES = {104: "three buffers + LDS weights, pieces SPREAD one per MFMA row of k-step 0 (timing only)", 106: "three buffers, global weights, pieces spread one per MFMA row of k-step 0", 98: "THREE staging buffers (pieces 2 chunks ahead) + LDS weights (timing only)", 102: "THREE staging buffers, global weights", 97: "LDS weights (timing only) + all waves stage, behind", 113: "LDS weights + all waves stage, start of k-step 0", 99: "LDS weights + waves 4-7 stage, behind", 115: "LDS weights + waves 4-7 stage, start of k-step 0", 103: "LDS weights + waves 4-7 stage, in front", 100: "LDS weights, no staging at all", 33: "two weight sets + all waves, behind (rolled)", 41: "two weight sets + all waves, behind, unrolled", 43: "two weight sets + waves 4-7, behind, unrolled", 45: "two weight sets + all waves, in front, unrolled", 49: "two weight sets + all waves, start of k-step 0 (rolled)", 63: "two weight sets, no staging at all (timing only)", 67: "no staging at all, one weight set (timing only)", 17: "all waves, at the start of k-step 0 (rolled)", 25: "all waves, start of k-step 0, unrolled", 27: "waves 4-7, start of k-step 0, unrolled", 35: "waves 4-7 variant without any staging (timing only)", 1: "all waves issue, behind k-step 0's MFMAs (rolled)", 3: "waves 4-7 issue, behind the MFMAs (rolled)",
         5: "all waves, in front of the MFMAs (rolled)", 7: "waves 4-7, in front of the MFMAs (rolled)",
         9: "all waves, behind, unrolled", 11: "waves 4-7, behind, unrolled", 13: "all waves, in front, unrolled", 15: "waves 4-7, in front, unrolled",
         19: "waves 4-7 variant without any staging (timing only)"}


def xblk(x, xg, ww, bias, y, dims, cin, cout, variant, s):
    h = lib()
    f = h.df_debug_wino_conv_fwd_xblk
    f.restype = ctypes.c_int
    f.argtypes = [P, P, P, P, P, I64, I64, I64, I64, I64, I64, F32, ctypes.c_int, P]
    rc = f(_ptr(x), _ptr(xg), _ptr(ww), _ptr(bias), _ptr(y), *dims, cin, cout, 0.2, variant, s)
    if rc:
        raise RuntimeError("df_debug_wino_conv_fwd_xblk(%d) failed: %d %s" % (variant, rc, h.df_last_error()))


def case(B, D, H, W, cin, cout, variants, time_it):
    s = _stream()
    torch.manual_seed(B + D + W + cin)
    x = torch.rand((B, D, H, W, cin), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, cin, cout), device="cuda") * 2 - 1) * (2.0 / (27 * cin)) ** 0.5
    ww = torch.empty(query("df_wino_packed_elems", cin, cout, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(ww), cin, cout, 0, s)
    bias = torch.rand(cout, device="cuda") * 0.1
    y0 = torch.empty((B, D, H, W, cout), device="cuda")
    y = torch.empty_like(y0)
    h = lib()
    h.df_debug_wino_xblk_elems.restype = I64
    h.df_debug_wino_xblk_elems.argtypes = [I64] * 5
    xg = torch.empty(h.df_debug_wino_xblk_elems(B, D, H, W, cin), device="cuda")
    dims = (B, D, H, W)
    prod = lambda: call("df_wino_conv_fwd", _ptr(x), _ptr(ww), _ptr(bias), None, None, _ptr(y0), B, D, H, W, cin, cout, 9, 0.2, s)
    prod()
    tag = "%dx%dx%dx%d C%d->%d" % (B, D, H, W, cin, cout)
    if time_it:
        t0 = timeit(prod, 4, 2)
        print("%s production                                                   %8.3f ms" % (tag, t0 * 1e3), flush=True)
    for v in variants:
        y.fill_(float("nan")); xg.fill_(float("nan"))
        xblk(x, xg, ww, bias, y, dims, cin, cout, v, s)
        torch.cuda.synchronize()
        same = bool(torch.equal(y, y0))
        err = float((y - y0).abs().max() / y0.abs().max())
        line = "%s xblk %2d %-52s identical=%s maxdiff=%.1e" % (tag, v, NAMES.get(v, "?"), same, err)
        if time_it:
            tf = timeit(lambda: xblk(x, xg, ww, bias, y, dims, cin, cout, v, s), 4, 2)
            tc = timeit(lambda: xblk(x, xg, ww, bias, y, dims, cin, cout, v | 256, s), 4, 2)
            line += "  conv %8.3f ms (%.3f of production)  + copy %6.3f ms" % (tc * 1e3, tc / t0, (tf - tc) * 1e3)
        print(line, flush=True)


def main():
    vs = [int(v) for v in os.environ.get("VARIANTS", "1,3,5,7,9,11,13,15").split(",")]
    for shp in ((1, 4, 8, 8, 32, 32), (2, 5, 7, 9, 64, 32), (1, 6, 10, 12, 32, 64), (1, 16, 24, 16, 128, 128)):
        case(*shp, variants=vs, time_it=False)
    for B in [int(a) for a in sys.argv[1:]] or [16]:
        case(B, 64, 96, 64, 128, 128, variants=vs + [63, 67, 100], time_it=True)


if __name__ == "__main__":
    main()
