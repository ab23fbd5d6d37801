#!/usr/bin/env python3
"""Round 5: the up-sampling-aware Winograd forward with the COARSE halo block staged as such (wino3d_kernel MODE 3, production) against the
fine-grid staging of rounds 2-4 (MODE 1, tuning-library variant 1000): results must be BIT-identical (same operands, same MFMA order);
then the time of both at cfg3's top level."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_fluids_amd import _lib as _libmod  # noqa: E402
_libmod.use_tuning_library()
from deep_fluids_amd._lib import call, query, lib  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402


def setv(v):
    lib().df_debug_set_wino(ctypes.c_int(v << 2))


def case(B, Dc, Hc, Wc, C, N, time_it=False):
    s = _stream()
    torch.manual_seed(B + Dc + Wc)
    xc = torch.rand((B, Dc, Hc, Wc, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, N), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    bias = torch.rand(N, device="cuda") * 0.1
    ww = torch.empty(query("df_wino_packed_elems", C, N, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(ww), C, N, 0, s)
    ys = []
    for v in (0, 1000):
        setv(v)
        y = torch.full((B, 2 * Dc, 2 * Hc, 2 * Wc, N), float("nan"), device="cuda")
        call("df_wino_upconv_fwd", _ptr(xc), _ptr(ww), _ptr(bias), _ptr(y), B, Dc, Hc, Wc, C, N, 9, 0.2, s)
        ys.append(y)
    torch.cuda.synchronize()
    same = torch.equal(ys[0], ys[1])
    # independent check: the plain Winograd conv on the materialised up-sampling
    xf = xc.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3).contiguous()
    yp = torch.empty_like(ys[0])
    setv(0)
    call("df_wino_conv_fwd", _ptr(xf), _ptr(ww), _ptr(bias), None, None, _ptr(yp), B, 2 * Dc, 2 * Hc, 2 * Wc, C, N, 9, 0.2, s)
    torch.cuda.synchronize()
    err = ((ys[0] - yp).abs().max() / yp.abs().max()).item()
    msg = "B%d coarse %dx%dx%d C%d->%d: MODE 3 == MODE 1 bitwise: %s; vs plain conv on the materialised up-sampling rel-linf %.2e" % (B, Dc, Hc, Wc, C, N, same, err)
    if time_it:
        ts = []
        for v in (0, 1000):
            setv(v)
            f = lambda: call("df_wino_upconv_fwd", _ptr(xc), _ptr(ww), _ptr(bias), _ptr(ys[0]), B, Dc, Hc, Wc, C, N, 9, 0.2, s)
            ts.append(timeit(f, 5, 2))
        fe = 2.0 * C * N * B * 8 * Dc * Hc * Wc * 27.0 / 8.0
        msg += " | coarse staging %.3f ms (executed MFMA fraction %.3f), fine staging %.3f ms (%.3f)" % (
            ts[0] * 1e3, fe / ts[0] / 157.3e12, ts[1] * 1e3, fe / ts[1] / 157.3e12)
    setv(0)
    print(msg, flush=True)
    assert same and err < 2e-5


if __name__ == "__main__":
    case(1, 3, 5, 6, 32, 32)
    case(2, 3, 9, 6, 64, 32)           # odd coarse extents: partial tile blocks
    case(1, 7, 10, 7, 128, 128)        # cfg4's x0
    case(2, 8, 12, 8, 128, 128)
    case(2, 6, 9, 12, 128, 128)        # run.bat liquid3_vis x0
    case(4, 32, 48, 32, 128, 128, True)
    case(16, 32, 48, 32, 128, 128, True)
    case(4, 56, 80, 56, 128, 128, True)    # cfg4 top level
    case(4, 64, 64, 64, 64, 64, True)      # cfg5 top level
