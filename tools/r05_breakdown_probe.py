#!/usr/bin/env python3
"""Round-5 time breakdown (tuning library; variant results wrong by construction -- timing only) of the second-tier Winograd kernels:
  * the 27-point up-sampling-aware forward `wino3d_kernel<.., 1>` and its pooled adjoint `<.., 2>` at cfg3's top level;
  * the plain F(2^3,3^3) kernel at 64 -> 64 channels (cfg5 / every auto-encoder run of the reference), beside 128 -> 128.
Each line: production time, then the time with one part removed (1 no input transform, 2 no LDS operand reads, 4 no staging, 8 no weight
loads, 7 = MFMA + epilogue only, 512 no output stores, 64 staging loads kept but no LDS writes, 128 staged zeros)."""
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

NAMES = {0: "production", 1: "no input transform", 2: "no LDS operand reads", 3: "no transform, no LDS reads", 4: "no staging", 7: "MFMA + epilogue only",
         8: "no weight loads", 64: "staging loads alive, no LDS writes", 128: "staged zeros (LDS writes only)", 512: "no output stores"}


def setv(v):
    lib().df_debug_set_wino(ctypes.c_int(v << 2))


def sweep(tag, f, variants, flops_exec):
    base = None
    for v in variants:
        setv(v)
        t = timeit(f, 4, 2)
        base = base or t
        print("%-34s variant %4d %-36s %8.3f ms (%.3f of production)%s" % (
            tag, v, NAMES.get(v, "?"), t * 1e3, t / base, "   executed MFMA fraction %.3f" % (flops_exec / t / 157.3e12) if t == base else ""), flush=True)
    setv(0)


def main():
    s = _stream()
    torch.manual_seed(0)
    # ---- 27-point forms at cfg3's top level: coarse 32x48x32 -> fine 64x96x64, 128 -> 128, batch 16
    B, Dc, Hc, Wc, C = int(os.environ.get("B", "16")), 32, 48, 32, 128
    xc = torch.rand((B, Dc, Hc, Wc, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    bias = torch.rand(C, device="cuda") * 0.1
    ww = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(ww), C, C, 0, s)
    y = torch.empty((B, 2 * Dc, 2 * Hc, 2 * Wc, C), device="cuda")
    nvox = B * 8 * Dc * Hc * Wc
    fe = 2.0 * 27 * C * C * nvox * (1.0 / 8.0)          # 27 of 64 points x 8/27 ... = 27 products per 2x2x2 tile and (cin, cout): 27/8 per voxel
    fe = 2.0 * C * C * nvox * 27.0 / 8.0
    sweep("up-fwd <..,1> 64x96x64 C128 B%d" % B, lambda: call("df_wino_upconv_fwd", _ptr(xc), _ptr(ww), _ptr(bias), _ptr(y), B, Dc, Hc, Wc, C, C, 9, 0.2, s),
          [0, 1, 2, 3, 4, 7, 8, 64, 128, 512], fe)
    wwd = torch.empty(query("df_wino_packed_elems", C, C, 1), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wwd), C, C, 1, s)
    acc = torch.zeros_like(xc)
    sweep("pool-dgrad <..,2> 64x96x64 C128 B%d" % B, lambda: call("df_wino_upconv_dgrad", _ptr(y), _ptr(wwd), _ptr(acc), B, Dc, Hc, Wc, C, C, s),
          [0, 1, 2, 3, 4, 7, 8, 64, 128], fe)
    del xc, y, acc
    # ---- plain kernel, 128 -> 128 at 64x96x64 B=4 and 64 -> 64 at 128^3 B=2 (same voxel count per launch x channel work ratio 4)
    for (Bp, D, H, W, Cc) in ((4, 64, 96, 64, 128), (2, 128, 128, 128, 64)):
        x = torch.rand((Bp, D, H, W, Cc), device="cuda") * 2 - 1
        w = (torch.rand((3, 3, 3, Cc, Cc), device="cuda") * 2 - 1) * (2.0 / (27 * Cc)) ** 0.5
        bias = torch.rand(Cc, device="cuda") * 0.1
        ww = torch.empty(query("df_wino_packed_elems", Cc, Cc, 0), device="cuda")
        call("df_wino_pack_weights", _ptr(w), _ptr(ww), Cc, Cc, 0, s)
        yy = torch.empty_like(x)
        fe = 2.0 * Cc * Cc * Bp * D * H * W * 8.0
        sweep("plain fwd %dx%dx%d C%d B%d" % (D, H, W, Cc, Bp),
              lambda: call("df_wino_conv_fwd", _ptr(x), _ptr(ww), _ptr(bias), None, None, _ptr(yy), Bp, D, H, W, Cc, Cc, 9, 0.2, s),
              [100, 1, 2, 4, 8, 7, 64, 128, 512], fe)
        del x, yy


if __name__ == "__main__":
    main()
