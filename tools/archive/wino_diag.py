#!/usr/bin/env python3
"""Timing of the diagnosis variants of wino3d_kernel (tuning library): which part of the main loop the time goes to.
Results of the variants are wrong by construction -- timing only."""
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

NAMES = {4194304: "staging loads FULLY COALESCED (1 KiB contiguous per instruction, walking the tensor: L1 misses)", 1048576: "staging loads one per MFMA row during k-step 1, stores before the barrier", 2097152: "staging loads one per MFMA row during k-step 0", 262144: "weight loads of the z-row-1 waves through a zero-length descriptor (half the weight stream)", 524288: "every weight load through a zero-length descriptor", 262144 | 4: "half the weight stream, no staging", 100: "production kernel, compile-time flags", 0: "production kernel", 1: "no transform (VALU)", 2: "no raw LDS reads", 4: "no staging (loads + LDS writes)", 8: "no weight loads",
         12: "no staging, no weight loads", 9: "no transform, no weight loads", 64: "staging loads alive, no LDS writes", 128: "staging loads -> zeros (LDS writes kept)",
         256: "staging loads read an always-cached address", 384: "staging loads confined to a 1 MB window (L2 hits, L1 misses)", 512: "no output stores", 7: "MFMA only (no transform / raw reads / staging)",
         }


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    D, H, W, C = 64, 96, 64, 128
    s = _stream()
    torch.manual_seed(0)
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    ww = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(ww), C, C, 0, s)
    bias = torch.rand(C, device="cuda") * 0.1
    y = torch.empty_like(x)
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,4,8,12,9,7,64,128,256,512,32").split(",")]
    base = None
    y0 = None
    for v in variants:
        lib().df_debug_set_wino(ctypes.c_int((v << 2) | (int(os.environ.get("SPX", "0")) << 26)))
        f = lambda: call("df_wino_conv_fwd", _ptr(x), _ptr(ww), _ptr(bias), None, None, _ptr(y), B, D, H, W, C, C, 9, 0.2, s)
        t = timeit(f, 4, 2)
        base = base or t
        if v == 0 and y0 is None:
            y0 = y.clone()
        err = ((y - y0).abs().max() / y0.abs().max()).item() if y0 is not None else float("nan")
        print("B=%d variant %5d  %-48s %8.3f ms  (%.3f of production)  max diff vs production %.1e" % (
            B, v, NAMES.get(v, "?"), t * 1e3, t / base, err), flush=True)
    lib().df_debug_set_wino(ctypes.c_int(0))


if __name__ == "__main__":
    main()
