#!/usr/bin/env python3
"""Winograd conv kernel vs the direct MFMA kernel: difference and timing at cfg3 shapes (gpurun tuning aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import _lib as _libmod  # noqa: E402
_libmod.use_tuning_library()      # needs `make -C deep_fluids_amd/csrc tuning`
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402


def run(B, D, H, W, C, N, flags=8 | 1, iters=5):
    torch.manual_seed(0)
    s = _stream()
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, N), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    bias = torch.rand(N, device="cuda") * 0.1
    res = torch.rand((B, D, H, W, N), device="cuda")
    for mode in (0, 1):
        K, NN = (C, N) if mode == 0 else (N, C)
        xin = x if mode == 0 else torch.rand((B, D, H, W, N), device="cuda") * 2 - 1
        wd = torch.empty(query("df_conv_packed_elems", 27, C, N, mode), device="cuda")
        call("df_conv_pack_weights", _ptr(w), _ptr(wd), 27, C, N, mode, s)
        ww = torch.empty(query("df_wino_packed_elems", C, N, mode), device="cuda")
        call("df_wino_pack_weights", _ptr(w), _ptr(ww), C, N, mode, s)
        b_ = bias if mode == 0 else torch.rand(NN, device="cuda") * 0.1
        r_ = res if mode == 0 else torch.rand((B, D, H, W, NN), device="cuda")
        fl = flags if mode == 0 else 4
        y0 = torch.empty((B, D, H, W, NN), device="cuda"); y1 = torch.full_like(y0, float("nan"))
        call("df_conv_fwd", _ptr(xin), _ptr(wd), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y0), B, D, H, W, K, NN, 3, fl, 0.2, s)
        call("df_wino_conv_fwd", _ptr(xin), _ptr(ww), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y1), B, D, H, W, K, NN, fl, 0.2, s)
        torch.cuda.synchronize()
        err = (y0 - y1).abs().max().item() / y0.abs().max().item()
        l1 = (y0 - y1).abs().sum().item() / y0.abs().sum().item()
        fl_ = 2.0 * B * D * H * W * 27 * C * N
        t0 = timeit(lambda: call("df_conv_fwd", _ptr(xin), _ptr(wd), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y0), B, D, H, W, K, NN, 3, fl, 0.2, s), iters, 2)
        t1 = timeit(lambda: call("df_wino_conv_fwd", _ptr(xin), _ptr(ww), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y1), B, D, H, W, K, NN, fl, 0.2, s), iters, 2)
        print("B%d %dx%dx%d C%d N%d mode%d: rel-linf %.2e rel-l1 %.2e | direct %.3f ms (%.0f TF)  wino %.3f ms (%.0f TF-eq, %.0f TF mfma)" % (
            B, D, H, W, C, N, mode, err, l1, t0 * 1e3, fl_ / t0 / 1e12, t1 * 1e3, fl_ / t1 / 1e12, fl_ * 8 / 27 / t1 / 1e12), flush=True)


def run_one(B, D, H, W, C, N, mode, fl, iters=3):
    torch.manual_seed(0)
    s = _stream()
    K, NN = (C, N) if mode == 0 else (N, C)
    x = torch.rand((B, D, H, W, K), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, N), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    ww = torch.empty(query("df_wino_packed_elems", C, N, mode), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(ww), C, N, mode, s)
    b_ = torch.rand(NN, device="cuda") * 0.1
    r_ = torch.rand((B, D, H, W, NN), device="cuda")
    y1 = torch.empty((B, D, H, W, NN), device="cuda")
    t1 = timeit(lambda: call("df_wino_conv_fwd", _ptr(x), _ptr(ww), _ptr(b_), _ptr(r_), _ptr(r_), _ptr(y1), B, D, H, W, K, NN, fl, 0.2, s), iters, 2)
    print("  mode %d flags %d: %.3f ms" % (mode, fl, t1 * 1e3), flush=True)


if __name__ == "__main__":
    import ctypes
    from deep_fluids_amd._lib import lib
    run(1, 6, 10, 12, 32, 32)
    run(2, 16, 24, 16, 128, 128)
    for dbg in [int(v) for v in os.environ.get("WINO_DBG", "0").split(",")]:
        lib().df_debug_set_wino(ctypes.c_int(dbg))
        print("dbg", dbg)
        if (dbg >> 2) & 16:
            for mode_flags in ((0, 9), (1, 4)):      # forward (bias+lrelu) and dgrad (mask) profiled separately
                lib().df_debug_wino_prof(None, 1)
                run_one(4, 64, 96, 64, 128, 128, *mode_flags)
                buf = (ctypes.c_ulonglong * 32)()
                lib().df_debug_wino_prof(buf, 0)
                n = max(buf[3], 1)
                print("  mode %d flags %d per block cycles: setup %.0f  main %.0f  epilogue %.0f  (n=%d)" % (mode_flags + (buf[0] / n, buf[1] / n, buf[2] / n, n)))
                print("    epilogue split (both cout blocks): emit %.0f  barrier %.0f  combine+stores %.0f  barrier(+store acks) %.0f" % tuple(buf[8 + i] / n for i in range(4)))
                print("    main split: wait-raw %.0f  transform+stage+rawissue %.0f  wait-B %.0f  mfma+Bissue %.0f" % tuple(buf[4 + i] / n for i in range(4)))
            continue
        run(4, 64, 96, 64, 128, 128, iters=3)
        if (dbg >> 2) & 16:
            buf = (ctypes.c_ulonglong * 32)()
            lib().df_debug_wino_prof(buf, 0)
            n = max(buf[3], 1)
            print("  per block cycles: setup %.0f  main %.0f  epilogue %.0f  (n=%d)" % (buf[0] / n, buf[1] / n, buf[2] / n, n))
            print("  epilogue split (both cout blocks): emit %.0f  barrier %.0f  combine+stores %.0f  barrier(+store acks) %.0f" % tuple(buf[8 + i] / n for i in range(4)))
            print("  main split: wait-raw %.0f  transform+stage+rawissue %.0f  wait-B %.0f  mfma+Bissue %.0f" % tuple(buf[4 + i] / n for i in range(4)))
    if len(sys.argv) > 1:
        run(16, 64, 96, 64, 128, 128, iters=3)
