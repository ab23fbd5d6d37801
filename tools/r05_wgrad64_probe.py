#!/usr/bin/env python3
"""Round 5: 64 -> 64 (x,y,z)-Winograd weight gradient (cfg5 / every AE run): interleaved tile-row pairs per wave (production since round 5) vs contiguous
sub-ranges (tuning variant 16, rounds 3-4) -- timing and agreement with the direct kernel."""
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


def run(B, D, H, W, C):
    torch.manual_seed(0)
    s = _stream()
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    g = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    nb = max(query("df_conv_wgrad_workspace_bytes", B, D, H, W, C, C, 3), 64 * 4 * 64 * 64 * 64 * 4 + (1 << 22))
    ws = torch.empty((nb + 3) // 4, device="cuda")
    res = {}
    for tag, algo, dbg in (("direct", 1, 0), ("xyz interleaved (production)", 4, 0), ("xyz contiguous (rounds 3-4)", 4, 16),
                           ("xyz interleaved, 64 ranges", 4 | (64 << 3), 0), ("xyz interleaved, 16 ranges", 4 | (16 << 3), 0)):
        lib().df_debug_set_wgrad(ctypes.c_int(dbg))
        gw = torch.empty((27, C, C), device="cuda"); gb = torch.empty(C, device="cuda")
        f = lambda: call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, algo, s)
        f(); torch.cuda.synchronize()
        t = timeit(f, 4, 1)
        res[tag] = (gw.clone(), t)
    lib().df_debug_set_wgrad(ctypes.c_int(0))
    fe = 2.0 * C * C * B * D * H * W * 8.0
    w0 = res["direct"][0]
    print("B%d %dx%dx%d C%d: " % (B, D, H, W, C) + " | ".join("%s %.3f ms (executed %.3f) rel-linf vs direct %.1e" % (
        k, v[1] * 1e3, fe / v[1] / 157.3e12, ((v[0] - w0).abs().max() / w0.abs().max()).item()) for k, v in res.items() if k != "direct"), flush=True)


if __name__ == "__main__":
    run(4, 128, 128, 128, 64)
    run(4, 64, 64, 64, 64)
    run(4, 48, 72, 48, 64) if False else None
