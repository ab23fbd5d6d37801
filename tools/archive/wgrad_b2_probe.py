#!/usr/bin/env python3
"""Weight-gradient algorithm choice at the SMALL per-GPU batches of the strong-scaling runs (global batch 16 over 4 | 8 GPUs: B = 4 | 2):
conv_wgrad.hip::wgrad_algo picks by image rows with thresholds measured at B = 16.  Times every `algo` of df_conv_wgrad_algo and
df_upconv_wgrad_algo per generator level (gpurun tuning aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402

NAMES = {0: "default", 1: "direct", 2: "x", 3: "xy", 4: "xyz"}
FORMS = {0: "direct", 1: "x", 2: "xy", 3: "xyz", 10: "thin-mfma", 11: "thin-valu"}


def conv(B, D, H, W, C=128):
    s = _stream()
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    g = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    nb = query("df_conv_wgrad_workspace_bytes", B, D, H, W, C, C, 3)
    ws = torch.empty((nb + 3) // 4, device="cuda")
    gw = torch.empty((27, C, C), device="cuda"); gb = torch.empty(C, device="cuda")
    msg = "conv   B%d %dx%dx%d: default form = %s |" % (B, D, H, W, FORMS.get(query("df_conv_wgrad_form", B, D, H, W, C, C, 3, 0)))
    for algo in (0, 1, 2, 3, 4):
        f = lambda: call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, algo, s)
        msg += " %s %.3f ms" % (NAMES[algo], timeit(f, 5, 2) * 1e3)
    print(msg, flush=True)


def upconv(B, Dc, Hc, Wc, C=128):
    s = _stream()
    xc = torch.rand((B, Dc, Hc, Wc, C), device="cuda") * 2 - 1
    g = torch.rand((B, 2 * Dc, 2 * Hc, 2 * Wc, C), device="cuda") * 2 - 1
    nb = query("df_upconv_wgrad_workspace_bytes", B, Dc, Hc, Wc, C, C, 3)
    ws = torch.empty((nb + 3) // 4, device="cuda")
    gw = torch.empty((27, C, C), device="cuda"); gb = torch.empty(C, device="cuda")
    msg = "upconv B%d fine %dx%dx%d: default form = %s |" % (B, 2 * Dc, 2 * Hc, 2 * Wc, "xyz-27pt" if query("df_upconv_wgrad_form", B, Dc, Hc, Wc, C, C, 3, 0) == 3 else "parity-class")
    for algo, name in ((0, "default"), (2, "parity-class"), (4, "xyz-27pt")):
        f = lambda: call("df_upconv_wgrad_algo", _ptr(xc), _ptr(g), _ptr(gw), _ptr(gb), B, Dc, Hc, Wc, C, C, 3, _ptr(ws), nb, algo, s)
        msg += " %s %.3f ms" % (name, timeit(f, 5, 2) * 1e3)
    print(msg, flush=True)


if __name__ == "__main__":
    for B in [int(v) for v in (sys.argv[1:] or ["2", "4"])]:
        for lvl in ((8, 12, 8), (16, 24, 16), (32, 48, 32), (64, 96, 64)):
            conv(B, *lvl)
        for lvl in ((8, 12, 8), (16, 24, 16), (32, 48, 32)):
            upconv(B, *lvl)
