#!/usr/bin/env python3
"""Diagnostic: per-shape time of the conv entry points in one AE3 (cfg5 shape) train step."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import ops, _lib
from deep_fluids_amd.trainer import AETrainer, default_config
B, R = 4, 128
cfg = default_config(is_3d=True, res_x=R, res_y=R, res_z=R, filters=64, batch_size=B, num_samples=5000, z_num=16, p_num=2)
tr = AETrainer(cfg)
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.rand((B, 2, 10), device="cuda", generator=g) * 2 - 1
x = ops.curl3(torch.rand((B, R, R, R, 3), device="cuda", generator=g) * 2 - 1)
x = (x / x.abs().max()).contiguous()
for _ in range(2):
    tr.train_step(x, y)


def select(name, args):
    if name in ("df_conv_wgrad", "df_upconv_wgrad"):
        return ("%s B%d %dx%dx%d %d->%d" % ((name,) + tuple(int(getattr(v, "value", v)) for v in args[4:10])), 1.0)
    if name in ("df_conv_fwd", "df_wino_conv_fwd", "df_conv_s2_fwd", "df_upconv_fwd", "df_wino_upconv_fwd", "df_upconv_dgrad", "df_wino_upconv_dgrad"):
        k = {"df_conv_fwd": 6, "df_wino_conv_fwd": 6, "df_conv_s2_fwd": 4, "df_upconv_fwd": 4, "df_wino_upconv_fwd": 4, "df_upconv_dgrad": 3,
             "df_wino_upconv_dgrad": 3}[name]
        return ("%s B%d %dx%dx%d %d->%d" % ((name,) + tuple(int(getattr(v, "value", v)) for v in args[k:k + 6])), 1.0)
    return None


_lib.TIMER = _lib.KernelTimer(select)
tr.train_step(x, y)
s = _lib.TIMER.summary()
_lib.TIMER = None
tot = sum(v["seconds"] for v in s.values())
print("timed total %.1f ms" % (tot * 1e3))
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["seconds"])[:30]:
    print("%8.2f ms  x%d  %s" % (v["seconds"] * 1e3, v["launches"], k))
