#!/usr/bin/env python3
"""Record every conv output of one F=128 train step with direct vs Winograd FORWARD convs and compare (gpurun aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import df_oracle as orc  # noqa: E402
from deep_fluids_amd import ops  # noqa: E402
from deep_fluids_amd.trainer import Trainer, default_config  # noqa: E402

spatial, filters, batch = (16, 24, 16), 128, 1
rng = np.random.RandomState(123)
oshape = list(spatial) + [3]
p = orc.generator_init(rng, 3, oshape, filters)
for k in p:
    if k.endswith("biases"):
        p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
x, y = orc.synthetic_batch(rng, batch, spatial)
orig_pack, orig_raw = ops._pack, ops._conv_raw
rec = {}
for enabled in ((), (0,)):
    log = []
    def pack(w, taps, cin, cout, mode, dims=None, _e=enabled):
        ops.CONV_ALGO = "auto" if mode in _e else "direct"
        return orig_pack(w, taps, cin, cout, mode, dims)
    def raw(x, wp, bias, residual, mask_src, dims, cin, cout, kz, flags, leak, _e=enabled, _log=log):
        ops.CONV_ALGO = "auto" if (0 if (flags & 8) else 1) in _e else "direct"
        out = orig_raw(x, wp, bias, residual, mask_src, dims, cin, cout, kz, flags, leak)
        _log.append((tuple(dims), cin, cout, flags, ops._use_wino(cin, cout, dims, kz), x.detach().clone(), out.detach().clone()))
        return out
    ops._pack, ops._conv_raw = pack, raw
    ops.reset_variables()
    cfg = default_config(is_3d=True, res_x=16, res_y=24, res_z=16, filters=filters, batch_size=batch, num_samples=1000)
    tr = Trainer(cfg)
    tr.load_variables(p)
    tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    rec[enabled] = log
for a, b in zip(rec[()], rec[(0,)]):
    dims, cin, cout, flags, _, xa, ya = a
    _, _, _, _, wino, xb, yb = b
    d = (ya - yb).abs()
    flips = ((ya > 0) != (yb > 0)).sum().item()
    din = (xa - xb).abs().max().item() / max(xa.abs().max().item(), 1e-30)
    print("dims %s %d->%d flags %d wino %d: in-diff %.1e  out max|d|/max %.2e  mean|d|/mean %.2e  sign flips %d of %d  (nan %d)" % (
        dims, cin, cout, flags, wino, din, d.max().item() / max(ya.abs().max().item(), 1e-30), d.mean().item() / max(ya.abs().mean().item(), 1e-30), flips, ya.numel(),
        torch.isnan(yb).sum().item()))
