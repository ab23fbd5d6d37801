#!/usr/bin/env python3
"""Per-variable gradient error of one F=128 3-D train step vs the fp64 oracle, direct vs Winograd convs (gpurun aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
p64 = {k: v.astype(np.float64) for k, v in p.items()}
opt = {"m": {k: np.zeros_like(v) for k, v in p64.items()}, "v": {k: np.zeros_like(v) for k, v in p64.items()}, "t": 0, "lr": 1e-4}
_, _, info = orc.train_step(y.astype(np.float64), x.astype(np.float64), p64, opt, oshape, filters, True)
gmax = max(np.abs(v).max() for v in info["grads"].values())
res = {}
for algo in ("direct", "auto"):
    ops.reset_variables()
    ops.CONV_ALGO = algo
    cfg = default_config(is_3d=True, res_x=16, res_y=24, res_z=16, filters=filters, batch_size=batch, num_samples=1000)
    tr = Trainer(cfg)
    tr.load_variables(p)
    m = tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    res[algo] = tr.grads_numpy()
    u = m.G_.detach().cpu().numpy().astype(np.float64)
    print(algo, "velocity rel-l1", np.abs(u - info["u"]).sum() / np.abs(info["u"]).sum() if "u" in info else "n/a", "loss", float(m.g_loss))
print("%-28s %10s | %10s %10s | %10s %10s" % ("variable", "|g|max/gmax", "dir linf", "dir l2", "wino linf", "wino l2"))
for k in sorted(info["grads"], key=lambda k: (int(k.split("/")[1].split("_")[0]), k)):
    g = info["grads"][k]
    row = [np.abs(g).max() / gmax]
    for algo in ("direct", "auto"):
        d = res[algo][k] - g
        row += [np.abs(d).max() / max(np.abs(g).max(), 1e-30), np.sqrt((d ** 2).sum() / max((g ** 2).sum(), 1e-300))]
    print("%-28s %10.2e | %10.2e %10.2e | %10.2e %10.2e" % ((k,) + tuple(row)))

# ---- localise: Winograd only in the forward convs / only in the dgrads
orig_pack, orig_raw = ops._pack, ops._conv_raw
for enabled in ({0}, {1}):
    def pack(w, taps, cin, cout, mode, dims=None, _e=enabled):
        ops.CONV_ALGO = "auto" if mode in _e else "direct"
        return orig_pack(w, taps, cin, cout, mode, dims)
    def raw(x, wp, bias, residual, mask_src, dims, cin, cout, kz, flags, leak, _e=enabled):
        ops.CONV_ALGO = "auto" if (0 if (flags & 8) else 1) in _e else "direct"
        return orig_raw(x, wp, bias, residual, mask_src, dims, cin, cout, kz, flags, leak)
    ops._pack, ops._conv_raw = pack, raw
    ops.reset_variables()
    cfg = default_config(is_3d=True, res_x=16, res_y=24, res_z=16, filters=filters, batch_size=batch, num_samples=1000)
    tr = Trainer(cfg)
    tr.load_variables(p)
    m = tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    gr = tr.grads_numpy()
    print("winograd in modes", enabled, {k: "%.1e" % (np.sqrt(((gr[k] - info["grads"][k]) ** 2).sum() / (info["grads"][k] ** 2).sum()))
                                          for k in ("G/8_conv/biases", "G/6_conv/biases", "G/4_conv/biases", "G/0_fc/weights")})
