#!/usr/bin/env python3
"""Diagnostic: cfg3 train step in the opt-in bf16x3 conv precision mode vs exact fp32 (timing + velocity error)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from deep_fluids_amd import ops
from deep_fluids_amd.trainer import Trainer, default_config
import df_oracle as orc

def step_time(mode, B=16, res=(64, 96, 64), n=4):
    ops.CONV_PRECISION = mode
    ops.reset_variables()
    cfg = default_config(is_3d=True, res_x=res[2], res_y=res[1], res_z=res[0], filters=128, batch_size=B, num_samples=6600)
    tr = Trainer(cfg)
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.rand((B, 3), device="cuda", generator=g) * 2 - 1
    x = ops.curl3(torch.rand((B,) + tuple(res) + (3,), device="cuda", generator=g) * 2 - 1)
    x = (x / x.abs().max()).contiguous()
    for _ in range(2):
        tr.train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        m = tr.train_step(x, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, float(m.g_loss.detach())

def velocity_err(mode):
    ops.CONV_PRECISION = mode
    ops.reset_variables()
    rng = np.random.RandomState(123)
    spatial = (16, 24, 16); oshape = list(spatial) + [3]
    p = orc.generator_init(rng, 3, oshape, 128)
    x, y = orc.synthetic_batch(rng, 1, spatial)
    cfg = default_config(is_3d=True, res_x=16, res_y=24, res_z=16, filters=128, batch_size=1, num_samples=100)
    tr = Trainer(cfg); tr.load_variables(p)
    u = tr.generate(torch.from_numpy(y).cuda()).cpu().numpy().astype(np.float64)
    ref = orc.curl3(orc.generator_fwd(y.astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()}, oshape, 128))
    return float(np.abs(u - ref).sum() / np.abs(ref).sum())

for mode in ("fp32", "bf16x3"):
    e = velocity_err(mode)
    t, loss = step_time(mode)
    print("%-7s  velocity rel-L1 vs fp64 oracle %.2e   cfg3 step %.1f ms  (%.2f Mvox/s)  loss %.6f" % (mode, e, t * 1e3, 16 * 64 * 96 * 64 / t / 1e6, loss))
ops.CONV_PRECISION = "fp32"
