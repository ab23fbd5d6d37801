#!/usr/bin/env python3
"""Diagnostic: the 3-D train step at BASELINE cfg4's grid (112x160x112, filters 128), per-GPU batch 4 -- not the bench metric."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_fluids_amd import ops
from deep_fluids_amd.trainer import Trainer, default_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = default_config(is_3d=True, res_x=112, res_y=160, res_z=112, filters=128, batch_size=B, num_samples=20000)
tr = Trainer(cfg)
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.rand((B, 3), device="cuda", generator=g) * 2 - 1
x = ops.curl3(torch.rand((B, 112, 160, 112, 3), device="cuda", generator=g) * 2 - 1)
x = (x / x.abs().max()).contiguous()
for _ in range(2):
    m = tr.train_step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n):
    m = tr.train_step(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("cfg4 grid 112x160x112 F=128 B=%d: %.1f ms/step, %.2f Mvox/s, loss %.5f, params %d, peak mem %.1f GiB" % (
    B, dt * 1e3, B * 112 * 160 * 112 / dt / 1e6, float(m.g_loss.detach()), tr.n_params, torch.cuda.max_memory_allocated() / 2 ** 30))
