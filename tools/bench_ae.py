#!/usr/bin/env python3
"""Diagnostic: AE3 train step at BASELINE cfg5 shape (128^3, filters 64, z_num 16), fp32 -- not the bench metric."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_fluids_amd import ops
from deep_fluids_amd.trainer import AETrainer, default_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = default_config(is_3d=True, res_x=R, res_y=R, res_z=R, filters=64, batch_size=B, num_samples=5000, z_num=16, p_num=2)
tr = AETrainer(cfg)
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.rand((B, 2, 10), device="cuda", generator=g) * 2 - 1
x = ops.curl3(torch.rand((B, R, R, R, 3), device="cuda", generator=g) * 2 - 1)
x = (x / x.abs().max()).contiguous()
for _ in range(2):
    tr.train_step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n):
    m = tr.train_step(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("cfg5 AE3 %d^3 F=64 B=%d: %.1f ms/step, %.2f Mvox/s, loss %.5f, params %d, peak mem %.1f GiB" % (
    R, B, dt * 1e3, B * R ** 3 / dt / 1e6, float(m.g_loss.detach()), tr.n_params, torch.cuda.max_memory_allocated() / 2 ** 30))
