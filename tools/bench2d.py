#!/usr/bin/env python3
"""Diagnostic: the 2-D train step at BASELINE cfg2 shape (128x96, batch 64, fp32) -- not the bench metric."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_fluids_amd import ops
from deep_fluids_amd.trainer import Trainer, default_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = default_config(is_3d=False, res_x=96, res_y=128, filters=128, batch_size=B, num_samples=21000)
tr = Trainer(cfg)
g = torch.Generator(device="cuda").manual_seed(1)
y = torch.rand((B, 3), device="cuda", generator=g) * 2 - 1
x = ops.curl(torch.rand((B, 128, 96, 1), device="cuda", generator=g) * 2 - 1)
x = (x / x.abs().max()).contiguous()
for _ in range(3):
    tr.train_step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n):
    m = tr.train_step(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
flops = 3.71e12 * B / 64
print("cfg2 2D 128x96 B=%d: %.2f ms/step, %.2f Mpx/s, conv %.1f TFLOP/s end-to-end, loss %.5f, params %d" % (B, dt * 1e3, B * 128 * 96 / dt / 1e6, flops / dt / 1e12, float(m.g_loss.detach()), tr.n_params))
