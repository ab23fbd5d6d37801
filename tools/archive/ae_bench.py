#!/usr/bin/env python3
"""BASELINE cfg5's shape (AE3 encoder + decoder train step, 128^3, F = 64, batch 4) on one GPU: ms/step + the dispatch log (gpurun aid;
bench.py reports the same workload as `extra_ae_cfg5`)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import ops  # noqa: E402
from deep_fluids_amd.trainer import AETrainer, default_config  # noqa: E402

B5, R = 4, 128
cfg5 = default_config(is_3d=True, res_x=R, res_y=R, res_z=R, filters=64, batch_size=B5, num_samples=5000, z_num=16, p_num=2)
tr = AETrainer(cfg5)
g5 = torch.Generator(device="cuda").manual_seed(1)
y5 = torch.rand((B5, 2, 10), device="cuda", generator=g5) * 2 - 1
x5 = ops.curl3(torch.rand((B5, R, R, R, 3), device="cuda", generator=g5) * 2 - 1)
x5 = (x5 / x5.abs().max()).contiguous()
for _ in range(2):
    tr.train_step(x5, y5)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    tr.train_step(x5, y5)
torch.cuda.synchronize()
print("AE3 128^3 B=4 F=64: %.1f ms/step" % ((time.perf_counter() - t0) / 3 * 1e3))
if "--dispatch" in sys.argv:
    ops.DISPATCH_COUNTS = {}
    tr.train_step(x5, y5)
    for k, v in sorted(ops.DISPATCH_COUNTS.items()):
        print(v, k)
