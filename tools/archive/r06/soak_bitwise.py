"""Soak: N train steps eager (weight-gradient lane on a second stream, direct gradient targets, sign words) vs Trainer(graph=True) (serial order) on
the reference's default-batch 2-D workload and a 3-D one, different batch every step: parameters must stay bitwise equal; loss finite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from deep_fluids_amd import ops
from deep_fluids_amd.trainer import Trainer, AETrainer, default_config
N = int(os.environ.get("N", "300"))
CASES = {"2d_b8": (Trainer, dict(is_3d=False, res_x=96, res_y=128, filters=128), 8, (128, 96)),
         "3d_b2": (Trainer, dict(is_3d=True, res_x=32, res_y=48, res_z=32, filters=128), 2, (32, 48, 32)),
         "ae2d_b8": (AETrainer, dict(is_3d=False, res_x=96, res_y=128, filters=64, z_num=16, p_num=1), 8, (128, 96))}
for name, (cls, kw, B, sp) in CASES.items():
    res = []
    for graph in (False, True):
        ops.reset_variables()
        tr = cls(default_config(batch_size=B, num_samples=B * 4000, graph=graph, **kw))
        g = torch.Generator(device="cuda").manual_seed(5)
        losses = []
        for step in range(N):
            nc = 1 if len(sp) == 2 else 3
            x = torch.rand((B,) + sp + ((2 if len(sp) == 2 else 3),), device="cuda", generator=g) * 2 - 1
            y = torch.rand((B, 3), device="cuda", generator=g) * 2 - 1
            if cls is AETrainer:
                y = torch.rand((B, 1, 10), device="cuda", generator=g) * 2 - 1
            m = tr.train_step(x, y)
            if step % 50 == 0 or step == N - 1:
                losses.append(float(m.g_loss.detach()))
        res.append((tr.flat_p.cpu().numpy().copy(), losses))
        del tr
    same = np.array_equal(res[0][0], res[1][0])
    print(name, "steps", N, "params bitwise equal eager vs graph:", same, "losses eager", [round(v, 5) for v in res[0][1]], "finite", bool(np.isfinite(res[0][0]).all()), flush=True)
    assert same and np.isfinite(res[0][0]).all()
