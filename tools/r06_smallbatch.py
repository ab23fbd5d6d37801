#!/usr/bin/env python3
"""Round 6: the reference's DEFAULT-batch workloads (config.py:40 batch_size=8; run.bat:13,56; 3-D B = 1 = the per-GPU share of cfg3 on 8 GPUs)
-- host time vs kernel time per step, eager vs Trainer(graph=True).

  r06_smallbatch.py time  [case ...]      wall ms/step eager and graph (same process, same inputs) -> one JSON object per case
  r06_smallbatch.py trace <case> <mode>   exactly WARM + STEPS steps in one mode, for `rocprofv3 --kernel-trace` (tools/summarize_trace.py --steps)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

CASES = {  # name: (kind, grid, batch, filters)
    "2d_b8": ("de", [128, 96], 8, 128),            # run.bat:13 smoke_pos_size, config.py:40
    "2d_b64": ("de", [128, 96], 64, 128),          # BASELINE cfg2
    "ae2d_b8": ("ae", [128, 96], 8, 64),           # run.bat:56 smoke_mov
    "dg2d_b8": ("dg", [128, 96], 8, 128),          # README's 2-D command (--arch=dg)
    "3d_b1": ("de", [64, 96, 64], 1, 128),         # cfg3's per-GPU share at 16 GPUs / strong-scaling floor
    "3d_b2": ("de", [64, 96, 64], 2, 128),         # cfg3's per-GPU share on 8 GPUs
    "3d_ref_b4": ("de", [32, 64, 112], 4, 128),    # run.bat:21 smoke3_vel_buo
    "ae3d_b4": ("ae", [48, 72, 48], 4, 64),        # run.bat:73
}
WARM, STEPS = 3, 20


def make(case, graph):
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, AETrainer, GANTrainer, default_config
    sys.path.insert(0, ROOT)
    import bench
    kind, grid, B, F = CASES[case]
    is3 = len(grid) == 3
    ops.reset_variables()
    kw = dict(is_3d=is3, res_x=grid[-1], res_y=grid[-2], res_z=grid[0] if is3 else 1, filters=F, batch_size=B, num_samples=6000, graph=graph)
    x, y = bench.make_inputs(B, grid, 3, ops)
    if kind == "ae":
        tr = AETrainer(default_config(z_num=16, p_num=2 if is3 else 1, **kw))
        y = torch.from_numpy(np.random.RandomState(4).uniform(-1, 1, (B, 2 if is3 else 1, 10)).astype(np.float32)).cuda()
    elif kind == "dg":
        tr = GANTrainer(default_config(arch="dg", **kw))
    else:
        tr = Trainer(default_config(**kw))
    return tr, x, y


def wall(tr, x, y, warm, n):
    for _ in range(warm):
        tr.train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(x, y)
    t1 = time.perf_counter()          # host done issuing
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3


def main():
    mode = sys.argv[1]
    if mode == "trace":
        case, m = sys.argv[2], sys.argv[3]
        tr, x, y = make(case, m == "graph")
        w, h = wall(tr, x, y, WARM, STEPS)
        print(json.dumps({"case": case, "mode": m, "steps_total": WARM + STEPS, "ms_per_step_under_trace": w}))
        return
    for case in (sys.argv[2:] or ["2d_b8", "ae2d_b8", "dg2d_b8", "3d_b1", "3d_b2", "2d_b64"]):
        rec = {"case": case, "workload": dict(zip(("arch", "grid", "batch", "filters"), CASES[case]))}
        for m in ("eager", "graph"):
            tr, x, y = make(case, m == "graph")
            w, h = wall(tr, x, y, WARM, STEPS)
            w2, h2 = wall(tr, x, y, 0, STEPS)
            rec[m] = {"ms_per_step": min(w, w2), "host_issue_ms_per_step": min(h, h2)}
            del tr
            torch.cuda.empty_cache()
        rec["speedup"] = rec["eager"]["ms_per_step"] / rec["graph"]["ms_per_step"]
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
