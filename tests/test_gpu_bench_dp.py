"""-m gpu: bench.py's N > 1 data-parallel path for EVERY BASELINE workload (--config cfg2 | cfg3 | cfg4 | cfg5), executed with two
ranks.  A gpurun box has one GPU, so the ranks share it and exchange over gloo (DF_DIST_BACKEND=gloo: `rccl_ranks` is 0 by
construction, `counted_ranks` 2); on a multi-GPU node the same commands run over RCCL.  Checked on the printed line: the contract
fields, both scaling legs, and the cross-rank check of every leg -- reduced flat gradient bit-identical on all ranks, mean of the
shard losses == one process's loss on the gathered global batch to 1e-6 (reference graph: trainer.py:136-184, trainer3.py:14-63,
240-309)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, tmp_path, timeout=1500):
    import torch
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["DF_DIST_BACKEND"] = "gloo"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    side = str(tmp_path / "side.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--other-steps", "2", "--sidecar", side] + args
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-4000:])
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0]), json.load(open(side))


def _check_leg(leg, world=2):
    assert leg["ms_per_step"] > 0 and leg["value"] > 0 and leg["loss"] == leg["loss"]
    cr = leg["cross_rank"]
    assert cr["ok"] is True and cr["grad_identical_on_all_ranks"] is True and cr["loss_rel_diff"] <= 1e-6, cr
    o = leg["other_scaling_leg"]
    if o is not None:
        assert o["scaling"] != leg["scaling"] and o["global_batch"] == o["batch_per_gpu"] * world and o["ms_per_step"] > 0


@pytest.mark.parametrize("config,args,unit", [
    ("cfg2", ["--res", "32", "24", "--batch", "4", "--filters", "32"], "pixels/s"),
    ("cfg4", ["--res", "28", "40", "28", "--batch", "2", "--filters", "32"], "voxels/s"),          # cfg4's odd geometry (x0 = 7x10x7) at a quarter grid
    ("cfg5", ["--res", "16", "16", "16", "--batch", "2", "--filters", "16"], "voxels/s"),
])
def test_bench_two_ranks_per_config(tmp_path, config, args, unit):
    line, full = _run(["--config", config] + args, tmp_path)
    assert line["n_gpus"] == 2 and line["counted_ranks"] == 2 and line["unit"] == unit and line["config"]["parallelism"] == "dp2"
    assert line["config"]["workload"].startswith(config) and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["config"]["global_batch"] == 2 * line["config"]["batch_per_gpu"]
    if line["dist_backend"] == "gloo":
        assert line["rccl_ranks"] == 0 and line["distinct_devices"] == 1
    else:
        assert line["rccl_ranks"] == 2 and line["distinct_devices"] == 2 and line["allreduce"]["comm_span_ms"] > 0
    assert line["l1_vs_ref"]["value"] <= 1e-4
    _check_leg(dict(line, scaling=line["scaling"]))
    assert line["other_scaling_leg"] is not None and line["other_scaling_leg"]["scaling"] == "strong"


def test_bench_two_ranks_cfg3_line_carries_cfg2_and_cfg4_legs(tmp_path):
    """The metric's own config at N > 1: ONE invocation yields cfg3 AND short data-parallel legs of the 2-D workload (128x96, batch 64 per
    GPU) and of cfg4's grid (112x160x112, batch 4 per GPU) at their BASELINE sizes, each with its cross-rank check."""
    line, full = _run(["--res", "16", "24", "16", "--batch", "2", "--extra-leg-steps", "2"], tmp_path)
    assert line["config"]["workload"].startswith("cfg3") and line["n_gpus"] == 2
    _check_leg(line)
    for name, grid, per in (("cfg2", [128, 96], 64), ("cfg4", [112, 160, 112], 4)):
        leg = full["extra_leg_" + name]
        assert "error" not in leg, leg
        assert leg["grid"] == grid and leg["batch_per_gpu"] == per and leg["global_batch"] == 2 * per
        _check_leg(leg)
        assert line["extra_leg_" + name]["cross_rank"]["ok"] is True
