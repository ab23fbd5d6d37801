"""Rank process of tests/test_gpu_dist.py: the REAL Trainer / GANTrainer with ``enable_data_parallel()`` and world > 1.

Launched by ``python -m torch.distributed.run --nproc-per-node N tests/dp_worker.py <case> <out.npz>``; with
DF_DIST_BACKEND=gloo the ranks may share one GPU (single-GPU boxes).  Every rank takes its shard of the SAME global batch
(``shard_batch``), runs ``steps`` train steps and rank 0 stores the scaled reduced gradients of the first step, the final
parameters and the bucket launch order."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_case(case):
    """-> (trainer class name, config kwargs, global batch, steps); shared with the single-process reference run."""
    if case == "de3_cfg4geom":      # cfg4's odd geometry at a quarter of the grid, F = 128 (default dispatch)
        return "Trainer", dict(is_3d=True, res_x=28, res_y=40, res_z=28, filters=128, num_samples=1000), 4, 2
    if case == "de2":
        return "Trainer", dict(is_3d=False, res_x=32, res_y=64, filters=32, num_samples=1000), 4, 2
    if case == "dg2":
        return "GANTrainer", dict(is_3d=False, res_x=32, res_y=64, filters=32, num_samples=1000, arch="dg"), 4, 2
    raise ValueError(case)


def make_batch(cfg_kw, global_batch, seed=5):
    import df_oracle as orc
    rng = np.random.RandomState(seed)
    spatial = ([cfg_kw["res_z"]] if cfg_kw["is_3d"] else []) + [cfg_kw["res_y"], cfg_kw["res_x"]]
    x, y = orc.synthetic_batch(rng, global_batch, tuple(spatial))
    return x, y


def run(case, world, rank, force_sync=False):
    from deep_fluids_amd import ops, trainer as T
    from deep_fluids_amd.dist import shard_batch
    name, kw, gb, steps = make_case(case)
    lo, n = shard_batch(gb, rank, world)
    graph = os.environ.get("DF_TEST_GRAPH") == "1"          # tests/test_gpu_graph.py: Trainer(graph=True) under data parallelism
    if graph:
        steps = 4                                           # eager warm-up, capture + replay, two more replays
    cfg = T.default_config(batch_size=gb, graph=graph, **kw)            # max_step (the cosine period) follows the GLOBAL batch
    ops.reset_variables()
    tr = getattr(T, name)(cfg)                              # same seed on every rank -> identical initial variables
    if world > 1:
        tr.enable_data_parallel()
    elif force_sync:
        tr.enable_data_parallel(profile=True, force=True)
    x, y = make_batch(kw, gb)
    xs = torch.from_numpy(x[lo:lo + n]).cuda(); ys = torch.from_numpy(y[lo:lo + n]).cuda()
    out = {"p_init": tr.flat_p.cpu().numpy()}
    for s in range(steps):
        m = tr.train_step(xs, ys)
        if s == 0:
            scale = 1.0 / world
            out["g0"] = (tr.flat_g * scale).cpu().numpy()
            if name == "GANTrainer":
                out["gd0"] = (tr.D.g * scale).cpu().numpy()
            if tr.grad_sync is not None:
                out["launch_order"] = np.asarray(tr.grad_sync.launch_order)
                out["n_buckets"] = np.int64(len(tr.grad_sync.buckets))
    out["p"] = tr.flat_p.cpu().numpy()
    out["n_graphs"] = np.int64(len(tr._graphs))
    if name == "GANTrainer":
        out["pd"] = tr.D.p.cpu().numpy()
    out["loss"] = np.float64(float(m.g_loss.detach()))
    out["logged_loss"] = np.float64(tr._scalars(m, 0.0)["loss/g_loss"])      # what Trainer.train logs: the global-batch mean on EVERY rank
    out["g_lr"] = np.float64(tr.g_lr)
    if dist.is_initialized():
        from deep_fluids_amd.dist import verify_world
        w = verify_world()
        out["counted_ranks"] = np.int64(w["ranks"]); out["distinct_devices"] = np.int64(w["distinct_devices"])
    if force_sync:
        t = tr.grad_sync.timing()
        out["comm_span_ms"] = np.float64(t["comm_span_ms"]); out["exposed_ms"] = np.float64(t["exposed_ms"])
        out["timed_steps"] = np.int64(t["steps"])
    return out


if __name__ == "__main__":
    from deep_fluids_amd.dist import init_from_env
    case, path = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 3 and sys.argv[3] == "rccl1":
        # ONE rank over RCCL ("nccl"), exchange forced on: the production communication path on a single-GPU box
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        res = run(case, 1, 0, force_sync=True)
        res["backend"] = np.asarray(dist.get_backend())
        np.savez(path, **res)
        dist.destroy_process_group()
        sys.exit(0)
    rank, local_rank, world = init_from_env()
    res = run(case, world, rank)
    if world > 1:
        # every rank must hold identical parameters after the exchange
        t = torch.from_numpy(res["p"]).double().cuda()
        lo = t.clone(); hi = t.clone()
        if dist.get_backend() == "gloo":
            lo = lo.cpu(); hi = hi.cpu()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        res["rank_spread"] = np.float64(float((hi - lo).abs().max()))
        lt = torch.tensor([float(res["loss"])], dtype=torch.float64)          # mean of the equal-sized shard losses == global loss
        if dist.get_backend() != "gloo":
            lt = lt.cuda()
        dist.all_reduce(lt)
        res["loss"] = np.float64(float(lt.item()) / world)
        dist.barrier()
    if rank == 0:
        np.savez(path, **res)
    if world > 1:
        dist.destroy_process_group()
