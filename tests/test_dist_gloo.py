"""world_size-2 `gloo` test (CPU) of the data-parallel gradient path (deep_fluids_amd/dist.py): bucketed,
hook-driven all-reduce of the flat gradient slab + the 1/world scale == the gradient of the un-sharded batch
(reduce_mean over the global batch == mean of equal-sized shard means)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deep_fluids_amd.dist import GradSync, shard_batch, verify_world, check_world


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _model_and_slabs(seed=0):
    """A toy 3-bucket model whose parameters are views into one flat slab (exactly the Trainer's arrangement)."""
    g = torch.Generator().manual_seed(seed)
    shapes = [(3, 8), (8,), (8, 8), (8,), (8, 2), (2,)]
    total = sum(int(np.prod(s)) for s in shapes)
    flat_p = torch.randn(total, generator=g)
    flat_g = torch.zeros(total)
    params, off = [], 0
    for s in shapes:
        n = int(np.prod(s))
        v = flat_p[off:off + n].view(s).detach().requires_grad_(True)
        v.grad = flat_g[off:off + n].view(s)
        params.append((off, n, v))
        off += n
    buckets = [(params[0][0], params[0][1] + params[1][1], [params[0][2], params[1][2]]),
               (params[2][0], params[2][1] + params[3][1], [params[2][2], params[3][2]]),
               (params[4][0], params[4][1] + params[5][1], [params[4][2], params[5][2]])]
    return [p[2] for p in params], flat_g, buckets


def _loss(ps, x, y):
    h = torch.nn.functional.leaky_relu(x @ ps[0] + ps[1], 0.2)
    h = torch.nn.functional.leaky_relu(h @ ps[2] + ps[3], 0.2) + h
    return ((h @ ps[4] + ps[5]) - y).abs().mean()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps, flat_g, buckets = _model_and_slabs()
    sync = GradSync(flat_g, buckets)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 3, generator=g); Y = torch.randn(8, 2, generator=g)
    lo, n = shard_batch(8, rank, world)
    for step in range(2):                                   # two steps: hooks / counters must re-arm
        flat_g.zero_()
        sync.begin_step()
        _loss(ps, X[lo:lo + n], Y[lo:lo + n]).backward()
        scale = sync.finish()
    out[rank] = (flat_g * scale).clone().numpy()
    out["world%d" % rank] = verify_world()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_allreduce_matches_full_batch(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ps, flat_g, _ = _model_and_slabs()
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 3, generator=g); Y = torch.randn(8, 2, generator=g)
    _loss(ps, X, Y).backward()
    ref = flat_g.numpy()
    np.testing.assert_allclose(out[0], ref, rtol=1e-5, atol=1e-7)
    for r in range(1, world):
        np.testing.assert_allclose(out[r], out[0], rtol=0, atol=0)  # every rank holds identical reduced gradients
    for r in range(world):      # bench.py's `rccl_ranks`: counted by an all-reduce over the backend, not read from the environment
        assert out["world%d" % r]["ranks"] == world and out["world%d" % r]["backend"] == "gloo" and len(out["world%d" % r]["devices"]) == world


def _gan_losses(pg, pd, x, y):
    """Generator loss through the discriminator + discriminator loss (the two `minimize` calls of arch='dg', trainer.py:174-184)."""
    fake = _loss_head(pg, x)
    d_fake = torch.nn.functional.leaky_relu(fake @ pd[0] + pd[1], 0.2) @ pd[2] + pd[3]
    d_real = torch.nn.functional.leaky_relu(y @ pd[0] + pd[1], 0.2) @ pd[2] + pd[3]
    g_loss = (fake - y).abs().mean() + 0.005 * ((d_fake - 1) ** 2).mean()
    d_loss = ((d_real - 1) ** 2).mean() + (d_fake ** 2).mean()
    return g_loss, d_loss


def _loss_head(ps, x):
    h = torch.nn.functional.leaky_relu(x @ ps[0] + ps[1], 0.2)
    h = torch.nn.functional.leaky_relu(h @ ps[2] + ps[3], 0.2) + h
    return h @ ps[4] + ps[5]


def _d_slab(seed=3):
    g = torch.Generator().manual_seed(seed)
    shapes = [(2, 6), (6,), (6, 1), (1,)]
    total = sum(int(np.prod(s)) for s in shapes)
    flat_p = torch.randn(total, generator=g)
    flat_g = torch.zeros(total)
    params, off = [], 0
    for s in shapes:
        n = int(np.prod(s))
        v = flat_p[off:off + n].view(s).detach().requires_grad_(True)
        v.grad = flat_g[off:off + n].view(s)
        params.append(v)
        off += n
    return params, flat_g, [(0, total, params)]


def _gan_step(pg, pd, X, Y):
    """GANTrainer.train_step's two backward passes on disjoint variable lists (each accumulates into its own slab only)."""
    g_loss, d_loss = _gan_losses(pg, pd, X, Y)
    g_loss.backward(inputs=pg, retain_graph=True)
    d_loss.backward(inputs=pd)


def _gan_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pg, flat_g, bg = _model_and_slabs()
    pd, flat_d, bd = _d_slab()
    sync_g, sync_d = GradSync(flat_g, bg), GradSync(flat_d, bd)       # two slabs -> two exchanges (GANTrainer.enable_data_parallel)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 3, generator=g); Y = torch.randn(8, 2, generator=g)
    lo, n = shard_batch(8, rank, world)
    for step in range(2):
        flat_g.zero_(); flat_d.zero_()
        sync_g.begin_step(); sync_d.begin_step()
        _gan_step(pg, pd, X[lo:lo + n], Y[lo:lo + n])
        sg = sync_g.finish(); sd = sync_d.finish()
    out[rank] = ((flat_g * sg).clone().numpy(), (flat_d * sd).clone().numpy(), list(sync_g.launch_order), list(sync_d.launch_order))
    dist.destroy_process_group()


def test_two_slab_gan_exchange_world4_matches_full_batch():
    """arch='dg' under data parallelism: the generator's bucketed slab and the discriminator's one-bucket slab are reduced by two
    GradSync objects fed by two backward passes over disjoint variable lists; 4 ranks x 2 samples == the un-sharded batch of 8."""
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gan_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    pg, flat_g, _ = _model_and_slabs()
    pd, flat_d, _ = _d_slab()
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 3, generator=g); Y = torch.randn(8, 2, generator=g)
    _gan_step(pg, pd, X, Y)
    np.testing.assert_allclose(out[0][0], flat_g.numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(out[0][1], flat_d.numpy(), rtol=2e-5, atol=1e-7)
    for r in range(1, world):
        np.testing.assert_array_equal(out[r][0], out[0][0]); np.testing.assert_array_equal(out[r][1], out[0][1])
    assert sorted(out[0][2]) == [0, 1, 2] and out[0][2][0] == 2 and out[0][3] == [0]       # last layer's bucket first; D's single bucket once


def test_shard_batch_and_single_process_passthrough():
    assert shard_batch(16, 3, 8) == (6, 2)
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)
    ps, flat_g, buckets = _model_and_slabs()
    sync = GradSync(flat_g, buckets)           # no process group: disabled, scale 1
    assert not sync.enabled and sync.finish() == 1.0


def test_check_world_refuses_rccl_on_shared_devices():
    ok = check_world("nccl", 2, 2, ["0:5:0", "0:6:0"])
    assert ok == {"ranks": 2, "backend": "nccl", "devices": ["0:5:0", "0:6:0"], "distinct_devices": 2}
    with pytest.raises(RuntimeError, match="one rank per GPU"):
        check_world("nccl", 2, 2, ["0:5:0", "0:5:0"])          # two RCCL ranks on one physical GPU: not a 2-GPU measurement
    assert check_world("gloo", 2, 2, ["0:5:0", "0:5:0"])["distinct_devices"] == 1      # the single-GPU sharing TEST configuration
    with pytest.raises(RuntimeError, match="returned 1"):
        check_world("nccl", 2, 1, ["a", "b"])
    assert verify_world() == {"ranks": 1, "backend": None, "devices": [verify_world()["devices"][0]], "distinct_devices": 1}


def _state_worker(rank, world, port, out):
    from deep_fluids_amd.dist import broadcast_trainer_state, all_equal_across_ranks
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 0 "restored a checkpoint" (step 7, decayed lr, trained slabs); the other ranks did not see the file: fresh init, step 0
    g = torch.Generator().manual_seed(100 + rank)
    slabs = [torch.randn(37, generator=g) for _ in range(3)]
    scalars = [7, 3.25e-5, 7] if rank == 0 else [0, 1e-4, 0]
    before = all_equal_across_ranks([float(s.double().sum()) for s in slabs])
    got = broadcast_trainer_state(slabs, scalars)
    after = all_equal_across_ranks([float(s.double().sum()) for s in slabs] + got)
    out[rank] = ([s.clone().numpy() for s in slabs], got, before, after)
    dist.destroy_process_group()


def test_state_broadcast_makes_every_rank_continue_from_rank0():
    """restore-on-start is per rank but only rank 0 writes checkpoints; enable_data_parallel() broadcasts rank 0's
    parameters / Adam slots / step / g_lr / Adam step count so that no rank trains a divergent replica or runs a different loop length."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_state_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    g = torch.Generator().manual_seed(100)
    ref = [torch.randn(37, generator=g).numpy() for _ in range(3)]
    for r in range(world):
        slabs, got, before, after = out[r]
        for a, b in zip(slabs, ref):
            np.testing.assert_array_equal(a, b)
        assert got == [7.0, 3.25e-5, 7.0]
        assert before is False and after is True
