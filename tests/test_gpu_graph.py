"""-m gpu: ``Trainer(graph=True)`` -- the whole train step (gradient clear, forward, loss tail, autograd backward, optimizer) captured once
into a hipGraph and replayed, the counterpart of ``sess.run(g_optim)`` on the reference's pre-built TF graph (trainer.py:265-269).

The bar is BITWISE equality with the eager step: the same kernels with the same arguments in the same order; only the optimizer's
per-step scalars (lr_t, grad_scale) come from device memory instead of kernel arguments.  Every case runs 5 steps with a DIFFERENT
batch per step (step 1 eager warm-up, step 2 capture + replay, steps 3-5 replay) so that stale inputs, a stale learning rate or stale beta
powers cannot go unnoticed, over the cosine schedule with a short period (g_lr changes visibly from step to step)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import df_oracle as orc
from gpu_util import dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # name: (trainer class, config kwargs, batch)
    "de2": ("Trainer", dict(is_3d=False, res_x=32, res_y=64, filters=32), 4),
    "de2_128x96_f128": ("Trainer", dict(is_3d=False, res_x=96, res_y=128, filters=128), 8),       # run.bat:13, config.py:40 (B = 8)
    "de3": ("Trainer", dict(is_3d=True, res_x=16, res_y=24, res_z=16, filters=32), 2),
    "de3_f128": ("Trainer", dict(is_3d=True, res_x=16, res_y=24, res_z=16, filters=128), 1),      # Winograd / sign-bit dispatch
    "de3_nocurl": ("Trainer", dict(is_3d=True, res_x=16, res_y=24, res_z=16, filters=32, use_curl=False), 2),
    "de3_unfused_tail": ("Trainer", dict(is_3d=True, res_x=16, res_y=24, res_z=16, filters=32, fused_tail=False), 2),
    "de2_gd_steplr": ("Trainer", dict(is_3d=False, res_x=32, res_y=64, filters=32, optimizer="gd", lr_update="step", lr_update_step=2), 4),
    "ae2": ("AETrainer", dict(is_3d=False, res_x=32, res_y=64, filters=32, z_num=16, p_num=1), 4),
    "ae3_sparse": ("AETrainer", dict(is_3d=True, res_x=16, res_y=16, res_z=16, filters=32, z_num=16, p_num=2, use_sparse=True), 2),
    "dg2": ("GANTrainer", dict(is_3d=False, res_x=32, res_y=64, filters=32, arch="dg"), 4),
}


def _batches(name, kw, B, n):
    rng = np.random.RandomState(11)
    spatial = ([kw["res_z"]] if kw["is_3d"] else []) + [kw["res_y"], kw["res_x"]]
    out = []
    for _ in range(n):
        x, y = orc.synthetic_batch(rng, B, tuple(spatial))
        if name.startswith("ae"):
            y = rng.uniform(-1, 1, (B, kw["p_num"], 6)).astype(np.float32)
        out.append((dev(x), dev(y)))
    return out


def _run(name, graph, steps=5, **extra):
    from deep_fluids_amd import ops, trainer as T
    cls, kw, B = CASES[name]
    ops.reset_variables()
    cfg = T.default_config(batch_size=B, num_samples=B * 8, max_epoch=1, graph=graph, **dict(kw, **extra))      # max_step = 8: a visible cosine
    tr = getattr(T, cls)(cfg)
    losses = []
    for x, y in _batches(name, kw, B, steps):
        m = tr.train_step(x, y)
        losses.append([float(m.g_loss.detach()), float(m.g_loss_l1.detach()), float(m.g_loss_j_l1.detach())])
    out = {"p": tr.flat_p.cpu().numpy(), "m": tr.flat_m.cpu().numpy(), "v": tr.flat_v.cpu().numpy(), "g": tr.flat_g.cpu().numpy(),
           "loss": np.asarray(losses), "lr": tr.g_lr, "step": tr.step, "t": tr._adam_t, "u": m.G_.detach().cpu().numpy()}
    if cls == "GANTrainer":
        out["pd"] = tr.D.p.cpu().numpy(); out["td"] = tr._adam_t_d
    out["n_graphs"] = len(tr._graphs)
    del tr
    ops.reset_variables()
    return out


@pytest.mark.parametrize("name", sorted(CASES))
def test_graph_replay_is_bitwise_the_eager_step(name):
    a = _run(name, False)
    b = _run(name, True)
    assert a["n_graphs"] == 0 and b["n_graphs"] == 1
    assert a["step"] == b["step"] == 5 and a["t"] == b["t"] and a["lr"] == b["lr"]
    np.testing.assert_array_equal(a["loss"], b["loss"])
    for k in ("p", "m", "v", "g", "u") + (("pd",) if "pd" in a else ()):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    if "td" in a:
        assert a["td"] == b["td"]
    assert np.isfinite(a["loss"]).all() and np.abs(a["p"]).sum() > 0


def test_graph_is_keyed_by_input_shape_and_survives_a_batch_size_change():
    """A second batch size gets its own warm-up + capture; returning to the first replays the first graph; parameters stay bitwise
    those of the eager run through the same sequence."""
    from deep_fluids_amd import ops, trainer as T
    kw = dict(is_3d=False, res_x=32, res_y=64, filters=32)
    rng = np.random.RandomState(3)
    seq = []
    for B in (4, 4, 4, 2, 2, 2, 4):
        x, y = orc.synthetic_batch(rng, B, (64, 32))
        seq.append((dev(x), dev(y)))
    res = []
    for graph in (False, True):
        ops.reset_variables()
        tr = T.Trainer(T.default_config(batch_size=4, num_samples=64, max_epoch=1, graph=graph, **kw))
        for x, y in seq:
            tr.train_step(x, y)
        res.append((tr.flat_p.cpu().numpy(), len(tr._graphs)))
        del tr
    ops.reset_variables()
    assert res[0][1] == 0 and res[1][1] == 2
    np.testing.assert_array_equal(res[0][0], res[1][0])


def test_graph_checkpoint_resume_continues_bitwise(tmp_path):
    """save after 3 graph steps, restore into a fresh graph trainer, continue: equals the uninterrupted eager run (the replayed graph reads
    the restored slabs -- they are the SAME storage the capture recorded -- and the restored step / beta powers feed the device scalars)."""
    from deep_fluids_amd import ops, trainer as T
    kw = dict(is_3d=True, res_x=16, res_y=16, res_z=16, filters=32)
    bs = _batches("de3", kw, 2, 6)
    ops.reset_variables()
    ref = T.Trainer(T.default_config(batch_size=2, num_samples=16, max_epoch=1, **kw))
    for x, y in bs:
        ref.train_step(x, y)
    p_ref = ref.flat_p.cpu().numpy()
    ops.reset_variables()
    tr = T.Trainer(T.default_config(batch_size=2, num_samples=16, max_epoch=1, graph=True, **kw))
    for x, y in bs[:3]:
        tr.train_step(x, y)
    ck = str(tmp_path / "model.ckpt-3.npz")
    tr.save(ck)
    ops.reset_variables()
    tr2 = T.Trainer(T.default_config(batch_size=2, num_samples=16, max_epoch=1, graph=True, **kw))
    tr2.load(ck)
    for x, y in bs[3:]:
        tr2.train_step(x, y)
    np.testing.assert_array_equal(p_ref, tr2.flat_p.cpu().numpy())
    # ... and loading INTO a trainer that already holds a captured graph: the graph reads the slabs, so it follows the restore
    tr.load(ck)
    for x, y in bs[3:]:
        tr.train_step(x, y)
    np.testing.assert_array_equal(p_ref, tr.flat_p.cpu().numpy())
    ops.reset_variables()


def test_graph_two_rank_data_parallel(tmp_path):
    """world 2 (ranks share the GPU over gloo on a one-GPU box): forward + backward replayed from the graph, ONE eager all-reduce of the
    flat slab, the optimizer launch.  Both ranks end bit-identical and agree with the single-process run on the global batch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker
    from test_gpu_dist import _free_port
    out = str(tmp_path / "de2_graph.npz")
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["DF_DIST_BACKEND"] = "gloo"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["DF_TEST_GRAPH"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_worker.py"), "de2", out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-4000:]
    dp = dict(np.load(out))
    os.environ["DF_TEST_GRAPH"] = "1"
    try:
        ref = dp_worker.run("de2", 1, 0)
    finally:
        del os.environ["DF_TEST_GRAPH"]
    from deep_fluids_amd import ops
    ops.reset_variables()
    assert dp["rank_spread"] == 0.0 and int(dp["n_graphs"]) == 1 and int(ref["n_graphs"]) == 1
    d_dp, d_ref = dp["p"] - dp["p_init"], ref["p"] - ref["p_init"]
    assert np.abs(d_dp - d_ref).sum() / np.abs(d_ref).sum() < 2e-2
    assert abs(dp["loss"] - ref["loss"]) < 1e-4 * abs(ref["loss"])
    assert abs(dp["g_lr"] - ref["g_lr"]) < 1e-15


@pytest.mark.parametrize("name", ["de2", "de3_f128", "ae2", "dg2"])
def test_concurrent_weight_gradient_lane_is_bitwise_the_serial_order(name):
    """ops.CONCURRENT_WGRAD_WORK: the weight gradients of the small levels run on a second stream next to the dgrad of the same
    layer (fork after the incoming gradient, join at the end of the backward node).  Same kernels and arguments: bitwise equal."""
    from deep_fluids_amd import ops
    with ops.options(concurrent_wgrad_work=0):
        a = _run(name, False, 3)
    with ops.options(concurrent_wgrad_work=1 << 40):
        b = _run(name, False, 3)
    np.testing.assert_array_equal(a["loss"], b["loss"])
    for k in ("p", "m", "v", "g"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("name", ["de2", "de3_f128", "de3_unfused_tail", "ae2", "ae3_sparse"])
@pytest.mark.parametrize("graph", [False, True])
def test_direct_gradient_targets_are_bitwise_the_accumulated_gradients(name, graph):
    """config.direct_grads: the weight / bias gradient kernels write their variable's slice of the flat gradient slab themselves instead of
    returning a tensor that AccumulateGrad adds to the zeroed slab (the reference's optimizer consumes tf.gradients' outputs directly,
    trainer.py:149-152).  0 + g == g: parameters, Adam slots, gradients and losses are bitwise equal, eager and captured."""
    from deep_fluids_amd import ops
    a = _run(name, graph, 3, direct_grads=False)
    assert not ops._DIRECT_GRADS
    b = _run(name, graph, 3, direct_grads=True)
    np.testing.assert_array_equal(a["loss"], b["loss"])
    for k in ("p", "m", "v", "g", "u"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_direct_gradient_targets_are_off_where_a_variable_gets_two_gradients_or_hooks():
    from deep_fluids_amd import ops, trainer as T
    ops.reset_variables()
    tr = T.GANTrainer(T.default_config(is_3d=False, res_x=32, res_y=64, filters=32, arch="dg", batch_size=2))
    assert not ops._DIRECT_GRADS
    del tr
    ops.reset_variables()
    tr = T.Trainer(T.default_config(is_3d=False, res_x=32, res_y=64, filters=32, batch_size=2))
    assert len(ops._DIRECT_GRADS) == len(tr.G_var)
    tr._unregister_direct_grads()
    assert not ops._DIRECT_GRADS
    ops.reset_variables()
