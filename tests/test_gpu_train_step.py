"""-m gpu: the generator and the full velocity-field train step (HIP path through ops/model/trainer) against
(a) the golden vectors produced by running the reference's own model.py and (b) the fp64 oracle.
North-star tolerance: relative L1 on the velocity field <= 1e-4."""
import json
import os

import numpy as np
import pytest
import torch

import df_oracle as orc
from conftest import GOLDEN
from gpu_util import dev, host, rel_l1, rel_linf

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["g2_small", "g3_small", "g3_odd", "g2_skip", "g3_skip"])
def test_generator_vs_reference_model_py(golden_generators, tag):
    from deep_fluids_amd import ops
    from deep_fluids_amd.model import GeneratorBE, GeneratorBE3
    g = golden_generators
    pl = json.load(open(os.path.join(GOLDEN, "layer_plans.json")))[tag]
    ops.reset_variables()
    for k, v in g.items():
        if k.startswith(tag + "|"):
            ops.set_variable(k.split("|", 1)[1], v)
    gen = GeneratorBE3 if pl["fn"] == "GeneratorBE3" else GeneratorBE
    out, variables = gen(dev(g[tag + "_z"]), pl["filters"], pl["output_shape"], skip_concat=pl.get("skip_concat", False), reuse=True)
    assert len(variables) == len(pl["variables"])
    assert rel_linf(host(out), g[tag + "_out"]) < 2e-5
    assert rel_l1(host(out), g[tag + "_out"]) < 1e-5
    ops.reset_variables()


def test_generator_skip_concat_gradients_vs_torch_oracle():
    """skip_concat=True (model.py:30-33 / :72-75): the concat-skip generator through autograd (stride-1 convs with Cin = 2F,
    up-sampling, channel concat) against the PyTorch-CPU oracle's autograd on the same weights."""
    import df_oracle_torch as ort
    from deep_fluids_amd import ops
    from deep_fluids_amd.model import GeneratorBE3
    rng = np.random.RandomState(21)
    oshape, filters = [8, 16, 8, 3], 16
    p = orc.generator_init(rng, 3, oshape, filters, skip_concat=True)
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
    z = rng.uniform(-1, 1, (2, 3)).astype(np.float32)
    go = rng.uniform(-1, 1, [2] + oshape).astype(np.float32)
    ops.reset_variables()
    vs = {k: ops.set_variable(k, v) for k, v in p.items()}
    out, variables = GeneratorBE3(dev(z), filters, oshape, skip_concat=True, reuse=True)
    assert len(variables) == len(p)
    (out * dev(go)).sum().backward()
    pt = ort.to_torch(p, torch.float64)
    for v in pt.values():
        v.requires_grad_(True)
    ref = ort.generator_fwd(torch.from_numpy(z).double(), pt, oshape, filters, skip_concat=True)
    (ref * torch.from_numpy(go).double()).sum().backward()
    assert rel_linf(host(out), ref.detach().numpy()) < 2e-5
    for k in p:
        assert rel_linf(host(vs[k].grad), pt[k].grad.numpy()) < 1e-4, k
    ops.reset_variables()


def _run_step_case(is_3d, spatial, filters, batch, steps=2, use_curl=True, optimizer="adam", lr_update="decay", lr_update_step=2,
                   lr_max=1e-4):
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, default_config
    ops.reset_variables()
    rng = np.random.RandomState(123)
    oshape = list(spatial) + [(3 if is_3d else 1) if use_curl else (3 if is_3d else 2)]      # trainer.py:48-55
    p = orc.generator_init(rng, 3, oshape, filters)
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
    x, y = orc.synthetic_batch(rng, batch, spatial)
    cfg = default_config(is_3d=is_3d, res_x=spatial[-1], res_y=spatial[-2], res_z=spatial[0] if is_3d else 1,
                         filters=filters, batch_size=batch, num_samples=1000, use_curl=use_curl, optimizer=optimizer,
                         lr_update=lr_update, lr_update_step=lr_update_step, lr_max=lr_max)
    tr = Trainer(cfg)
    assert tr.output_shape == oshape
    tr.load_variables(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in p64.items()}, "v": {k: np.zeros_like(v) for k, v in p64.items()},
           "t": 0, "lr": cfg.lr_max}
    out = {"lrs": []}
    for s in range(steps):
        # the oracle's reverse pass uses the lrelu sign pattern the GPU took (fetched from the fused blocks; the layer-by-layer
        # path of thin models keeps the oracle's own): the network is piecewise linear and a pre-activation within rounding
        # error of zero may pick either slope -- an O(1) change of one dp element that says nothing about the kernels
        with ops.options(activation_fetch=[]):
            m = tr.train_step(dev(x), dev(y))
            fetched = list(ops.ACTIVATION_FETCH)
        masks = {i + 1: host(t) > 0 for i, t in enumerate(fetched)} if fetched else None
        p64, opt, info = orc.train_step(y.astype(np.float64), x.astype(np.float64), p64, opt, oshape, filters, is_3d, masks=masks,
                                        sign_u=host(m.G_), use_curl=use_curl, optimizer=optimizer)
        opt["lr"] = orc.lr_cosine(s + 1, tr.max_step) if lr_update == "decay" else orc.lr_step(opt["lr"], s, lr_update_step, cfg.lr_min)
        out["lrs"].append(tr.g_lr)
        assert abs(tr.g_lr - opt["lr"]) < 1e-12, (s, tr.g_lr, opt["lr"])
        out["velocity_rel_l1_step%d" % s] = rel_l1(host(m.G_), info["u"])
        out["loss_rel_step%d" % s] = abs(float(m.g_loss) - info["loss"]) / abs(info["loss"])
        if s == 0:
            gr = tr.grads_numpy()
            # per-variable L-inf error relative to that variable's gradient scale, floored at 1e-3 of the global
            # scale: the last conv's bias gradient is mathematically ZERO (curl kills constants), i.e. pure
            # roundoff in any finite precision (5e-18 in the fp64 oracle, 1e-8 in fp32)
            gmax = max(np.abs(v).max() for v in info["grads"].values())
            out["grad_rel_linf"] = max(
                float(np.abs(gr[k] - info["grads"][k]).max() / max(np.abs(info["grads"][k]).max(), 1e-3 * gmax)) for k in gr)
            out["grad_worst"] = max(gr, key=lambda k: float(np.abs(gr[k] - info["grads"][k]).max() / max(np.abs(info["grads"][k]).max(), 1e-3 * gmax)))
    newp = tr.variables_numpy()
    # Adam divides by sqrt(v): early updates are ~ +-lr whatever |g| is, so compare the parameter DELTAS in the
    # mean (elements whose gradient is roundoff-level flip sign; the zero-gradient last bias is excluded)
    last_bias = sorted((k for k in p if k.endswith("biases")), key=lambda k: int(k.split("/")[1].split("_")[0]))[-1]
    num = sum(np.abs((newp[k] - p[k]) - (p64[k] - p[k])).sum() for k in p if k != last_bias)
    den = sum(np.abs(p64[k] - p[k]).sum() for k in p if k != last_bias)
    out["param_delta_rel_l1"] = float(num / den)
    ops.reset_variables()
    return out


def test_train_step_3d_vs_oracle():
    r = _run_step_case(True, (8, 16, 8), 16, 2)
    assert r["velocity_rel_l1_step0"] <= 1e-4 and r["velocity_rel_l1_step1"] <= 1e-4, r
    assert r["loss_rel_step0"] < 1e-5 and r["loss_rel_step1"] < 1e-4, r
    assert r["grad_rel_linf"] < 1e-3, r          # sign(a-b) gradients flip on fp32-vs-fp64 ties; see DESIGN.md
    assert r["param_delta_rel_l1"] < 1e-2, r


@pytest.mark.parametrize("is_3d,spatial,filters,batch", [(False, (16, 8), 16, 3), (True, (8, 16, 8), 16, 2), (True, (16, 24, 16), 128, 1)])
def test_train_step_without_curl_vs_oracle(is_3d, spatial, filters, batch):
    """use_curl=False (trainer.py:48-55,141-143 / trainer3.py:19-21; run.bat's liquid scenes): the generator's last conv emits the
    2- | 3-channel velocity itself, the Jacobian / L1 tail runs op by op (no stream function, no curl).  F = 128 at 16x24x16 puts the
    Winograd forward / dgrad, the Winograd weight gradients and the matrix-core thin 128 -> 3 layer under it."""
    r = _run_step_case(is_3d, spatial, filters, batch, steps=2 if filters < 128 else 1, use_curl=False)
    assert r["velocity_rel_l1_step0"] <= 1e-4, r
    assert r["loss_rel_step0"] < 1e-5, r
    assert r["grad_rel_linf"] < 1e-3, r
    if filters < 128:
        assert r["velocity_rel_l1_step1"] <= 1e-4 and r["param_delta_rel_l1"] < 1e-2, r


@pytest.mark.parametrize("is_3d,spatial", [(False, (16, 8)), (True, (8, 16, 8))])
def test_lr_update_step_and_gd_optimizer_vs_oracle(is_3d, spatial):
    """`--lr_update step` (trainer.py:77-78, 285-286: g_lr halved after iterations lr_update_step-1, 2*lr_update_step-1, ..., floored at
    lr_min) and `--optimizer gd` (tf.train.GradientDescentOptimizer, trainer.py:163-165): five steps against the fp64 oracle."""
    r = _run_step_case(is_3d, spatial, 16, 2, steps=5, optimizer="gd", lr_update="step", lr_update_step=2, lr_max=1e-2)
    assert r["lrs"] == [1e-2, 5e-3, 5e-3, 2.5e-3, 2.5e-3], r["lrs"]
    for s in range(5):
        assert r["velocity_rel_l1_step%d" % s] <= 1e-4, r
    assert r["grad_rel_linf"] < 1e-3 and r["param_delta_rel_l1"] < 1e-3, r
    # the schedule alone, down to the floor, with Adam
    r = _run_step_case(is_3d, spatial, 16, 2, steps=4, lr_update="step", lr_update_step=1, lr_max=8e-6)
    assert r["lrs"] == [4e-6, 2.5e-6, 2.5e-6, 2.5e-6], r["lrs"]
    assert r["param_delta_rel_l1"] < 1e-2, r


def test_train_step_2d_vs_oracle():
    r = _run_step_case(False, (16, 8), 16, 3)
    assert r["velocity_rel_l1_step0"] <= 1e-4 and r["velocity_rel_l1_step1"] <= 1e-4, r
    assert r["loss_rel_step0"] < 1e-5, r
    assert r["grad_rel_linf"] < 1e-3, r
    assert r["param_delta_rel_l1"] < 1e-2, r


@pytest.mark.parametrize("algo", ["direct", "auto"])
def test_train_step_cfg3_geometry_filters128(algo):
    """One real-width (F=128) 3-D step at a reduced grid (16x24x16: same 4-level geometry as 64x96x64 / 4), with the direct
    MFMA convs and with the default dispatch (Winograd forward / dgrad at the 8x12x8 and 16x24x16 levels)."""
    from deep_fluids_amd import ops
    with ops.options(conv_algo=algo):
        r = _run_step_case(True, (16, 24, 16), 128, 1, steps=1)
    assert r["velocity_rel_l1_step0"] <= 1e-4, r
    assert r["loss_rel_step0"] < 1e-5, r
    # (lrelu sign pattern taken from the GPU, see _run_step_case: both algorithms are pinned at the same 1e-3)
    assert r["grad_rel_linf"] < 1e-3, r


def test_train_step_is_run_to_run_deterministic():
    """Two trainers started from the same variables, fed the same batch, must end two steps with bit-identical parameters and loss: every
    reduction in the path is fixed-order (range partials + fixed-order combines, per-wave LDS accumulation in program order, no global
    atomics).  F = 128 at 16x24x16 exercises the Winograd forward / dgrad, the Winograd weight gradients, the up-sampling-aware kernels
    and the matrix-core thin layer."""
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, default_config
    spatial, filters, batch = (16, 24, 16), 128, 2
    rng = np.random.RandomState(7)
    p = orc.generator_init(rng, 3, list(spatial) + [3], filters)
    x, y = orc.synthetic_batch(rng, batch, spatial)
    cfg = default_config(is_3d=True, res_x=spatial[2], res_y=spatial[1], res_z=spatial[0], filters=filters, batch_size=batch, num_samples=1000)
    results = []
    for _ in range(2):
        ops.reset_variables()
        tr = Trainer(cfg)
        tr.load_variables(p)
        for _ in range(2):
            m = tr.train_step(dev(x), dev(y))
        results.append((float(m.g_loss), tr.variables_numpy()))
    ops.reset_variables()
    assert results[0][0] == results[1][0]
    for k in results[0][1]:
        assert np.array_equal(results[0][1][k], results[1][1][k]), k


def test_checkpoint_resume_and_dataset_reader(tmp_path):
    """read a dataset in the reference's on-disk format, train, save, restore into a fresh trainer
    (slim variable names + Adam slots + step + g_lr) and continue: the next step must be bit-identical."""
    from types import SimpleNamespace
    from deep_fluids_amd import ops
    from deep_fluids_amd.data import BatchManager, write_synthetic_dataset
    from deep_fluids_amd.trainer import Trainer, default_config
    root = str(tmp_path / "smoke_tiny")
    n = write_synthetic_dataset(root, (16, 8), num_p=(3, 2), num_frames=4)
    cfg = default_config(is_3d=False, res_x=8, res_y=16, filters=16, batch_size=4, num_samples=n)
    dcfg = SimpleNamespace(random_seed=123, data_path=root, is_3d=False, arch="de", data_type="velocity", batch_size=4,
                           res_x=8, res_y=16, res_z=1, num_worker=2)
    bm = BatchManager(dcfg)
    ops.reset_variables()
    tr = Trainer(cfg)
    assert tr.max_step == int(100 // (4 / float(n)))
    for _ in range(2):
        x, y = bm.batch()
        tr.train_step(x, y)
    x, y = bm.batch()
    bm.stop_thread()
    ck = str(tmp_path / "model.ckpt.npz")
    tr.save(ck)
    la = float(tr.train_step(x, y).g_loss.detach())
    pa = tr.variables_numpy()
    ops.reset_variables()
    tr2 = Trainer(cfg)
    tr2.load(ck)
    assert tr2.step == 2 and tr2._adam_t == 2
    lb = float(tr2.train_step(x, y).g_loss.detach())
    pb = tr2.variables_numpy()
    assert la == lb
    for k in pa:
        np.testing.assert_array_equal(pa[k], pb[k])
    with np.load(ck) as d:
        assert "G/0_fc/weights" in d and "G/1_conv/biases/Adam_1" in d
    ops.reset_variables()


def test_train_loop_scalars_nan_guard_and_test_sweep(tmp_path):
    """`Trainer.train` (trainer.py:228-293: loop, JSON-lines scalars in place of the TensorBoard event file, NaN guard, periodic
    parameter sweeps, final checkpoint) and `Trainer.test_` (trainer.py:314-354: (p1,p2) fixed, frame swept, de-normalised,
    one compressed .npz per frame under <model_dir>/<p1>_<p2>/)."""
    import json as js
    from types import SimpleNamespace
    from deep_fluids_amd import ops
    from deep_fluids_amd.data import BatchManager, write_synthetic_dataset
    from deep_fluids_amd.trainer import Trainer, default_config
    root = str(tmp_path / "smoke_tiny")
    n = write_synthetic_dataset(root, (16, 8), num_p=(3, 2), num_frames=4)
    model_dir = str(tmp_path / "run")
    cfg = default_config(is_3d=False, res_x=8, res_y=16, filters=16, batch_size=4, num_samples=n, model_dir=model_dir,
                         log_step=2, test_step=2, test_batch_size=2)
    dcfg = SimpleNamespace(random_seed=123, data_path=root, is_3d=False, arch="de", data_type="velocity", batch_size=4,
                           res_x=8, res_y=16, res_z=1, num_worker=2)
    bm = BatchManager(dcfg)
    ops.reset_variables()
    tr = Trainer(cfg)
    recs = tr.train(bm, max_step=5)
    assert [r["step"] for r in recs] == [0, 2, 4] and tr.step == 5
    lines = [js.loads(l) for l in open(os.path.join(model_dir, "scalars.jsonl"))]
    assert len(lines) == 3 and {"loss/g_loss", "loss/g_loss_l1", "loss/g_loss_j_l1", "misc/epoch", "misc/g_lr", "step"} <= set(lines[0])
    assert abs(lines[1]["misc/epoch"] - 2 * 4 / float(n)) < 1e-12
    assert abs(lines[0]["loss/g_loss"] - (lines[0]["loss/g_loss_l1"] + lines[0]["loss/g_loss_j_l1"])) < 1e-5
    assert os.path.exists(os.path.join(model_dir, "model.ckpt-5.npz"))
    assert js.load(open(os.path.join(model_dir, "params.json")))["filters"] == 16                  # util.py:52-59
    with np.load(os.path.join(model_dir, "4_G.npz")) as d:
        assert d["G"].shape == (3, 4, 16, 8, 2) and d["z"].shape == (3, 4, 3)
    # a trainer of another scope started on this model_dir must refuse the checkpoint it finds there, with a clear message
    with pytest.raises(ValueError, match="does not hold this trainer's variables"):
        Trainer(cfg, name="G9")
    # NaN guard: poison the parameters -> the next logged step must raise the reference's assertion
    cfg_nan = default_config(is_3d=False, res_x=8, res_y=16, filters=16, batch_size=4, num_samples=n, log_step=2, test_step=2)
    tr2 = Trainer(cfg_nan, name="G2")                   # (no model_dir in ITS config: a trainer auto-restores the model_dir it is given)
    tr2.flat_p.fill_(float("nan"))
    bm2 = BatchManager(dcfg)
    with pytest.raises(AssertionError, match="Model diverged with loss = NaN"):
        tr2.train(bm2, max_step=1, model_dir=str(tmp_path / "nan"))
    bm2.stop_thread()
    # test_: 4 frames in batches of 2
    bm3 = BatchManager(dcfg)
    out_dir = tr.test_(bm3, p1=1, p2=1)
    assert out_dir == os.path.join(model_dir, "1_1")
    files = sorted(os.listdir(out_dir), key=lambda f: int(f[:-4]))
    assert files == ["0.npz", "1.npz", "2.npz", "3.npz"]
    z = np.zeros((4, 3), np.float32); z[:, 0] = 1 / 2.0 * 2 - 1; z[:, 1] = 1 / 1.0 * 2 - 1; z[:, 2] = np.linspace(-1, 1, 4)
    ref = host(tr.generate(dev(z))).copy()
    ref *= bm3.x_range                                   # the same in-place float32 op as BatchManager.denorm (data.py:188-189)
    for i, f in enumerate(files):
        with np.load(os.path.join(out_dir, f)) as d:
            np.testing.assert_array_equal(d["x"], ref[i])
    with pytest.raises(ValueError):
        tr.test_(bm3, p1=1, p2=1, test_b_num=3)
    ops.reset_variables()


class _FixedBatches(object):
    """A deterministic batch source (the reference's queue threads draw at random, data.py:124-144): batch i is a function of i."""

    def __init__(self, xs, ys, start, epochs_per_step):
        self.xs, self.ys, self.i, self.epochs_per_step = xs, ys, start, epochs_per_step

    def batch(self):
        i = self.i
        self.i += 1
        return self.xs[i], self.ys[i]


@pytest.mark.parametrize("arch", ["de", "dg"])
def test_periodic_checkpoint_and_auto_restore_resume_is_bit_identical(tmp_path, arch):
    """SURVEY 8(f)-3 (trainer.py:107-123: Supervisor(save_model_secs=save_sec) + prepare_or_wait_for_session): a 6-step run interrupted
    after step 3 and restarted on the same model_dir -- which auto-restores the latest model.ckpt-<step>.npz: variables, Adam slots,
    global step, g_lr (and the discriminator's slab for arch='dg') -- ends bit-identical to the uninterrupted run."""
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, GANTrainer, default_config, latest_checkpoint
    cls = GANTrainer if arch == "dg" else Trainer
    rng = np.random.RandomState(3)
    xs, ys = [], []
    for _ in range(6):
        x, y = orc.synthetic_batch(rng, 2, (16, 16))
        xs.append(dev(x)); ys.append(dev(y))

    def cfg(d, **kw):
        return default_config(is_3d=False, res_x=16, res_y=16, filters=16, batch_size=2, num_samples=40, model_dir=d, log_step=100,
                              test_step=100, arch=arch, **kw)

    def final_state(tr):
        st = {"p": tr.flat_p.clone(), "m": tr.flat_m.clone(), "v": tr.flat_v.clone()}
        if arch == "dg":
            st.update(pd=tr.D.p.clone(), md=tr.D.m.clone(), vd=tr.D.v.clone())
        return st, (tr.step, tr.g_lr, tr._adam_t)

    dir_a, dir_b = str(tmp_path / "a"), str(tmp_path / "b")
    ops.reset_variables()
    tr = cls(cfg(dir_a, save_sec=None))
    assert tr.restored_from is None
    init = tr.flat_p.clone(); init_d = tr.D.p.clone() if arch == "dg" else None
    tr.train(_FixedBatches(xs, ys, 0, 2 / 40.0), max_step=6)
    ref, ref_meta = final_state(tr)
    assert sorted(f for f in os.listdir(dir_a) if f.startswith("model.ckpt")) == ["model.ckpt-6.npz"]      # save_sec=None: only the last one
    # interrupted run: a checkpoint after every step (save_sec = 0), stopped after 3 steps
    ops.reset_variables()
    tr = cls(cfg(dir_b, save_sec=0))
    tr.flat_p.copy_(init)
    if arch == "dg":
        tr.D.p.copy_(init_d)
    tr.train(_FixedBatches(xs, ys, 0, 2 / 40.0), max_step=3)
    assert sorted(f for f in os.listdir(dir_b) if f.startswith("model.ckpt")) == ["model.ckpt-1.npz", "model.ckpt-2.npz", "model.ckpt-3.npz"]
    assert latest_checkpoint(dir_b).endswith("model.ckpt-3.npz")
    del tr
    # restart on the same model_dir (or with --load_path pointing at it): continues from step 3
    ops.reset_variables()
    tr = cls(cfg(str(tmp_path / "elsewhere"), load_path=dir_b) if arch == "de" else cfg(dir_b))
    assert tr.restored_from.endswith("model.ckpt-3.npz") and tr.step == 3 and tr._adam_t == 3
    tr.train(_FixedBatches(xs, ys, 3, 2 / 40.0), max_step=6, model_dir=dir_b)
    got, got_meta = final_state(tr)
    assert got_meta == ref_meta, (got_meta, ref_meta)
    for k in ref:
        assert torch.equal(got[k], ref[k]), k
    with np.load(os.path.join(dir_a, "model.ckpt-6.npz")) as da, np.load(os.path.join(dir_b, "model.ckpt-6.npz")) as db:
        assert sorted(da.files) == sorted(db.files) and ("D/Conv/weights/Adam_1" in da.files) == (arch == "dg")
        for k in da.files:
            np.testing.assert_array_equal(da[k], db[k])
    ops.reset_variables()
