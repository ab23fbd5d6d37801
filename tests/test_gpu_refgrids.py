"""-m gpu: parity at the reference's OWN documented workloads (run.bat) and on the corners of the generator surface its trainers
never visit.

(a) Full default-dispatch train steps at the grids of run.bat's commands, batch 1-2, against the PyTorch-CPU oracle (same
    assertions as the BASELINE-grid tests of test_gpu_fullsize.py):
      run.bat:21  smoke3_vel_buo   res 112x64x32  -> [Z,Y,X] = 32x64x112,  F = 128, use_curl           (x0 = 4x8x14)
      run.bat:37  liquid3_d_r      res 96x48x96   -> 96x48x96,             F = 128, use_curl=False     (x0 = 12x6x12)
      run.bat:42  liquid3_vis      res 96x72x48   -> 48x72x96,             F = 128, use_curl=False     (x0 = 6x9x12: odd coarse extent)
      run.bat:73  smoke3_mov (AE)  res 48x72x48   -> AE3 48x72x48,         F = 64,  z_num 16           (encoder 48x72x48 -> 6x9x6)
      run.bat:32  liquid_pos_size  res 128x64     -> [Y,X] = 64x128 2-D,   F = 128, use_curl=False     (x0 = 4x8)
      run.bat:56  smoke_mov (AE)   res 96x128     -> AE 128x96 2-D,        F = 64
(b) GeneratorBE / GeneratorBE3 with num_conv in {2, 3, 5}, repeat = 2 | 3, conv_k = 5, last_k = 1 (model.py:5-6,48-49): forward vs
    goldens produced by the reference's own model.py with the same keyword arguments (tests/golden/generators_args.npz), forward +
    every gradient vs the fp64 PyTorch-CPU oracle, and a Trainer step with config.num_conv / config.repeat away from the defaults."""
import json
import os

import numpy as np
import pytest
import torch

import df_oracle as orc
import df_oracle_torch as ort
from conftest import GOLDEN
from gpu_util import dev, host, rel_l1, rel_linf
from test_gpu_fullsize import _step_vs_torch_oracle, _assert_production_dispatch_identical, _ae_step_vs_torch_oracle

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- (a) run.bat grids
def test_runbat_smoke3_vel_buo_32x64x112_train_step():
    """run.bat:21: the grid BASELINE names 'smoke3_vel_buo' really has (scene/smoke3_vel_buo.py:38-40), reference batch 4 -> 2 here."""
    r = _step_vs_torch_oracle((32, 64, 112), 128, 2, seed=51, unsteered=True)
    assert r["n_layers_fetched"] == 16, r
    _assert_production_dispatch_identical(r, 3)
    assert r["velocity_rel_l1"] <= 1e-4 and r["loss_rel"] < 1e-5, r
    assert r["grad_rel_linf"] < 3e-5, r                                # measured 9.4e-6
    # un-steered: measured 6.1e-3 (x3); sign decisions that differ: 2.2e-7 of 2.7e8 lrelu's, 0 of the |.| terms
    assert r["unsteered_grad_rel_linf"] < 2e-2 and r["lrelu_sign_disagree_frac"] < 1e-5 and r["l1_sign_disagree_frac"] < 1e-5, r
    assert r["last_bias_abs"] < 1e-3, r


@pytest.mark.parametrize("spatial,seed,unsteered_bound", [((48, 72, 96), 52, 4e-3), ((96, 48, 96), 53, 2e-3)])
def test_runbat_liquid3_without_curl_train_step(spatial, seed, unsteered_bound):
    """run.bat:42 (liquid3_vis 96x72x48: x0 = 6x9x12, an ODD coarse extent through the Winograd tiles and the 27-point up-sampling
    forms) and run.bat:37 (liquid3_d_r 96x48x96): --use_curl=False, the generator emits the 3-channel velocity itself."""
    r = _step_vs_torch_oracle(spatial, 128, 1, seed=seed, unsteered=True, use_curl=False)
    assert r["n_layers_fetched"] == 16, r
    _assert_production_dispatch_identical(r, 3)
    assert r["velocity_rel_l1"] <= 1e-4 and r["loss_rel"] < 1e-5, r
    assert r["grad_rel_linf"] < 1.5e-4, r                              # measured 2.8e-5 / 5.0e-5 (the last conv's bias, included here: no curl)
    assert r["unsteered_grad_rel_linf"] < unsteered_bound and r["lrelu_sign_disagree_frac"] < 1e-5, r      # measured 1.2e-3 / 6.5e-4 (x3)


def test_runbat_liquid_pos_size_2d_64x128_without_curl_train_step():
    """run.bat:32: 2-D liquid, res_x=128 res_y=64 -> [Y,X] = 64x128 (x0 = 4x8, 5 levels, W = 128 rows), --use_curl=False: 2 output channels."""
    r = _step_vs_torch_oracle((64, 128), 128, 2, seed=54, unsteered=True, use_curl=False)
    assert r["n_layers_fetched"] == 20, r
    assert r["velocity_rel_l1"] <= 1e-4 and r["loss_rel"] < 1e-5, r
    assert r["grad_rel_linf"] < 1e-5, r                                # measured 1.8e-6
    assert r["unsteered_grad_rel_linf"] < 2e-3 and r["lrelu_sign_disagree_frac"] < 1e-5, r      # measured 1.7e-5 (one flip away from ~1e-3)


@pytest.mark.parametrize("spatial,seed,unsteered_bound", [((48, 72, 48), 55, 1.5e-2), ((128, 96), 56, 5e-2)])
def test_runbat_smoke_mov_autoencoder_train_step(spatial, seed, unsteered_bound):
    """run.bat:73 (AE3 48x72x48, --filter=64 --z_num=16, batch 4 -> 1 here) and run.bat:56 (2-D AE 128x96): build_model_ae
    (trainer3.py:240-279 / trainer.py:357-423) as a full train step vs the PyTorch-CPU oracle's autograd."""
    r = _ae_step_vs_torch_oracle(spatial, 64, 16, seed)
    assert r["velocity_rel_l1"] <= 1e-4 and r["z_rel_linf"] < 1e-4, r
    assert r["loss_rel"] < 1e-5 and r["loss_p_ok"], r
    assert r["grad_worst"][0][1] < 3e-5, r                             # measured 7.5e-6 / 8.1e-6
    assert r["unsteered_grad_worst"][0][1] < unsteered_bound and r["lrelu_sign_disagree_frac"] < 1e-5, r     # measured 4.3e-3 / 1.9e-2 (x3)


# ---------------------------------------------------------------- (b) non-default generator arguments
@pytest.mark.parametrize("tag", ["g3_nc2_rep3", "g2_nc5_lastk1", "g3_nc3_convk5", "g2_convk5_lastk1_rep2"])
def test_generator_nondefault_arguments_vs_reference_model_py(tag):
    from deep_fluids_amd import ops
    from deep_fluids_amd.model import GeneratorBE, GeneratorBE3
    g = dict(np.load(os.path.join(GOLDEN, "generators_args.npz")))
    pl = json.load(open(os.path.join(GOLDEN, "layer_plans_args.json")))[tag]
    ops.reset_variables()
    for k, v in g.items():
        if k.startswith(tag + "|"):
            ops.set_variable(k.split("|", 1)[1], v)
    gen = GeneratorBE3 if pl["fn"] == "GeneratorBE3" else GeneratorBE
    out, variables = gen(dev(g[tag + "_z"]), pl["filters"], pl["output_shape"], reuse=True, **pl["kwargs"])
    assert len(variables) == len(pl["variables"])
    assert rel_linf(host(out), g[tag + "_out"]) < 2e-5
    assert rel_l1(host(out), g[tag + "_out"]) < 1e-5
    ops.reset_variables()


@pytest.mark.parametrize("is_3d,oshape,filters,kw", [
    (True, [16, 16, 16, 3], 32, dict(num_conv=2)),                 # fused block nodes with a 2-name list (matrix-core path, F = 32)
    (True, [16, 32, 16, 3], 32, dict(num_conv=3, repeat=3)),       # 3 levels forced: x0 = 4x8x4
    (True, [8, 16, 8, 3], 16, dict(num_conv=5)),
    (False, [32, 32, 1], 32, dict(num_conv=5, last_k=1)),          # 1x1 last conv (general kernels)
    (False, [32, 16, 2], 16, dict(conv_k=5, last_k=1, repeat=2)),  # 5x5 convs: layer-by-layer path on the general kernels
    (True, [8, 8, 8, 3], 8, dict(num_conv=3, conv_k=5)),
])
def test_generator_nondefault_arguments_forward_and_gradients_vs_fp64_oracle(is_3d, oshape, filters, kw):
    from deep_fluids_amd import ops
    from deep_fluids_amd.model import GeneratorBE, GeneratorBE3
    rng = np.random.RandomState(77)
    p = orc.generator_init(rng, 3, oshape, filters, **kw)
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
    z = rng.uniform(-1, 1, (2, 3)).astype(np.float32)
    go = rng.uniform(-1, 1, [2] + oshape).astype(np.float32)
    ops.reset_variables()
    vs = {k: ops.set_variable(k, v) for k, v in p.items()}
    gen = GeneratorBE3 if is_3d else GeneratorBE
    out, variables = gen(dev(z), filters, oshape, reuse=True, **kw)
    assert len(variables) == len(p)
    (out * dev(go)).sum().backward()
    pt = ort.to_torch(p, torch.float64)
    for v in pt.values():
        v.requires_grad_(True)
    ref = ort.generator_fwd(torch.from_numpy(z).double(), pt, oshape, filters, num_conv=kw.get("num_conv", 4), repeat=kw.get("repeat", 0))
    (ref * torch.from_numpy(go).double()).sum().backward()
    assert rel_linf(host(out), ref.detach().numpy()) < 2e-5
    gmax = max(float(v.grad.abs().max()) for v in pt.values())
    worst = sorted(((float((vs[k].grad.cpu().double() - pt[k].grad).abs().max()) / max(float(pt[k].grad.abs().max()), 1e-3 * gmax), k)
                    for k in p), reverse=True)[:3]
    print("generator %s %s F=%d: worst gradients %s" % (kw, oshape, filters, worst))
    assert worst[0][0] < 1e-3, worst
    ops.reset_variables()


@pytest.mark.parametrize("is_3d,spatial,filters,kw", [(True, (16, 16, 16), 32, dict(num_conv=2)), (True, (16, 32, 16), 32, dict(num_conv=3, repeat=3)),
                                                     (False, (32, 32), 32, dict(num_conv=5))])
def test_trainer_step_with_nondefault_num_conv_and_repeat_vs_fp64_oracle(is_3d, spatial, filters, kw):
    """--num_conv / --repeat (config.py:20,22) through the whole train step (fused tail, bucket ids, Adam) vs the fp64 NumPy oracle."""
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, default_config
    ops.reset_variables()
    rng = np.random.RandomState(78)
    oshape = list(spatial) + [3 if is_3d else 1]
    p = orc.generator_init(rng, 3, oshape, filters, **kw)
    x, y = orc.synthetic_batch(rng, 2, spatial)
    cfg = default_config(is_3d=is_3d, res_x=spatial[-1], res_y=spatial[-2], res_z=spatial[0] if is_3d else 1, filters=filters, batch_size=2,
                         num_samples=1000, **kw)
    tr = Trainer(cfg)
    assert sorted(tr.var_names) == sorted(p)
    tr.load_variables(p)
    # gradients are compared on the linear pieces the GPU is on (lrelu slopes from its fetched activations, |.| signs from its velocity):
    # un-steered, one pre-activation within rounding of zero moves single gradients by 1-2e-2 here (measured), see test_gpu_fullsize.py
    with ops.options(activation_fetch=[]):
        m = tr.train_step(dev(x), dev(y))
        masks = {i + 1: host(t) > 0 for i, t in enumerate(ops.ACTIVATION_FETCH)}
    assert len(masks) == len(p) // 2 - 2
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in p64.items()}, "v": {k: np.zeros_like(v) for k, v in p64.items()}, "t": 0, "lr": cfg.lr_max}
    _, _, info = orc.train_step(y.astype(np.float64), x.astype(np.float64), p64, opt, oshape, filters, is_3d, num_conv=kw.get("num_conv", 4),
                                repeat=kw.get("repeat", 0), masks=masks, sign_u=host(m.G_))
    assert rel_l1(host(m.G_), info["u"]) <= 1e-4
    assert abs(float(m.g_loss.detach()) - info["loss"]) < 1e-5 * abs(info["loss"])
    gr = tr.grads_numpy()
    gmax = max(np.abs(v).max() for v in info["grads"].values())
    last_bias = sorted((k for k in gr if k.endswith("biases")), key=lambda k: int(k.split("/")[1].split("_")[0]))[-1]
    worst = sorted(((float(np.abs(gr[k] - info["grads"][k]).max() / max(np.abs(info["grads"][k]).max(), 1e-3 * gmax)), k) for k in gr if k != last_bias),
                   reverse=True)[:3]
    print("trainer step %s %s F=%d: worst gradients (steered) %s" % (kw, spatial, filters, worst))
    assert worst[0][0] < 2e-4, worst
    if tr.grad_sync is None:      # bucket ids follow num_conv: fc | one bucket per block | last conv
        ids = sorted({tr._bucket_id(k) for k in tr.var_names})
        assert ids == list(range(len(ids)))
    ops.reset_variables()
