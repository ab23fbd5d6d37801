"""The oracle (oracle/df_oracle.py) against the golden vectors captured from the reference's
own ops.py / model.py (tests/golden/make_golden.py).  Bit-exact for the stencils."""
import json
import os

import numpy as np
import pytest

import df_oracle as orc
from conftest import GOLDEN

TAGS2 = ["a", "edge2", "tall", "wide"]
TAGS3 = ["a", "edge2", "slab", "b"]


@pytest.mark.parametrize("tag", TAGS2)
def test_curl_jacobian_2d_bit_exact(golden_stencils, tag):
    g = golden_stencils
    s = g["curl_%s_in" % tag]
    out = orc.curl(s)
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out, g["curl_%s_out" % tag])
    np.testing.assert_array_equal(out, g["curl_np_%s_out" % tag])       # reference's TF graph == its numpy twin
    np.testing.assert_array_equal(orc.grad_np(s), g["grad_np_%s_out" % tag])
    v = g["jacobian_%s_in" % tag]
    j, w = orc.jacobian(v)
    np.testing.assert_array_equal(j, g["jacobian_%s_j" % tag])
    np.testing.assert_array_equal(w, g["jacobian_%s_w" % tag])
    np.testing.assert_array_equal(orc.vort_np(v), g["vort_np_%s_out" % tag])
    np.testing.assert_array_equal(orc.divergence(v), g["divergence_%s_out" % tag])
    np.testing.assert_array_equal(orc.pgrad(s), g["pgrad_%s_out" % tag])                 # ops.py:292-303


def test_nchw_entry_points(golden_stencils):
    g = golden_stencils
    s = g["curl_nchw_in"]
    out = orc.curl(s.transpose(0, 2, 3, 1)).transpose(0, 3, 1, 2)
    np.testing.assert_array_equal(out, g["curl_nchw_out"])
    v = g["jacobian_nchw_in"]
    j, w = orc.jacobian(v.transpose(0, 2, 3, 1))
    np.testing.assert_array_equal(j.transpose(0, 3, 1, 2), g["jacobian_nchw_j"])
    np.testing.assert_array_equal(w.transpose(0, 3, 1, 2), g["jacobian_nchw_w"])


@pytest.mark.parametrize("tag", TAGS3)
def test_jacobian3_bit_exact(golden_stencils, tag):
    g = golden_stencils
    v = g["jacobian3_%s_in" % tag]
    j, c = orc.jacobian3(v)
    np.testing.assert_array_equal(j, g["jacobian3_%s_j" % tag])
    np.testing.assert_array_equal(c, g["jacobian3_%s_c" % tag])
    np.testing.assert_array_equal(j, g["jacobian_np3_%s_j" % tag])
    np.testing.assert_array_equal(c, g["jacobian_np3_%s_c" % tag])
    np.testing.assert_array_equal(orc.curl3(v), g["jacobian3_%s_c" % tag])
    np.testing.assert_array_equal(orc.divergence3(v), g["divergence3_%s_out" % tag])
    # c == [j7-j5, j2-j6, j3-j1]
    np.testing.assert_array_equal(c, np.stack([j[..., 7] - j[..., 5], j[..., 2] - j[..., 6], j[..., 3] - j[..., 1]], -1))


def test_composites_and_invariants(golden_stencils):
    g = golden_stencils
    u = orc.curl3(g["composite3_psi"])
    np.testing.assert_array_equal(u, g["composite3_u"])
    np.testing.assert_array_equal(orc.jacobian3(u)[0], g["composite3_ju"])
    np.testing.assert_array_equal(orc.divergence3(u), g["composite3_div"])
    assert np.abs(orc.divergence3(u)).max() < 5e-6              # div(curl) == 0 up to fp32 roundoff
    u2 = orc.curl(g["composite2_psi"])
    np.testing.assert_array_equal(u2, g["composite2_u"])
    np.testing.assert_array_equal(orc.jacobian(u2)[0], g["composite2_ju"])
    assert np.abs(orc.divergence(u2)).max() < 5e-6
    # boundary rule: last row of every forward difference equals the previous row
    j, _ = orc.jacobian3(g["jacobian3_a_in"])
    np.testing.assert_array_equal(j[:, :, :, -1, 0], j[:, :, :, -2, 0])
    np.testing.assert_array_equal(j[:, :, -1, :, 1], j[:, :, -2, :, 1])
    np.testing.assert_array_equal(j[:, -1, :, :, 2], j[:, -2, :, :, 2])


def test_layers_vs_reference_source(golden_stencils):
    g = golden_stencils
    np.testing.assert_array_equal(orc.lrelu(g["lrelu_in"]), g["lrelu_out"])
    np.testing.assert_array_equal(orc.upscale_nn(g["upscale_in"]), g["upscale_out"])      # ops.py:75-77
    np.testing.assert_array_equal(orc.upscale_nn(g["upscale3_in"]), g["upscale3_out"])    # ops.py:79-91


@pytest.mark.parametrize("tag", ["g2_small", "g3_small", "g3_odd", "g2_skip", "g3_skip"])
def test_generator_graph_structure(golden_generators, tag):
    """generator_fwd (restated) == the reference's model.py executed under the stub, bit for bit
    (same layer arithmetic underneath, so any difference is a graph-structure difference)."""
    g = golden_generators
    plans = json.load(open(os.path.join(GOLDEN, "layer_plans.json")))
    pl = plans[tag]
    p = {k.split("|", 1)[1]: v for k, v in g.items() if k.startswith(tag + "|")}
    out = orc.generator_fwd(g[tag + "_z"], p, pl["output_shape"], pl["filters"], skip_concat=pl.get("skip_concat", False))
    np.testing.assert_array_equal(out, g[tag + "_out"])
    assert sorted(p) == pl["variables"]
    rep, x0, nl = orc.generator_plan(pl["output_shape"], pl["filters"])
    assert nl == len(pl["layers"])
    assert [l["name"] for l in pl["layers"]] == ["G/0_fc"] + ["G/%d_conv" % i for i in range(1, nl)]


@pytest.mark.parametrize("tag", ["g3_nc2_rep3", "g2_nc5_lastk1", "g3_nc3_convk5", "g2_convk5_lastk1_rep2"])
def test_generator_nondefault_arguments_graph_structure(tag):
    """num_conv / repeat / conv_k / last_k away from their defaults (model.py:5-6,48-49; --repeat / --num_conv are reference flags,
    config.py:20,22): the restated generator == the reference's model.py executed with the same keyword arguments, bit for bit."""
    g = dict(np.load(os.path.join(GOLDEN, "generators_args.npz")))
    pl = json.load(open(os.path.join(GOLDEN, "layer_plans_args.json")))[tag]
    kw = pl["kwargs"]
    p = {k.split("|", 1)[1]: v for k, v in g.items() if k.startswith(tag + "|")}
    out = orc.generator_fwd(g[tag + "_z"], p, pl["output_shape"], pl["filters"], num_conv=kw.get("num_conv", 4), repeat=kw.get("repeat", 0))
    np.testing.assert_array_equal(out, g[tag + "_out"])
    assert sorted(p) == pl["variables"]
    rep, x0, nl = orc.generator_plan(pl["output_shape"], pl["filters"], kw.get("num_conv", 4), kw.get("repeat", 0))
    assert nl == len(pl["layers"]) == 2 + rep * kw.get("num_conv", 4)
    if "repeat" in kw:
        assert rep == kw["repeat"]
    ks = [l["k"] for l in pl["layers"][1:]]
    assert ks[:-1] == [kw.get("conv_k", 3)] * (nl - 2) and ks[-1] == kw.get("last_k", 3)
    # the same initialiser reproduces the variable shapes the reference's layers asserted on
    q = orc.generator_init(np.random.RandomState(0), pl["c_num"], pl["output_shape"], pl["filters"], **kw)
    assert {k: v.shape for k, v in q.items()} == {k: v.shape for k, v in p.items()}


def test_parameter_counts_at_baseline_shapes():
    plans = json.load(open(os.path.join(GOLDEN, "layer_plans.json")))
    assert plans["cfg2_2d_128x96"]["n_params"] == 2977409
    assert plans["cfg3_3d_64x96x64"]["n_params"] == 7483523
    assert plans["cfg4_3d_112x160x112"]["n_params"] == 9111171
    rng = np.random.RandomState(0)
    p = orc.generator_init(rng, 3, [16, 24, 16, 3], 16)
    n0 = 8 * 12 * 8 * 16                      # repeat_num = int(log2(24)) - 2 = 2 -> x0 = [8,12,8,16]
    assert sum(v.size for v in p.values()) == 3 * n0 + n0 + 2 * 4 * (27 * 16 * 16 + 16) + 27 * 16 * 3 + 3


def _numgrad(f, x, eps=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        xp = x.copy(); xp[i] += eps
        xm = x.copy(); xm[i] -= eps
        g[i] = (f(xp) - f(xm)) / (2 * eps)
    return g


def test_adjoints_fp64():
    rng = np.random.RandomState(1)
    x = rng.randn(1, 3, 4, 3, 3)
    gj = rng.randn(1, 3, 4, 3, 9); gc = rng.randn(1, 3, 4, 3, 3)
    f = lambda a: (orc.jacobian3(a)[0] * gj).sum() + (orc.jacobian3(a)[1] * gc).sum()
    np.testing.assert_allclose(orc.jacobian3_bwd(gj, gc), _numgrad(f, x), atol=1e-7)
    x = rng.randn(2, 4, 3, 2); gj = rng.randn(2, 4, 3, 4); gw = rng.randn(2, 4, 3, 1)
    f = lambda a: (orc.jacobian(a)[0] * gj).sum() + (orc.jacobian(a)[1] * gw).sum()
    np.testing.assert_allclose(orc.jacobian_bwd(gj, gw), _numgrad(f, x), atol=1e-7)
    s = rng.randn(2, 4, 3, 1); g = rng.randn(2, 4, 3, 2)
    f = lambda a: (orc.curl(a) * g).sum()
    np.testing.assert_allclose(orc.curl_bwd(g), _numgrad(f, s), atol=1e-7)


def test_conv_backward_fp64():
    rng = np.random.RandomState(2)
    x = rng.randn(1, 3, 4, 2, 3); w = rng.randn(3, 3, 3, 3, 2); b = rng.randn(2)
    go = rng.randn(1, 3, 4, 2, 2)
    dx, dw, db = orc.conv_same_bwd(x, w, go)
    np.testing.assert_allclose(dx, _numgrad(lambda a: (orc.conv_same(a, w, b) * go).sum(), x), atol=1e-6)
    np.testing.assert_allclose(dw, _numgrad(lambda a: (orc.conv_same(x, a, b) * go).sum(), w), atol=1e-6)
    np.testing.assert_allclose(db, go.reshape(-1, 2).sum(0), atol=1e-12)


def test_generator_backward_and_step_fp64():
    rng = np.random.RandomState(3)
    oshape = [4, 8, 4, 3]
    p = {k: v.astype(np.float64) for k, v in orc.generator_init(rng, 2, oshape, 4).items()}
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-.1, .1, p[k].shape)
    z = rng.uniform(-1, 1, (2, 2))
    x, _ = orc.synthetic_batch(rng, 2, oshape[:-1])
    x = x.astype(np.float64)

    def loss_of(pp):
        return orc.velocity_loss(orc.generator_fwd(z, pp, oshape, 4), x, True, need_grad=False)["loss"]

    psi, cache = orc.generator_fwd(z, p, oshape, 4, keep=True)
    res = orc.velocity_loss(psi, x, True)
    grads = orc.generator_bwd(res["dpsi"], cache, p)
    for key in ["G/0_fc/weights", "G/1_conv/biases", "G/3_conv/weights", "G/%d_conv/weights" % cache["last_ln"]]:
        idx = tuple(rng.randint(0, s) for s in p[key].shape)
        pp = dict(p); pp[key] = p[key].copy(); pp[key][idx] += 1e-6
        pm = dict(p); pm[key] = p[key].copy(); pm[key][idx] -= 1e-6
        num = (loss_of(pp) - loss_of(pm)) / 2e-6
        assert abs(num - grads[key][idx]) < 1e-6 + 1e-4 * abs(num), (key, num, grads[key][idx])


def test_adam_tf1_and_cosine_lr():
    p, m, v = orc.adam_tf1(np.array([1.0]), np.array([0.5]), np.zeros(1), np.zeros(1), 1, 1e-4)
    # t=1: lr_t = lr*sqrt(1-b2)/(1-b1); m = .25, v = 2.5e-4
    lr_t = 1e-4 * np.sqrt(1 - 0.999) / 0.5
    np.testing.assert_allclose(p, 1.0 - lr_t * 0.25 / (np.sqrt(0.00025) + 1e-8), rtol=1e-12)
    assert orc.lr_cosine(0, 100) == pytest.approx(1e-4)
    assert orc.lr_cosine(100, 100) == pytest.approx(2.5e-6)
    # reference max_step arithmetic (trainer.py:67): float floor-division lands one below
    assert int(100 // (8 / 21000.0)) == 262499
