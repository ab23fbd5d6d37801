"""-m gpu: SURVEY 8(f)-1 -- encoder / auto-encoder path: stride-2 TF-SAME conv, channel concat, large-K FC, sigmoid,
mean-squared loss, AE/AE3 graphs vs the reference's model.py (golden) and the AE train step vs the fp64 oracle."""
import json
import os

import numpy as np
import pytest
import torch

import df_oracle as orc
from conftest import GOLDEN
from gpu_util import dev, host, rel_l1, rel_linf

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.mark.parametrize("shape,cin,cout,leak", [
    ((1, 4, 8, 32), 16, 128, 0.2), ((2, 8, 8, 8), 32, 64, 0.2), ((1, 6, 10, 14), 64, 192, None),
    ((2, 16, 32), 16, 128, 0.2), ((1, 8, 8), 128, 128, None), ((1, 2, 2, 2), 16, 32, 0.2),
    # output rows of 8 | 16 | 32 | 64 voxels, even channels >= 32: the NATIVE stride-2 weight gradient (df_conv_s2_wgrad), incl. channel
    # counts that are not multiples of the 128-wide workgroup tile (192 = 128 + 64) and a single-plane output (Do = 1)
    ((1, 4, 4, 16), 32, 64, 0.2), ((2, 4, 8, 32), 64, 32, 0.2), ((1, 2, 4, 64), 32, 32, None), ((1, 2, 2, 128), 32, 64, 0.2),
    ((1, 4, 4, 16), 192, 192, 0.2), ((2, 8, 64), 64, 64, 0.2), ((1, 6, 4, 32), 128, 128, 0.2),
    # [r5] df_conv_s2_dgrad's live-tap specialisations (Cin > 64, Cout % 64 == 0; both tile shapes, 3-D and 2-D, ragged tiles, the AE's widths)
    ((1, 6, 10, 28), 128, 128, 0.2), ((2, 12, 32), 192, 128, None), ((1, 4, 4, 8), 320, 320, 0.2), ((1, 2, 6, 40), 256, 64, 0.2)])
def test_conv_stride2_fwd_bwd(shape, cin, cout, leak):
    from deep_fluids_amd import ops
    from deep_fluids_amd.ops import _ConvSame3S2
    counts = {}
    rng = np.random.RandomState(cin + cout + sum(shape))
    nd = len(shape) - 1
    x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
    w = (rng.uniform(-1, 1, (3,) * nd + (cin, cout)) / np.sqrt(cin * 3 ** nd)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    oshape = (shape[0],) + tuple(s // 2 for s in shape[1:])
    go = rng.uniform(-1, 1, oshape + (cout,)).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    with ops.options(dispatch_counts=counts):
        y = _ConvSame3S2.apply(xt, wt, bt, leak)
        (y * dev(go)).sum().backward()
    x64, w64, b64 = x.astype(np.float64), w.astype(np.float64), b.astype(np.float64)
    pre = orc.conv_same(x64, w64, b64, stride=2)
    ref = orc.lrelu(pre, leak) if leak is not None else pre
    dpre = go * (np.where(ref > 0, 1.0, leak) if leak is not None else 1.0)
    dx, dw, db = orc.conv_same_bwd(x64, w64, dpre, stride=2)
    errs = {"y": rel_linf(host(y), ref), "dx": rel_linf(host(xt.grad), dx), "dw": rel_linf(host(wt.grad), dw),
            "db": rel_linf(host(bt.grad), db)}
    assert max(errs.values()) < TOL, errs
    native = shape[-1] // 2 in (8, 16, 32, 64) and cin >= 32 and cout >= 32
    assert any(k.startswith("wgrad-s2 native") for k in counts) == native, counts
    # the counter names the kernel the LIBRARY took (df_conv_s2_dgrad_form): the live-tap classes need a 128-wide N tile (the forward
    # conv's Cin > 64) and Cout a multiple of 64; every other channel count runs the generic 2x2(x2)-tap class kernel
    live = cin > 64 and cout % 64 == 0
    assert any(k.startswith("dgrad-s2 parity-class-" + ("live-taps" if live else "generic")) for k in counts), counts


@pytest.mark.parametrize("oshape,cin,cout", [((2, 8, 12, 16), 128, 128), ((1, 5, 7, 9), 192, 192), ((2, 12, 20), 128, 64), ((1, 3, 5, 6), 96, 48)])
def test_conv_s2_dgrad_live_taps_vs_generic_parity_class_kernel(oshape, cin, cout):
    """df_conv_s2_dgrad (kernels specialised on the 8,4,4,2,4,2,2,1 live taps of the stride-2 adjoint's parity classes) == df_upconv_fwd on the
    same mode-2 operand (the generic 2x2x2-tap kernel multiplying the structural zeros too) up to fp32 summation order; channel counts
    without a specialisation (96 -> 48) take that generic kernel inside df_conv_s2_dgrad: bit-identical there."""
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    nd = len(oshape) - 1
    kz = 3 if nd == 3 else 1
    rng = np.random.RandomState(cin + cout + sum(oshape))
    w = dev((rng.uniform(-1, 1, (3,) * nd + (cin, cout)) / np.sqrt(cout * 3 ** nd)).astype(np.float32))
    g = dev(rng.uniform(-1, 1, oshape + (cout,)).astype(np.float32))
    s = _stream()
    B = oshape[0]
    Do, Ho, Wo = (oshape[1], oshape[2], oshape[3]) if nd == 3 else (1, oshape[1], oshape[2])
    wp = torch.empty(query("df_upconv_packed_elems", cin, cout, kz, 2), device="cuda")
    call("df_upconv_pack_weights", _ptr(w), _ptr(wp), cin, cout, kz, 2, s)
    fine = (B,) + tuple(2 * d for d in oshape[1:]) + (cin,)
    a = torch.full(fine, float("nan"), device="cuda"); b = torch.full(fine, float("nan"), device="cuda")
    call("df_conv_s2_dgrad", _ptr(g), _ptr(wp), _ptr(a), B, Do, Ho, Wo, cin, cout, kz, s)
    call("df_upconv_fwd", _ptr(g), _ptr(wp), None, _ptr(b), B, Do, Ho, Wo, cout, cin, kz, 0, 0.0, s)
    assert not torch.isnan(a).any()
    if cin <= 64 or cout % 64:
        assert torch.equal(a, b)
    else:
        assert ((a - b).abs().max() / b.abs().max()).item() < 5e-6


@pytest.mark.parametrize("shape,cin,cout", [((1, 4, 8, 32), 16, 128), ((2, 16, 32), 32, 64), ((1, 4, 4, 16), 16, 20)])
def test_conv_stride2_in_bf16x3_mode(shape, cin, cout):
    """CONV_PRECISION = 'bf16x3' must not change the operand format of the stride-2 forward (df_conv_s2_fwd has no bf16x3
    variant: the weights are packed fp32), and its backward (stride-1 dgrad / wgrad kernels on the zero-inserted gradient) stays
    within the split-precision bound."""
    from deep_fluids_amd import ops
    from deep_fluids_amd.ops import _ConvSame3S2
    rng = np.random.RandomState(cin + cout)
    nd = len(shape) - 1
    x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
    w = (rng.uniform(-1, 1, (3,) * nd + (cin, cout)) / np.sqrt(cin * 3 ** nd)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    oshape = (shape[0],) + tuple(s // 2 for s in shape[1:])
    go = rng.uniform(-1, 1, oshape + (cout,)).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    with ops.options(conv_precision="bf16x3"):
        y = _ConvSame3S2.apply(xt, wt, bt, 0.2)
        (y * dev(go)).sum().backward()
    x64, w64, b64 = x.astype(np.float64), w.astype(np.float64), b.astype(np.float64)
    ref = orc.lrelu(orc.conv_same(x64, w64, b64, stride=2), 0.2)
    dpre = go * np.where(host(y) > 0, 1.0, 0.2)
    dx, dw, db = orc.conv_same_bwd(x64, w64, dpre, stride=2)
    assert rel_linf(host(y), ref) < TOL                       # exact-fp32 forward
    assert rel_linf(host(xt.grad), dx) < 1e-4 and rel_linf(host(wt.grad), dw) < 1e-4 and rel_linf(host(bt.grad), db) < TOL


def test_conv_bf16x3_mode_asymmetric_channel_counts():
    """16 -> 18 (Cin a multiple of 4, Cout not): forward and dgrad must agree on ONE operand format -- such a layer stays on
    the exact-fp32 kernels in bf16x3 mode (ops._sfx is symmetric in cin / cout)."""
    from deep_fluids_amd import ops
    from deep_fluids_amd.ops import _ConvSame3
    rng = np.random.RandomState(9)
    shape, cin, cout = (1, 3, 5, 8), 16, 18
    x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
    w = (rng.uniform(-1, 1, (3, 3, 3, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    go = rng.uniform(-1, 1, shape + (cout,)).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    with ops.options(conv_precision="bf16x3"):
        assert ops._sfx(cin, cout) == "" and ops._sfx(cout, cin) == ""
        y = _ConvSame3.apply(xt, wt, bt, None)
        (y * dev(go)).sum().backward()
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    dx, dw, db = orc.conv_same_bwd(x64, w64, go.astype(np.float64))
    assert rel_linf(host(y), orc.conv_same(x64, w64, b.astype(np.float64))) < TOL
    assert rel_linf(host(xt.grad), dx) < TOL and rel_linf(host(wt.grad), dw) < TOL and rel_linf(host(bt.grad), db) < TOL


def test_concat_sigmoid_mse_bigfc():
    from deep_fluids_amd import ops
    from deep_fluids_amd.ops import _Linear
    rng = np.random.RandomState(4)
    a = rng.uniform(-1, 1, (2, 3, 5, 8)).astype(np.float32); b = rng.uniform(-1, 1, (2, 3, 5, 12)).astype(np.float32)
    g = rng.uniform(-1, 1, (2, 3, 5, 20)).astype(np.float32)
    at, bt = dev(a).requires_grad_(True), dev(b).requires_grad_(True)
    y = ops.concat([at, bt], axis=-1)
    np.testing.assert_array_equal(host(y), np.concatenate([a, b], -1))
    (y * dev(g)).sum().backward()
    np.testing.assert_array_equal(host(at.grad), g[..., :8]); np.testing.assert_array_equal(host(bt.grad), g[..., 8:])
    z = rng.uniform(-3, 3, (4, 7)).astype(np.float32); gz = rng.uniform(-1, 1, (4, 7)).astype(np.float32)
    zt = dev(z).requires_grad_(True)
    s = ops.sigmoid(zt)
    ref = 1 / (1 + np.exp(-z.astype(np.float64)))
    assert rel_linf(host(s), ref) < 1e-6
    (s * dev(gz)).sum().backward()
    assert rel_linf(host(zt.grad), gz * ref * (1 - ref)) < 1e-6
    p = rng.uniform(-1, 1, (4, 2)).astype(np.float32); q = rng.uniform(-1, 1, (4, 2)).astype(np.float32)
    pt = dev(p).requires_grad_(True)
    l = ops.mse_mean(pt, dev(q))
    (l * 2.0).backward()
    assert abs(float(l.detach()) - ((p.astype(np.float64) - q) ** 2).mean()) < 1e-7
    np.testing.assert_allclose(host(pt.grad), 2 * 2 * (p - q) / p.size, rtol=1e-5, atol=1e-8)
    # encoder head geometry: K = 24576 -> N = 16
    B, K, N = 3, 24576 + 100, 16
    x = rng.uniform(-1, 1, (B, K)).astype(np.float32); w = (rng.uniform(-1, 1, (K, N)) / 100).astype(np.float32)
    bias = rng.uniform(-1, 1, N).astype(np.float32); go = rng.uniform(-1, 1, (B, N)).astype(np.float32)
    xt, wt, bt2 = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(bias).requires_grad_(True)
    yy = _Linear.apply(xt, wt, bt2)
    (yy * dev(go)).sum().backward()
    assert rel_linf(host(yy), x.astype(np.float64) @ w + bias) < 2e-6
    assert rel_linf(host(wt.grad), x.astype(np.float64).T @ go) < 2e-6
    assert rel_linf(host(xt.grad), go.astype(np.float64) @ w.T) < 2e-6
    assert rel_linf(host(bt2.grad), go.astype(np.float64).sum(0)) < 2e-6


@pytest.mark.parametrize("tag", ["ae3_small", "ae2_small"])
def test_ae_vs_reference_model_py(tag):
    from deep_fluids_amd import ops
    from deep_fluids_amd.model import AE, AE3
    g = dict(np.load(os.path.join(GOLDEN, "autoencoders.npz")))
    pl = json.load(open(os.path.join(GOLDEN, "layer_plans.json")))[tag]
    ops.reset_variables()
    for k, v in g.items():
        if k.startswith(tag + "|"):
            ops.set_variable(k.split("|", 1)[1], v)
    ae = AE3 if pl["fn"] == "AE3" else AE
    out, z, variables = ae(dev(g[tag + "_x"]), pl["filters"], pl["z_num"], use_sparse=pl["use_sparse"], reuse=True)
    assert len(variables) == len(pl["variables"])
    assert rel_linf(host(z), g[tag + "_z"]) < 2e-5
    assert rel_linf(host(out), g[tag + "_out"]) < 5e-5
    ops.reset_variables()


@pytest.mark.parametrize("is_3d,spatial,filters", [(True, (8, 16, 8), 16), (False, (16, 16), 16),
                                                   # wide enough for the matrix-core paths: Winograd forward / dgrad with Cin != Cout
                                                   # (64 -> 32, 128 -> 64, 192 -> 64), stride-2 dgrad as parity-class convs, W = 32 rows
                                                   (True, (16, 16, 16), 32), (True, (16, 16, 32), 64)])
def test_ae_train_step_vs_oracle(is_3d, spatial, filters):
    # F = 64 at 16x16x32: one pre-activation within rounding of zero already moves an encoder gradient by 3e-3 (measured when the thin
    # first / last layers moved to the matrix-core kernels: different summation order, same values to 1e-6) -- back-propagate the
    # oracle on the GPU's linear pieces there (6e-5 then) as the full-size tests do
    steer = filters >= 64
    _ae_step_case(is_3d, spatial, filters, False, steer=steer, grad_tol=2e-4 if steer else 1e-3)


@pytest.mark.parametrize("is_3d,spatial,filters", [(True, (8, 16, 8), 8), (False, (16, 16), 16)])
def test_ae_train_step_use_sparse_vs_oracle(is_3d, spatial, filters):
    """use_sparse=True (trainer3.py:272-277): sigmoid on the code (model.py:196,210) + w5 * Bernoulli-KL on its first
    z_num - p_num columns."""
    _ae_step_case(is_3d, spatial, filters, True)


@pytest.mark.parametrize("is_3d,spatial,filters", [(True, (8, 16, 8), 16), (False, (16, 16), 16), (True, (16, 16, 32), 64)])
def test_ae_train_step_without_curl_vs_oracle(is_3d, spatial, filters):
    """use_curl=False (trainer.py:362-364 / trainer3.py:245-247; BASELINE cfg5 names liquid3, and run.bat trains every liquid scene
    with --use_curl=False): x_ is the decoder's own output."""
    steer = filters >= 64
    _ae_step_case(is_3d, spatial, filters, False, steer=steer, grad_tol=2e-4 if steer else 1e-3, use_curl=False)


def _ae_step_case(is_3d, spatial, filters, use_sparse, steer=False, grad_tol=1e-3, use_curl=True):
    """``steer``: back-propagate the oracle on the linear pieces the GPU is on (lrelu slopes from the fetched GPU activations, the
    signs of the |.| terms from the GPU's velocity) -- needed from ~1e5 voxels x 64 channels on, where a handful of pre-activations
    within rounding error of zero moves the cancelling gradient sums at the 1e-2 level (tests/test_gpu_fullsize.py)."""
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import AETrainer, default_config
    ops.reset_variables()
    rng = np.random.RandomState(123)
    z_num, p_num, batch = 8, 2, 2
    ch = 3 if is_3d else 2
    xshape = list(spatial) + [ch]
    p = orc.ae_init(rng, xshape, filters, z_num)
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
    x, _ = orc.synthetic_batch(rng, batch, spatial)
    y = rng.uniform(-1, 1, (batch, p_num, 5)).astype(np.float32)
    cfg = default_config(is_3d=is_3d, res_x=spatial[-1], res_y=spatial[-2], res_z=spatial[0] if is_3d else 1,
                         filters=filters, batch_size=batch, num_samples=1000, z_num=z_num, p_num=p_num,
                         use_sparse=use_sparse, sparsity=0.05, w5=0.7, use_curl=use_curl)
    tr = AETrainer(cfg)
    assert sorted(tr.var_names) == sorted(p)
    tr.load_variables(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in p64.items()}, "v": {k: np.zeros_like(v) for k, v in p64.items()},
           "t": 0, "lr": cfg.lr_max}
    kw = {}
    if steer:
        with ops.options(activation_fetch=[]):
            m = tr.train_step(dev(x), dev(y))
            acts = [host(t) > 0 for t in ops.ACTIVATION_FETCH]
        n_enc = sum(1 for k in p if "/enc/" in k and k.endswith("conv/weights"))
        n_dec = sum(1 for k in p if "/dec/" in k and k.endswith("conv/weights")) - 1           # the last decoder conv has no lrelu
        assert len(acts) == n_enc + n_dec, (len(acts), n_enc, n_dec)
        kw = dict(enc_masks={i: a for i, a in enumerate(acts[:n_enc])}, dec_masks={i + 1: a for i, a in enumerate(acts[n_enc:])},
                  sign_u=host(m.G_))
    else:
        m = tr.train_step(dev(x), dev(y))
    _, _, info = orc.ae_train_step(x.astype(np.float64), y[:, :, -1].astype(np.float64), p64, opt, filters, z_num, p_num, is_3d,
                                   use_sparse=use_sparse, sparsity=0.05, w5=0.7, use_curl=use_curl, **kw)
    if use_sparse:
        assert abs(float(m.loss_kl.detach()) - info["loss_kl"]) < 1e-5 * abs(info["loss_kl"])
    assert rel_l1(host(m.G_), info["u"]) <= 1e-4
    assert abs(float(m.g_loss.detach()) - info["loss"]) < 1e-5 * abs(info["loss"])
    assert abs(float(m.loss_p.detach()) - info["loss_p"]) < 1e-5 * abs(info["loss_p"]) + 1e-8
    gr = tr.grads_numpy()
    gmax = max(np.abs(v).max() for v in info["grads"].values())
    errs = {k: float(np.abs(gr[k] - info["grads"][k]).max() / max(np.abs(info["grads"][k]).max(), 1e-3 * gmax)) for k in gr}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print("AE step %s F=%d%s: velocity rel-L1 %.2e, worst gradients %s" % ("x".join(map(str, spatial)), filters, " (steered)" if steer else "",
                                                                       rel_l1(host(m.G_), info["u"]), worst))
    assert worst[0][1] < grad_tol, worst
    ops.reset_variables()


@pytest.mark.parametrize("tag", ["d2_small", "d3_small"])
def test_discriminator_vs_reference_model_py(tag):
    """DiscriminatorPatch(3) (three stride-2 convs from the wrapper default, slim default layer names)."""
    from deep_fluids_amd import ops
    from deep_fluids_amd.model import DiscriminatorPatch, DiscriminatorPatch3
    g = dict(np.load(os.path.join(GOLDEN, "autoencoders.npz")))
    pl = json.load(open(os.path.join(GOLDEN, "layer_plans.json")))[tag]
    ops.reset_variables()
    for k, v in g.items():
        if k.startswith(tag + "|"):
            ops.set_variable(k.split("|", 1)[1], v)
    d = DiscriminatorPatch3 if pl["fn"].endswith("3") else DiscriminatorPatch
    out, variables = d(dev(g[tag + "_x"]), pl["filters"], reuse=True)
    assert len(variables) == len(pl["variables"])
    assert rel_linf(host(out), g[tag + "_out"]) < 2e-5
    ops.reset_variables()


@pytest.mark.parametrize("is_3d,spatial", [(False, (16, 16)), (True, (8, 16, 8))])
def test_gan_train_step_vs_oracle(is_3d, spatial):
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import GANTrainer, default_config
    ops.reset_variables()
    rng = np.random.RandomState(5)
    filters, batch = 16, 2
    oshape = list(spatial) + [3 if is_3d else 1]
    pG = orc.generator_init(rng, 3, oshape, filters)
    pD = orc.discriminator_init(rng, 6 if is_3d else 3, filters, len(spatial))
    for p in (pG, pD):
        for k in p:
            if k.endswith("biases"):
                p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
    x, y = orc.synthetic_batch(rng, batch, spatial)
    cfg = default_config(is_3d=is_3d, res_x=spatial[-1], res_y=spatial[-2], res_z=spatial[0] if is_3d else 1,
                         filters=filters, batch_size=batch, num_samples=1000, arch="dg", w3=0.5)
    tr = GANTrainer(cfg)
    tr.load_variables(pG)
    for k, v in pD.items():
        o, n = tr.D.slices[k]
        tr.D.p[o:o + n].copy_(torch.from_numpy(v.reshape(-1)))
    m = tr.train_step(dev(x), dev(y))
    ref = orc.gan_losses_and_grads(y.astype(np.float64), x.astype(np.float64), {k: v.astype(np.float64) for k, v in pG.items()},
                                   {k: v.astype(np.float64) for k, v in pD.items()}, oshape, filters, is_3d, w3=0.5)
    assert abs(float(m.g_loss.detach()) - ref["g_loss"]) < 1e-5 * abs(ref["g_loss"])
    assert abs(float(m.d_loss.detach()) - ref["d_loss"]) < 1e-5 * abs(ref["d_loss"])
    assert rel_l1(host(m.G_), ref["u"]) <= 1e-4
    gG = tr.grads_numpy()
    gmax = max(np.abs(v).max() for v in ref["gG"].values())
    worst = max(float(np.abs(gG[k] - ref["gG"][k]).max() / max(np.abs(ref["gG"][k]).max(), 1e-3 * gmax)) for k in gG)
    assert worst < 2e-3, ("G", worst)
    dmax = max(np.abs(v).max() for v in ref["gD"].values())
    for k, (o, n) in tr.D.slices.items():
        got = tr.D.g[o:o + n].view(ops._VARS[k].shape).cpu().numpy()
        err = float(np.abs(got - ref["gD"][k]).max() / max(np.abs(ref["gD"][k]).max(), 1e-3 * dmax))
        assert err < 2e-3, (k, err)
    ops.reset_variables()


@pytest.mark.parametrize("is_3d,spatial,use_curl", [(True, (8, 16, 8), True), (False, (16, 16), True), (True, (8, 16, 8), False)])
def test_test_ae_latent_dump_and_reconstruction_vs_oracle(tmp_path, is_3d, spatial, use_curl):
    """``--arch=ae --is_train=False`` (trainer.py:475-583, trainer3.py:311-367).  (1) without code_path: every frame of the dataset is
    encoded in file order into ``code<z_num>.npz`` (x = codes of frames 0..F-2 per scene, y = frames 1..F-1, p = source-position
    increments from n.npz, s, f) -- checked against the fp64 oracle's encoder on the same normalised frames; (2) with code_path:
    ``code_out.npz`` (z_out, z_gt) is decoded (+ curl), de-normalised and stored per scene -- checked against the oracle's decoder."""
    from types import SimpleNamespace
    from deep_fluids_amd import ops
    from deep_fluids_amd.data import BatchManager, write_synthetic_ae_dataset, preprocess
    from deep_fluids_amd.trainer import AETrainer, default_config
    ops.reset_variables()
    root = str(tmp_path / "data")
    scenes, frames, z_num, filters = 2, 4, 6, 8
    n = write_synthetic_ae_dataset(root, spatial, num_scenes=scenes, num_frames=frames, seed=3)
    assert n == scenes * frames
    dcfg = SimpleNamespace(random_seed=1, data_path=root, is_3d=is_3d, arch="ae", data_type="velocity", batch_size=2,
                           res_x=spatial[-1], res_y=spatial[-2], res_z=spatial[0] if is_3d else 1, num_worker=1)
    bm = BatchManager(dcfg, device="cuda")
    rng = np.random.RandomState(9)
    xshape = list(spatial) + [3 if is_3d else 2]
    p = orc.ae_init(rng, xshape, filters, z_num)
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
    model_dir = str(tmp_path / "model")
    cfg = default_config(is_3d=is_3d, res_x=spatial[-1], res_y=spatial[-2], res_z=spatial[0] if is_3d else 1, filters=filters, batch_size=2,
                         num_samples=n, z_num=z_num, p_num=2 if is_3d else 1, use_curl=use_curl, model_dir=model_dir, test_batch_size=2)
    tr = AETrainer(cfg)
    tr.load_variables(p)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}

    # ---- (1) latent dump
    out = tr.test_(bm)
    assert out == os.path.join(model_dir, "code%d.npz" % z_num)
    d = dict(np.load(out))
    assert int(d["s"]) == scenes and int(d["f"]) == frames
    assert d["x"].shape == (scenes * (frames - 1), z_num) and d["y"].shape == d["x"].shape
    assert d["p"].shape == (scenes * (frames - 1), 2 if is_3d else 1)
    nn = dict(np.load(os.path.join(root, "n.npz")))
    np.testing.assert_array_equal(d["p"][:, 0], (nn["nx"][:, 1:] - nn["nx"][:, :-1]).reshape(-1))
    if is_3d:
        np.testing.assert_array_equal(d["p"][:, 1], (nn["nz"][:, 1:] - nn["nz"][:, :-1]).reshape(-1))
    zs = []
    for path in bm.paths:                 # file order = (scene, frame) (data.py:32-38)
        xf, _ = preprocess(path, bm.data_type, bm.x_range, bm.y_range)
        zs.append(orc.encoder_fwd(xf[None].astype(np.float64), p64, filters, z_num, "AE/enc", 3, 0)[0])
    zs = np.stack(zs).reshape(scenes, frames, z_num)
    assert rel_linf(d["x"], zs[:, :-1].reshape(-1, z_num)) < 2e-5
    assert rel_linf(d["y"], zs[:, 1:].reshape(-1, z_num)) < 2e-5

    # ---- (2) reconstruction from (predicted, ground-truth) codes
    code_dir = str(tmp_path / "nn")
    os.makedirs(code_dir)
    z_gt = zs.astype(np.float32)
    z_out = (z_gt + rng.uniform(-0.05, 0.05, z_gt.shape)).astype(np.float32)
    np.savez_compressed(os.path.join(code_dir, "code_out.npz"), z_out=z_out, z_gt=z_gt)
    paths = tr.test_(bm, code_path=code_dir)
    assert paths == [os.path.join(model_dir, "v%d.npz" % s) for s in range(scenes)]
    for s, path in enumerate(paths):
        v = dict(np.load(path))
        for key, zz in (("v", z_out[s]), ("v_gt", z_gt[s])):
            psi = orc.generator_fwd(zz.astype(np.float64), p64, xshape, filters, "AE/dec", 4, 0)
            if use_curl:
                ref = orc.curl3(psi) if is_3d else orc.curl(psi[..., :1])
            else:
                ref = psi
            ref = ref * bm.x_range                        # batch_manager.denorm (data.py:186-189)
            assert v[key].shape == ref.shape
            assert rel_l1(v[key], ref) <= 1e-4, (s, key)
    bm.stop_thread()
    ops.reset_variables()
