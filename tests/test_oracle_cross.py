"""The two oracle implementations against each other (CPU): df_oracle.py's hand-written NumPy conv / generator /
reverse pass vs the independent PyTorch-CPU restatement (F.conv*, autograd).  This is the second opinion for the
arithmetic whose reference implementation (TF 1.15) is unavailable ("parity unpinned")."""
import numpy as np
import pytest
import torch

import df_oracle as orc
import df_oracle_torch as ort


@pytest.mark.parametrize("is_3d,spatial,filters", [(True, (4, 8, 4), 4), (False, (16, 8), 8)])
def test_train_step_numpy_vs_torch_fp64(is_3d, spatial, filters):
    rng = np.random.RandomState(7)
    oshape = list(spatial) + [3 if is_3d else 1]
    p = {k: v.astype(np.float64) for k, v in orc.generator_init(rng, 3, oshape, filters).items()}
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-0.1, 0.1, p[k].shape)
    x, y = orc.synthetic_batch(rng, 2, spatial)
    x = x.astype(np.float64); y = y.astype(np.float64)
    opt = {"m": {k: np.zeros_like(v) for k, v in p.items()}, "v": {k: np.zeros_like(v) for k, v in p.items()},
           "t": 0, "lr": 1e-4}
    pt = ort.to_torch(p, torch.float64)
    ot = ort.new_opt(pt)
    for step in range(2):
        p, opt, info = orc.train_step(y, x, p, opt, oshape, filters, is_3d)
        it = ort.train_step(torch.tensor(y), torch.tensor(x), pt, ot, oshape, filters, is_3d)
        assert abs(info["loss"] - it["loss"]) < 1e-12
        np.testing.assert_allclose(info["u"], it["u"].numpy(), atol=1e-12)
        for k in p:
            np.testing.assert_allclose(info["grads"][k], it["grads"][k].numpy(), atol=1e-12, err_msg=k)
            np.testing.assert_allclose(p[k], pt[k].detach().numpy(), atol=1e-12, err_msg=k)


def test_stencils_numpy_vs_torch():
    rng = np.random.RandomState(8)
    v = rng.randn(2, 4, 5, 6, 3)
    j, c = orc.jacobian3(v)
    jt, ct = ort.jacobian3(torch.tensor(v))
    np.testing.assert_array_equal(j, jt.numpy()); np.testing.assert_array_equal(c, ct.numpy())
    s = rng.randn(2, 5, 6, 1)
    np.testing.assert_array_equal(orc.curl(s), ort.curl(torch.tensor(s)).numpy())


def test_kl_bernoulli_gradient_numeric():
    """oracle kl_bernoulli / kl_bernoulli_bwd (trainer3.py:272-277): closed form vs a central difference."""
    rng = np.random.RandomState(4)
    z = rng.uniform(0.05, 0.95, (5, 8))
    n, rho = 6, 0.07
    q = z[:, :n].mean(0)
    brute = sum(rho * np.log(rho / qq) + (1 - rho) * np.log((1 - rho) / (1 - qq)) for qq in q)
    assert abs(orc.kl_bernoulli(z, n, rho) - brute) < 1e-12
    g = orc.kl_bernoulli_bwd(z, n, rho)
    for (b, j) in [(0, 0), (3, 5), (4, 7)]:
        zp = z.copy(); zp[b, j] += 1e-6
        zm = z.copy(); zm[b, j] -= 1e-6
        num = (orc.kl_bernoulli(zp, n, rho) - orc.kl_bernoulli(zm, n, rho)) / 2e-6
        assert abs(num - g[b, j]) < 1e-6 * max(1.0, abs(num))


@pytest.mark.parametrize("xshape,filters", [([8, 16, 8, 3], 8), ([16, 16, 2], 8)])
def test_ae_forward_numpy_vs_torch_fp64(xshape, filters):
    """Encoder (stride-2 TF-SAME convs: pad 0 before / 1 after, concat skips, flatten + FC) + decoder: NumPy tap-shift restatement vs
    F.pad + F.conv*(stride=2).  The torch version is the full-size (128^3) checker of tests/test_gpu_fullsize.py."""
    rng = np.random.RandomState(1)
    p = orc.ae_init(rng, xshape, filters, 8)
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
    x, _ = orc.synthetic_batch(rng, 2, xshape[:-1])
    x = x[..., :xshape[-1]].astype(np.float64)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    z = orc.encoder_fwd(x, p64, filters, 8, "AE/enc", 3)
    out = orc.generator_fwd(z, p64, xshape, filters, "AE/dec", 4)
    with torch.no_grad():
        o2, z2 = ort.ae_fwd(torch.from_numpy(x), ort.to_torch(p, torch.float64), filters, 8)
    np.testing.assert_allclose(z2.numpy(), z, atol=1e-12)
    np.testing.assert_allclose(o2.numpy(), out, atol=1e-12)


def test_kl_bernoulli_closed_form_equals_definition_and_logit_form():
    """The one loss term whose reference implementation lives in an absent dependency (tf.distributions.kl_divergence of two Bernoullis,
    trainer3.py:272-277; TensorFlow 1.15 is not in this image): the oracle's closed form against (a) the definition
    sum_k P(k) log(P(k) / Q(k)) over the two outcomes in extended precision and (b) the logits / softplus form TensorFlow's
    `_kl_bernoulli_bernoulli` evaluates (sigmoid(a) (softplus(-b) - softplus(-a)) + sigmoid(-a) (softplus(b) - softplus(a)), restated);
    and its analytic gradient against central differences.  Still a restatement -- DESIGN.md section 2 keeps the term 'parity unpinned'."""
    import df_oracle as orc
    rng = np.random.RandomState(5)
    z = rng.uniform(0.02, 0.98, (6, 9))
    n, rho = 7, 0.05
    got = orc.kl_bernoulli(z, n, rho)
    q = z[:, :n].mean(axis=0).astype(np.longdouble)
    p = np.longdouble(rho)
    definition = sum(float(pk * np.log(pk / qk)) for qj in q for pk, qk in ((p, qj), (1 - p, 1 - qj)))
    assert abs(got - definition) <= 1e-13 * abs(definition)
    softplus = lambda t: np.logaddexp(0.0, t)
    la, lb = np.log(rho) - np.log1p(-rho), np.log(q.astype(np.float64)) - np.log1p(-q.astype(np.float64))
    sig = lambda t: 1.0 / (1.0 + np.exp(-t))
    logit_form = float((sig(la) * (softplus(-lb) - softplus(-la)) + sig(-la) * (softplus(lb) - softplus(la))).sum())
    assert abs(got - logit_form) <= 1e-12 * abs(got)
    g = orc.kl_bernoulli_bwd(z, n, rho, scale=0.7)
    assert np.all(g[:, n:] == 0.0)
    for (b, j) in ((0, 0), (3, 4), (5, 6)):
        zp, zm = z.copy(), z.copy()
        zp[b, j] += 1e-6; zm[b, j] -= 1e-6
        fd = 0.7 * (orc.kl_bernoulli(zp, n, rho) - orc.kl_bernoulli(zm, n, rho)) / 2e-6
        assert abs(g[b, j] - fd) <= 1e-6 * abs(fd)


@pytest.mark.parametrize("is_3d,spatial,filters", [(True, (4, 8, 4), 4), (False, (16, 8), 8)])
def test_train_step_without_curl_numpy_vs_torch_fp64(is_3d, spatial, filters):
    """use_curl=False (trainer.py:141-143 / trainer3.py:19-21; every liquid scene of run.bat): the generator emits the velocity
    itself (2 | 3 channels); hand-written reverse pass vs autograd, and the 'gd' optimizer (trainer.py:163-165)."""
    rng = np.random.RandomState(9)
    oshape = list(spatial) + [3 if is_3d else 2]
    p = {k: v.astype(np.float64) for k, v in orc.generator_init(rng, 3, oshape, filters).items()}
    x, y = orc.synthetic_batch(rng, 2, spatial)
    x = x.astype(np.float64); y = y.astype(np.float64)
    opt = {"m": {k: np.zeros_like(v) for k, v in p.items()}, "v": {k: np.zeros_like(v) for k, v in p.items()}, "t": 0, "lr": 1e-2}
    pt = ort.to_torch(p, torch.float64)
    it = ort.train_step(torch.tensor(y), torch.tensor(x), pt, None, oshape, filters, is_3d, update=False, use_curl=False)
    newp, _, info = orc.train_step(y, x, p, opt, oshape, filters, is_3d, use_curl=False, optimizer="gd")
    assert info["u"].shape[-1] == oshape[-1] and np.array_equal(info["u"], info["psi"])
    assert abs(info["loss"] - it["loss"]) < 1e-12
    for k in p:
        np.testing.assert_allclose(info["grads"][k], it["grads"][k].numpy(), atol=1e-12, err_msg=k)
        np.testing.assert_allclose(newp[k], p[k] - 1e-2 * info["grads"][k], atol=0, rtol=0)


def test_lr_step_schedule_restatement():
    """lr_update='step' (trainer.py:77-78, 285-286): halved after iterations lr_update_step-1, 2*lr_update_step-1, ..., floored."""
    lr, seen = 1e-4, []
    for step in range(10):
        lr = orc.lr_step(lr, step, 3, lr_min=2e-5)
        seen.append(lr)
    assert seen == [1e-4, 1e-4, 5e-5, 5e-5, 5e-5, 2.5e-5, 2.5e-5, 2.5e-5, 2e-5, 2e-5]


@pytest.mark.parametrize("xshape,filters,use_curl", [([8, 16, 8, 3], 8, True), ([8, 16, 8, 3], 8, False), ([16, 16, 2], 8, True),
                                                     ([16, 16, 2], 8, False)])
def test_ae_gradients_numpy_vs_torch_fp64(xshape, filters, use_curl):
    """The AE train step's gradients (build_model_ae, trainer3.py:240-279; with and without curl): hand-written reverse pass of
    df_oracle.ae_train_step vs autograd of df_oracle_torch.ae_grads -- the latter is the full-size (cfg5) checker."""
    rng = np.random.RandomState(2)
    z_num, p_num = 8, 2
    is_3d = len(xshape) == 4
    p = {k: v.astype(np.float64) for k, v in orc.ae_init(rng, xshape, filters, z_num).items()}
    for k in p:
        if k.endswith("biases"):
            p[k] = rng.uniform(-0.05, 0.05, p[k].shape)
    x, _ = orc.synthetic_batch(rng, 2, xshape[:-1])
    x = x.astype(np.float64)
    y_last = rng.uniform(-1, 1, (2, p_num))
    opt = {"m": {k: np.zeros_like(v) for k, v in p.items()}, "v": {k: np.zeros_like(v) for k, v in p.items()}, "t": 0, "lr": 1e-4}
    _, _, info = orc.ae_train_step(x, y_last, p, opt, filters, z_num, p_num, is_3d, use_curl=use_curl)
    it = ort.ae_grads(torch.tensor(x), torch.tensor(y_last), ort.to_torch(p, torch.float64), filters, z_num, p_num, is_3d,
                      use_curl=use_curl)
    assert abs(info["loss"] - it["loss"]) < 1e-12 and abs(info["loss_p"] - it["loss_p"]) < 1e-12
    np.testing.assert_allclose(info["u"], it["u"].numpy(), atol=1e-12)
    for k in p:
        np.testing.assert_allclose(info["grads"][k], it["grads"][k].numpy(), atol=1e-11, err_msg=k)


@pytest.mark.parametrize("shape,cin,cout,k,s", [((2, 16, 12), 3, 5, 4, 2), ((1, 9, 7, 5), 2, 3, 3, 2), ((1, 11, 13), 3, 4, 2, 3), ((1, 3, 4), 1, 2, 7, 4),
                                                ((1, 6, 5, 7), 2, 2, 5, 1)])
def test_general_kernel_and_stride_same_conv_numpy_vs_torch(shape, cin, cout, k, s):
    """The wrappers' generality (ops.py:12-16: any k, any stride, 'SAME'; defaults k=4, s=2): the NumPy oracle's TF-'SAME' rule
    (out = ceil(n/s), pad_before = pad_total // 2) against explicit F.pad + F.conv*(stride) and autograd."""
    import torch.nn.functional as F
    rng = np.random.RandomState(k * 10 + s)
    nd = len(shape) - 1
    x = rng.randn(*shape, cin); w = rng.randn(*((k,) * nd), cin, cout) / k ** nd; b = rng.randn(cout)
    ref = orc.conv_same(x, w, b, stride=s)
    pads = []
    for a in reversed(range(nd)):                       # F.pad takes the LAST axis first
        n = shape[1 + a]
        o = -(-n // s); pt = max((o - 1) * s + k - n, 0)
        pads += [pt // 2, pt - pt // 2]
    xt = torch.tensor(x, requires_grad=True); wt = torch.tensor(w, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
    perm_in = (0, nd + 1) + tuple(range(1, nd + 1))
    xp = F.pad(xt.permute(*perm_in), pads)
    wk = wt.permute(nd + 1, nd, *range(nd))
    y = (F.conv3d if nd == 3 else F.conv2d)(xp, wk, bt, stride=s).permute(0, *range(2, nd + 2), 1)
    np.testing.assert_allclose(y.detach().numpy(), ref, atol=1e-12)
    go = rng.randn(*ref.shape)
    (y * torch.tensor(go)).sum().backward()
    dx, dw, db = orc.conv_same_bwd(x, w, go, stride=s)
    np.testing.assert_allclose(xt.grad.numpy(), dx, atol=1e-12)
    np.testing.assert_allclose(wt.grad.numpy(), dw, atol=1e-11)
    np.testing.assert_allclose(bt.grad.numpy(), db, atol=1e-11)


@pytest.mark.parametrize("shape,new", [((2, 5, 7, 3), (11, 4)), ((1, 4, 4, 2), (12, 12)), ((1, 3, 4, 5, 2), (7, 4, 13))])
def test_resize_nearest_numpy_vs_torch_legacy_nearest(shape, new):
    """tf.image.resize_nearest_neighbor(align_corners=False) restated (src = floor(dst in / out)) against torch's legacy 'nearest'
    interpolation, which uses the same index rule, and its autograd."""
    import torch.nn.functional as F
    rng = np.random.RandomState(3)
    x = rng.randn(*shape)
    nd = len(shape) - 2
    xt = torch.tensor(x, requires_grad=True)
    y = F.interpolate(xt.permute(0, nd + 1, *range(1, nd + 1)), size=new, mode="nearest").permute(0, *range(2, nd + 2), 1)
    np.testing.assert_array_equal(y.detach().numpy(), orc.resize_nn(x, new))
    go = rng.randn(*y.shape)
    (y * torch.tensor(go)).sum().backward()
    np.testing.assert_allclose(xt.grad.numpy(), orc.resize_nn_bwd(go, x.shape), atol=1e-12)
