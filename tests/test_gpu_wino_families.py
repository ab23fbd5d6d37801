"""-m gpu: the two 3-D Winograd families behind ``ops.WINO3D_FAMILY`` -- "f224" = F(2,3) x F(2,3) x F(4,3) (conv_wino43.hip, the default
since round 6) and "f222" = F(2,3)^3 (conv_wino.hip) -- for the stride-1 128 -> 128-class convs (reference: slim.conv3d, ops.py:15-16,
model.py:66-70).  The other -m gpu files run through the default family; this one keeps the F(2,3)^3 plain kernels covered, pins the
new family's fused epilogues one by one at the C-ABI, and checks that the two families are interchangeable: same sign-word layout, same
results within fp32 rounding, either one against the fp64 oracle."""
import numpy as np
import pytest
import torch

import df_oracle as orc
from gpu_util import dev, host, rel_l1, rel_linf
from test_gpu_layers import WINO2D_CASES, WINO_CASES, _conv_case

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from deep_fluids_amd import ops as o
    return o


@pytest.mark.parametrize("family", ["f222", "f224"])
@pytest.mark.parametrize("shape,cin,cout,leak", WINO_CASES)
def test_conv3d_fwd_bwd_both_families_vs_oracle(ops, family, shape, cin, cout, leak):
    counts = {}
    with ops.options(conv_algo="winograd", wino3d_family=family, dispatch_counts=counts):
        errs = _conv_case(ops, shape, cin, cout, leak, seed=cin * 3 + cout + sum(shape), mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs
    want = "conv winograd-" + ("f2x2x4" if family == "f224" else "f2x2x2")
    assert any(k.startswith(want) for k in counts), counts          # the family asked for is the one that ran


@pytest.mark.parametrize("family", ["f22", "f24"])
@pytest.mark.parametrize("shape,cin,cout,leak", WINO2D_CASES)
def test_conv2d_fwd_bwd_both_families_vs_oracle(ops, family, shape, cin, cout, leak):
    """The 2-D twins: F(2,3)^2 (conv_wino2d.hip) and F(2,3) x F(4,3) (conv_wino2d43.hip, the default since round 6), forward and dgrad
    through _ConvSame3, against the fp64 oracle (reference: slim.conv2d, ops.py:12-13, model.py:24-28)."""
    counts = {}
    with ops.options(conv_algo="winograd", wino2d_family=family, dispatch_counts=counts):
        errs = _conv_case(ops, shape, cin, cout, leak, seed=cin * 3 + cout + sum(shape), mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs
    assert any(k.startswith("conv winograd-" + ("f2x4" if family == "f24" else "f2x2 ")) for k in counts), counts


@pytest.mark.parametrize("dims", [(2, 24, 40, 64, 32), (1, 16, 32, 32, 32), (3, 33, 47, 32, 64), (2, 16, 64, 128, 128), (1, 10, 12, 32, 32)])
def test_wino2d43_fused_epilogues_match_the_direct_kernel(dims):
    """bias / lrelu / residual / lrelu-mask epilogues of df_wino2d43_conv vs df_conv_fwd on the same inputs (full, ragged and sub-block images)."""
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    B, H, W, C, N = dims
    rng = np.random.RandomState(sum(dims))
    x = dev(rng.uniform(-1, 1, (B, H, W, C)).astype(np.float32))
    w = dev((rng.uniform(-1, 1, (3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32))
    bias = dev(rng.uniform(-0.5, 0.5, N).astype(np.float32))
    res = dev(rng.uniform(-1, 1, (B, H, W, N)).astype(np.float32))
    msk = dev(rng.uniform(-1, 1, (B, H, W, N)).astype(np.float32))
    wd = torch.empty(query("df_conv_packed_elems", 9, C, N, 0), device="cuda")
    call("df_conv_pack_weights", _ptr(w), _ptr(wd), 9, C, N, 0, _stream())
    ww = torch.empty(query("df_wino2d43_packed_elems", C, N, 0), device="cuda")
    call("df_wino2d43_pack_weights", _ptr(w), _ptr(ww), C, N, 0, _stream())
    for flags in (0, 8, 8 | 1, 2, 4, 8 | 1 | 2, 2 | 4, 8 | 1 | 2 | 4):
        y0 = torch.empty((B, H, W, N), device="cuda"); y1 = torch.full_like(y0, float("nan"))
        call("df_conv_fwd", _ptr(x), _ptr(wd), _ptr(bias), _ptr(res), _ptr(msk), _ptr(y0), B, 1, H, W, C, N, 1, flags, 0.2, _stream())
        call("df_wino2d43_conv", _ptr(x), _ptr(ww), _ptr(bias), _ptr(res) if flags & 2 else None, _ptr(msk) if flags & 4 else None, _ptr(y1),
             B, H, W, C, N, flags, 0.2, _stream())
        err = rel_linf(host(y1), host(y0))
        assert err < 2e-5, (flags, err)


def test_train_step_2d_both_families_vs_oracle(ops):
    """One 2-D velocity-field train step (128 x 96: the Winograd levels of GeneratorBE) under both 2-D families against the fp64 oracle."""
    from deep_fluids_amd.trainer import Trainer, default_config
    rng = np.random.RandomState(6)
    spatial, filters, batch = (128, 96), 32, 2
    oshape = list(spatial) + [1]
    p = orc.generator_init(rng, 3, oshape, filters)
    x, y = orc.synthetic_batch(rng, batch, spatial)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in p64.items()}, "v": {k: np.zeros_like(v) for k, v in p64.items()}, "t": 0, "lr": 1e-4}
    _, _, info = orc.train_step(y.astype(np.float64), x.astype(np.float64), p64, opt, oshape, filters, False)
    for family in ("f22", "f24"):
        counts = {}
        with ops.options(wino2d_family=family, dispatch_counts=counts):
            ops.reset_variables()
            tr = Trainer(default_config(is_3d=False, res_x=96, res_y=128, filters=filters, batch_size=batch, num_samples=100))
            tr.load_variables(p)
            m = tr.train_step(dev(x), dev(y))
            u = host(m.G_)
        ops.reset_variables()
        assert any("winograd-" + ("f2x4" if family == "f24" else "f2x2 ") in k for k in counts), counts
        assert rel_l1(u, info["u"]) < 1e-5, (family, rel_l1(u, info["u"]))
        assert abs(float(m.g_loss.detach()) - info["loss"]) < 1e-5 * abs(info["loss"])


def _setup(rng, B, D, H, W, C, N):
    x = dev(rng.uniform(-1, 1, (B, D, H, W, C)).astype(np.float32))
    w = dev((rng.uniform(-1, 1, (3, 3, 3, C, N)) / np.sqrt(27 * C)).astype(np.float32))
    bias = dev(rng.uniform(-0.5, 0.5, N).astype(np.float32))
    return x, w, bias


@pytest.mark.parametrize("dims", [(2, 6, 8, 12, 64, 32), (1, 4, 8, 8, 32, 32), (1, 5, 7, 9, 32, 64), (1, 8, 16, 8, 128, 128)])
def test_wino43_fused_epilogues_match_the_direct_kernel(dims):
    """bias / lrelu / residual / fp32 lrelu-mask epilogues of df_wino43_conv vs df_conv_fwd on the same inputs (full and ragged blocks)."""
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    B, D, H, W, C, N = dims
    rng = np.random.RandomState(sum(dims))
    x, w, bias = _setup(rng, B, D, H, W, C, N)
    res = dev(rng.uniform(-1, 1, (B, D, H, W, N)).astype(np.float32))
    msk = dev(rng.uniform(-1, 1, (B, D, H, W, N)).astype(np.float32))
    wd = torch.empty(query("df_conv_packed_elems", 27, C, N, 0), device="cuda")
    call("df_conv_pack_weights", _ptr(w), _ptr(wd), 27, C, N, 0, _stream())
    ww = torch.empty(query("df_wino43_packed_elems", C, N, 0), device="cuda")
    call("df_wino43_pack_weights", _ptr(w), _ptr(ww), C, N, 0, _stream())
    for flags in (0, 8, 8 | 1, 2, 4, 8 | 1 | 2, 2 | 4, 8 | 1 | 2 | 4):
        y0 = torch.empty((B, D, H, W, N), device="cuda"); y1 = torch.full_like(y0, float("nan"))
        call("df_conv_fwd", _ptr(x), _ptr(wd), _ptr(bias), _ptr(res), _ptr(msk), _ptr(y0), B, D, H, W, C, N, 3, flags, 0.2, _stream())
        call("df_wino43_conv", _ptr(x), _ptr(ww), _ptr(bias), _ptr(res) if flags & 2 else None, _ptr(msk) if flags & 4 else None, None,
             _ptr(y1), None, None, B, D, H, W, C, N, flags, 0.2, _stream())
        err = rel_linf(host(y1), host(y0))
        assert err < 2e-5, (flags, err)


@pytest.mark.parametrize("dims", [(1, 8, 16, 8, 32), (2, 6, 10, 12, 64), (1, 4, 8, 16, 128)])
def test_wino43_sign_words_addup_and_mask_bits(ops, dims):
    """The sign-word epilogues of the new family and their interchangeability with the F(2,3)^3 family:
    * BIAS | LRELU + sign_bits: the words decode (ops.sign_bits_to_mask) to exactly (y > 0) of the kernel's own output;
    * MASK from mask_bits == MASK from the fp32 activation those bits were taken from, bit for bit, and the F(2,3)^3 kernel reading
      the SAME words produces the mask the new kernel produced (one layout);
    * ADDUP (+ sign words, with and without the primary output): y2 == y + nearest_up2x(xc) of the plain launch, bit for bit."""
    from deep_fluids_amd._lib import call, query, DF_CONV_ADDUP
    from deep_fluids_amd.ops import _ptr, _stream, _new_bits, sign_bits_to_mask
    B, D, H, W, C = dims
    rng = np.random.RandomState(sum(dims))
    x, w, bias = _setup(rng, B, D, H, W, C, C)
    s = _stream()
    ww = torch.empty(query("df_wino43_packed_elems", C, C, 0), device="cuda")
    call("df_wino43_pack_weights", _ptr(w), _ptr(ww), C, C, 0, s)
    wo = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wo), C, C, 0, s)
    fd = (B, D, H, W)
    # plain forward, with and without sign words
    y = torch.empty((B, D, H, W, C), device="cuda"); yb = torch.empty_like(y)
    bits = _new_bits(fd, C, x)
    call("df_wino43_conv", _ptr(x), _ptr(ww), _ptr(bias), None, None, None, _ptr(y), None, None, B, D, H, W, C, C, 9, 0.2, s)
    call("df_wino43_conv", _ptr(x), _ptr(ww), _ptr(bias), None, None, None, _ptr(yb), None, _ptr(bits), B, D, H, W, C, C, 9, 0.2, s)
    assert torch.equal(y, yb)
    assert torch.equal(sign_bits_to_mask(bits, fd, C), y > 0)
    # the F(2,3)^3 family's words on the same input: same layout (the masks differ only where the two families' outputs straddle zero)
    yo = torch.empty_like(y); bo = _new_bits(fd, C, x)
    call("df_wino_conv_fwd_bits", _ptr(x), _ptr(wo), _ptr(bias), None, _ptr(yo), _ptr(bo), B, D, H, W, C, C, 9, 0.2, s)
    assert torch.equal(sign_bits_to_mask(bo, fd, C), yo > 0)
    differ = (sign_bits_to_mask(bo, fd, C) != sign_bits_to_mask(bits, fd, C))
    assert float(differ.float().mean()) < 1e-4 and float(y[differ].abs().max() if differ.any() else 0.0) < 1e-5
    # masked dgrad: bits vs fp32 mask, new family; and the old family fed with the new family's words
    g = dev(rng.uniform(-1, 1, (B, D, H, W, C)).astype(np.float32))
    wd4 = torch.empty(query("df_wino43_packed_elems", C, C, 1), device="cuda")
    call("df_wino43_pack_weights", _ptr(w), _ptr(wd4), C, C, 1, s)
    wdo = torch.empty(query("df_wino_packed_elems", C, C, 1), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wdo), C, C, 1, s)
    d_bits = torch.empty_like(y); d_f32 = torch.empty_like(y); d_old = torch.empty_like(y)
    call("df_wino43_conv", _ptr(g), _ptr(wd4), None, None, None, _ptr(bits), _ptr(d_bits), None, None, B, D, H, W, C, C, 4, 0.2, s)
    call("df_wino43_conv", _ptr(g), _ptr(wd4), None, None, _ptr(y), None, _ptr(d_f32), None, None, B, D, H, W, C, C, 4, 0.2, s)
    call("df_wino_conv_fwd_bits", _ptr(g), _ptr(wdo), None, _ptr(bits), _ptr(d_old), None, B, D, H, W, C, C, 4, 0.2, s)
    assert torch.equal(d_bits, d_f32)
    assert rel_linf(host(d_old), host(d_bits)) < 2e-5
    # ADDUP: xc = a coarse tensor; y2 = y + up2x(xc)
    xc = dev(rng.uniform(-1, 1, (B, D // 2, H // 2, W // 2, C)).astype(np.float32))
    up = xc.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    ya = torch.empty_like(y); y2a = torch.empty_like(y); y2b = torch.empty_like(y); tb = _new_bits(fd, C, x)
    fl = 9 | DF_CONV_ADDUP
    call("df_wino43_conv", _ptr(x), _ptr(ww), _ptr(bias), _ptr(xc), None, None, _ptr(ya), _ptr(y2a), None, B, D, H, W, C, C, fl, 0.2, s)
    call("df_wino43_conv", _ptr(x), _ptr(ww), _ptr(bias), _ptr(xc), None, None, None, _ptr(y2b), _ptr(tb), B, D, H, W, C, C, fl, 0.2, s)
    assert torch.equal(ya, y) and torch.equal(y2a, y + up) and torch.equal(y2b, y2a)
    assert torch.equal(sign_bits_to_mask(tb, fd, C), y > 0)


def test_wino43_argument_checks():
    from deep_fluids_amd import _lib
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    x = torch.zeros((1, 4, 8, 8, 32), device="cuda"); y = torch.empty_like(x)
    w = torch.zeros(query("df_wino43_packed_elems", 32, 32, 0), device="cuda")
    for bad in (dict(flags=8), dict(flags=2), dict(flags=4), dict(flags=16), dict(flags=1 << 10), dict(cin=48)):      # BIAS without bias, ...
        with pytest.raises(_lib.DeepFluidsHipError):
            call("df_wino43_conv", _ptr(x), _ptr(w), None, None, None, None, _ptr(y), None, None, 1, 4, 8, 8, bad.get("cin", 32), 32,
                 bad.get("flags", 0), 0.2, _stream())
    with pytest.raises(_lib.DeepFluidsHipError):      # y may be null only with ADDUP + sign words
        call("df_wino43_conv", _ptr(x), _ptr(w), None, None, None, None, None, None, None, 1, 4, 8, 8, 32, 32, 0, 0.2, _stream())


@pytest.mark.parametrize("shape,C,up", [((1, 8, 12, 8), 32, False), ((1, 5, 7, 9), 64, False), ((2, 3, 5, 6), 64, True), ((1, 8, 12, 8), 128, True)])
def test_generator_blocks_agree_between_the_families(ops, shape, C, up):
    """A fused generator block (plain and up-sampling: the latter mixes the F(2,3)^3 27-point forms with the chosen family's plain convs)
    forward + backward under both families: outputs and all gradients agree to fp32 rounding."""
    from deep_fluids_amd.ops import _GenBlock, _UpGenBlock
    rng = np.random.RandomState(sum(shape) + C)
    n = 4
    x = rng.uniform(-1, 1, shape + (C,)).astype(np.float32)
    ws = [(rng.uniform(-1, 1, (3, 3, 3, C, C)) / np.sqrt(C * 27)).astype(np.float32) for _ in range(n)]
    bs = [rng.uniform(-0.3, 0.3, C).astype(np.float32) for _ in range(n)]
    fshape = tuple(shape[:1]) + tuple(2 * d for d in shape[1:]) if up else shape
    go = rng.uniform(-1, 1, fshape + (C,)).astype(np.float32)
    res = []
    for fam in ("f222", "f224"):
        with ops.options(conv_algo="winograd", wino3d_family=fam):
            xt = dev(x).requires_grad_(True)
            args = []
            for w, b in zip(ws, bs):
                args += [dev(w).requires_grad_(True), dev(b).requires_grad_(True)]
            y = (_UpGenBlock if up else _GenBlock).apply(xt, 0.2, *args)
            (y * dev(go)).sum().backward()
            res.append([host(y), host(xt.grad)] + [host(a.grad) for a in args])
    assert rel_linf(res[0][0], res[1][0]) < 1e-5
    for a, b in zip(res[0][1:], res[1][1:]):
        # (a pre-activation within rounding of zero may take the other lrelu branch in one family: compare in the mean)
        assert rel_l1(a, b) < 2e-4, rel_l1(a, b)


@pytest.mark.parametrize("family", ["f222", "f224"])
def test_train_step_both_families_vs_oracle(ops, family):
    """One 3-D velocity-field train step (F = 32: every 3-D conv of the generator on the chosen Winograd family) against the fp64 oracle:
    velocity rel-L1 within the north star's 1e-4 with a wide margin for either family."""
    from deep_fluids_amd.trainer import Trainer, default_config
    rng = np.random.RandomState(5)
    spatial, filters, batch = (16, 24, 16), 32, 2
    oshape = list(spatial) + [3]
    p = orc.generator_init(rng, 3, oshape, filters)
    x, y = orc.synthetic_batch(rng, batch, spatial)
    with ops.options(conv_algo="winograd", wino3d_family=family):
        ops.reset_variables()
        tr = Trainer(default_config(is_3d=True, res_x=16, res_y=24, res_z=16, filters=filters, batch_size=batch, num_samples=100))
        tr.load_variables(p)
        m = tr.train_step(dev(x), dev(y))
        u = host(m.G_)
    ops.reset_variables()
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    opt = {"m": {k: np.zeros_like(v) for k, v in p64.items()}, "v": {k: np.zeros_like(v) for k, v in p64.items()}, "t": 0, "lr": 1e-4}
    _, _, info = orc.train_step(y.astype(np.float64), x.astype(np.float64), p64, opt, oshape, filters, True)
    assert rel_l1(u, info["u"]) < 1e-5, rel_l1(u, info["u"])
    assert abs(float(m.g_loss) - info["loss"]) < 1e-5 * abs(info["loss"])


@pytest.mark.parametrize("dims", [(2, 24, 40, 64, 64), (1, 16, 32, 32, 32), (3, 33, 47, 32, 64), (1, 10, 12, 32, 32), (2, 50, 70, 96, 32), (8, 128, 96, 128, 128)])
def test_wino2d43_sign_words_and_mask_bits(ops, dims):
    """The 2-D twin of the sign-word epilogues (df_wino2d43_conv_bits):
    * BIAS | LRELU + sign_bits: same output as the launch without them, and the words decode (ops.sign_words2d_to_mask) to exactly (y > 0);
    * MASK from mask_bits == MASK from the fp32 activation the words were taken from, bit for bit -- also when the dgrad's own input has another
      channel count than the masked tensor (full, ragged and sub-block images)."""
    from deep_fluids_amd import _lib
    from deep_fluids_amd._lib import call, query, DF_CONV_BIAS, DF_CONV_LRELU, DF_CONV_MASK
    from deep_fluids_amd.ops import _ptr, _stream, _new_bits, sign_words2d_to_mask
    B, H, W, C, C2 = dims
    rng = np.random.RandomState(sum(dims))
    s = _stream()
    x = dev(rng.uniform(-1, 1, (B, H, W, C)).astype(np.float32))
    w = dev((rng.uniform(-1, 1, (3, 3, C, C)) / np.sqrt(9 * C)).astype(np.float32))
    bias = dev(rng.uniform(-0.5, 0.5, C).astype(np.float32))
    ww = torch.empty(query("df_wino2d43_packed_elems", C, C, 0), device="cuda")
    call("df_wino2d43_pack_weights", _ptr(w), _ptr(ww), C, C, 0, s)
    y = torch.empty((B, H, W, C), device="cuda"); yb = torch.full_like(y, float("nan"))
    bits = _new_bits((B, 1, H, W), C, x, 2)
    assert bits.numel() * 8 == query("df_wino2d43_signbits_bytes", B, H, W, C) > 0
    FW = DF_CONV_BIAS | DF_CONV_LRELU
    call("df_wino2d43_conv", _ptr(x), _ptr(ww), _ptr(bias), None, None, _ptr(y), B, H, W, C, C, FW, 0.2, s)
    call("df_wino2d43_conv_bits", _ptr(x), _ptr(ww), _ptr(bias), None, _ptr(yb), _ptr(bits), B, H, W, C, C, FW, 0.2, s)
    assert torch.equal(y, yb)
    assert torch.equal(sign_words2d_to_mask(bits, (B, 1, H, W), C), y > 0)
    # the dgrad of the layer above: its input gradient has C2 channels, its output (and the mask) C
    g = dev(rng.uniform(-1, 1, (B, H, W, C2)).astype(np.float32))
    w2 = dev((rng.uniform(-1, 1, (3, 3, C, C2)) / np.sqrt(9 * C)).astype(np.float32))
    wd = torch.empty(query("df_wino2d43_packed_elems", C, C2, 1), device="cuda")
    call("df_wino2d43_pack_weights", _ptr(w2), _ptr(wd), C, C2, 1, s)
    d_bits = torch.full_like(y, float("nan")); d_f32 = torch.empty_like(y)
    call("df_wino2d43_conv_bits", _ptr(g), _ptr(wd), None, _ptr(bits), _ptr(d_bits), None, B, H, W, C2, C, DF_CONV_MASK, 0.2, s)
    call("df_wino2d43_conv", _ptr(g), _ptr(wd), None, None, _ptr(y), _ptr(d_f32), B, H, W, C2, C, DF_CONV_MASK, 0.2, s)
    assert torch.equal(d_bits, d_f32)
    # argument checks: neither / both bit buffers, a flag combination without a sign-word variant
    for args in ((None, None, FW), (bits, bits, FW), (None, bits, DF_CONV_BIAS)):
        with pytest.raises(_lib.DeepFluidsHipError):
            call("df_wino2d43_conv_bits", _ptr(x), _ptr(ww), _ptr(bias), _ptr(args[0]), _ptr(yb), _ptr(args[1]), B, H, W, C, C, args[2], 0.2, s)


@pytest.mark.parametrize("shape,C,up", [((2, 32, 48), 32, False), ((1, 20, 30), 64, False), ((8, 32, 24), 128, False), ((2, 16, 24), 32, True), ((1, 9, 13), 64, True),
                                        ((4, 32, 24), 128, True)])
def test_sign_word_masks_2d_are_bit_identical_to_activation_masks(ops, shape, C, up):
    """2-D generator blocks (plain and up-sampling) forward + backward with the dgrads' lrelu masks read from the F(2,3) x F(4,3) kernel's sign words vs from
    the fp32 activations: outputs and every gradient bit for bit; the words are really used (dispatch through df_wino2d43_conv_bits)."""
    from deep_fluids_amd import _lib as L
    from deep_fluids_amd.ops import _GenBlock, _UpGenBlock
    rng = np.random.RandomState(sum(shape) + C)
    n = 4
    x = rng.uniform(-1, 1, shape + (C,)).astype(np.float32)
    ws = [(rng.uniform(-1, 1, (3, 3, C, C)) / np.sqrt(C * 9)).astype(np.float32) for _ in range(n)]
    bs = [rng.uniform(-0.3, 0.3, C).astype(np.float32) for _ in range(n)]
    fshape = tuple(shape[:1]) + tuple(2 * d for d in shape[1:]) if up else shape
    go = rng.uniform(-1, 1, fshape + (C,)).astype(np.float32)
    res, ncalls = [], []
    orig = L.call
    for use_bits in (True, False):
        seen = []

        def spy(name, *a):
            seen.append(name)
            return orig(name, *a)
        with ops.options(conv_algo="winograd", sign_bit_masks=use_bits):
            ops.call = spy
            try:
                xt = dev(x).requires_grad_(True)
                args = []
                for w, b in zip(ws, bs):
                    args += [dev(w).requires_grad_(True), dev(b).requires_grad_(True)]
                y = (_UpGenBlock if up else _GenBlock).apply(xt, 0.2, *args)
                (y * dev(go)).sum().backward()
            finally:
                ops.call = orig
            res.append([host(y), host(xt.grad)] + [host(a.grad) for a in args])
        ncalls.append(seen.count("df_wino2d43_conv_bits"))
    # plain block: convs 1..3 emit words, dgrads of convs 2..4 read them (6 launches); up block: conv 1 is the 9-point form (no words): 2 + 2
    assert ncalls == [4 if up else 6, 0], ncalls
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("dims", [(2, 32, 64, 64), (1, 16, 32, 32), (3, 34, 46, 32), (1, 10, 12, 64), (8, 128, 96, 128)])
def test_wino2d43_block_tail_on_sign_words_equals_the_fp32_path(ops, dims):
    """df_wino2d43_conv_addup_bits (no fp32 activation written, sign words instead) + df_lrelu_words2d_bwd_pool2x ==
    df_wino2d43_conv + df_add_up2x / df_lrelu_bwd_pool2x bit for bit; full, ragged and sub-block images (tile block 16 x 32 pixels)."""
    from deep_fluids_amd._lib import call, query, DF_CONV_BIAS, DF_CONV_LRELU
    from deep_fluids_amd.ops import _ptr, _stream, _new_bits, sign_words2d_to_mask
    B, H, W, C = dims
    rng = np.random.RandomState(sum(dims))
    s = _stream()
    x = dev(rng.uniform(-1, 1, (B, H, W, C)).astype(np.float32))
    w = dev((rng.uniform(-1, 1, (3, 3, C, C)) / np.sqrt(9 * C)).astype(np.float32))
    bias = dev(rng.uniform(-0.5, 0.5, C).astype(np.float32))
    xc = dev(rng.uniform(-1, 1, (B, H // 2, W // 2, C)).astype(np.float32))
    ww = torch.empty(query("df_wino2d43_packed_elems", C, C, 0), device="cuda")
    call("df_wino2d43_pack_weights", _ptr(w), _ptr(ww), C, C, 0, s)
    act = torch.empty((B, H, W, C), device="cuda"); y_ref = torch.empty_like(act); y2 = torch.full_like(act, float("nan"))
    call("df_wino2d43_conv", _ptr(x), _ptr(ww), _ptr(bias), None, None, _ptr(act), B, H, W, C, C, DF_CONV_BIAS | DF_CONV_LRELU, 0.2, s)
    call("df_add_up2x", _ptr(act), _ptr(xc), _ptr(y_ref), B, 1, H // 2, W // 2, C, 0, s)
    bits = _new_bits((B, 1, H, W), C, x, 2)
    call("df_wino2d43_conv_addup_bits", _ptr(x), _ptr(ww), _ptr(bias), _ptr(xc), _ptr(y2), _ptr(bits), B, H, W, C, C, 0.2, s)
    assert torch.equal(y2, y_ref)
    assert torch.equal(sign_words2d_to_mask(bits, (B, 1, H, W), C), act > 0)
    gy = dev(rng.uniform(-1, 1, (B, H, W, C)).astype(np.float32))
    gx1 = torch.empty_like(gy); p1 = torch.empty_like(xc); gx2 = torch.full_like(gy, float("nan")); p2 = torch.full_like(xc, float("nan"))
    call("df_lrelu_bwd_pool2x", _ptr(gy), _ptr(act), _ptr(gx1), _ptr(p1), 0.2, B, 1, H // 2, W // 2, C, 0, s)
    call("df_lrelu_words2d_bwd_pool2x", _ptr(gy), _ptr(bits), _ptr(gx2), _ptr(p2), 0.2, B, H // 2, W // 2, C, s)
    assert torch.equal(gx1, gx2) and torch.equal(p1, p2)
