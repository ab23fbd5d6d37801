"""-m gpu: conv (fp32 MFMA) fwd / dgrad / wgrad, linear, up-sampling, lrelu, add, L1, Adam, column sums
against the fp64 oracle on identical seeded inputs.  Tolerance: fp32 accumulation over K <= 3456 products
-> relative L-inf <= 2e-5 of the oracle's dynamic range (the north star asks 1e-4 relative L1 end to end)."""
import ctypes

import numpy as np
import pytest
import torch

import df_oracle as orc
from gpu_util import dev, host, rel_linf, rel_l1

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from deep_fluids_amd import ops as o
    return o


def _conv_case(ops, shape, cin, cout, leak, seed, mask_from_gpu=False):
    from deep_fluids_amd.ops import _ConvSame3
    rng = np.random.RandomState(seed)
    nd = len(shape) - 1
    x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
    w = (rng.uniform(-1, 1, (3,) * nd + (cin, cout)) / np.sqrt(cin * 3 ** nd)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    go = rng.uniform(-1, 1, shape + (cout,)).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = _ConvSame3.apply(xt, wt, bt, leak)
    (y * dev(go)).sum().backward()
    x64, w64, b64 = x.astype(np.float64), w.astype(np.float64), b.astype(np.float64)
    pre = orc.conv_same(x64, w64, b64)
    ref = orc.lrelu(pre, leak) if leak is not None else pre
    # (mask_from_gpu: take the lrelu sign pattern from the GPU output so that a near-zero pre-activation whose sign differs at
    #  the 1e-6 level does not turn into an O(1) difference of one gradient element -- used by the reduced-precision mode)
    msrc = host(y) if mask_from_gpu else ref
    dpre = go * (np.where(msrc > 0, 1.0, leak) if leak is not None else 1.0)
    dx, dw, db = orc.conv_same_bwd(x64, w64, dpre)
    return {"y": rel_linf(host(y), ref), "dx": rel_linf(host(xt.grad), dx), "dw": rel_linf(host(wt.grad), dw),
            "db": rel_linf(host(bt.grad), db)}


CASES_3D = [
    ((1, 4, 8, 16), 16, 128, 0.2),      # one exact tile set of the (2,4,16) shape
    ((2, 8, 12, 8), 32, 128, 0.2),      # W < 12 -> the (4,4,8) tile shape; generator level 0 geometry
    ((1, 5, 7, 19), 16, 128, None),     # ragged in every axis, no activation
    ((1, 3, 6, 18), 128, 128, 0.2),     # the hot layer's channel counts
    ((2, 4, 6, 16), 128, 3, None),      # last conv 128 -> 3 (N tile 32, masked columns)
    ((1, 4, 5, 14), 3, 128, 0.2),       # 3 input channels (scalar staging path; dgrad of the last conv)
    ((1, 2, 3, 4), 64, 64, 0.2),        # N tile 64
    ((1, 7, 10, 7), 16, 32, 0.2),       # cfg4 level-0 geometry (odd extents)
    ((1, 2, 3, 32), 32, 64, 0.2),       # W = 32 / 64 / 112: the fully unrolled wgrad row variants
    ((1, 2, 2, 64), 16, 16, None),
    ((1, 2, 2, 112), 16, 16, 0.2),
    ((3, 2, 2, 64), 128, 128, 0.2),     # odd row count (12 rows -> 6 pairs; 3 batches) at W = 64
    ((2, 3, 5, 16), 32, 96, 0.2),       # W = 16: the third Winograd-in-x wgrad row variant; Cout not a multiple of 64
    ((1, 4, 4, 64), 64, 64, None),      # single 64x64 quadrant (the spare waves split the voxel range)
]


@pytest.mark.parametrize("shape,cin,cout,leak", CASES_3D)
def test_conv3d_fwd_bwd(ops, shape, cin, cout, leak):
    errs = _conv_case(ops, shape, cin, cout, leak, seed=cin * 7 + cout + sum(shape))
    assert max(errs.values()) < TOL, errs


CASES_2D = [
    ((2, 8, 16), 16, 128, 0.2),
    ((1, 8, 6), 128, 128, 0.2),         # generator level 0 (W < 12 -> (1,16,8) tiles)
    ((3, 11, 21), 32, 128, None),
    ((2, 16, 24), 128, 1, None),        # last conv 128 -> 1 (2-D stream function)
    ((1, 9, 17), 1, 128, 0.2),
]


@pytest.mark.parametrize("algo", [1, 2, 3])
@pytest.mark.parametrize("shape,cin,cout,leak", [((3, 2, 2, 64), 128, 128, 0.2), ((1, 2, 6, 32), 32, 64, 0.2), ((2, 3, 4, 16), 32, 96, None),
                                                 ((2, 8, 32), 64, 64, 0.2), ((1, 2, 4, 56), 32, 32, 0.2), ((1, 1, 2, 112), 32, 32, None), ((1, 2, 2, 128), 64, 64, 0.2), ((2, 8, 96), 128, 128, 0.2), ((1, 4, 48), 32, 64, None),
                                                 ((1, 2, 4, 96), 32, 32, 0.2), ((1, 6, 24), 64, 64, 0.2)])
def test_conv_wgrad_algorithms(ops, shape, cin, cout, leak, algo):
    """df_conv_wgrad forced to the direct kernel (1), Winograd in x (2) and Winograd in (x,y) (3) -- the default picks by size --
    against the fp64 oracle; the last case is 2-D (kz = 1)."""
    with ops.options(wgrad_algo=algo):
        errs = _conv_case(ops, shape, cin, cout, leak, seed=algo + cin + cout)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("shape,leak", [((1, 4, 4, 32), 0.2), ((2, 6, 4, 64), None), ((3, 4, 6, 32), 0.2), ((1, 8, 4, 64), 0.2), ((2, 4, 6, 16), 0.2), ((1, 4, 4, 56), 0.2), ((1, 4, 4, 112), None), ((1, 4, 4, 128), 0.2)])
def test_conv_wgrad_winograd_xyz(ops, shape, leak):
    """df_conv_wgrad forced to the Winograd F(2x2x2,3x3x3) form (algo 4: 64 transform-domain products per 2x2x2 positions, four launches of
    (xi_z, xi_y) workgroup types, z/y/x G^T in the reduce) against the fp64 oracle; odd tile-row counts, several batches."""
    with ops.options(wgrad_algo=4):
        errs = _conv_case(ops, shape, 128, 128, leak, seed=sum(shape), mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("shape,leak", [((1, 4, 4, 16), 0.2), ((2, 6, 4, 32), None), ((1, 4, 6, 64), 0.2), ((1, 4, 4, 128), 0.2), ((3, 4, 6, 32), 0.2)])
def test_conv_wgrad_winograd_xyz_64_channels(ops, shape, leak):
    """The (x,y,z) form for the auto-encoder's 64 -> 64 layers (cfg5: F = 64; one 64x64 quadrant, the spare waves split the voxel range),
    the default there since round 3, against the fp64 oracle."""
    from deep_fluids_amd._lib import query
    B, D, H, W = shape
    assert query("df_conv_wgrad_form", B, D, H, W, 64, 64, 3, 0) == 3
    errs = _conv_case(ops, shape, 64, 64, leak, seed=sum(shape) + 1, mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("shape,leak", [((1, 4, 4, 28), 0.2), ((2, 4, 6, 14), None), ((1, 6, 4, 28), None)])
def test_conv_wgrad_winograd_xyz_on_zero_padded_rows(ops, shape, leak):
    """Row lengths without an (x,y,z) instantiation (W = 28, 14: levels of cfg4's 112-wide grid) run the W = 32 / 16 kernel on copies with
    zero columns appended (conv_wgrad.hip::wxyz_padded_w: the weight gradient of a SAME conv is unchanged by that) -- against the fp64
    oracle, and against the direct kernel the shape took before."""
    from deep_fluids_amd._lib import query
    B, D, H, W = shape
    assert query("df_conv_wgrad_form", B, D, H, W, 128, 128, 3, 0) == 3 and query("df_conv_wgrad_form", B, D, H, W, 128, 128, 3, 1) == 0
    assert query("df_conv_wgrad_workspace_bytes", B, D, H, W, 128, 128, 3) > query("df_conv_wgrad_workspace_bytes", B, D, H, W + (4 if W == 28 else 2), 128, 128, 3)
    errs = _conv_case(ops, shape, 128, 128, leak, seed=sum(shape) + 2, mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs
    with ops.options(wgrad_algo=1):
        errs1 = _conv_case(ops, shape, 128, 128, leak, seed=sum(shape) + 2, mask_from_gpu=True)
    assert max(errs1.values()) < TOL, errs1


WINO_CASES = [
    ((1, 4, 8, 8), 32, 32, 0.2),        # exactly one tile block
    ((2, 8, 16, 8), 64, 32, 0.2),       # Cin != Cout (forward and dgrad swap them)
    ((1, 6, 10, 12), 32, 64, None),     # partial tile blocks in z, y and x; no activation
    ((1, 5, 7, 9), 32, 32, 0.2),        # odd extents: half-filled 2x2x2 tiles at the upper faces
    ((1, 8, 12, 8), 128, 128, 0.2),     # the hot layer's channel counts (8 chunks, 4 cout slices)
    ((3, 4, 8, 16), 96, 96, 0.2),       # 3 cout slices: the generic (non XCD-pinned) worker mapping; odd chunk count
]


WINO2D_CASES = [
    ((1, 16, 32), 32, 32, 0.2),         # exactly one tile block
    ((2, 32, 48), 64, 32, 0.2),         # Cin != Cout; half-filled block column
    ((3, 33, 47), 32, 64, None),        # ragged in both axes (odd extents), no activation
    ((1, 10, 12), 32, 32, 0.2),         # smaller than a block
    ((2, 16, 64), 128, 128, 0.2),       # the hot layer's channel counts (8 chunks, 4 cout slices)
    ((2, 24, 32), 96, 96, 0.2),         # 3 cout slices: generic worker mapping; 6 chunks
]


@pytest.mark.parametrize("shape,cin,cout,leak", WINO2D_CASES)
def test_conv2d_winograd_fwd_bwd(ops, shape, cin, cout, leak):
    """conv_wino2d.hip (forward and dgrad through _ConvSame3) against the fp64 oracle, same tolerance as the direct kernel."""
    with ops.options(conv_algo="winograd"):
        errs = _conv_case(ops, shape, cin, cout, leak, seed=cin * 3 + cout + sum(shape), mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("shape,cin,cout,leak", WINO_CASES)
def test_conv3d_winograd_fwd_bwd(ops, shape, cin, cout, leak):
    """conv_wino.hip (forward and dgrad through _ConvSame3; the wgrad stays direct) against the fp64 oracle, same tolerance
    as the direct kernel."""
    with ops.options(conv_algo="winograd"):
        errs = _conv_case(ops, shape, cin, cout, leak, seed=cin * 3 + cout + sum(shape), mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs


def test_winograd_fused_epilogues_match_direct(ops):
    """bias / lrelu / residual / lrelu-mask epilogues of df_wino_conv_fwd vs df_conv_fwd on the same inputs."""
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    rng = np.random.RandomState(11)
    B, D, H, W, C, N = 2, 6, 8, 12, 64, 32
    x = dev(rng.uniform(-1, 1, (B, D, H, W, C)).astype(np.float32))
    w = dev((rng.uniform(-1, 1, (3, 3, 3, C, N)) / np.sqrt(27 * C)).astype(np.float32))
    bias = dev(rng.uniform(-0.5, 0.5, N).astype(np.float32))
    res = dev(rng.uniform(-1, 1, (B, D, H, W, N)).astype(np.float32))
    msk = dev(rng.uniform(-1, 1, (B, D, H, W, N)).astype(np.float32))
    wd = torch.empty(query("df_conv_packed_elems", 27, C, N, 0), device="cuda")
    call("df_conv_pack_weights", _ptr(w), _ptr(wd), 27, C, N, 0, _stream())
    ww = torch.empty(query("df_wino_packed_elems", C, N, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(ww), C, N, 0, _stream())
    for flags in (0, 8, 8 | 1, 2, 4, 8 | 1 | 2, 2 | 4, 8 | 1 | 2 | 4):
        y0 = torch.empty((B, D, H, W, N), device="cuda"); y1 = torch.full_like(y0, float("nan"))
        call("df_conv_fwd", _ptr(x), _ptr(wd), _ptr(bias), _ptr(res), _ptr(msk), _ptr(y0), B, D, H, W, C, N, 3, flags, 0.2, _stream())
        call("df_wino_conv_fwd", _ptr(x), _ptr(ww), _ptr(bias), _ptr(res), _ptr(msk), _ptr(y1), B, D, H, W, C, N, flags, 0.2, _stream())
        err = rel_linf(host(y1), host(y0))
        # (an lrelu applied to a pre-activation within rounding of zero may pick the other branch: difference <= 1e-6 * scale)
        assert err < 2e-5, (flags, err)


@pytest.mark.parametrize("shape,cin,cout,leak", CASES_2D)
def test_conv2d_fwd_bwd(ops, shape, cin, cout, leak):
    errs = _conv_case(ops, shape, cin, cout, leak, seed=cin * 5 + cout + sum(shape))
    assert max(errs.values()) < TOL, errs


def test_conv_fused_epilogue_flags(ops):
    """DF_CONV_RESIDUAL / DF_CONV_MASK straight through the C-ABI (used by the fused backward chain)."""
    from deep_fluids_amd.ops import _pack, _conv_raw, DF_CONV_BIAS, DF_CONV_LRELU, DF_CONV_RESIDUAL, DF_CONV_MASK
    rng = np.random.RandomState(5)
    shape, cin, cout = (1, 4, 8, 16), 32, 128
    x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
    w = (rng.uniform(-1, 1, (3, 3, 3, cin, cout)) / 30).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    res = rng.uniform(-1, 1, shape + (cout,)).astype(np.float32)
    msk = rng.uniform(-1, 1, shape + (cout,)).astype(np.float32)
    wp = _pack(dev(w), 27, cin, cout, 0)
    y = _conv_raw(dev(x), wp, dev(b), dev(res), dev(msk), shape, cin, cout, 3,
                  DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_RESIDUAL | DF_CONV_MASK, 0.2)
    ref = orc.lrelu(orc.conv_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))) + res
    ref = ref * np.where(msk > 0, 1.0, 0.2)
    assert rel_linf(host(y), ref) < TOL


def test_linear_upsample_lrelu_add(ops):
    from deep_fluids_amd.ops import _Linear, _Upsample2x
    rng = np.random.RandomState(11)
    x = rng.uniform(-1, 1, (5, 3)).astype(np.float32)
    w = rng.uniform(-1, 1, (3, 1000)).astype(np.float32)
    b = rng.uniform(-1, 1, 1000).astype(np.float32)
    go = rng.uniform(-1, 1, (5, 1000)).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = _Linear.apply(xt, wt, bt)
    (y * dev(go)).sum().backward()
    assert rel_linf(host(y), x.astype(np.float64) @ w + b) < 1e-6
    assert rel_linf(host(wt.grad), x.astype(np.float64).T @ go) < 1e-6
    assert rel_linf(host(bt.grad), go.astype(np.float64).sum(0)) < 1e-6
    assert rel_linf(host(xt.grad), go.astype(np.float64) @ w.T) < 1e-6
    for shp in [(2, 3, 5, 8), (1, 2, 3, 2, 12)]:
        a = rng.uniform(-1, 1, shp).astype(np.float32)
        g = rng.uniform(-1, 1, tuple([shp[0]] + [2 * s for s in shp[1:-1]] + [shp[-1]])).astype(np.float32)
        at = dev(a).requires_grad_(True)
        up = _Upsample2x.apply(at)
        np.testing.assert_array_equal(host(up), orc.upscale_nn(a))
        (up * dev(g)).sum().backward()
        assert rel_linf(host(at.grad), orc.upscale_nn_bwd(g.astype(np.float64))) < 1e-6
    # sizes around the element-wise kernels' tile (4 x 256 float4 per workgroup): fewer than one float4, a ragged float4 tail, exactly
    # one tile, one element more, several tiles with a partial last one
    for n in (1, 3, 5, 1003, 4096, 4097, 3 * 4096 + 7, 1 << 20):
        a = rng.uniform(-1, 1, (n,)).astype(np.float32)
        c = rng.uniform(-1, 1, (n,)).astype(np.float32)
        at = dev(a).requires_grad_(True)
        y = ops.lrelu(at)
        np.testing.assert_array_equal(host(y), orc.lrelu(a).astype(np.float32))
        (y * dev(c)).sum().backward()
        np.testing.assert_array_equal(host(at.grad), (c * np.where(a > 0, 1.0, 0.2).astype(np.float32)))
        np.testing.assert_array_equal(host(ops.add(dev(a), dev(c))), a + c)


def test_l1_mean_and_adam(ops):
    from deep_fluids_amd._lib import call
    from deep_fluids_amd.ops import _ptr, _stream
    rng = np.random.RandomState(12)
    for n in (7, 4096, 1000003):
        a = rng.uniform(-1, 1, n).astype(np.float32)
        b = rng.uniform(-1, 1, n).astype(np.float32)
        b[:3] = a[:3]                                             # sign(0) == 0 branch
        at = dev(a).requires_grad_(True)
        l = ops.l1_mean(at, dev(b))
        (l * 3.0).backward()
        assert abs(float(l) - np.abs(a.astype(np.float64) - b).mean()) < 1e-7
        np.testing.assert_allclose(host(at.grad), np.sign(a - b) * np.float32(3.0 / n), rtol=1e-6, atol=0)
    n = 100003
    p = rng.uniform(-1, 1, n).astype(np.float32); g = rng.uniform(-1, 1, n).astype(np.float32)
    m = rng.uniform(-.1, .1, n).astype(np.float32); v = rng.uniform(0, .1, n).astype(np.float32)
    pt, gt, mt, vt = dev(p), dev(g), dev(m), dev(v)
    t, lr = 7, 1e-4
    lr_t = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.5 ** t)
    call("df_adam_tf1_step", _ptr(pt), _ptr(gt), _ptr(mt), _ptr(vt), n, float(lr_t), 0.5, 0.999, 1e-8, 0.5, _stream())
    rp, rm, rv = orc.adam_tf1(p.astype(np.float64), 0.5 * g.astype(np.float64), m.astype(np.float64),
                              v.astype(np.float64), t, lr)
    assert rel_linf(host(pt), rp) < 1e-6 and rel_linf(host(mt), rm) < 1e-6 and rel_linf(host(vt), rv) < 1e-6


def test_colsum(ops):
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    rng = np.random.RandomState(13)
    g = rng.uniform(-1, 1, (5000, 128)).astype(np.float32)
    gt = dev(g)
    out = torch.empty(128, device="cuda")
    nb = query("df_colsum_workspace_bytes", 5000, 128)
    ws = torch.empty(nb // 4 + 1, device="cuda")
    call("df_colsum", _ptr(gt), _ptr(out), 5000, 128, _ptr(ws), nb, _stream())
    assert rel_linf(host(out), g.astype(np.float64).sum(0)) < 1e-6


@pytest.mark.parametrize("cshape,C", [((1, 2, 4, 16), 16), ((2, 4, 6, 8), 32), ((1, 3, 5, 7), 128), ((2, 8, 16), 16),
                                      ((1, 5, 9), 128), ((1, 2, 2, 32), 64), ((1, 2, 2, 32), 128), ((2, 1, 3, 16), 64),
                                      ((1, 3, 24), 128), ((1, 2, 48), 128), ((1, 8, 12), 32), ((2, 9, 20), 64), ((1, 2, 3, 8), 64), ((1, 2, 2, 64), 64)])
def test_upconv_block_vs_materialised_upsample(ops, cshape, C):
    """The up-sampling-aware fused block (parity-class convs on the coarse grid) == upscale + conv + ... + add of the
    oracle, forward and every gradient (input, 27-tap weights, biases)."""
    from deep_fluids_amd.ops import _UpGenBlock
    rng = np.random.RandomState(C + sum(cshape))
    nd = len(cshape) - 1
    n = 2
    xc = rng.uniform(-1, 1, cshape + (C,)).astype(np.float32)
    ws = [(rng.uniform(-1, 1, (3,) * nd + (C, C)) / np.sqrt(C * 3 ** nd)).astype(np.float32) for _ in range(n)]
    bs = [rng.uniform(-0.3, 0.3, C).astype(np.float32) for _ in range(n)]
    fshape = (cshape[0],) + tuple(2 * s for s in cshape[1:])
    go = rng.uniform(-1, 1, fshape + (C,)).astype(np.float32)
    xt = dev(xc).requires_grad_(True)
    wts = [dev(w).requires_grad_(True) for w in ws]; bts = [dev(b).requires_grad_(True) for b in bs]
    args = []
    for w, b in zip(wts, bts):
        args += [w, b]
    fetch = []      # the block's post-lrelu activations: the oracle's backward takes the lrelu branch the GPU took (a pre-activation within
    with ops.options(activation_fetch=fetch):      # rounding of zero would otherwise turn into an O(1) difference of a few gradient elements)
        y = _UpGenBlock.apply(xt, 0.2, *args)
        (y * dev(go)).sum().backward()
    assert len(fetch) == n
    # oracle: materialise the up-sampled tensor, plain convs, residual add
    x0 = orc.upscale_nn(xc.astype(np.float64))
    x = x0; ins, outs = [], []
    for w, b in zip(ws, bs):
        ins.append(x)
        x = orc.lrelu(orc.conv_same(x, w.astype(np.float64), b.astype(np.float64)))
        outs.append(x)
    ref = x + x0
    assert rel_linf(host(y), ref) < TOL
    dx = go.astype(np.float64)
    for i in reversed(range(n)):
        gmask = host(fetch[i]) > 0
        assert np.mean(gmask != (outs[i] > 0)) < 1e-4      # (the two sign patterns differ at near-zero pre-activations only)
        dpre = dx * np.where(gmask, 1.0, 0.2)
        dx, dw, db = orc.conv_same_bwd(ins[i], ws[i].astype(np.float64), dpre)
        assert rel_linf(host(wts[i].grad), dw) < TOL, ("dw", i)
        assert rel_linf(host(bts[i].grad), db) < TOL, ("db", i)
    dxc = orc.upscale_nn_bwd(dx + go)
    assert rel_linf(host(xt.grad), dxc) < TOL


@pytest.mark.parametrize("cshape,cin,cout", [((1, 2, 4, 16), 32, 32), ((2, 4, 6, 8), 32, 64), ((1, 3, 5, 7), 128, 128),
                                             ((1, 1, 1, 1), 64, 32), ((1, 5, 2, 3), 96, 160)])
def test_wino_upconv_fwd_vs_oracle(ops, cshape, cin, cout):
    """df_wino_upconv_fwd (the 27-point up-sampling-aware Winograd form, conv_wino.hip UP variant) == lrelu(conv_same(upscale(xc)) + b)
    of the oracle, including ragged tile blocks (fine extents that are not multiples of 4/8/8) and Cin != Cout."""
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    rng = np.random.RandomState(cin + cout + sum(cshape))
    xc = rng.uniform(-1, 1, cshape + (cin,)).astype(np.float32)
    w = (rng.uniform(-1, 1, (3, 3, 3, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    b = rng.uniform(-0.3, 0.3, cout).astype(np.float32)
    s = _stream()
    xt, wt, bt = dev(xc), dev(w), dev(b)
    wp = torch.empty(query("df_wino_packed_elems", cin, cout, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(wt), _ptr(wp), cin, cout, 0, s)
    B, D, H, W = cshape
    y = torch.full((B, 2 * D, 2 * H, 2 * W, cout), float("nan"), device="cuda")
    call("df_wino_upconv_fwd", _ptr(xt), _ptr(wp), _ptr(bt), _ptr(y), B, D, H, W, cin, cout, 9, 0.2, s)
    ref = orc.lrelu(orc.conv_same(orc.upscale_nn(xc.astype(np.float64)), w.astype(np.float64), b.astype(np.float64)))
    assert rel_linf(host(y), ref) < TOL
    # [r5] the coarse-block staging (wino3d_kernel MODE 3) feeds the matrix cores the SAME operand values in the same order as the plain
    # Winograd kernel sees on the materialised up-sampling (the skipped xi = 2 products are exact zeros): bit-identical outputs, also the
    # sign words of the _bits variant
    xf = xt.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3).contiguous()
    yp = torch.full_like(y, float("nan"))
    call("df_wino_conv_fwd", _ptr(xf), _ptr(wp), _ptr(bt), None, None, _ptr(yp), B, 2 * D, 2 * H, 2 * W, cin, cout, 9, 0.2, s)
    assert torch.equal(y, yp)
    nb = query("df_wino_signbits_bytes", B, 2 * D, 2 * H, 2 * W, cout)
    b1 = torch.zeros(nb, dtype=torch.uint8, device="cuda"); b2 = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    y1 = torch.full_like(y, float("nan")); y2 = torch.full_like(y, float("nan"))
    call("df_wino_upconv_fwd_bits", _ptr(xt), _ptr(wp), _ptr(bt), _ptr(y1), _ptr(b1), B, D, H, W, cin, cout, 0.2, s)
    call("df_wino_conv_fwd_bits", _ptr(xf), _ptr(wp), _ptr(bt), None, _ptr(y2), _ptr(b2), B, 2 * D, 2 * H, 2 * W, cin, cout, 9, 0.2, s)
    assert torch.equal(y1, y) and torch.equal(y2, y)
    full = (2 * D) % 4 == 0 and (2 * H) % 8 == 0 and (2 * W) % 8 == 0      # (sign bits of outputs outside the tensor are don't-care)
    if full:
        assert torch.equal(b1, b2)


@pytest.mark.parametrize("cshape,cin,cout", [((1, 2, 4, 16), 32, 32), ((2, 4, 6, 8), 32, 64), ((1, 3, 5, 7), 128, 128),
                                             ((1, 1, 1, 1), 64, 32), ((1, 5, 2, 3), 96, 160)])
def test_wino_upconv_dgrad_vs_oracle(ops, cshape, cin, cout):
    """df_wino_upconv_dgrad (pooled-output Winograd form) accumulates d/d(xc) of conv_same(upscale(xc), w) into acc: oracle =
    upscale_nn_bwd(conv_same_bwd(...).dx) on top of the previous contents of acc."""
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    rng = np.random.RandomState(cin + 2 * cout + sum(cshape))
    B, D, H, W = cshape
    g = rng.uniform(-1, 1, (B, 2 * D, 2 * H, 2 * W, cout)).astype(np.float32)
    w = (rng.uniform(-1, 1, (3, 3, 3, cin, cout)) / np.sqrt(cout * 27)).astype(np.float32)
    acc0 = rng.uniform(-1, 1, cshape + (cin,)).astype(np.float32)
    s = _stream()
    gt, wt, acc = dev(g), dev(w), dev(acc0)
    wp = torch.empty(query("df_wino_packed_elems", cin, cout, 1), device="cuda")
    call("df_wino_pack_weights", _ptr(wt), _ptr(wp), cin, cout, 1, s)
    call("df_wino_upconv_dgrad", _ptr(gt), _ptr(wp), _ptr(acc), B, D, H, W, cin, cout, s)
    xf = np.zeros((B, 2 * D, 2 * H, 2 * W, cin))
    dx, _, _ = orc.conv_same_bwd(xf, w.astype(np.float64), g.astype(np.float64))
    ref = acc0.astype(np.float64) + orc.upscale_nn_bwd(dx)
    assert rel_linf(host(acc), ref) < TOL


@pytest.mark.parametrize("cshape", [(1, 2, 2, 32), (1, 2, 3, 16), (2, 3, 2, 32), (1, 4, 5, 16), (1, 3, 2, 8), (1, 2, 2, 28), (1, 2, 2, 56)])
def test_upconv_block_wgrad_winograd_xyz_27point(ops, cshape):
    """The up-sampling-aware weight gradient in its 27-point Winograd-(x,y,z) form (wgrad_wxyz_kernel<.., UP>: coarse operand reads,
    9 workgroup types, xi_x = 2 skipped), forced on small grids; whole fused block against the oracle."""
    with ops.options(wgrad_algo=4):
        test_upconv_block_vs_materialised_upsample(ops, cshape, 128)


@pytest.mark.parametrize("cshape,C", [((1, 2, 4, 16), 32), ((1, 3, 5, 7), 128), ((1, 2, 2, 32), 64), ((1, 5, 9), 128), ((2, 8, 16), 32),
                                      ((1, 3, 24), 128), ((1, 1, 1), 32)])
def test_upconv_block_winograd_forced(ops, cshape, C):
    """The fused up-sampling block with the Winograd kernels forced on small/ragged grids (forward through df_wino_upconv_fwd, adjoint
    through df_wino_upconv_dgrad; 2-D: df_wino2d_upconv_fwd / _dgrad, 9 of 16 products)."""
    with ops.options(conv_algo="winograd"):
        test_upconv_block_vs_materialised_upsample(ops, cshape, C)


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("shape,cout", [((1, 3, 5, 32), 1), ((1, 3, 5, 32), 2), ((2, 4, 6, 16), 3), ((1, 3, 5, 32), 4), ((2, 2, 3, 64), 3),
                                        ((1, 1, 7, 16), 3), ((1, 5, 3, 48), 4)])
def test_thin_wgrad_mfma_and_valu_vs_oracle(ops, shape, cout, algo):
    """Weight gradient of the 128 -> Cout <= 4 conv (the generator's last layer): algo 0 = matrix-core form with the taps folded into the
    GEMM's M side (wgrad_thin_mfma_kernel), algo 1 = the vector-ALU kernel; both against the fp64 oracle, incl. row counts that leave
    streams empty or ragged."""
    from deep_fluids_amd._lib import call, query, lib
    from deep_fluids_amd.ops import _ptr, _stream
    rng = np.random.RandomState(cout + sum(shape))
    cin = 128
    x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
    g = rng.uniform(-1, 1, shape + (cout,)).astype(np.float32)
    B, D, H, W = shape
    s = _stream()
    xt, gt = dev(x), dev(g)
    gw = torch.full((3, 3, 3, cin, cout), float("nan"), device="cuda"); gb = torch.full((cout,), float("nan"), device="cuda")
    nb = query("df_conv_wgrad_workspace_bytes", B, D, H, W, cin, cout, 3)
    ws = torch.empty((nb + 3) // 4, device="cuda")
    call("df_conv_wgrad_algo", _ptr(xt), _ptr(gt), _ptr(gw), _ptr(gb), B, D, H, W, cin, cout, 3, _ptr(ws), nb, algo, s)
    w0 = np.zeros((3, 3, 3, cin, cout))
    _, dw, db = orc.conv_same_bwd(x.astype(np.float64), w0, g.astype(np.float64))
    assert rel_linf(host(gw), dw) < TOL
    assert rel_linf(host(gb), db) < TOL


@pytest.mark.parametrize("shape,cin,leak,wide", [((1, 3, 5, 32), 3, 0.2, 128), ((2, 2, 3, 64), 3, None, 128), ((1, 4, 2, 32), 1, 0.2, 128),
                                                 ((1, 2, 3, 32), 4, 0.2, 128), ((1, 1, 5, 32), 2, 0.2, 128), ((1, 5, 1, 64), 3, 0.2, 128),
                                                 # [r3] rows that end in a shifted chunk (W % 32 != 0), two load passes (W * Cin > 256: LDS opt-in),
                                                 # 64 output channels (the auto-encoder's first layer / the dgrad of its last)
                                                 ((1, 2, 3, 112), 3, 0.2, 128), ((1, 3, 2, 48), 3, None, 64), ((1, 2, 2, 128), 3, 0.2, 64),
                                                 ((1, 1, 3, 40), 2, 0.2, 128), ((2, 2, 3, 64), 3, 0.2, 64), ((1, 2, 2, 56), 4, None, 64)])
def test_thin_k_conv_mfma_vs_oracle(ops, shape, cin, leak, wide):
    """Cin <= 4 -> 128 | 64 conv on the matrix cores (conv_thin_k_mfma_kernel: taps on the GEMM's K side, rows of 32-voxel chunks), forward
    and all gradients against the fp64 oracle."""
    errs = _conv_case(ops, shape, cin, wide, leak, seed=cin + sum(shape), mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("shape,cout,leak,wide", [((1, 3, 5, 32), 3, None, 128), ((2, 2, 4, 64), 3, 0.2, 128), ((1, 4, 9, 32), 1, None, 128),
                                                  ((1, 2, 3, 32), 2, None, 128), ((1, 1, 6, 32), 3, None, 128), ((1, 2, 16, 32), 3, None, 128),
                                                  ((1, 1, 13, 64), 3, 0.2, 128),
                                                  # [r3] shifted last chunk, two store passes (W * Cout > 256), 64 input channels
                                                  ((1, 2, 3, 112), 3, None, 128), ((1, 3, 5, 48), 3, 0.2, 64), ((1, 2, 2, 128), 3, None, 64),
                                                  ((1, 1, 4, 40), 2, None, 128), ((2, 2, 4, 64), 3, 0.2, 64), ((1, 2, 9, 112), 1, None, 64)])
def test_thin_n_conv_mfma_vs_oracle(ops, shape, cout, leak, wide):
    """128 | 64 -> Cout <= 3 conv forward on the matrix cores (conv_thin_n_mfma_kernel: z taps on the GEMM's K side, in-plane taps on its N
    side, LDS shift-add ring), incl. single-plane volumes, y ranges with halo rows (few planes) and the gradients of the same layer."""
    errs = _conv_case(ops, shape, wide, cout, leak, seed=cout + sum(shape), mask_from_gpu=True)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("flags,W,N", [(0, 64, 128), (9, 64, 128), (4, 64, 128), (2, 64, 128), (15, 64, 128), (15, 112, 128), (4, 112, 128),
                                       (15, 48, 64), (6, 128, 64)])
def test_thin_k_conv_mfma_epilogues_match_valu_kernel(ops, flags, W, N):
    """Every fused epilogue of the thin-K matrix-core kernel (bias, lrelu, residual, lrelu-mask of another tensor: the dgrad of the
    generator's last layer uses the mask) against the vector-ALU kernel it replaces; [r3] incl. rows with a shifted last chunk (the
    overlapped voxels are computed and stored twice: same values), two load passes and 64 output channels."""
    from deep_fluids_amd._lib import call, query, lib
    from deep_fluids_amd.ops import _ptr, _stream
    torch.manual_seed(flags)
    B, D, H, C = 2, 3, 5, 3
    s = _stream()
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, N, C), device="cuda") * 2 - 1) * 0.2      # weights of the 128 -> 3 layer whose dgrad this is
    bias = torch.rand(N, device="cuda") - 0.5
    res = torch.rand((B, D, H, W, N), device="cuda") * 2 - 1
    msk = torch.rand((B, D, H, W, N), device="cuda") * 2 - 1
    wp = torch.empty(query("df_conv_packed_elems", 27, N, C, 1), device="cuda")
    call("df_conv_pack_weights", _ptr(w), _ptr(wp), 27, N, C, 1, s)
    ys = []
    for valu in (0, 32):      # 32 = DF_CONV_VALU_ONLY
        y = torch.full((B, D, H, W, N), float("nan"), device="cuda")
        call("df_conv_fwd", _ptr(x), _ptr(wp), _ptr(bias), _ptr(res), _ptr(msk), _ptr(y), B, D, H, W, C, N, 3, flags | valu, 0.2, s)
        ys.append(y)
    assert ((ys[0] - ys[1]).abs().max() / ys[1].abs().max()).item() < 2e-6


@pytest.mark.parametrize("shape,cin,cout", [((1, 3, 5, 32), 64, 3), ((1, 2, 2, 128), 64, 3), ((1, 2, 3, 112), 128, 3),      # F -> 3, 64 channels / two passes
                                            ((1, 3, 5, 32), 3, 64), ((2, 2, 3, 64), 3, 128), ((1, 2, 2, 128), 3, 64),          # 3 -> F (SWAP)
                                            ((1, 2, 3, 112), 3, 128), ((1, 1, 7, 16), 3, 64)])
def test_thin_wgrad_mfma_orientations_vs_oracle(ops, shape, cin, cout):
    """The matrix-core thin weight gradient in its other instantiations: 64-channel wide side, rows that need two 64-lane passes (LDS
    opt-in above 64 KB), and the mirrored orientation 3 -> F (thin = x gathered with +offsets, wide = the gradient stream, bias gradient
    = its column sums) -- the first layer of the auto-encoder."""
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream
    rng = np.random.RandomState(cin + 2 * cout + sum(shape))
    x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
    g = rng.uniform(-1, 1, shape + (cout,)).astype(np.float32)
    B, D, H, W = shape
    s = _stream()
    xt, gt = dev(x), dev(g)
    gw = torch.full((3, 3, 3, cin, cout), float("nan"), device="cuda"); gb = torch.full((cout,), float("nan"), device="cuda")
    nb = query("df_conv_wgrad_workspace_bytes", B, D, H, W, cin, cout, 3)
    ws = torch.empty((nb + 3) // 4, device="cuda")
    call("df_conv_wgrad", _ptr(xt), _ptr(gt), _ptr(gw), _ptr(gb), B, D, H, W, cin, cout, 3, _ptr(ws), nb, s)
    w0 = np.zeros((3, 3, 3, cin, cout))
    _, dw, db = orc.conv_same_bwd(x.astype(np.float64), w0, g.astype(np.float64))
    assert rel_linf(host(gw), dw) < TOL
    assert rel_linf(host(gb), db) < TOL


@pytest.fixture
def bf16x3(ops):
    with ops.options(conv_precision="bf16x3"):
        yield


@pytest.mark.parametrize("shape,cin,cout,leak", [((1, 4, 8, 16), 32, 128, 0.2), ((2, 8, 12, 8), 64, 128, 0.2),
                                                  ((1, 5, 7, 19), 16, 32, None), ((1, 3, 6, 18), 128, 128, 0.2),
                                                  ((2, 8, 16), 32, 128, 0.2), ((1, 9, 17), 128, 64, None), ((1, 3, 2, 64), 128, 128, 0.2),
                                                  ((2, 2, 3, 32), 64, 64, None), ((1, 5, 96), 32, 64, 0.2)])
def test_conv_bf16x3_mode(ops, bf16x3, shape, cin, cout, leak):
    """Opt-in split-precision mode: forward and dgrad on the bf16 matrix pipe (hi*hi + hi*lo + lo*hi); 16 significand
    bits per operand -> relative L-inf ~1e-5 (bound 1e-4); rows of 16/32/64/96 voxels use the bf16x3 wgrad kernel (112 takes the fp32 Winograd forms)."""
    errs = _conv_case(ops, shape, cin, cout, leak, seed=cin + cout + sum(shape), mask_from_gpu=True)
    assert errs["y"] < 1e-4 and errs["dx"] < 1e-4, errs
    assert errs["y"] > 1e-7, "suspiciously exact: is the bf16x3 kernel really running?"
    assert errs["dw"] < 1e-4 and errs["db"] < TOL, errs


def test_upconv_block_bf16x3_mode(ops, bf16x3):
    from deep_fluids_amd.ops import _UpGenBlock
    rng = np.random.RandomState(77)
    cshape, C, n = (1, 3, 5, 7), 64, 2
    xc = rng.uniform(-1, 1, cshape + (C,)).astype(np.float32)
    ws = [(rng.uniform(-1, 1, (3, 3, 3, C, C)) / np.sqrt(C * 27)).astype(np.float32) for _ in range(n)]
    bs = [rng.uniform(-0.3, 0.3, C).astype(np.float32) for _ in range(n)]
    xt = dev(xc).requires_grad_(True)
    args = []
    for w, b in zip(ws, bs):
        args += [dev(w).requires_grad_(True), dev(b).requires_grad_(True)]
    y = _UpGenBlock.apply(xt, 0.2, *args)
    fshape = (1, 6, 10, 14)
    go = rng.uniform(-1, 1, fshape + (C,)).astype(np.float32)
    (y * dev(go)).sum().backward()
    x0 = orc.upscale_nn(xc.astype(np.float64)); x = x0; ins, outs = [], []
    for w, b in zip(ws, bs):
        ins.append(x); x = orc.lrelu(orc.conv_same(x, w.astype(np.float64), b.astype(np.float64))); outs.append(x)
    assert rel_linf(host(y), x + x0) < 1e-4
    dx = go.astype(np.float64)
    for i in reversed(range(n)):
        dx, _, _ = orc.conv_same_bwd(ins[i], ws[i].astype(np.float64), dx * np.where(outs[i] > 0, 1.0, 0.2))
    assert rel_linf(host(xt.grad), orc.upscale_nn_bwd(dx + go)) < 1e-4


def test_conv_wgrad_range_override_is_bounds_checked(ops):
    """df_conv_wgrad_algo's optional partial-range override (algo >> 3) must never write past the caller's workspace: a count that
    needs more than df_conv_wgrad_workspace_bytes promises is rejected (DF_EWORKSPACE), a smaller one gives the same gradient."""
    from deep_fluids_amd._lib import call, query, DeepFluidsHipError
    from deep_fluids_amd.ops import _ptr, _stream
    torch.manual_seed(2)
    B, D, H, W, C = 2, 16, 24, 32, 128
    s = _stream()
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    g = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    nb = query("df_conv_wgrad_workspace_bytes", B, D, H, W, C, C, 3)
    ws = torch.empty((nb + 3) // 4, device="cuda")
    ref = torch.empty((27, C, C), device="cuda"); gb = torch.empty(C, device="cuda")
    call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(ref), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, 4, s)
    gw = torch.full((27, C, C), float("nan"), device="cuda")
    call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, 4 | (32 << 3), s)
    assert ((gw - ref).abs().max() / ref.abs().max()).item() < 2e-5
    with pytest.raises(DeepFluidsHipError, match="workspace too small"):
        call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, 4 | (256 << 3), s)
    with pytest.raises(DeepFluidsHipError):
        call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, 5, s)       # algo out of range


@pytest.mark.parametrize("shape,C,up", [((1, 8, 12, 8), 32, False), ((2, 6, 10, 12), 32, False), ((1, 5, 7, 9), 64, False), ((1, 4, 6, 4), 32, True),
                                        ((2, 3, 5, 6), 64, True), ((1, 8, 12, 8), 128, True)])
def test_sign_bit_masks_are_bit_identical_to_activation_masks(ops, shape, C, up):
    """The masked dgrads of a fused generator block read the lrelu mask either from the fp32 activation of the layer below
    (DF_CONV_MASK + mask_src) or from the sign-bit words its forward conv emitted (df_wino_conv_fwd_bits / df_wino_upconv_fwd_bits:
    1/32 of the bytes).  Same arithmetic: outputs and every gradient must agree bit for bit -- full and ragged tile blocks, plain and
    up-sampling blocks."""
    from deep_fluids_amd.ops import _GenBlock, _UpGenBlock
    rng = np.random.RandomState(sum(shape) + C)
    n = 4
    x = rng.uniform(-1, 1, shape + (C,)).astype(np.float32)
    ws = [(rng.uniform(-1, 1, (3, 3, 3, C, C)) / np.sqrt(C * 27)).astype(np.float32) for _ in range(n)]
    bs = [rng.uniform(-0.3, 0.3, C).astype(np.float32) for _ in range(n)]
    fshape = tuple(shape[:1]) + tuple(2 * d for d in shape[1:]) if up else shape
    go = rng.uniform(-1, 1, fshape + (C,)).astype(np.float32)
    res = []
    for use_bits in (True, False):
        with ops.options(conv_algo="winograd", sign_bit_masks=use_bits):
            xt = dev(x).requires_grad_(True)
            args = []
            for w, b in zip(ws, bs):
                args += [dev(w).requires_grad_(True), dev(b).requires_grad_(True)]
            y = (_UpGenBlock if up else _GenBlock).apply(xt, 0.2, *args)
            (y * dev(go)).sum().backward()
            res.append([host(y), host(xt.grad)] + [host(a.grad) for a in args])
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("shape", [(2, 3, 5, 7, 8), (1, 1, 4, 6, 12), (3, 2, 2, 2, 128), (2, 8, 12, 8, 128)])
def test_lrelu_bwd_pool2x_equals_the_two_separate_passes(ops, shape):
    """df_lrelu_bwd_pool2x (the backward tail of an up-sampling generator block in one pass over dy) == df_lrelu_bwd + df_upsample2x_bwd,
    bit for bit, 3-D and 2-D (D = 1 shapes run both ways)."""
    from deep_fluids_amd._lib import call
    from deep_fluids_amd.ops import _ptr, _stream
    B, D, H, W, C = shape
    for is3d in ((1, 0) if D == 1 else (1,)):
        fD = 2 * D if is3d else 1
        g = torch.Generator(device="cuda").manual_seed(sum(shape) + is3d)
        dy = torch.rand((B, fD, 2 * H, 2 * W, C), device="cuda", generator=g) * 2 - 1
        y = torch.rand((B, fD, 2 * H, 2 * W, C), device="cuda", generator=g) * 2 - 1
        y[0, 0, 0, 0, :4] = 0.0                                                     # y == 0 takes the leak branch (y > 0 is false)
        gx1 = torch.empty_like(dy); p1 = torch.empty((B, D if is3d else 1, H, W, C), device="cuda")
        gx2 = torch.empty_like(dy); p2 = torch.empty_like(p1)
        call("df_lrelu_bwd_pool2x", _ptr(dy), _ptr(y), _ptr(gx1), _ptr(p1), 0.2, B, D if is3d else 1, H, W, C, is3d, _stream())
        call("df_lrelu_bwd", _ptr(dy), _ptr(y), _ptr(gx2), 0.2, dy.numel(), _stream())
        call("df_upsample2x_bwd", _ptr(dy), _ptr(p2), B, D if is3d else 1, H, W, C, is3d, _stream())
        assert torch.equal(gx1, gx2) and torch.equal(p1, p2)
        ref = torch.where(y > 0, dy, 0.2 * dy)
        assert torch.equal(gx1, ref)


@pytest.mark.parametrize("dims", [(2, 8, 16, 16), (1, 6, 10, 12), (3, 4, 8, 24)])
def test_block_tail_on_sign_bits_equals_the_fp32_mask_path(ops, dims):
    """df_wino_conv_fwd_addup_bits (no fp32 activation written, sign bits instead) + df_lrelu_bits_bwd_pool2x ==
    df_wino_conv_fwd_addup + df_lrelu_bwd_pool2x bit for bit; full and ragged tile blocks (4 x 8 x 8 voxels)."""
    from deep_fluids_amd._lib import call, query
    from deep_fluids_amd.ops import _ptr, _stream, _pack
    B, D, H, W = dims
    C = 64
    g = torch.Generator(device="cuda").manual_seed(sum(dims))
    x = torch.rand((B, D, H, W, C), device="cuda", generator=g) - 0.5
    xc = torch.rand((B, D // 2, H // 2, W // 2, C), device="cuda", generator=g) - 0.5
    w = (torch.rand((3, 3, 3, C, C), device="cuda", generator=g) - 0.5) * 0.1
    b = torch.rand(C, device="cuda", generator=g) - 0.5
    dy = torch.rand((B, D, H, W, C), device="cuda", generator=g) - 0.5
    wp = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wp), C, C, 0, _stream())
    y1 = torch.empty_like(x); y2a = torch.empty_like(x); y2b = torch.empty_like(x)
    call("df_wino_conv_fwd_addup", _ptr(x), _ptr(wp), _ptr(b), _ptr(xc), _ptr(y1), _ptr(y2a), B, D, H, W, C, C, 0.2, _stream())
    bits = torch.zeros(query("df_wino_signbits_bytes", B, D, H, W, C) // 8, dtype=torch.int64, device="cuda")
    call("df_wino_conv_fwd_addup_bits", _ptr(x), _ptr(wp), _ptr(b), _ptr(xc), _ptr(y2b), _ptr(bits), B, D, H, W, C, C, 0.2, _stream())
    assert torch.equal(y2a, y2b)
    gx1 = torch.empty_like(dy); p1 = torch.empty_like(xc); gx2 = torch.empty_like(dy); p2 = torch.empty_like(xc)
    call("df_lrelu_bwd_pool2x", _ptr(dy), _ptr(y1), _ptr(gx1), _ptr(p1), 0.2, B, D // 2, H // 2, W // 2, C, 1, _stream())
    call("df_lrelu_bits_bwd_pool2x", _ptr(dy), _ptr(bits), _ptr(gx2), _ptr(p2), 0.2, B, D // 2, H // 2, W // 2, C, _stream())
    assert torch.equal(gx1, gx2) and torch.equal(p1, p2)


@pytest.mark.parametrize("dims", [(8, 8, 6, 128, 128), (8, 16, 12, 128, 128), (3, 7, 5, 32, 96), (5, 16, 16, 64, 32), (1, 3, 3, 32, 32), (2, 9, 11, 96, 160)])
def test_tiny_2d_conv_kernel_vs_fp64_and_batch_invariance(ops, dims):
    """conv_tiny2d_kernel (conv.hip: the 2-D levels of at most 256 pixels per image -- 16 x 16 blocks on v_mfma_f32_16x16x4_f32, operands straight from L1 / L2)
    behind df_conv_fwd: forward and dgrad operands against an fp64 reference, every fused epilogue, pixel counts that do not fill the last 16-pixel tile, couts
    that do not fill a workgroup's 64 -- and the result of an image does not depend on the batch it sits in (the kernel is chosen by the image size alone)."""
    import torch.nn.functional as F
    from deep_fluids_amd._lib import call, query, DF_CONV_BIAS, DF_CONV_LRELU, DF_CONV_MASK, DF_CONV_RESIDUAL
    from deep_fluids_amd.ops import _ptr, _stream
    B, H, W, Ci, Co = dims
    rng = np.random.RandomState(sum(dims))
    s = _stream()
    x = dev(rng.uniform(-1, 1, (B, H, W, Ci)).astype(np.float32))
    w = dev((rng.uniform(-1, 1, (3, 3, Ci, Co)) / np.sqrt(9 * Ci)).astype(np.float32))
    bias = dev(rng.uniform(-0.5, 0.5, Co).astype(np.float32))
    aux = dev(rng.uniform(-1, 1, (B, H, W, Co)).astype(np.float32))
    wp = torch.empty(query("df_conv_packed_elems", 9, Ci, Co, 0), device="cuda")
    call("df_conv_pack_weights", _ptr(w), _ptr(wp), 9, Ci, Co, 0, s)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().permute(3, 2, 0, 1).cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
    y = torch.full((B, H, W, Co), float("nan"), device="cuda")
    call("df_conv_fwd", _ptr(x), _ptr(wp), _ptr(bias), None, None, _ptr(y), B, 1, H, W, Ci, Co, 1, DF_CONV_BIAS, 0.2, s)
    assert rel_l1(host(y), ref.numpy()) < 2e-6
    # epilogues: lrelu;  residual + mask on the bias-free launch
    y1 = torch.empty_like(y); y2 = torch.empty_like(y); y3 = torch.empty_like(y)
    call("df_conv_fwd", _ptr(x), _ptr(wp), _ptr(bias), None, None, _ptr(y1), B, 1, H, W, Ci, Co, 1, DF_CONV_BIAS | DF_CONV_LRELU, 0.2, s)
    assert torch.equal(y1, torch.maximum(y, 0.2 * y))
    call("df_conv_fwd", _ptr(x), _ptr(wp), None, None, None, _ptr(y2), B, 1, H, W, Ci, Co, 1, 0, 0.2, s)
    call("df_conv_fwd", _ptr(x), _ptr(wp), None, _ptr(aux), _ptr(aux), _ptr(y3), B, 1, H, W, Ci, Co, 1, DF_CONV_MASK | DF_CONV_RESIDUAL, 0.2, s)
    e = y2 + aux
    assert torch.equal(y3, torch.where(aux > 0, e, 0.2 * e))
    # dgrad operand (mode 1): dx = conv_transpose(g, w)
    g = dev(rng.uniform(-1, 1, (B, H, W, Co)).astype(np.float32))
    wd = torch.empty(query("df_conv_packed_elems", 9, Ci, Co, 1), device="cuda")
    call("df_conv_pack_weights", _ptr(w), _ptr(wd), 9, Ci, Co, 1, s)
    dx = torch.full((B, H, W, Ci), float("nan"), device="cuda")
    call("df_conv_fwd", _ptr(g), _ptr(wd), None, None, None, _ptr(dx), B, 1, H, W, Co, Ci, 1, 0, 0.2, s)
    dref = F.conv_transpose2d(g.double().permute(0, 3, 1, 2).cpu(), w.double().permute(3, 2, 0, 1).cpu(), padding=1).permute(0, 2, 3, 1)
    assert rel_l1(host(dx), dref.numpy()) < 2e-6
    # batch invariance: image B - 1 alone
    ys = torch.empty((1, H, W, Co), device="cuda")
    xs = x[B - 1:].contiguous()
    call("df_conv_fwd", _ptr(xs), _ptr(wp), _ptr(bias), None, None, _ptr(ys), 1, 1, H, W, Ci, Co, 1, DF_CONV_BIAS, 0.2, s)
    assert torch.equal(ys[0], y[B - 1])
