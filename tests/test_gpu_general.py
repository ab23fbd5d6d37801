"""-m gpu: the WRAPPER GENERALITY of the reference's call surface that its trainers never use (ops.py:12-16: `conv2d / conv3d` default to
k=4, s=2; any cubic kernel / stride / extent is legal; ops.py:66-73: `resize_nearest_neighbor` to any size) -- the general-shape kernels of
conv_general.hip through the public wrappers, forward and every gradient against the fp64 oracle (TF 'SAME' padding, pad_before =
pad_total // 2; TF1 nearest index rule src = floor(dst in / out))."""
import numpy as np
import pytest
import torch

import df_oracle as orc
from gpu_util import dev, host, rel_linf

pytestmark = pytest.mark.gpu
TOL = 2e-5

CASES = [
    # (input shape without channels, cin, cout, k, s, act)
    ((2, 16, 12), 3, 8, 4, 2, True),          # the wrapper's own defaults (k=4, s=2), 2-D
    ((1, 8, 6, 10), 4, 6, 4, 2, True),        # ... 3-D
    ((2, 9, 7), 5, 4, 3, 2, True),            # k=3, s=2 on ODD extents (the fast stride-2 kernel asserts even ones)
    ((1, 5, 7, 9), 3, 5, 3, 2, False),
    ((1, 7, 5), 2, 3, 5, 1, True),            # k=5, s=1
    ((1, 6, 6, 6), 2, 2, 1, 1, False),        # 1x1x1
    ((2, 11, 13), 3, 4, 2, 3, True),          # even kernel, stride 3: asymmetric SAME pads
    ((1, 4, 9, 5), 2, 3, 4, 3, True),
    ((1, 3, 4), 1, 70, 7, 4, False),          # kernel larger than the image, Cout > 64 (two channel passes of the weight gradient)
]


@pytest.mark.parametrize("shape,cin,cout,k,s,act", CASES)
def test_conv_wrappers_any_kernel_and_stride_vs_oracle(shape, cin, cout, k, s, act):
    from deep_fluids_amd import ops
    nd = len(shape) - 1
    rng = np.random.RandomState(sum(shape) + cin + cout + 7 * k + s)
    x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
    w = (rng.uniform(-1, 1, (k,) * nd + (cin, cout)) / np.sqrt(cin * k ** nd)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    ops.reset_variables()
    wt, bt = ops.set_variable("T/c/weights", w), ops.set_variable("T/c/biases", b)
    xt = dev(x).requires_grad_(True)
    counts = {}
    with ops.options(dispatch_counts=counts), ops.variable_scope("T", reuse=True):
        conv = ops.conv3d if nd == 3 else ops.conv2d
        kw = {} if (k, s) == (4, 2) else dict(k=k, s=s)            # (4, 2): exercised as the wrapper's DEFAULTS
        y = conv(xt, cout, name="c", act=ops.lrelu if act else None, **kw)
    assert any("general-valu k%d s%d" % (k, s) in key for key in counts), counts
    x64, w64, b64 = x.astype(np.float64), w.astype(np.float64), b.astype(np.float64)
    pre = orc.conv_same(x64, w64, b64, stride=s)
    ref = orc.lrelu(pre) if act else pre
    assert tuple(y.shape) == ref.shape
    go = rng.uniform(-1, 1, ref.shape).astype(np.float32)
    (y * dev(go)).sum().backward()
    dpre = go * (np.where(ref > 0, 1.0, 0.2) if act else 1.0)
    dx, dw, db = orc.conv_same_bwd(x64, w64, dpre, stride=s)
    errs = {"y": rel_linf(host(y), ref), "dx": rel_linf(host(xt.grad), dx), "dw": rel_linf(host(wt.grad), dw), "db": rel_linf(host(bt.grad), db)}
    assert max(errs.values()) < TOL, errs
    ops.reset_variables()


def test_fast_call_sites_do_not_take_the_general_kernels():
    from deep_fluids_amd import ops
    ops.reset_variables()
    counts = {}
    with ops.options(dispatch_counts=counts):
        ops.conv3d(torch.zeros((1, 4, 8, 8, 32), device="cuda"), 32, k=3, s=1, name="a")
        ops.conv3d(torch.zeros((1, 4, 8, 8, 32), device="cuda"), 32, k=3, s=2, name="b")
        ops.conv2d(torch.zeros((1, 8, 8, 32), device="cuda"), 32, k=3, s=1, name="c")
    assert counts and not any("general-valu" in k for k in counts), counts
    ops.reset_variables()


@pytest.mark.parametrize("shape,new", [((2, 5, 7, 3), (11, 4)), ((1, 8, 6, 2), (8, 6)), ((1, 4, 4, 5), (12, 12)), ((2, 6, 9, 1), (4, 3)),
                                       ((1, 3, 4, 5, 2), (7, 4, 13)), ((2, 2, 3, 2, 4), (6, 9, 6))])
def test_resize_nearest_neighbor_any_size_vs_oracle(shape, new):
    from deep_fluids_amd import ops
    from deep_fluids_amd.ops import _ResizeNN
    rng = np.random.RandomState(sum(shape) + sum(new))
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = ops.resize_nearest_neighbor(xt, new) if len(shape) == 4 else _ResizeNN.apply(xt, new)
    ref = orc.resize_nn(x, new)
    np.testing.assert_array_equal(host(y), ref)
    go = rng.uniform(-1, 1, ref.shape).astype(np.float32)
    (y * dev(go)).sum().backward()
    assert rel_linf(host(xt.grad), orc.resize_nn_bwd(go.astype(np.float64), x.shape)) < 2e-6


def test_upscale_by_other_integer_factors():
    from deep_fluids_amd import ops
    rng = np.random.RandomState(5)
    x2 = rng.uniform(-1, 1, (2, 4, 5, 3)).astype(np.float32)
    np.testing.assert_array_equal(host(ops.upscale(dev(x2), 3)), orc.upscale_nn(x2, 3))
    np.testing.assert_array_equal(host(ops.upscale(dev(x2), 2)), orc.upscale_nn(x2, 2))
    np.testing.assert_array_equal(host(ops.upscale(dev(x2), 2, data_format="NHWC")), host(ops.resize_nearest_neighbor(dev(x2), (8, 10))))
    x3 = rng.uniform(-1, 1, (1, 2, 3, 4, 2)).astype(np.float32)
    np.testing.assert_array_equal(host(ops.upscale3(dev(x3), 3)), orc.upscale_nn(x3, 3))
    np.testing.assert_array_equal(host(ops.upscale3(dev(x3), 1)), x3)


def test_random_shapes_property_sweep():
    """A seeded random sweep over (dimension, extents, channels, kernel, stride): general conv forward / gradients and the stencils against the
    oracle on shapes nobody hand-picked (40 cases, a few ms each)."""
    from deep_fluids_amd import ops
    from deep_fluids_amd.ops import _ConvGeneral
    rng = np.random.RandomState(2024)
    for case in range(40):
        nd = int(rng.randint(2, 4))
        shape = (int(rng.randint(1, 3)),) + tuple(int(rng.randint(2, 10)) for _ in range(nd))
        cin, cout = int(rng.randint(1, 7)), int(rng.randint(1, 7))
        k, s = int(rng.randint(1, 6)), int(rng.randint(1, 4))
        x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
        w = (rng.uniform(-1, 1, (k,) * nd + (cin, cout)) / np.sqrt(cin * k ** nd)).astype(np.float32)
        b = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
        xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
        y = _ConvGeneral.apply(xt, wt, bt, None, k, s)
        ref = orc.conv_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), stride=s)
        go = rng.uniform(-1, 1, ref.shape).astype(np.float32)
        (y * dev(go)).sum().backward()
        dx, dw, db = orc.conv_same_bwd(x.astype(np.float64), w.astype(np.float64), go.astype(np.float64), stride=s)
        tag = (case, shape, cin, cout, k, s)
        assert rel_linf(host(y), ref) < TOL and rel_linf(host(xt.grad), dx) < TOL, tag
        assert rel_linf(host(wt.grad), dw) < TOL and rel_linf(host(bt.grad), db) < TOL, tag
        # stencils on the same random extents (>= 2 per axis): bit-exact
        if nd == 3:
            v = rng.uniform(-1, 1, shape + (3,)).astype(np.float32)
            j, c = ops.jacobian3(dev(v))
            oj, oc = orc.jacobian3(v)
            np.testing.assert_array_equal(host(j), oj); np.testing.assert_array_equal(host(c), oc)
        else:
            p = rng.uniform(-1, 1, shape + (1,)).astype(np.float32)
            np.testing.assert_array_equal(host(ops.curl(dev(p))), orc.curl(p))
