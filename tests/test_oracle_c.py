"""The oracle's plain-C restatement (oracle/df_oracle.c) against the golden vectors captured from the reference's
ops.py and against the NumPy oracle (CPU only)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import df_oracle as orc
from conftest import ROOT

LIB = os.environ.get("DF_ORACLE_LIBRARY") or os.path.join(ROOT, "oracle", "libdf_oracle.so")      # (tools/run_asan.sh: the ASAN build)
I64 = ctypes.c_int64


@pytest.fixture(scope="module")
def clib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    h = ctypes.CDLL(LIB)
    h.dfo_l1_mean.restype = ctypes.c_double
    return h


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("tag", ["a", "edge2", "slab", "b"])
def test_c_jacobian3_golden(clib, golden_stencils, tag):
    x = np.ascontiguousarray(golden_stencils["jacobian3_%s_in" % tag])
    B, Z, Y, X, _ = x.shape
    j = np.empty((B, Z, Y, X, 9), np.float32); c = np.empty((B, Z, Y, X, 3), np.float32)
    clib.dfo_jacobian3d(_p(x), _p(j), _p(c), I64(B), I64(Z), I64(Y), I64(X))
    np.testing.assert_array_equal(j, golden_stencils["jacobian3_%s_j" % tag])
    np.testing.assert_array_equal(c, golden_stencils["jacobian3_%s_c" % tag])


@pytest.mark.parametrize("tag", ["a", "edge2", "tall", "wide"])
def test_c_2d_golden(clib, golden_stencils, tag):
    s = np.ascontiguousarray(golden_stencils["curl_%s_in" % tag])
    B, Y, X, _ = s.shape
    u = np.empty((B, Y, X, 2), np.float32)
    clib.dfo_curl2d(_p(s), _p(u), I64(B), I64(Y), I64(X))
    np.testing.assert_array_equal(u, golden_stencils["curl_%s_out" % tag])
    v = np.ascontiguousarray(golden_stencils["jacobian_%s_in" % tag])
    j = np.empty((B, Y, X, 4), np.float32); w = np.empty((B, Y, X, 1), np.float32)
    clib.dfo_jacobian2d(_p(v), _p(j), _p(w), I64(B), I64(Y), I64(X))
    np.testing.assert_array_equal(j, golden_stencils["jacobian_%s_j" % tag])
    np.testing.assert_array_equal(w, golden_stencils["jacobian_%s_w" % tag])


def test_c_conv_and_l1_vs_numpy(clib):
    rng = np.random.RandomState(3)
    for shape, kz in (((2, 3, 4, 5), 3), ((2, 1, 6, 7), 1)):
        cin, cout = 5, 4
        x = rng.uniform(-1, 1, shape + (cin,)).astype(np.float32)
        w = rng.uniform(-1, 1, (kz, 3, 3, cin, cout)).astype(np.float32)
        b = rng.uniform(-1, 1, cout).astype(np.float32)
        y = np.empty(shape + (cout,), np.float32)
        clib.dfo_conv_same(_p(x), _p(w), _p(b), _p(y), *[I64(s) for s in shape], I64(cin), I64(cout), kz, 1,
                           ctypes.c_float(0.2))
        xn = x if kz == 3 else x[:, 0]
        wn = w if kz == 3 else w[0]
        ref = orc.lrelu(orc.conv_same(xn.astype(np.float64), wn.astype(np.float64), b.astype(np.float64)))
        np.testing.assert_allclose(y.reshape(ref.shape), ref, rtol=0, atol=2e-6)
    a = rng.uniform(-1, 1, 1001).astype(np.float32); c = rng.uniform(-1, 1, 1001).astype(np.float32)
    assert abs(clib.dfo_l1_mean(_p(a), _p(c), I64(1001)) - np.abs(a.astype(np.float64) - c).mean()) < 1e-12
