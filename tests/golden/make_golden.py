#!/usr/bin/env python3
"""Capture golden vectors by EXECUTING the reference's own ``ops.py`` / ``model.py``.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  TensorFlow 1.15 is not installable here, so a minimal NumPy-backed ``tensorflow``
stub is injected into ``sys.modules``:

* stencil functions (``curl``, ``jacobian``, ``jacobian3``, ``divergence``,
  ``divergence3``, ``pgrad``, ``lrelu``) use only Python slicing plus
  ``tf.concat / expand_dims / stack / transpose / maximum`` -> the reference's own source
  lines run unmodified on ndarrays;  the ``*_np`` twins run as-is.
* ``model.py`` (GeneratorBE / GeneratorBE3) runs unmodified too; the slim layer
  *arithmetic* (conv / fully_connected / nearest resize) is supplied by
  ``oracle/df_oracle.py`` (TF semantics restated; "parity unpinned" for that arithmetic),
  so these vectors pin the reference's GRAPH STRUCTURE: layer order and names, residual
  adds, up-sampling placement, reshape order.

Only numbers (inputs + outputs) are written; no reference source or bytecode is copied.

Usage:  python tests/golden/make_golden.py   (writes tests/golden/*.npz, *.json)
"""
import contextlib
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True   # never write __pycache__ into /root/reference

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import df_oracle as orc  # noqa: E402


class _Shape(object):
    def __init__(self, shp):
        self._s = [int(s) for s in shp]
        self.ndims = len(self._s)

    def as_list(self):
        return list(self._s)


class T(np.ndarray):
    """ndarray that also answers the TF tensor calls the reference makes."""

    def get_shape(self):
        return _Shape(self.shape)


def _t(a):
    return np.asarray(a).view(T)


PLAN = []          # layer plan recorded while model.py runs
WEIGHTS = {}       # injected variables, slim names
_SCOPE = []


def _scope_name(n):
    return "/".join(_SCOPE + [n])


@contextlib.contextmanager
def _variable_scope(name, reuse=False):
    _SCOPE.append(name)
    yield "/".join(_SCOPE)
    _SCOPE.pop()


_UNNAMED = {}


def _default_name(scope, kind):
    if scope is not None:
        return scope
    key = ("/".join(_SCOPE), kind)
    n = _UNNAMED.get(key, 0)
    _UNNAMED[key] = n + 1
    return kind if n == 0 else "%s_%d" % (kind, n)


def _conv(x, o_dim, k, stride=1, activation_fn=None, scope=None, data_format=None):
    full = _scope_name(_default_name(scope, "Conv"))
    w = WEIGHTS[full + "/weights"]; b = WEIGHTS[full + "/biases"]
    nd = x.ndim - 2
    assert w.shape == (k,) * nd + (x.shape[-1], o_dim), (full, w.shape)
    PLAN.append({"name": full, "kind": "conv%dd" % nd, "in": list(x.shape), "cout": int(o_dim), "k": int(k),
                 "stride": int(stride), "act": getattr(activation_fn, "__name__", None)})
    out = _t(orc.conv_same(np.asarray(x), w, b, stride))
    return activation_fn(out) if activation_fn is not None else out


def _fc(x, o_dim, activation_fn=None, scope=None):
    full = _scope_name(scope)
    w = WEIGHTS[full + "/weights"]; b = WEIGHTS[full + "/biases"]
    PLAN.append({"name": full, "kind": "fc", "in": list(x.shape), "cout": int(o_dim),
                 "act": getattr(activation_fn, "__name__", None)})
    out = _t(orc.linear(np.asarray(x), w, b))
    return activation_fn(out) if activation_fn is not None else out


def _resize_nn(x, new_size):
    x = np.asarray(x)
    h, w = x.shape[1:3]
    nh, nw = new_size
    iy = np.minimum((np.arange(nh) * h) // nh, h - 1)     # align_corners=False rule, SURVEY A.3
    ix = np.minimum((np.arange(nw) * w) // nw, w - 1)
    return _t(x[:, iy][:, :, ix])


def install_stub():
    tf = types.ModuleType("tensorflow")
    tf.float32 = np.float32
    tf.nn = types.SimpleNamespace(elu=lambda x: x)
    tf.concat = lambda xs, axis: _t(np.concatenate([np.asarray(x) for x in xs], axis=axis))
    tf.expand_dims = lambda x, axis: _t(np.expand_dims(np.asarray(x), axis))
    tf.stack = lambda xs, axis=0: _t(np.stack([np.asarray(x) for x in xs], axis=axis))
    tf.transpose = lambda x, perm: _t(np.transpose(np.asarray(x), perm))
    tf.maximum = lambda a, b: _t(np.maximum(a, b))
    tf.reshape = lambda x, shp: _t(np.reshape(np.asarray(x), shp))
    tf.sigmoid = lambda x: _t(1.0 / (1.0 + np.exp(-np.asarray(x))))
    tf.variable_scope = _variable_scope
    tf.image = types.SimpleNamespace(resize_nearest_neighbor=_resize_nn)
    contrib = types.ModuleType("tensorflow.contrib")
    slim = types.ModuleType("tensorflow.contrib.slim")
    slim.conv2d = _conv
    slim.conv3d = _conv
    slim.fully_connected = _fc
    contrib.slim = slim
    contrib.framework = types.SimpleNamespace(get_variables=lambda vs: sorted(k for k in WEIGHTS if k.startswith(vs)))
    tf.contrib = contrib
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.contrib"] = contrib
    sys.modules["tensorflow.contrib.slim"] = slim


def capture_stencils(ops):
    rng = np.random.RandomState(123)          # the reference's seed, config.py:69
    out = {}

    def rnd(*shape):
        return rng.uniform(-1, 1, size=shape).astype(np.float32)

    cases2 = {"a": (2, 8, 6), "edge2": (1, 2, 2), "tall": (3, 5, 2), "wide": (1, 2, 7)}
    for tag, (b, y, x) in cases2.items():
        s = rnd(b, y, x, 1); v = rnd(b, y, x, 2)
        out["curl_%s_in" % tag] = s
        out["curl_%s_out" % tag] = np.asarray(ops.curl(_t(s)))
        out["curl_np_%s_out" % tag] = ops.curl_np(s)
        out["grad_np_%s_out" % tag] = ops.grad_np(s)
        out["jacobian_%s_in" % tag] = v
        j, w = ops.jacobian(_t(v))
        out["jacobian_%s_j" % tag] = np.asarray(j); out["jacobian_%s_w" % tag] = np.asarray(w)
        out["vort_np_%s_out" % tag] = ops.vort_np(v)
        out["pgrad_%s_out" % tag] = np.asarray(ops.pgrad(_t(s), "NHWC"))
        if y > 1 and x > 1:
            out["divergence_%s_out" % tag] = np.asarray(ops.divergence(_t(v)))
    # NCHW entry points (transposes in/out, ops.py:206-207,222-224,265,273)
    s = rnd(2, 1, 8, 6); v = rnd(2, 2, 8, 6)
    out["curl_nchw_in"] = s; out["curl_nchw_out"] = np.asarray(ops.curl(_t(s), data_format="NCHW"))
    out["pgrad_nchw_out"] = np.asarray(ops.pgrad(_t(s), "NCHW"))
    out["jacobian_nchw_in"] = v
    j, w = ops.jacobian(_t(v), data_format="NCHW")
    out["jacobian_nchw_j"] = np.asarray(j); out["jacobian_nchw_w"] = np.asarray(w)

    cases3 = {"a": (2, 4, 6, 5), "edge2": (1, 2, 2, 2), "slab": (1, 2, 5, 3), "b": (1, 6, 3, 8)}
    for tag, (b, z, y, x) in cases3.items():
        v = rnd(b, z, y, x, 3)
        out["jacobian3_%s_in" % tag] = v
        j, c = ops.jacobian3(_t(v))
        out["jacobian3_%s_j" % tag] = np.asarray(j); out["jacobian3_%s_c" % tag] = np.asarray(c)
        jn, cn = ops.jacobian_np3(v)
        out["jacobian_np3_%s_j" % tag] = jn; out["jacobian_np3_%s_c" % tag] = cn
        out["divergence3_%s_out" % tag] = np.asarray(ops.divergence3(_t(v)))
    # the composite the 3-D trainer builds (trainer3.py:18,24): J(curl3(psi))
    psi = rnd(2, 4, 6, 5, 3)
    _, u = ops.jacobian3(_t(psi))
    ju, wu = ops.jacobian3(u)
    out["composite3_psi"] = psi; out["composite3_u"] = np.asarray(u)
    out["composite3_ju"] = np.asarray(ju); out["composite3_div"] = np.asarray(ops.divergence3(u))
    s = rnd(2, 8, 6, 1)
    u2 = ops.curl(_t(s)); j2, w2 = ops.jacobian(u2)
    out["composite2_psi"] = s; out["composite2_u"] = np.asarray(u2)
    out["composite2_ju"] = np.asarray(j2); out["composite2_div"] = np.asarray(ops.divergence(u2))
    x = rnd(3, 4, 5)
    out["lrelu_in"] = x; out["lrelu_out"] = np.asarray(ops.lrelu(_t(x)))
    x = rnd(1, 2, 3, 2, 4)
    out["upscale3_in"] = x; out["upscale3_out"] = np.asarray(ops.upscale3(_t(x), 2))
    x = rnd(2, 3, 2, 4)
    out["upscale_in"] = x; out["upscale_out"] = np.asarray(ops.upscale(_t(x), 2))
    np.savez_compressed(os.path.join(HERE, "stencils.npz"), **out)
    print("stencils.npz: %d arrays" % len(out))


def capture_generators(model):
    plans = {}
    res = {}
    cases = {
        # tag: (fn, c_num, output_shape, filters, batch)
        "g2_small": ("GeneratorBE", 3, [16, 8, 1], 8, 2),
        "g3_small": ("GeneratorBE3", 3, [8, 16, 8, 3], 8, 2),
        "g3_odd": ("GeneratorBE3", 2, [12, 8, 4, 3], 4, 1),
        # skip_concat=True (model.py:30-33 / :72-75; never enabled by the reference trainers): concat skips instead of residual adds
        "g2_skip": ("GeneratorBE", 3, [16, 8, 1], 8, 2),
        "g3_skip": ("GeneratorBE3", 3, [8, 16, 8, 3], 8, 1),
    }
    for tag, (fn, c_num, oshape, filters, batch) in cases.items():
        rng = np.random.RandomState(123)
        WEIGHTS.clear(); del PLAN[:]
        skip = tag.endswith("_skip")
        WEIGHTS.update(orc.generator_init(rng, c_num, oshape, filters, skip_concat=skip))
        for k in list(WEIGHTS):                      # non-zero biases so bias handling is exercised
            if k.endswith("biases"):
                WEIGHTS[k] = rng.uniform(-0.1, 0.1, size=WEIGHTS[k].shape).astype(np.float32)
        z = rng.uniform(-1, 1, size=(batch, c_num)).astype(np.float32)
        out, var_names = getattr(model, fn)(_t(z), filters, oshape, skip_concat=skip)
        res[tag + "_z"] = z; res[tag + "_out"] = np.asarray(out)
        for k, v in WEIGHTS.items():
            res[tag + "|" + k] = v
        plans[tag] = {"fn": fn, "c_num": c_num, "output_shape": oshape, "filters": filters, "skip_concat": skip,
                      "layers": list(PLAN), "variables": list(var_names)}
    np.savez_compressed(os.path.join(HERE, "generators.npz"), **res)

    # layer plans + parameter counts at the BASELINE shapes (shape-only run: tiny batch, no arithmetic kept)
    for tag, (fn, oshape) in {"cfg2_2d_128x96": ("GeneratorBE", [128, 96, 1]),
                              "cfg3_3d_64x96x64": ("GeneratorBE3", [64, 96, 64, 3]),
                              "cfg4_3d_112x160x112": ("GeneratorBE3", [112, 160, 112, 3])}.items():
        rep, x0, nl = orc.generator_plan(oshape, 128)
        nd = len(oshape) - 1
        n_params = 3 * int(np.prod(x0)) + int(np.prod(x0)) + rep * 4 * (3 ** nd * 128 * 128 + 128) \
            + 3 ** nd * 128 * oshape[-1] + oshape[-1]
        plans[tag] = {"fn": fn, "output_shape": oshape, "filters": 128, "repeat_num": rep, "x0_shape": x0,
                      "n_layers": nl, "n_params": n_params}
    # auto-encoders (model.py:118-216): encoder with concat skips + stride-2 convs, decoder = GeneratorBE(3)
    ae = {}
    for tag, (fn, xshape, filters, z_num, batch, sparse) in {
            "ae3_small": ("AE3", [8, 16, 8, 3], 4, 6, 2, False),
            "ae2_small": ("AE", [16, 8, 1], 4, 5, 2, True)}.items():
        rng = np.random.RandomState(123)
        WEIGHTS.clear(); del PLAN[:]
        WEIGHTS.update(orc.ae_init(rng, xshape, filters, z_num))
        for k in list(WEIGHTS):
            if k.endswith("biases"):
                WEIGHTS[k] = rng.uniform(-0.1, 0.1, size=WEIGHTS[k].shape).astype(np.float32)
        x = rng.uniform(-1, 1, size=[batch] + xshape).astype(np.float32)
        out, z, var_names = getattr(model, fn)(_t(x), filters, z_num, use_sparse=sparse)
        ae[tag + "_x"] = x; ae[tag + "_out"] = np.asarray(out); ae[tag + "_z"] = np.asarray(z)
        for k, v in WEIGHTS.items():
            ae[tag + "|" + k] = v
        plans[tag] = {"fn": fn, "x_shape": xshape, "filters": filters, "z_num": z_num, "use_sparse": sparse,
                      "layers": list(PLAN), "variables": list(var_names)}
    for tag, (fn, xshape, filters, batch) in {"d2_small": ("DiscriminatorPatch", [16, 16, 3], 8, 2),
                                              "d3_small": ("DiscriminatorPatch3", [8, 16, 8, 6], 8, 1)}.items():
        rng = np.random.RandomState(123)
        WEIGHTS.clear(); del PLAN[:]; _UNNAMED.clear()
        WEIGHTS.update(orc.discriminator_init(rng, xshape[-1], filters, len(xshape) - 1))
        for k in list(WEIGHTS):
            if k.endswith("biases"):
                WEIGHTS[k] = rng.uniform(-0.1, 0.1, size=WEIGHTS[k].shape).astype(np.float32)
        x = rng.uniform(-1, 1, size=[batch] + xshape).astype(np.float32)
        out, var_names = getattr(model, fn)(_t(x), filters)
        ae[tag + "_x"] = x; ae[tag + "_out"] = np.asarray(out)
        for k, v in WEIGHTS.items():
            ae[tag + "|" + k] = v
        plans[tag] = {"fn": fn, "x_shape": xshape, "filters": filters, "layers": list(PLAN), "variables": list(var_names)}
    np.savez_compressed(os.path.join(HERE, "autoencoders.npz"), **ae)

    with open(os.path.join(HERE, "layer_plans.json"), "w") as f:
        json.dump(plans, f, indent=1, sort_keys=True)
    print("generators.npz: %d arrays; layer_plans.json: %s" % (len(res), sorted(plans)))


def capture_generator_args(model):
    """Non-default generator arguments the surface accepts (model.py:5-6,48-49; config.py:20,22 expose --repeat / --num_conv): num_conv,
    repeat, conv_k, last_k.  Separate files (generators_args.npz, layer_plans_args.json) so that the round-1 fixtures stay byte-identical."""
    cases = {
        # tag: (fn, c_num, output_shape, filters, batch, kwargs)
        "g3_nc2_rep3": ("GeneratorBE3", 3, [8, 16, 8, 3], 8, 2, dict(num_conv=2, repeat=3)),
        "g2_nc5_lastk1": ("GeneratorBE", 3, [16, 8, 1], 8, 2, dict(num_conv=5, last_k=1)),
        "g3_nc3_convk5": ("GeneratorBE3", 2, [8, 8, 8, 3], 4, 1, dict(num_conv=3, conv_k=5)),
        "g2_convk5_lastk1_rep2": ("GeneratorBE", 3, [16, 8, 2], 8, 2, dict(conv_k=5, last_k=1, repeat=2)),
    }
    res, plans = {}, {}
    for tag, (fn, c_num, oshape, filters, batch, kw) in cases.items():
        rng = np.random.RandomState(321)
        WEIGHTS.clear(); del PLAN[:]
        WEIGHTS.update(orc.generator_init(rng, c_num, oshape, filters, **kw))
        for k in list(WEIGHTS):
            if k.endswith("biases"):
                WEIGHTS[k] = rng.uniform(-0.1, 0.1, size=WEIGHTS[k].shape).astype(np.float32)
        z = rng.uniform(-1, 1, size=(batch, c_num)).astype(np.float32)
        out, var_names = getattr(model, fn)(_t(z), filters, oshape, **kw)
        res[tag + "_z"] = z; res[tag + "_out"] = np.asarray(out)
        for k, v in WEIGHTS.items():
            res[tag + "|" + k] = v
        plans[tag] = {"fn": fn, "c_num": c_num, "output_shape": oshape, "filters": filters, "kwargs": kw,
                      "layers": list(PLAN), "variables": list(var_names)}
    np.savez_compressed(os.path.join(HERE, "generators_args.npz"), **res)
    with open(os.path.join(HERE, "layer_plans_args.json"), "w") as f:
        json.dump(plans, f, indent=1, sort_keys=True)
    print("generators_args.npz: %d arrays; layer_plans_args.json: %s" % (len(res), sorted(plans)))


def main():
    install_stub()
    sys.path.insert(0, REF)
    import ops      # the reference's own module
    import model    # the reference's own module
    capture_stencils(ops)
    capture_generators(model)
    capture_generator_args(model)


if __name__ == "__main__":
    main()
