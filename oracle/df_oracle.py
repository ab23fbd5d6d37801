"""CPU oracle for the Deep Fluids velocity-field train step  --  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``deep_fluids_amd``) must never route through it.

It is a NumPy restatement (index form, written from SURVEY.md Appendix A, not copied)
of the reference algorithm on the north-star path.  Each function cites the reference
file:line it follows (paths relative to the upstream repo root).

Parity status
-------------
* Stencils (curl / jacobian / jacobian3 / divergence / divergence3 and the *_np twins):
  PINNED -- checked bit-exactly against golden vectors captured by executing the
  reference's own ``ops.py`` source under a NumPy-backed ``tensorflow`` stub
  (``tests/golden/make_golden.py`` -> ``tests/golden/stencils_*.npz``).
* Generator graph structure (layer order, residual adds, up-sampling placement, variable
  names): PINNED -- the reference's own ``model.py`` is executed under the same stub with
  the slim layer arithmetic supplied by this module (``tests/golden/generator_*.npz``).
* conv / fully-connected / nearest-resize / Adam arithmetic: lives in TensorFlow 1.15,
  which is absent here and not installable -> "parity unpinned" for that arithmetic.
  It follows the published TF-1.15 semantics listed in SURVEY.md A.3-A.5 and is
  cross-checked against PyTorch-CPU (``oracle/df_oracle_torch.py``) as an independent
  second implementation.
"""
from __future__ import annotations

import math
import numpy as np

# ----------------------------------------------------------------------------------------
# Forward-difference stencils (ops.py:205-290)
# ----------------------------------------------------------------------------------------

def fdiff(f, axis):
    """D_a f: forward difference along ``axis``; the LAST DIFFERENCE (not the last value)
    is replicated so the output keeps the input extent.  ops.py:214-217, 243-253, 269-270."""
    f = np.asarray(f)
    n = f.shape[axis]
    if n < 2:
        raise ValueError("forward difference needs extent >= 2 along axis %d" % axis)
    d = np.diff(f, axis=axis)
    last = np.take(d, [n - 2], axis=axis)
    return np.concatenate([d, last], axis=axis)


def fdiff_adj(g, axis):
    """Adjoint of :func:`fdiff`: fold the replicated row into the
    difference it copies, then apply the transpose of the plain difference."""
    g = np.moveaxis(np.asarray(g), axis, 0)
    n = g.shape[0]
    gp = g[: n - 1].copy()
    gp[n - 2] += g[n - 1]
    out = np.empty_like(g)
    out[0] = -gp[0]
    out[1 : n - 1] = gp[: n - 2] - gp[1 : n - 1]
    out[n - 1] = gp[n - 2]
    return np.moveaxis(out, 0, axis)


def curl(x):
    """2-D curl of a scalar stream function, NHWC.  ops.py:264-274.
    x [B,Y,X,1] -> [B,Y,X,2] with u = D_y psi, v = -D_x psi."""
    x = np.asarray(x)
    psi = x[..., 0]
    u = fdiff(psi, 1)
    v = -fdiff(psi, 2)
    return np.stack([u, v], axis=-1)


def curl_bwd(g):
    """Adjoint of :func:`curl`: g [B,Y,X,2] -> dpsi [B,Y,X,1]."""
    g = np.asarray(g)
    d = fdiff_adj(g[..., 0], 1) - fdiff_adj(g[..., 1], 2)
    return d[..., None]


def jacobian(x):
    """2-D Jacobian + vorticity, NHWC.  ops.py:205-225.
    x [B,Y,X,2] -> j [B,Y,X,4] = (dudx,dudy,dvdx,dvdy), w [B,Y,X,1] = dvdx-dudy."""
    x = np.asarray(x)
    u, v = x[..., 0], x[..., 1]
    dudx, dudy = fdiff(u, 2), fdiff(u, 1)
    dvdx, dvdy = fdiff(v, 2), fdiff(v, 1)
    j = np.stack([dudx, dudy, dvdx, dvdy], axis=-1)
    w = (dvdx - dudy)[..., None]
    return j, w


def jacobian_bwd(gj, gw=None):
    """Adjoint of :func:`jacobian`: (gj [..,4], gw [..,1] or None) -> dx [..,2]."""
    gj = np.asarray(gj)
    g = [gj[..., i] for i in range(4)]
    if gw is not None:
        gw = np.asarray(gw)[..., 0]
        g[2] = g[2] + gw
        g[1] = g[1] - gw
    du = fdiff_adj(g[0], 2) + fdiff_adj(g[1], 1)
    dv = fdiff_adj(g[2], 2) + fdiff_adj(g[3], 1)
    return np.stack([du, dv], axis=-1)


# axis numbers of a [B,Z,Y,X,C] tensor ("x: bzyxd", ops.py:228)
_AX3 = {"x": 3, "y": 2, "z": 1}


def jacobian3(x):
    """3-D Jacobian + curl, NDHWC.  ops.py:227-262.
    x [B,Z,Y,X,3] -> j [..,9] = (dudx,dudy,dudz,dvdx,dvdy,dvdz,dwdx,dwdy,dwdz),
                     c [..,3] = (dwdy-dvdz, dudz-dwdx, dvdx-dudy)."""
    x = np.asarray(x)
    d = {}
    for ci, cn in enumerate("uvw"):
        for an in "xyz":
            d[cn + an] = fdiff(x[..., ci], _AX3[an])
    j = np.stack([d["ux"], d["uy"], d["uz"], d["vx"], d["vy"], d["vz"],
                  d["wx"], d["wy"], d["wz"]], axis=-1)
    c = np.stack([d["wy"] - d["vz"], d["uz"] - d["wx"], d["vx"] - d["uy"]], axis=-1)
    return j, c


def curl3(x):
    """North-star alias: ``jacobian3(x)[1]`` (trainer3.py:18 consumes only c)."""
    return jacobian3(x)[1]


def jacobian3_bwd(gj=None, gc=None):
    """Adjoint of :func:`jacobian3`: (gj [..,9] or None, gc [..,3] or None) -> dx [..,3]."""
    ref = gj if gj is not None else gc
    shp = np.asarray(ref).shape[:-1]
    dt = np.asarray(ref).dtype
    names = ["ux", "uy", "uz", "vx", "vy", "vz", "wx", "wy", "wz"]
    g = {n: np.zeros(shp, dt) for n in names}
    if gj is not None:
        gj = np.asarray(gj)
        for i, n in enumerate(names):
            g[n] = g[n] + gj[..., i]
    if gc is not None:
        gc = np.asarray(gc)
        g["wy"] = g["wy"] + gc[..., 0]; g["vz"] = g["vz"] - gc[..., 0]
        g["uz"] = g["uz"] + gc[..., 1]; g["wx"] = g["wx"] - gc[..., 1]
        g["vx"] = g["vx"] + gc[..., 2]; g["uy"] = g["uy"] - gc[..., 2]
    out = []
    for cn in "uvw":
        acc = 0
        for an in "xyz":
            acc = acc + fdiff_adj(g[cn + an], _AX3[an])
        out.append(acc)
    return np.stack(out, axis=-1)


def divergence(x):
    """Interior forward-difference divergence, output shrinks by one per axis. ops.py:276-284."""
    x = np.asarray(x)
    dudx = x[:, :-1, 1:, 0] - x[:, :-1, :-1, 0]
    dvdy = x[:, 1:, :-1, 1] - x[:, :-1, :-1, 1]
    return (dudx + dvdy)[..., None]


def divergence3(x):
    """ops.py:286-290."""
    x = np.asarray(x)
    dudx = x[:, :-1, :-1, 1:, 0] - x[:, :-1, :-1, :-1, 0]
    dvdy = x[:, :-1, 1:, :-1, 1] - x[:, :-1, :-1, :-1, 1]
    dwdz = x[:, 1:, :-1, :-1, 2] - x[:, :-1, :-1, :-1, 2]
    return (dudx + dvdy + dwdz)[..., None]


def pgrad(x):
    """ops.py:292-303: pressure gradient (D_x p, D_y p) of channel 0, last difference replicated; x [B,Y,X,C] -> [B,Y,X,2]."""
    p = np.asarray(x)[..., 0]
    return np.stack([fdiff(p, 2), fdiff(p, 1)], axis=-1)


def vort_np(x):
    """ops.py:305-310."""
    x = np.asarray(x)
    return (fdiff(x[..., 1], 2) - fdiff(x[..., 0], 1))[..., None]


def grad_np(x):
    """ops.py:319-324 (pressure gradient)."""
    x = np.asarray(x)
    return np.stack([fdiff(x[..., 0], 2), fdiff(x[..., 0], 1)], axis=-1)


# ----------------------------------------------------------------------------------------
# Losses (trainer.py:170-172, trainer3.py:49-51)
# ----------------------------------------------------------------------------------------

def l1_mean(a, b):
    """mean(|a-b|) over all elements.  trainer.py:170-171."""
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).mean(dtype=np.float64 if a.dtype == np.float64 else a.dtype)


def l1_mean_bwd(a, b, gscale=1.0):
    """d/da mean|a-b| = sign(a-b)/N  (TF Abs grad uses sign; 0 at 0)."""
    a = np.asarray(a); b = np.asarray(b)
    return (np.sign(a - b) * (gscale / a.size)).astype(a.dtype)


# ----------------------------------------------------------------------------------------
# Layers (ops.py:9-24, 66-91; TF-1.15 slim semantics SURVEY.md A.3)
# ----------------------------------------------------------------------------------------

def lrelu(x, leak=0.2):
    """ops.py:9-10."""
    return np.maximum(x, leak * x)


def _same_pads(n, k, s):
    o = -(-n // s)
    pt = max((o - 1) * s + k - n, 0)
    return o, pt // 2, pt - pt // 2


def conv_same(x, w, b=None, stride=1):
    """slim.conv2d / conv3d, channels-last, padding='SAME' (ops.py:12-16).
    x [B,*S,Cin], w [*k,Cin,Cout] (HWIO / DHWIO), b [Cout] or None."""
    x = np.asarray(x); w = np.asarray(w)
    nd = x.ndim - 2
    k = w.shape[:nd]
    geo = [_same_pads(x.shape[1 + a], k[a], stride) for a in range(nd)]
    outs = [g[0] for g in geo]
    xp = np.pad(x, [(0, 0)] + [(g[1], g[2]) for g in geo] + [(0, 0)])
    out = np.zeros([x.shape[0]] + outs + [w.shape[-1]], dtype=np.result_type(x, w))
    for tap in np.ndindex(*k):
        sl = tuple(slice(t, t + (o - 1) * stride + 1, stride) for t, o in zip(tap, outs))
        out += xp[(slice(None),) + sl] @ w[tap]
    if b is not None:
        out += b
    return out


def conv_same_bwd(x, w, dout, stride=1, need_dx=True):
    """Backward of :func:`conv_same`: returns (dx, dw, db)."""
    x = np.asarray(x); w = np.asarray(w); dout = np.asarray(dout)
    nd = x.ndim - 2
    k = w.shape[:nd]
    geo = [_same_pads(x.shape[1 + a], k[a], stride) for a in range(nd)]
    outs = [g[0] for g in geo]
    pads = [(0, 0)] + [(g[1], g[2]) for g in geo] + [(0, 0)]
    xp = np.pad(x, pads)
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    d2 = dout.reshape(-1, w.shape[-1])
    for tap in np.ndindex(*k):
        sl = (slice(None),) + tuple(slice(t, t + (o - 1) * stride + 1, stride) for t, o in zip(tap, outs))
        dw[tap] = xp[sl].reshape(-1, w.shape[-2]).T @ d2
        if need_dx:
            dxp[sl] += dout @ w[tap].T
    unpad = (slice(None),) + tuple(slice(g[1], g[1] + x.shape[1 + a]) for a, g in enumerate(geo))
    db = d2.sum(axis=0)
    return (dxp[unpad] if need_dx else None), dw, db


def linear(x, w, b=None):
    """slim.fully_connected, W [in,out] (ops.py:23-24)."""
    out = np.asarray(x) @ np.asarray(w)
    return out + b if b is not None else out


def upscale_nn(x, scale=2):
    """Nearest-neighbour up-sampling of every spatial axis of a channels-last tensor;
    src = dst // scale  (tf.image.resize_nearest_neighbor, align_corners=False).
    2-D: ops.py:75-77; 3-D: ops.py:79-91 (two 2-D resizes == one 3-D nearest resize)."""
    x = np.asarray(x)
    for a in range(1, x.ndim - 1):
        x = np.repeat(x, scale, axis=a)
    return x


def resize_nn(x, new_size):
    """tf.image.resize_nearest_neighbor(x, new_size), align_corners=False, per spatial axis (ops.py:66-73; TF 1.15 absent: restated from
    its documented index rule src = min(floor(dst * in / out), in - 1))."""
    x = np.asarray(x)
    for a, out in enumerate(new_size):
        n = x.shape[1 + a]
        idx = np.minimum((np.arange(out, dtype=np.int64) * n) // out, n - 1)
        x = np.take(x, idx, axis=1 + a)
    return x


def resize_nn_bwd(g, in_shape):
    """Adjoint of :func:`resize_nn`: scatter-add of g onto the source cells."""
    g = np.asarray(g)
    for a in reversed(range(g.ndim - 2)):
        n, out = in_shape[1 + a], g.shape[1 + a]
        idx = np.minimum((np.arange(out, dtype=np.int64) * n) // out, n - 1)
        shp = list(g.shape); shp[1 + a] = n
        acc = np.zeros(shp, g.dtype)
        np.add.at(acc, (slice(None),) * (1 + a) + (idx,), g)
        g = acc
    return g


def upscale_nn_bwd(g, scale=2):
    g = np.asarray(g)
    nd = g.ndim - 2
    shp = [g.shape[0]]
    for a in range(nd):
        shp += [g.shape[1 + a] // scale, scale]
    shp += [g.shape[-1]]
    return g.reshape(shp).sum(axis=tuple(2 + 2 * a for a in range(nd)))


# ----------------------------------------------------------------------------------------
# Generator (model.py:5-87)
# ----------------------------------------------------------------------------------------

def generator_plan(output_shape, filters, num_conv=4, repeat=0):
    """Layer bookkeeping of GeneratorBE / GeneratorBE3 (model.py:8-19, 51-61)."""
    spatial = list(output_shape[:-1])
    repeat_num = int(np.log2(np.max(spatial))) - 2 if repeat == 0 else repeat
    f = 2 ** (repeat_num - 1)
    assert repeat_num > 0 and all(s % f == 0 for s in spatial), "model.py:12 / :55"
    x0_shape = [s // f for s in spatial] + [filters]
    n_layers = 1 + repeat_num * num_conv + 1
    return repeat_num, x0_shape, n_layers


def xavier_uniform(rng, shape):
    """slim xavier_initializer() (uniform): limit = sqrt(6/(fan_in+fan_out)), SURVEY A.4."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def generator_init(rng, c_num, output_shape, filters, name="G", num_conv=4, conv_k=3, last_k=3, repeat=0, skip_concat=False):
    """Variables in TF/slim naming order: '<name>/<n>_fc|_conv/{weights,biases}'.  skip_concat (model.py:30-33 / :72-75): the first
    conv of every block after the first reads concat([upscale(x), upscale(x0)]) = 2 x filters channels."""
    nd = len(output_shape) - 1
    repeat_num, x0_shape, _ = generator_plan(output_shape, filters, num_conv, repeat)
    p = {}
    p["%s/0_fc/weights" % name] = xavier_uniform(rng, (c_num, int(np.prod(x0_shape))))
    p["%s/0_fc/biases" % name] = np.zeros(int(np.prod(x0_shape)), np.float32)
    ln = 1
    for i in range(repeat_num * num_conv):
        cin = 2 * filters if (skip_concat and i >= num_conv and i % num_conv == 0) else filters
        p["%s/%d_conv/weights" % (name, ln)] = xavier_uniform(rng, (conv_k,) * nd + (cin, filters))
        p["%s/%d_conv/biases" % (name, ln)] = np.zeros(filters, np.float32)
        ln += 1
    p["%s/%d_conv/weights" % (name, ln)] = xavier_uniform(rng, (last_k,) * nd + (filters, output_shape[-1]))
    p["%s/%d_conv/biases" % (name, ln)] = np.zeros(output_shape[-1], np.float32)
    return p


def generator_fwd(z, p, output_shape, filters, name="G", num_conv=4, repeat=0, leak=0.2, keep=False, skip_concat=False):
    """GeneratorBE (model.py:5-46) / GeneratorBE3 (model.py:48-87), skip_concat=False.
    Returns out (and, if keep, the cache needed by :func:`generator_bwd`)."""
    repeat_num, x0_shape, _ = generator_plan(output_shape, filters, num_conv, repeat)
    W = lambda n, kind: p["%s/%d_%s/weights" % (name, n, kind)]
    Bv = lambda n, kind: p["%s/%d_%s/biases" % (name, n, kind)]
    cache = {"z": z, "acts": [], "blocks": []}
    x = linear(z, W(0, "fc"), Bv(0, "fc")).reshape([-1] + x0_shape)
    ln = 1
    x0 = x
    for idx in range(repeat_num):
        blk = {"x0": x0, "ins": [], "outs": [], "ln": []}
        for _ in range(num_conv):
            blk["ins"].append(x); blk["ln"].append(ln)
            x = lrelu(conv_same(x, W(ln, "conv"), Bv(ln, "conv")), leak)
            blk["outs"].append(x)
            ln += 1
        blk["up"] = idx < repeat_num - 1
        if skip_concat:                              # model.py:30-33 / :72-75 (forward only: no cache for generator_bwd)
            assert not keep, "oracle: skip_concat reverse pass not restated (use df_oracle_torch)"
            if blk["up"]:
                x = upscale_nn(x, 2); x0 = upscale_nn(x0, 2)
                x = np.concatenate([x, x0], axis=-1)
            continue
        x = x + x0                                   # model.py:35 / :40
        if blk["up"]:
            x = upscale_nn(x, 2)                     # model.py:36 / :78
            x0 = x
        cache["blocks"].append(blk)
    cache["last_in"] = x; cache["last_ln"] = ln
    out = conv_same(x, W(ln, "conv"), Bv(ln, "conv"))
    return (out, cache) if keep else out


def generator_bwd(dout, cache, p, name="G", leak=0.2, masks=None):
    """Manual reverse pass of :func:`generator_fwd`; returns grads keyed like ``p``.  ``masks`` ({layer number: bool array},
    optional) replaces the oracle's own lrelu sign pattern by the one the implementation under test took: the network is
    piecewise linear, and a pre-activation within rounding error of zero may pick either slope."""
    g = {}
    ln = cache["last_ln"]
    dx, dw, db = conv_same_bwd(cache["last_in"], p["%s/%d_conv/weights" % (name, ln)], dout)
    g["%s/%d_conv/weights" % (name, ln)] = dw; g["%s/%d_conv/biases" % (name, ln)] = db
    for blk in reversed(cache["blocks"]):
        if blk["up"]:
            dx = upscale_nn_bwd(dx, 2)
        dy = dx
        for xin, xout, l in zip(reversed(blk["ins"]), reversed(blk["outs"]), reversed(blk["ln"])):
            pos = (xout > 0) if masks is None else masks[l]
            dpre = dx * np.where(pos, 1.0, leak).astype(dx.dtype)
            dx, dw, db = conv_same_bwd(xin, p["%s/%d_conv/weights" % (name, l)], dpre)
            g["%s/%d_conv/weights" % (name, l)] = dw; g["%s/%d_conv/biases" % (name, l)] = db
        dx = dx + dy
    d2 = dx.reshape(dx.shape[0], -1)
    g["%s/0_fc/weights" % name] = np.asarray(cache["z"]).T @ d2
    g["%s/0_fc/biases" % name] = d2.sum(axis=0)
    return g


# ----------------------------------------------------------------------------------------
# Encoder / auto-encoder (model.py:118-216)
# ----------------------------------------------------------------------------------------

def encoder_plan(x_shape, filters, repeat=0):
    spatial = list(x_shape[:-1])
    repeat_num = int(np.log2(np.max(spatial))) - 2 if repeat == 0 else repeat
    assert repeat_num > 0 and all(s % (2 ** (repeat_num - 1)) == 0 for s in spatial), "model.py:125 / :161"
    return repeat_num


def encoder_init(rng, x_shape, filters, z_num, name="enc", num_conv=3, conv_k=3, repeat=0):
    """Variables of EncoderBE / EncoderBE3 in slim naming order."""
    nd = len(x_shape) - 1
    repeat_num = encoder_plan(x_shape, filters, repeat)
    p = {}
    def add_conv(ln, cin, cout):
        p["%s/%d_conv/weights" % (name, ln)] = xavier_uniform(rng, (conv_k,) * nd + (cin, cout))
        p["%s/%d_conv/biases" % (name, ln)] = np.zeros(cout, np.float32)
    ch = filters
    add_conv(0, x_shape[-1], ch)
    ln = 1
    spatial = list(x_shape[:-1])
    cin = ch
    for idx in range(repeat_num):
        for _ in range(num_conv):
            add_conv(ln, cin, filters); cin = filters; ln += 1
        ch += filters
        cin = ch
        if idx < repeat_num - 1:
            add_conv(ln, ch, ch); ln += 1
            spatial = [s // 2 for s in spatial]
    flat = int(np.prod(spatial)) * ch
    p["%s/%d_fc/weights" % (name, ln)] = xavier_uniform(rng, (flat, z_num))
    p["%s/%d_fc/biases" % (name, ln)] = np.zeros(z_num, np.float32)
    return p


def encoder_fwd(x, p, filters, z_num, name="enc", num_conv=3, repeat=0, leak=0.2, keep=False):
    """EncoderBE (model.py:118-152) / EncoderBE3 (model.py:154-188)."""
    repeat_num = encoder_plan(x.shape[1:], filters, repeat)
    W = lambda n, kind: p["%s/%d_%s/weights" % (name, n, kind)]
    Bv = lambda n, kind: p["%s/%d_%s/biases" % (name, n, kind)]
    tape = []                                           # (kind, payload) in execution order
    def conv(xx, ln, stride):
        y = lrelu(conv_same(xx, W(ln, "conv"), Bv(ln, "conv"), stride), leak)
        tape.append(("conv", (xx, y, ln, stride)))
        return y
    x = conv(x, 0, 1)
    x0 = x
    ln = 1
    for idx in range(repeat_num):
        for _ in range(num_conv):
            x = conv(x, ln, 1); ln += 1
        tape.append(("concat", (x.shape[-1], x0.shape[-1])))
        x = np.concatenate([x, x0], axis=-1)            # model.py:138 / :174
        if idx < repeat_num - 1:
            x = conv(x, ln, 2); ln += 1                # model.py:141-143
            tape.append(("mark_x0", None))
            x0 = x
    flat = x.reshape(x.shape[0], -1)
    z = linear(flat, W(ln, "fc"), Bv(ln, "fc"))
    cache = {"tape": tape, "flat": flat, "xshape": x.shape, "fc_ln": ln}
    return (z, cache) if keep else z


def encoder_bwd(dz, cache, p, name="enc", leak=0.2, masks=None):
    """Reverse pass of :func:`encoder_fwd` (gradient w.r.t. the input is not needed: x is data).  ``masks`` ({layer number: bool
    array}, optional): the lrelu sign pattern of the implementation under test (see :func:`generator_bwd`)."""
    g = {}
    ln = cache["fc_ln"]
    g["%s/%d_fc/weights" % (name, ln)] = cache["flat"].T @ dz
    g["%s/%d_fc/biases" % (name, ln)] = dz.sum(axis=0)
    dx = (dz @ p["%s/%d_fc/weights" % (name, ln)].T).reshape(cache["xshape"])
    dx0 = None                                          # gradient flowing into the current skip source x0
    for kind, payload in reversed(cache["tape"]):
        if kind == "mark_x0":                           # x0 = x after the stride-2 conv: merge the skip gradient
            if dx0 is not None:
                dx = dx + dx0
            dx0 = None
        elif kind == "concat":
            ca, cb = payload
            dx0 = dx[..., ca:] if dx0 is None else dx0 + dx[..., ca:]
            dx = dx[..., :ca]
        else:
            xin, y, l, stride = payload
            if l == 0 and dx0 is not None:              # x0 = output of the first conv
                dx = dx + dx0
                dx0 = None
            dpre = dx * np.where((y > 0) if masks is None else masks[l], 1.0, leak).astype(dx.dtype)
            dx, dw, db = conv_same_bwd(xin, p["%s/%d_conv/weights" % (name, l)], dpre, stride, need_dx=(l != 0))
            g["%s/%d_conv/weights" % (name, l)] = dw; g["%s/%d_conv/biases" % (name, l)] = db
    return g


def ae_init(rng, x_shape, filters, z_num, name="AE", num_conv=4, repeat=0):
    """AE / AE3 variables: encoder 'AE/enc/*' with num_conv-1 convs per block, decoder 'AE/dec/*' (model.py:190-216)."""
    p = encoder_init(rng, x_shape, filters, z_num, name + "/enc", num_conv - 1, repeat=repeat)
    p.update(generator_init(rng, z_num, list(x_shape), filters, name + "/dec", num_conv, repeat=repeat))
    return p


def kl_bernoulli(z, n, rho):
    """trainer3.py:272-277 / trainer.py:389-394: sum_j KL(Bernoulli(rho) || Bernoulli(mean_b z[b, j])), j < n.
    (tf.distributions.kl_divergence for two Bernoullis -- TensorFlow 1.15 is absent here, so this line is a restatement of its
    documented closed form p log(p/q) + (1-p) log((1-p)/(1-q)): parity unpinned for this one term.)"""
    q = z[:, :n].mean(axis=0)
    return float((rho * np.log(rho / q) + (1 - rho) * np.log((1 - rho) / (1 - q))).sum())


def kl_bernoulli_bwd(z, n, rho, scale=1.0):
    q = z[:, :n].mean(axis=0)
    g = np.zeros_like(z)
    g[:, :n] = scale * (-rho / q + (1 - rho) / (1 - q)) / z.shape[0]
    return g


def ae_train_step(x, y_last, p, opt, filters, z_num, p_num, is_3d, num_conv=4, repeat=0, use_curl=True, w1=1.0, w2=1.0,
                  w4=1.0, name="AE", use_sparse=False, sparsity=0.01, w5=1.0, enc_masks=None, dec_masks=None, sign_u=None):
    """build_model_ae + one optimizer step (trainer.py:357-423 / trainer3.py:240-279).
    ``y_last`` = y[:, :, -1]  [B, p_num];  loss = w1*L1 + w2*J-L1 + w4*mean((y_last - z[:, -p_num:])^2)
    (+ w5 * Bernoulli-KL of the sigmoid code's first z_num - p_num columns when use_sparse, model.py:196,210).
    ``enc_masks`` / ``dec_masks`` / ``sign_u``: back-propagate on the linear pieces (lrelu slopes per layer, signs of the |.| terms) the
    implementation under test is on; forward values and the loss stay this oracle's own."""
    oshape = list(x.shape[1:])                           # model.py:197,211: the decoder emits x's own shape
    zpre, ecache = encoder_fwd(x, p, filters, z_num, name + "/enc", num_conv - 1, repeat, keep=True)
    z = 1.0 / (1.0 + np.exp(-zpre)) if use_sparse else zpre
    s, dcache = generator_fwd(z, p, oshape, filters, name + "/dec", num_conv, repeat, keep=True)
    if use_curl:
        if is_3d:
            res = velocity_loss(s, x, True, w1, w2, sign_u=sign_u)
            ds = res["dpsi"]
        else:                                            # curl reads channel 0 only (ops.py:267-268)
            res = velocity_loss(s[..., :1], x, False, w1, w2, sign_u=sign_u)
            ds = np.zeros_like(s); ds[..., :1] = res["dpsi"]
    else:                                                # trainer.py:362-364 / trainer3.py:245-247: x_ = the decoder's own output
        res = velocity_loss(s, x, is_3d, w1, w2, sign_u=sign_u, use_curl=False)
        ds = res["dpsi"]
    zp = z[:, -p_num:]
    loss_p = ((y_last - zp) ** 2).mean()
    dzp = -2.0 * (y_last - zp) / zp.size * w4
    grads = generator_bwd(ds, dcache, p, name + "/dec", masks=dec_masks)
    dz = _generator_dz(ds, dcache, p, name + "/dec", masks=dec_masks).copy()      # dL/dz through the decoder (z is the encoder's output)
    dz[:, -p_num:] += dzp
    loss_kl = 0.0
    if use_sparse:
        loss_kl = kl_bernoulli(z, z_num - p_num, sparsity)
        dz = dz + kl_bernoulli_bwd(z, z_num - p_num, sparsity, w5)
        dz = dz * z * (1.0 - z)                                   # through the sigmoid
    grads.update(encoder_bwd(dz, ecache, p, name + "/enc", masks=enc_masks))
    t = opt["t"] + 1
    new_p, new_m, new_v = {}, {}, {}
    for k in p:
        new_p[k], new_m[k], new_v[k] = adam_tf1(p[k], grads[k], opt["m"][k], opt["v"][k], t, opt["lr"])
    info = {"loss": res["loss"] + w4 * loss_p + w5 * loss_kl, "l1": res["l1"], "j_l1": res["j_l1"], "loss_p": loss_p,
            "loss_kl": loss_kl, "u": res["u"], "z": z, "grads": grads}
    return new_p, {"m": new_m, "v": new_v, "t": t, "lr": opt["lr"]}, info


def _generator_dz(dout, cache, p, name, leak=0.2, masks=None):
    """dL/dz of the generator input (needed when z comes from an encoder): the reverse pass of generator_fwd down
    to the FC layer's input."""
    ln = cache["last_ln"]
    dx, _, _ = conv_same_bwd(cache["last_in"], p["%s/%d_conv/weights" % (name, ln)], dout)
    for blk in reversed(cache["blocks"]):
        if blk["up"]:
            dx = upscale_nn_bwd(dx, 2)
        dy = dx
        for xin, xout, l in zip(reversed(blk["ins"]), reversed(blk["outs"]), reversed(blk["ln"])):
            dpre = dx * np.where((xout > 0) if masks is None else masks[l], 1.0, leak).astype(dx.dtype)
            dx, _, _ = conv_same_bwd(xin, p["%s/%d_conv/weights" % (name, l)], dpre)
        dx = dx + dy
    return dx.reshape(dx.shape[0], -1) @ p["%s/0_fc/weights" % name].T


# ----------------------------------------------------------------------------------------
# Discriminator (model.py:89-116) and the LSGAN terms of arch='dg' (trainer.py:149-156,174-182)
# ----------------------------------------------------------------------------------------

def discriminator_init(rng, cin, filters, nd, name="D"):
    p = {}
    d = int(filters / 2)
    c = cin
    names = ["Conv", "Conv_1", "Conv_2", "Conv_3", "Conv_4"]          # slim's default scope names (no `name=` given)
    chans = [d, 2 * d, 4 * d, 8 * d, 1]
    for n, co in zip(names, chans):
        p["%s/%s/weights" % (name, n)] = xavier_uniform(rng, (3,) * nd + (c, co))
        p["%s/%s/biases" % (name, n)] = np.zeros(co, np.float32)
        c = co
    return p


_D_LAYERS = [("Conv", 2, True), ("Conv_1", 2, True), ("Conv_2", 2, True), ("Conv_3", 1, True), ("Conv_4", 1, False)]


def discriminator_fwd(x, p, name="D", leak=0.2, keep=False):
    tape = []
    for n, stride, act in _D_LAYERS:
        pre = conv_same(x, p["%s/%s/weights" % (name, n)], p["%s/%s/biases" % (name, n)], stride)
        y = lrelu(pre, leak) if act else pre
        tape.append((n, stride, act, x, y))
        x = y
    return (x, tape) if keep else x


def discriminator_bwd(dout, tape, p, name="D", leak=0.2, need_dx=True):
    g = {}
    dx = dout
    for n, stride, act, xin, y in reversed(tape):
        dpre = dx * np.where(y > 0, 1.0, leak).astype(dx.dtype) if act else dx
        dx, dw, db = conv_same_bwd(xin, p["%s/%s/weights" % (name, n)], dpre, stride)
        g["%s/%s/weights" % (name, n)] = dw; g["%s/%s/biases" % (name, n)] = db
    return g, (dx if need_dx else None)


def gan_losses_and_grads(z, x, pG, pD, output_shape, filters, is_3d, w1=1.0, w2=1.0, w3=0.005, num_conv=4, repeat=0):
    """arch='dg' (trainer.py:136-184 / trainer3.py:14-63): losses and the gradients the two `minimize` calls apply:
    g_loss w.r.t. G_var (through the discriminator), d_loss w.r.t. D_var."""
    psi, cache = generator_fwd(z, pG, output_shape, filters, "G", num_conv, repeat, keep=True)
    if is_3d:
        u = curl3(psi); ju, vu = jacobian3(u); jx, vx = jacobian3(x)
    else:
        u = curl(psi); ju, vu = jacobian(u); jx, vx = jacobian(x)
    l1 = l1_mean(u, x); jl1 = l1_mean(ju, jx)
    D_x, tape_x = discriminator_fwd(np.concatenate([x, vx], -1), pD, keep=True)
    D_G, tape_G = discriminator_fwd(np.concatenate([u, vu], -1), pD, keep=True)
    g_real = ((D_G - 1) ** 2).mean(); d_fake = (D_G ** 2).mean(); d_real = ((D_x - 1) ** 2).mean()
    g_loss = w1 * l1 + w2 * jl1 + w3 * g_real
    d_loss = d_real + d_fake
    gD1, _ = discriminator_bwd(2 * (D_x - 1) / D_x.size, tape_x, pD, need_dx=False)
    gD2, _ = discriminator_bwd(2 * D_G / D_G.size, tape_G, pD, need_dx=False)
    gD = {k: gD1[k] + gD2[k] for k in gD1}
    _, dGin = discriminator_bwd(w3 * 2 * (D_G - 1) / D_G.size, tape_G, pD)
    c = u.shape[-1]
    du = l1_mean_bwd(u, x, w1) + dGin[..., :c]
    dj = l1_mean_bwd(ju, jx, w2)
    if is_3d:
        du = du + jacobian3_bwd(gj=dj, gc=dGin[..., c:])
        dpsi = jacobian3_bwd(gc=du)
    else:
        du = du + jacobian_bwd(dj, dGin[..., c:])
        dpsi = curl_bwd(du)
    gG = generator_bwd(dpsi, cache, pG, "G")
    return {"g_loss": g_loss, "d_loss": d_loss, "l1": l1, "j_l1": jl1, "g_loss_real": g_real, "u": u, "gG": gG, "gD": gD}


# ----------------------------------------------------------------------------------------
# Train step (trainer.py:136-184, trainer3.py:14-63) + TF1 Adam + LR schedule
# ----------------------------------------------------------------------------------------

def velocity_loss(psi, x, is_3d, w1=1.0, w2=1.0, need_grad=True, sign_u=None, use_curl=True):
    """G_ = curl(psi) | jacobian3(psi)[1]  (use_curl=False, trainer.py:141-143 / trainer3.py:19-21: G_ = the generator's own
    2- | 3-channel output, every liquid scene of run.bat);  loss = w1*mean|G_-x| + w2*mean|J(G_)-J(x)|.
    Returns dict(loss, l1, j_l1, u, dpsi).  ``sign_u`` (optional): take the sign pattern of the two |.| terms in the reverse
    pass from this velocity field (the one the implementation under test produced) -- |.| is piecewise linear and values within
    rounding error of zero may sit on either piece, which moves the heavily cancelling parameter gradients at the 1e-3 level."""
    if is_3d:
        u = curl3(psi) if use_curl else psi
        ju, _ = jacobian3(u); jx, _ = jacobian3(x)
    else:
        u = curl(psi) if use_curl else psi
        ju, _ = jacobian(u); jx, _ = jacobian(x)
    l1 = l1_mean(u, x); jl1 = l1_mean(ju, jx)
    res = {"loss": w1 * l1 + w2 * jl1, "l1": l1, "j_l1": jl1, "u": u, "ju": ju, "jx": jx}
    if need_grad:
        us = u if sign_u is None else np.asarray(sign_u)
        js = ju if sign_u is None else (jacobian3(us)[0] if is_3d else jacobian(us)[0])
        du = l1_mean_bwd(us, x, w1)
        dj = l1_mean_bwd(js, jx, w2)
        if is_3d:
            du = du + jacobian3_bwd(gj=dj)
            res["dpsi"] = jacobian3_bwd(gc=du) if use_curl else du
        else:
            du = du + jacobian_bwd(dj)
            res["dpsi"] = curl_bwd(du) if use_curl else du
    return res


def adam_tf1(p, g, m, v, t, lr, beta1=0.5, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    theta -= lr_t * m / (sqrt(v) + eps)  ("epsilon-hat" form).  ``t`` is the 1-based step."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v


def lr_step(lr, step, lr_update_step, lr_min=2.5e-6):
    """lr_update='step' (trainer.py:77-78, 285-286): after loop iteration ``step`` (0-based) the rate is halved, floored at lr_min,
    when ``step % lr_update_step == lr_update_step - 1``."""
    return max(lr * 0.5, lr_min) if step % lr_update_step == lr_update_step - 1 else lr


def lr_cosine(step, max_step, lr_max=1e-4, lr_min=2.5e-6):
    """trainer.py:74-75: assigned AFTER each step with the already-incremented step."""
    return lr_min + 0.5 * (lr_max - lr_min) * (math.cos(step * math.pi / max_step) + 1.0)


def train_step(z, x, p, opt, output_shape, filters, is_3d, num_conv=4, repeat=0, w1=1.0, w2=1.0,
               name="G", masks=None, sign_u=None, use_curl=True, optimizer="adam"):
    """One full step: G fwd -> curl -> Jacobian -> L1 losses -> bwd -> TF1 Adam (in place on
    copies).  ``opt`` = dict(m, v, t, lr).  Returns (new_p, new_opt, info).  ``use_curl=False``: ``output_shape`` carries the
    velocity's own 2 | 3 channels (trainer.py:48-55).  ``optimizer='gd'``: tf.train.GradientDescentOptimizer (trainer.py:163-165)."""
    psi, cache = generator_fwd(z, p, output_shape, filters, name, num_conv, repeat, keep=True)
    res = velocity_loss(psi, x, is_3d, w1, w2, sign_u=sign_u, use_curl=use_curl)
    grads = generator_bwd(res["dpsi"], cache, p, name, masks=masks)
    t = opt["t"] + 1
    new_p, new_m, new_v = {}, {}, {}
    for k in p:
        if optimizer == "gd":
            new_p[k], new_m[k], new_v[k] = p[k] - opt["lr"] * grads[k], opt["m"][k], opt["v"][k]
            continue
        new_p[k], new_m[k], new_v[k] = adam_tf1(p[k], grads[k], opt["m"][k], opt["v"][k], t, opt["lr"])
    info = {k: res[k] for k in ("loss", "l1", "j_l1", "u")}
    info["psi"] = psi; info["grads"] = grads
    return new_p, {"m": new_m, "v": new_v, "t": t, "lr": opt["lr"]}, info


# ----------------------------------------------------------------------------------------
# Synthetic inputs
# ----------------------------------------------------------------------------------------

def synthetic_batch(rng, batch, spatial, c_num=3):
    """y ~ U(-1,1) [B,c_num];  x = curl(psi_gt) rescaled to max|x| = 1 (divergence-free,
    in [-1,1] like the reference's normalised data, data.py:87-88,329)."""
    is_3d = len(spatial) == 3
    y = rng.uniform(-1, 1, size=(batch, c_num)).astype(np.float32)
    psi = rng.uniform(-1, 1, size=[batch] + list(spatial) + [3 if is_3d else 1]).astype(np.float32)
    x = curl3(psi) if is_3d else curl(psi)
    x = (x / np.abs(x).max()).astype(np.float32)
    return x, y
