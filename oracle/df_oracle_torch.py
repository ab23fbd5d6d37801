"""PyTorch-CPU restatement of the velocity-field train step  --  TEST INFRASTRUCTURE ONLY.

Second, independent implementation of the conv / FC / up-sampling arithmetic that lives in TensorFlow 1.15 in
the reference (absent here -> "parity unpinned", see df_oracle.py): F.conv2d/conv3d(padding=1), F.linear,
nearest 2x repeat, leaky-relu 0.2, autograd for the reverse pass.  Used by
  * tests: cross-check of df_oracle.py's hand-written conv forward/backward and generator gradients;
  * bench.py's ``cpu_baseline`` leg: the train step timed on the GPU box's host cores (kind "port").
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Follows model.py:5-87, ops.py:9-24,66-91,205-274, trainer.py:136-184, trainer3.py:14-63.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _fdiff(f, axis):
    """ops.py:214-217: forward difference, last DIFFERENCE replicated."""
    n = f.shape[axis]
    d = f.narrow(axis, 1, n - 1) - f.narrow(axis, 0, n - 1)
    return torch.cat([d, d.narrow(axis, n - 2, 1)], dim=axis)


def curl(x):
    psi = x[..., 0]
    return torch.stack([_fdiff(psi, 1), -_fdiff(psi, 2)], dim=-1)


def jacobian(x):
    u, v = x[..., 0], x[..., 1]
    dudx, dudy, dvdx, dvdy = _fdiff(u, 2), _fdiff(u, 1), _fdiff(v, 2), _fdiff(v, 1)
    return torch.stack([dudx, dudy, dvdx, dvdy], dim=-1), (dvdx - dudy).unsqueeze(-1)


def jacobian3(x):
    d = {}
    for ci, cn in enumerate("uvw"):
        for an, ax in (("x", 3), ("y", 2), ("z", 1)):
            d[cn + an] = _fdiff(x[..., ci], ax)
    j = torch.stack([d["ux"], d["uy"], d["uz"], d["vx"], d["vy"], d["vz"], d["wx"], d["wy"], d["wz"]], dim=-1)
    c = torch.stack([d["wy"] - d["vz"], d["uz"] - d["wx"], d["vx"] - d["uy"]], dim=-1)
    return j, c


def conv_same(x, w, b):
    """channels-last x [B,*S,Cin], TF weights [*k,Cin,Cout]; odd cubic k (3 on the trainers' path; conv_k / last_k = 1, 5 are accepted
    by the generator surface, model.py:5-6), stride 1, TF 'SAME' = symmetric pad k // 2."""
    nd = x.dim() - 2
    pad = int(w.shape[0]) // 2
    assert int(w.shape[0]) % 2 == 1
    if nd == 3:
        y = F.conv3d(x.permute(0, 4, 1, 2, 3), w.permute(4, 3, 0, 1, 2).contiguous(), b, padding=pad)
        return y.permute(0, 2, 3, 4, 1)
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1).contiguous(), b, padding=pad)
    return y.permute(0, 2, 3, 1)


def upscale_nn(x):
    for a in range(1, x.dim() - 1):
        x = x.repeat_interleave(2, dim=a)
    return x


class _LreluMasked(torch.autograd.Function):
    """lrelu whose BACKWARD slope pattern is given (``mask`` = where the implementation under test took slope 1).  The
    network is piecewise linear; a pre-activation within rounding error of zero can pick the other slope in two correct
    implementations, which moves single gradient elements by O(1).  Gradient parity is therefore checked on the linear
    region the GPU actually took; the forward value is this oracle's own lrelu."""

    @staticmethod
    def forward(ctx, pre, mask, leak):
        ctx.save_for_backward(mask)
        ctx.leak = leak
        return F.leaky_relu(pre, leak)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, g * ctx.leak), None, None


def generator_fwd(z, p, output_shape, filters, name="G", num_conv=4, repeat=0, leak=0.2, masks=None, skip_concat=False,
                  own_masks=None):
    """model.py:5-87 (skip_concat=False).  ``masks``: optional {layer number: bool tensor} for :class:`_LreluMasked`;
    ``own_masks``: optional dict that receives {layer number: (this oracle's own pre-activation > 0)}."""
    spatial = list(output_shape[:-1])
    repeat_num = int(np.log2(np.max(spatial))) - 2 if repeat == 0 else repeat
    f = 2 ** (repeat_num - 1)
    x0_shape = [s // f for s in spatial] + [filters]
    x = F.linear(z, p["%s/0_fc/weights" % name].t(), p["%s/0_fc/biases" % name]).reshape([-1] + x0_shape)
    ln = 1
    x0 = x
    for idx in range(repeat_num):
        for _ in range(num_conv):
            pre = conv_same(x, p["%s/%d_conv/weights" % (name, ln)], p["%s/%d_conv/biases" % (name, ln)])
            if own_masks is not None:
                own_masks[ln] = (pre > 0).detach()
            x = F.leaky_relu(pre, leak) if masks is None else _LreluMasked.apply(pre, masks[ln], leak)
            ln += 1
        if skip_concat:                              # model.py:30-33 / :72-75
            if idx < repeat_num - 1:
                x = upscale_nn(x); x0 = upscale_nn(x0)
                x = torch.cat([x, x0], dim=-1)
            continue
        x = x + x0
        if idx < repeat_num - 1:
            x = upscale_nn(x)
            x0 = x
    return conv_same(x, p["%s/%d_conv/weights" % (name, ln)], p["%s/%d_conv/biases" % (name, ln)])


def conv_same_s2(x, w, b):
    """k=3, stride 2, TF 'SAME' on even extents: pad 0 before / 1 after per axis -- NOT torch's padding=1."""
    nd = x.dim() - 2
    if nd == 3:
        xp = F.pad(x.permute(0, 4, 1, 2, 3), (0, 1, 0, 1, 0, 1))
        return F.conv3d(xp, w.permute(4, 3, 0, 1, 2).contiguous(), b, stride=2).permute(0, 2, 3, 4, 1)
    xp = F.pad(x.permute(0, 3, 1, 2), (0, 1, 0, 1))
    return F.conv2d(xp, w.permute(3, 2, 0, 1).contiguous(), b, stride=2).permute(0, 2, 3, 1)


def encoder_fwd(x, p, filters, z_num, name="enc", num_conv=3, repeat=0, leak=0.2, masks=None, own_masks=None):
    """EncoderBE / EncoderBE3 (model.py:118-188): conv, [num_conv convs, concat skip, stride-2 conv] x repeat_num, flatten, FC.
    ``masks`` / ``own_masks``: as in :func:`generator_fwd`, keyed by the encoder's layer number (0 = the first conv)."""
    spatial = list(x.shape[1:-1])
    repeat_num = int(np.log2(np.max(spatial))) - 2 if repeat == 0 else repeat
    W = lambda n, kind: p["%s/%d_%s/weights" % (name, n, kind)]
    Bv = lambda n, kind: p["%s/%d_%s/biases" % (name, n, kind)]

    def act(pre, ln):
        if own_masks is not None:
            own_masks[ln] = (pre > 0).detach()
        return F.leaky_relu(pre, leak) if masks is None else _LreluMasked.apply(pre, masks[ln], leak)

    x = act(conv_same(x, W(0, "conv"), Bv(0, "conv")), 0)
    x0 = x
    ln = 1
    for idx in range(repeat_num):
        for _ in range(num_conv):
            x = act(conv_same(x, W(ln, "conv"), Bv(ln, "conv")), ln); ln += 1
        x = torch.cat([x, x0], dim=-1)                      # model.py:138 / :174
        if idx < repeat_num - 1:
            x = act(conv_same_s2(x, W(ln, "conv"), Bv(ln, "conv")), ln); ln += 1      # model.py:141-143
            x0 = x
    flat = x.reshape(x.shape[0], -1)
    return F.linear(flat, W(ln, "fc").t(), Bv(ln, "fc"))


def ae_fwd(x, p, filters, z_num, name="AE", num_conv=4, repeat=0, enc_masks=None, dec_masks=None, own_enc=None, own_dec=None):
    """AE / AE3 (model.py:190-216), use_sparse=False: returns (out, z)."""
    z = encoder_fwd(x, p, filters, z_num, name + "/enc", num_conv - 1, repeat, masks=enc_masks, own_masks=own_enc)
    out = generator_fwd(z, p, list(x.shape[1:]), filters, name + "/dec", num_conv, repeat, masks=dec_masks, own_masks=own_dec)
    return out, z


def ae_grads(x, y_last, p, filters, z_num, p_num, is_3d, num_conv=4, repeat=0, use_curl=True, w1=1.0, w2=1.0, w4=1.0,
             enc_masks=None, dec_masks=None, sign_u=None, own_enc=None, own_dec=None):
    """build_model_ae (trainer.py:357-423 / trainer3.py:240-279, use_sparse=False) through autograd: loss = w1*L1 + w2*J-L1 +
    w4*mean((y_last - z[:, -p_num:])^2); returns losses, the velocity field, the code and d loss / d every variable."""
    for v in p.values():
        v.requires_grad_(True)
        v.grad = None
    s, z = ae_fwd(x, p, filters, z_num, num_conv=num_conv, repeat=repeat, enc_masks=enc_masks, dec_masks=dec_masks,
                  own_enc=own_enc, own_dec=own_dec)
    psi = s if (is_3d or not use_curl) else s[..., :1]      # 2-D curl reads channel 0 only (ops.py:267-268)
    loss_v, l1, jl1, u = velocity_loss(psi, x, is_3d, w1, w2, sign_u=sign_u, use_curl=use_curl)
    loss_p = ((y_last - z[:, -p_num:]) ** 2).mean()
    loss = loss_v + w4 * loss_p
    loss.backward()
    return {"loss": float(loss.detach()), "l1": float(l1.detach()), "j_l1": float(jl1.detach()), "loss_p": float(loss_p.detach()),
            "u": u.detach(), "z": z.detach(), "grads": {k: v.grad for k, v in p.items()}}


def velocity_loss(psi, x, is_3d, w1=1.0, w2=1.0, sign_u=None, use_curl=True):
    """``sign_u`` (optional): the velocity field of the implementation under test.  |.| is piecewise linear: where u - x or
    J(u) - J(x) lies within rounding error of zero two correct implementations may sit on different linear pieces, and the
    parameter gradients -- sums of ~1e7 sign terms with heavy cancellation -- then differ at the 1e-3 level for reasons that say
    nothing about the kernels.  With ``sign_u`` the returned loss keeps its value but back-propagates on the pieces ``sign_u`` is on."""
    if is_3d:
        u = jacobian3(psi)[1] if use_curl else psi         # use_curl=False: trainer3.py:19-21
        ju = jacobian3(u)[0]; jx = jacobian3(x)[0]
    else:
        u = curl(psi) if use_curl else psi                 # trainer.py:141-143
        ju = jacobian(u)[0]; jx = jacobian(x)[0]
    l1 = (u - x).abs().mean(); jl1 = (ju - jx).abs().mean()
    loss = l1 * w1 + jl1 * w2
    if sign_u is not None:
        with torch.no_grad():
            s1 = torch.sign(sign_u - x)
            sj = torch.sign((jacobian3(sign_u)[0] if is_3d else jacobian(sign_u)[0]) - jx)
        surrogate = (s1 * (u - x)).mean() * w1 + (sj * (ju - jx)).mean() * w2      # same gradient as the L1 terms on sign_u's pieces
        loss = surrogate + (loss - surrogate).detach()
    return loss, l1, jl1, u


def train_step(z, x, p, opt, output_shape, filters, is_3d, num_conv=4, repeat=0, w1=1.0, w2=1.0, beta1=0.5,
               beta2=0.999, eps=1e-8, masks=None, sign_u=None, own_masks=None, update=True, use_curl=True):
    """One step with TF1 Adam, in place on ``p`` (dict of leaf tensors) and ``opt`` (m, v, t, lr); ``update=False`` stops after the
    gradients (parity runs that evaluate the same weights twice)."""
    for v in p.values():
        v.requires_grad_(True)
        v.grad = None
    psi = generator_fwd(z, p, output_shape, filters, num_conv=num_conv, repeat=repeat, masks=masks, own_masks=own_masks)
    loss, l1, jl1, u = velocity_loss(psi, x, is_3d, w1, w2, sign_u=sign_u, use_curl=use_curl)
    loss.backward()
    if not update:
        return {"loss": float(loss.detach()), "l1": float(l1.detach()), "j_l1": float(jl1.detach()), "u": u.detach(),
                "psi": psi.detach(), "grads": {k: v.grad for k, v in p.items()}}
    opt["t"] += 1
    t = opt["t"]
    lr_t = opt["lr"] * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    grads = {}
    with torch.no_grad():
        for k, v in p.items():
            g = v.grad
            grads[k] = g
            opt["m"][k].mul_(beta1).add_(g, alpha=1.0 - beta1)
            opt["v"][k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            v.sub_(lr_t * opt["m"][k] / (opt["v"][k].sqrt() + eps))
    return {"loss": float(loss.detach()), "l1": float(l1.detach()), "j_l1": float(jl1.detach()), "u": u.detach(), "psi": psi.detach(), "grads": grads}


def to_torch(params, dtype=torch.float32):
    return {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in params.items()}


def new_opt(p, lr=1e-4):
    return {"m": {k: torch.zeros_like(v) for k, v in p.items()}, "v": {k: torch.zeros_like(v) for k, v in p.items()},
            "t": 0, "lr": lr}
