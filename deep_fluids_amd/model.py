"""Drop-in counterpart of the reference's ``model.py`` generators (GeneratorBE / GeneratorBE3).

Same signatures, defaults and ``(out, variables)`` return as ``byungsook/deep-fluids/model.py:5-87``;
variables live in the ``ops`` registry under slim names (``G/0_fc/weights``, ``G/1_conv/biases`` ...)
and ``reuse=True`` returns the same parameters (trainer.py:299-300).
"""
import numpy as np

from . import ops as ops_mod
from .ops import *          # noqa: F401,F403  (the reference does `from ops import *`, model.py:3)
from .ops import (variable_scope, get_variables, add, lrelu, linear, reshape, conv2d, conv3d, upscale, upscale3, concat,
                  sigmoid, get_conv_shape)

__all__ = ["GeneratorBE", "GeneratorBE3", "EncoderBE", "EncoderBE3", "AE", "AE3", "DiscriminatorPatch",
           "DiscriminatorPatch3"]


def _generator(z, filters, output_shape, name, num_conv, conv_k, last_k, repeat, skip_concat, act, reuse, is_3d):
    conv = conv3d if is_3d else conv2d
    up = upscale3 if is_3d else upscale
    with variable_scope(name, reuse=reuse) as vs:
        if repeat == 0:
            repeat_num = int(np.log2(np.max(output_shape[:-1]))) - 2          # model.py:9 / :52
        else:
            repeat_num = repeat
        assert repeat_num > 0 and np.sum([i % np.power(2, repeat_num - 1) for i in output_shape[:-1]]) == 0

        x0_shape = [int(i / np.power(2, repeat_num - 1)) for i in output_shape[:-1]] + [filters]
        num_output = int(np.prod(x0_shape))
        layer_num = 0
        x = linear(z, num_output, name=str(layer_num) + "_fc")
        layer_num += 1
        if is_3d:
            x = x.reshape([-1] + x0_shape)                                     # model.py:63
        else:
            x = reshape(x, x0_shape[0], x0_shape[1], x0_shape[2])              # model.py:21
        x0 = x

        # skip_concat=True (model.py:30-33 / :72-75; never enabled by the reference trainers) runs layer by layer: the fused block
        # nodes assume the residual form
        fused = (ops_mod.FUSED_BLOCKS and not skip_concat and act is lrelu and conv_k == 3 and int(x.shape[-1]) == int(filters)
                 and int(filters) >= 8 and int(filters) % 4 == 0)     # (the fused kernels are MFMA-only: no thin-channel path)
        pending_up = False      # fused path: the 2x up-sampling is folded into the NEXT block (never materialised)
        for idx in range(repeat_num):
            if fused:       # same layers, variables and arithmetic as the loop below; one autograd node per block
                names = [str(layer_num + i) + "_conv" for i in range(num_conv)]
                if pending_up:
                    x = ops_mod.up_gen_block(x, filters, names, 3 if is_3d else 2)      # model.py:36-40 / :78-82
                else:
                    x = ops_mod.gen_block(x, filters, names, 3 if is_3d else 2)
                layer_num += num_conv
                pending_up = idx < repeat_num - 1
            else:
                for _ in range(num_conv):
                    x = conv(x, filters, k=conv_k, s=1, act=act, name=str(layer_num) + "_conv")
                    layer_num += 1
                if idx < repeat_num - 1:
                    if skip_concat:                                            # model.py:30-33 / :72-75
                        x = up(x, 2)
                        x0 = up(x0, 2)
                        x = concat([x, x0], axis=-1)
                    else:
                        x = add(x, x0)                                         # model.py:35 / :77
                        x = up(x, 2)                                           # model.py:36 / :78
                        x0 = x
                elif not skip_concat:
                    x = add(x, x0)                                             # model.py:40 / :82

        out = conv(x, output_shape[-1], k=last_k, s=1, name=str(layer_num) + "_conv")
    variables = get_variables(vs)
    return out, variables


def GeneratorBE(z, filters, output_shape, name="G", num_conv=4, conv_k=3, last_k=3, repeat=0, skip_concat=False,
                act=lrelu, reuse=False):
    """model.py:5-46."""
    return _generator(z, filters, output_shape, name, num_conv, conv_k, last_k, repeat, skip_concat, act, reuse, False)


def GeneratorBE3(z, filters, output_shape, name="G", num_conv=4, conv_k=3, last_k=3, repeat=0, skip_concat=False,
                 act=lrelu, reuse=False):
    """model.py:48-87."""
    return _generator(z, filters, output_shape, name, num_conv, conv_k, last_k, repeat, skip_concat, act, reuse, True)


def _encoder(x, filters, z_num, name, num_conv, conv_k, repeat, act, reuse, is_3d):
    conv = conv3d if is_3d else conv2d
    with variable_scope(name, reuse=reuse) as vs:
        x_shape = get_conv_shape(x)[1:]
        if repeat == 0:
            repeat_num = int(np.log2(np.max(x_shape[:-1]))) - 2                 # model.py:121 / :157
        else:
            repeat_num = repeat
        assert repeat_num > 0 and np.sum([i % np.power(2, repeat_num - 1) for i in x_shape[:-1]]) == 0

        ch = filters
        layer_num = 0
        x = conv(x, ch, k=conv_k, s=1, act=act, name=str(layer_num) + "_conv")
        x0 = x
        layer_num += 1
        for idx in range(repeat_num):
            if ops_mod.FUSED_BLOCKS and act is lrelu and conv_k == 3 and num_conv > 1 and int(filters) >= 8 and int(filters) % 4 == 0:
                # one autograd node per level's conv stack (same kernels; the reverse chain fuses the lrelu slopes into the dgrads)
                names = [str(layer_num + i) + "_conv" for i in range(num_conv)]
                x = ops_mod.conv_chain(x, filters, names, 3 if is_3d else 2)
                layer_num += num_conv
            else:
                for _ in range(num_conv):
                    x = conv(x, filters, k=conv_k, s=1, act=act, name=str(layer_num) + "_conv")
                    layer_num += 1
            x = concat([x, x0], axis=-1)                                        # model.py:138 / :174 skip connection
            ch += filters
            if idx < repeat_num - 1:
                x = conv(x, ch, k=conv_k, s=2, act=act, name=str(layer_num) + "_conv")
                layer_num += 1
                x0 = x
        b = get_conv_shape(x)[0]
        flat = x.reshape(b, -1)
        out = linear(flat, z_num, name=str(layer_num) + "_fc")
    variables = get_variables(vs)
    return out, variables


def EncoderBE(x, filters, z_num, name="enc", num_conv=4, conv_k=3, repeat=0, act=lrelu, reuse=False):
    """model.py:118-152."""
    return _encoder(x, filters, z_num, name, num_conv, conv_k, repeat, act, reuse, False)


def EncoderBE3(x, filters, z_num, name="enc", num_conv=3, conv_k=3, repeat=0, act=lrelu, reuse=False):
    """model.py:154-188."""
    return _encoder(x, filters, z_num, name, num_conv, conv_k, repeat, act, reuse, True)


def _ae(x, filters, z_num, name, num_conv, conv_k, last_k, repeat, act, skip_concat, use_sparse, reuse, is_3d):
    enc = EncoderBE3 if is_3d else EncoderBE
    dec = GeneratorBE3 if is_3d else GeneratorBE
    with variable_scope(name, reuse=reuse) as vs:
        z, _ = enc(x, filters, z_num, "enc", num_conv=num_conv - 1, conv_k=conv_k, repeat=repeat, act=act, reuse=reuse)
        if use_sparse:
            z = sigmoid(z)
        out, _ = dec(z, filters, get_conv_shape(x)[1:], "dec", num_conv=num_conv, conv_k=conv_k, last_k=last_k,
                     repeat=repeat, skip_concat=skip_concat, act=act, reuse=reuse)
    variables = get_variables(vs)
    return out, z, variables


def AE(x, filters, z_num, name="AE", num_conv=4, conv_k=3, last_k=3, repeat=0, act=lrelu, skip_concat=False,
       use_sparse=False, reuse=False):
    """model.py:190-202."""
    return _ae(x, filters, z_num, name, num_conv, conv_k, last_k, repeat, act, skip_concat, use_sparse, reuse, False)


def AE3(x, filters, z_num, name="AE", num_conv=4, conv_k=3, last_k=3, repeat=0, act=lrelu, skip_concat=False,
        use_sparse=False, reuse=False):
    """model.py:204-216."""
    return _ae(x, filters, z_num, name, num_conv, conv_k, last_k, repeat, act, skip_concat, use_sparse, reuse, True)


def _discriminator(x, filters, name, reuse, is_3d):
    conv = conv3d if is_3d else conv2d
    with variable_scope(name, reuse=reuse) as vs:
        repeat_num = 3                                         # model.py:91 / :107
        d = int(filters / 2)
        for _ in range(repeat_num):
            x = conv(x, d, k=3, act=lrelu)                     # the wrapper's default stride 2 (ops.py:12,15)
            d *= 2
        x = conv(x, d, k=3, s=1, act=lrelu)
        out = conv(x, 1, k=3, s=1)
    variables = get_variables(vs)
    return out, variables


def DiscriminatorPatch(x, filters, name="D", train=True, reuse=False):
    """model.py:89-103 (PatchGAN-style, used by arch='dg')."""
    return _discriminator(x, filters, name, reuse, False)


def DiscriminatorPatch3(x, filters, name="D", train=True, reuse=False):
    """model.py:105-116."""
    return _discriminator(x, filters, name, reuse, True)
