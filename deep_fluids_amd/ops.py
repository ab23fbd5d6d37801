"""Drop-in counterpart of the reference's ``ops.py`` call surface on MI355X.

Same function names, argument order, defaults, return-tuple structure and channels-last layouts
as ``byungsook/deep-fluids/ops.py`` (cited per function), operating on ``torch`` CUDA(=HIP) tensors.
Every bit of arithmetic is done by hand-written gfx950 kernels in ``libdeepfluids_hip.so`` reached
through the C-ABI of ``include/deepfluids_hip.h`` (ctypes, PyTorch's current HIP stream);  PyTorch
supplies device memory, streams and autograd bookkeeping only.  There is no CPU / eager fallback:
a missing library or a non-GPU tensor raises.

TF-1 style hidden state (``slim`` variables inside ``tf.variable_scope``) is reproduced by a
name-keyed registry: ``variable_scope(name, reuse)`` + ``get_variables(scope)``; variable names
follow slim: ``<scope>/<layer>/weights`` ``[k,(k,)k,Cin,Cout]`` / ``[in,out]`` and ``.../biases``.
"""
import contextlib
import contextlib as _contextlib
import math
import os as _os
import threading as _threading

import numpy as np
import torch

from . import _lib
from ._lib import call, query, DF_CONV_BIAS, DF_CONV_LRELU, DF_CONV_MASK, DF_CONV_RESIDUAL  # noqa: F401

__all__ = [
    "lrelu", "conv2d", "conv3d", "linear", "upscale", "upscale3", "resize_nearest_neighbor", "reshape",
    "int_shape", "get_conv_shape", "nchw_to_nhwc", "nhwc_to_nchw", "add", "concat", "sigmoid", "mse_mean",
    "jacobian", "jacobian3", "curl", "curl3", "divergence", "divergence3", "pgrad",
    "vort_np", "curl_np", "grad_np", "jacobian_np3", "l1_mean", "velocity_loss",
    "variable_scope", "get_variables", "get_variable", "reset_variables", "set_random_seed", "all_variables",
]


# --------------------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------------------
def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _prep(t, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor on the MI355X (got %r)" % (name, type(t)))
    if not t.is_cuda:
        raise _lib.DeepFluidsHipError("%s is on %s: deep_fluids_amd has no CPU path (HIP kernels only)" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (the reference is fp32 end-to-end), got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# --------------------------------------------------------------------------------------------------
# variable registry (tf.variable_scope / slim variables)
# --------------------------------------------------------------------------------------------------
_VARS = {}            # full name -> torch.Tensor (requires_grad leaf)
_UNNAMED = {}         # (scope prefix, layer kind) -> how many unnamed layers of that kind were created in this scope entry
_SCOPES = []          # stack of (name, reuse)
_RNG = np.random.RandomState(123)     # main.py:12 tf.set_random_seed(123) / config.py:69
_DEFAULT_DEVICE = "cuda"


class VariableScope(object):
    def __init__(self, name):
        self.name = name


def set_random_seed(seed):
    global _RNG
    _RNG = np.random.RandomState(seed)


def reset_variables():
    _VARS.clear()
    _DIRECT_GRADS.clear()


# Direct gradient targets: data_ptr of a variable -> the tensor its gradient kernel writes (the variable's slice of a trainer's flat gradient slab).
# A backward node then hands the weight / bias gradient kernels that slice as their OUTPUT and returns None for the input, instead of allocating a
# tensor that autograd's AccumulateGrad adds to the (zeroed) slab in a second launch: 42 launches and passes per 2-D step saved.  Valid only where
# every registered variable receives exactly ONE gradient per backward pass and nobody listens for post-accumulate hooks -- the single-process
# `de` / `ae` trainers register theirs (Trainer._build_variables); data parallelism and the GAN trainer (D is applied twice) do not.
_DIRECT_GRADS = {}


def _grad_out(ptr, shape, device):
    """-> (tensor the gradient kernel writes, what backward returns for that input)."""
    g = _DIRECT_GRADS.get(ptr)
    if g is not None:
        return g, None
    t = torch.empty(shape, dtype=torch.float32, device=device)
    return t, t


def all_variables():
    return dict(_VARS)


@contextlib.contextmanager
def variable_scope(name, reuse=False):
    inherited = bool(_SCOPES and _SCOPES[-1][1])
    _SCOPES.append((name, bool(reuse) or inherited))
    prefix = "/".join(s[0] for s in _SCOPES)
    for key in [k for k in _UNNAMED if k[0] == prefix]:      # slim's default layer names (Conv, Conv_1, ...) restart on
        del _UNNAMED[key]                                     # every entry of the scope, so reuse=True finds the same names
    try:
        yield VariableScope("/".join(s[0] for s in _SCOPES))
    finally:
        _SCOPES.pop()


def _scope_prefix():
    return "/".join(s[0] for s in _SCOPES)


def _reusing():
    return bool(_SCOPES and _SCOPES[-1][1])


def get_variable(name, shape, init="xavier", device=None):
    """slim model variable: Xavier-uniform weights / zero biases, created on first use,
    returned as-is under ``reuse=True`` (trainer.py:299-300 builds the test model that way)."""
    full = (_scope_prefix() + "/" if _SCOPES else "") + name
    if full in _VARS:
        if not _reusing():
            raise ValueError("Variable %s already exists; did you mean reuse=True?" % full)
        v = _VARS[full]
        if tuple(v.shape) != tuple(shape):
            raise ValueError("Variable %s has shape %s, requested %s" % (full, tuple(v.shape), tuple(shape)))
        return v
    if _reusing():
        raise ValueError("Variable %s does not exist (reuse=True)" % full)
    if init == "xavier":
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        lim = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
        host = _RNG.uniform(-lim, lim, size=shape).astype(np.float32)
    else:
        host = np.zeros(shape, np.float32)
    v = torch.from_numpy(host).to(device or _DEFAULT_DEVICE).requires_grad_(True)
    _VARS[full] = v
    return v


def set_variable(full_name, value):
    """Inject a variable by its full slim name (parity tests / checkpoint restore)."""
    t = torch.as_tensor(np.asarray(value, np.float32)).to(_DEFAULT_DEVICE).contiguous().requires_grad_(True)
    _VARS[full_name] = t
    return t


def get_variables(scope):
    """tf.contrib.framework.get_variables(vs): variables under a scope, creation order."""
    prefix = (scope.name if isinstance(scope, VariableScope) else str(scope)) + "/"
    return [v for k, v in _VARS.items() if k.startswith(prefix)]


def _layer_name(name, kind):
    if name is not None:
        return name
    key = (_scope_prefix(), kind)
    n = _UNNAMED.get(key, 0)
    _UNNAMED[key] = n + 1
    return kind if n == 0 else "%s_%d" % (kind, n)


# --------------------------------------------------------------------------------------------------
# autograd bindings of the HIP kernels
# --------------------------------------------------------------------------------------------------
class _Lrelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, leak):
        x = _prep(x, "x")
        y = torch.empty_like(x)
        call("df_lrelu_fwd", _ptr(x), _ptr(y), float(leak), x.numel(), _stream())
        ctx.save_for_backward(y)
        ctx.leak = float(leak)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = _prep(gy, "grad")
        gx = torch.empty_like(gy)
        call("df_lrelu_bwd", _ptr(gy), _ptr(y), _ptr(gx), ctx.leak, gy.numel(), _stream())
        return gx, None


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = _prep(a, "a"); b = _prep(b, "b")
        if a.shape != b.shape:
            raise ValueError("add: shapes differ %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        y = torch.empty_like(a)
        call("df_add", _ptr(a), _ptr(b), _ptr(y), a.numel(), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


# Convolution precision of the 128->128-class layers (forward and dgrad):
#   "fp32"   exact fp32 MFMA (v_mfma_f32_32x32x2_f32) -- the default, what the reference computes and bench.py reports;
#   "bf16x3" opt-in: operands split into bf16 hi/lo words, 3 bf16 MFMAs per product block (16 significand bits per
#            operand, fp32 accumulate): ~5x the matrix rate, velocity-field error ~1e-5 (tolerance 1e-4).
CONV_PRECISION = "fp32"


def _sfx(cin, cout):
    """'_bf16x3' when that mode is on and the layer is wide enough for it (thin layers stay on the fp32 VALU kernels)."""
    # (symmetric in cin / cout: the dgrad runs the same layer with the two swapped and must agree with the packed format)
    return "_bf16x3" if (CONV_PRECISION == "bf16x3" and cin >= 16 and cout >= 16 and cin % 4 == 0 and cout % 4 == 0) else ""


# Algorithm of the wide 3-D stride-1 convs (forward and dgrad), fp32 mode only:
#   "auto"     Winograd F(2x2x2,3x3x3) / F(2x2,3x3) (conv_wino.hip / conv_wino2d.hip: same fp32 arithmetic, 3.4x / 2.25x fewer
#              matrix FLOPs) where it applies (Cin and Cout multiples of 32; 3-D extents >= 6, 2-D extents >= 16 x 24) and the direct
#              implicit-GEMM kernel everywhere else;
#   "direct"   always the direct kernel;   "winograd"  Winograd wherever the channel counts allow (tests).
CONV_ALGO = "auto"


# Weight-gradient algorithm request handed to df_conv_wgrad_algo / df_upconv_wgrad_algo (a call argument of the C-ABI, not a
# library global): 0 best available (what df_conv_wgrad does), 1 direct kernels only, 2 at most Winograd in x, 3 Winograd in
# (x,y), 4 Winograd in (x,y,z) wherever instantiated.  Tests pin the variants against each other and the oracle.
WGRAD_ALGO = 0
# True: thin layers (Cin or Cout <= 4) take the general-shape vector-ALU kernels (DF_CONV_VALU_ONLY) instead of the
# matrix-core forms -- the two are compared in the tests.
THIN_VALU_ONLY = False


# Which algorithm every conv / weight-gradient call took, counted per (op, form, shape) when set to a dict (bench.py `dispatch`, tests):
# the library chooses by size and row length, and a shape outside the instantiated variants falls back to a slower kernel SILENTLY --
# this makes it visible.  None (default) = no bookkeeping.
DISPATCH_COUNTS = None
_WGRAD_FORMS = {0: "direct-mfma", 1: "winograd-x", 2: "winograd-xy", 3: "winograd-xyz", 10: "thin-mfma", 11: "thin-valu"}


def _count(op, form, dims, cin, cout):
    if DISPATCH_COUNTS is not None:
        key = "%s %s %s C%d->%d" % (op, form, "x".join(str(int(d)) for d in dims[1:] if int(d) > 1 or len(dims) < 4), cin, cout)
        DISPATCH_COUNTS[key] = DISPATCH_COUNTS.get(key, 0) + 1


def _wgrad(x, dp, gw, gb, B, D, H, W, cin, cout, kz, sfx=""):
    if DISPATCH_COUNTS is not None:
        fid = query("df_conv_wgrad_form", B, D, H, W, cin, cout, kz, int(WGRAD_ALGO))
        # (bf16x3 mode: the library keeps the fp32 (x,y,z) Winograd form where it exists -- it is faster than the split-operand kernel)
        form = "bf16x3" if (sfx and fid != 3) else _WGRAD_FORMS.get(fid, "?")
        _count("wgrad", form, (B, D, H, W), cin, cout)
    nbytes = query("df_conv_wgrad_workspace_bytes", B, D, H, W, cin, cout, kz)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    if sfx:
        call("df_conv_wgrad" + sfx, _ptr(x), _ptr(dp), _ptr(gw), _ptr(gb), B, D, H, W, cin, cout, kz, _ptr(ws), nbytes, _stream())
    else:
        call("df_conv_wgrad_algo", _ptr(x), _ptr(dp), _ptr(gw), _ptr(gb), B, D, H, W, cin, cout, kz, _ptr(ws), nbytes,
             int(WGRAD_ALGO), _stream())


# Levels whose kernels cannot fill the chip (B x voxels x taps <= this) run the weight gradient of a layer on a SECOND stream, concurrently
# with the same layer's dgrad: both only read the incoming gradient and each launches few workgroups with a long serial chain (the
# low-resolution levels, and every level of the 2-D net at the reference's default batch 8).  Same kernels, same arguments: results are
# bitwise those of the serial order.  0 = always serial.  Measured (profiles/r06_probes.md section 1): 2-D 128x96 B = 8 4.95 -> 4.46 ms;
# above ~1M the two kernels each fill the chip and only contend (cfg3's 16x24x16 level at B = 16: +1.7 % on the step).
CONCURRENT_WGRAD_WORK = int(_os.environ.get("DF_CONCURRENT_WGRAD_WORK", str(1 << 20)))
_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.index if device.index is not None else torch.cuda.current_device())
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return s


class _WgradLane(object):
    """The second stream of one backward node: ``run(fn, *tensors)`` launches ``fn`` there after everything issued so far on the
    node's own stream; ``tensors`` (what fn reads or writes) stay referenced until ``join()`` makes the node's stream wait for the
    lane -- the caching allocator would otherwise hand a freed block to the next allocation while the lane still uses it."""

    def __init__(self, dims, like, taps):
        n = int(taps)
        for d in dims:
            n *= int(d)
        # (not while a hipGraph is being captured: ROCm 7.2 replays a forked graph no faster than the serial one, and its launch
        #  costs the host 0.5-6 ms instead of 0.1 -- profiles/r06_probes.md, section 1)
        self.on = 0 < n <= CONCURRENT_WGRAD_WORK and not torch.cuda.is_current_stream_capturing()
        self.keep = []
        if self.on:
            self.main = torch.cuda.current_stream()
            self.side = _side_stream(like.device)

    def run(self, fn, *tensors):
        if not self.on:
            return fn()
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            fn()
        self.keep.extend(tensors)

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)
            self.keep = []


# Which 3-D Winograd family the plain stride-1 convs (forward and dgrad) take where `_use_wino` says 3:
#   "f224"  F(2,3) x F(2,3) x F(4,3) (conv_wino43.hip, round 6): 6 matrix multiply-adds per output voxel and channel pair -- the default;
#   "f222"  F(2,3)^3 (conv_wino.hip): 8 of them, about one bit more accurate.
# The 27-point forms of an up-sampling block's first conv (forward / pooled adjoint) exist in the F(2,3)^3 family only; both families write /
# read the same sign-word layout, so they mix freely inside a block.
WINO3D_FAMILY = _os.environ.get("DF_WINO3D_FAMILY", "f224")
# ... and the 2-D twin: "f24" (default) = F(2,3) x F(4,3) (conv_wino2d43.hip: 3 multiply-adds per output pixel and channel pair) for forward convs
# and dgrads, "f22" = F(2,3)^2 (conv_wino2d.hip: 4), "auto" = f24 for the forward convs, f22 for the dgrads (the default before the f24
# epilogue moved to 16-byte accesses along the channels -- its 32 scalar fp32 mask reads per lane were not hidden; now 8 float4 reads: cfg2
# step 16.93 ms with "auto", 16.46 with "f24", profiles/r06_probes.md section 7).  The 9-point forms of an up-sampling block's first conv
# exist in the F(2,3)^2 family only.
WINO2D_FAMILY = _os.environ.get("DF_WINO2D_FAMILY", "f24")


def _w2fam(mode):
    """The 2-D Winograd family of a conv with pack mode `mode` (0 forward, 1 dgrad)."""
    return WINO2D_FAMILY if WINO2D_FAMILY != "auto" else ("f24" if mode == 0 else "f22")


def _use_wino(cin, cout, dims, kz):
    """0: direct kernel; 3: 3-D Winograd F(2x2x2,3x3x3) (conv_wino.hip); 2: 2-D Winograd F(2x2,3x3) (conv_wino2d.hip)."""
    if CONV_ALGO == "direct" or CONV_PRECISION != "fp32" or cin % 32 or cout % 32:
        return 0
    if kz == 3:
        # (from 6 voxels per axis on: at 7x10x7 -- cfg4's lowest level -- the direct kernel has 24 workgroups of 184 us each, the Winograd
        #  kernel 64 of a sixth of the work)
        return 3 if (CONV_ALGO == "winograd" or min(dims[1], dims[2], dims[3]) >= 6) else 0
    return 2 if (CONV_ALGO == "winograd" or (dims[2] >= 16 and dims[3] >= 24)) else 0


def _pack(w, taps, cin, cout, mode, dims=None, fp32=False, family=None):
    """Packed MFMA operand of w for the stride-1 conv on `dims` (mode 0: forward, mode 1: dgrad).  ``fp32=True`` forces the exact-fp32
    operand format whatever CONV_PRECISION says (kernels that have no bf16x3 variant: the stride-2 forward).  ``family="f222"``: the
    F(2,3)^3 operand whatever WINO3D_FAMILY says (the 27-point forms)."""
    algo = _use_wino(cin, cout, dims, 3 if taps == 27 else 1) if dims is not None else 0
    if algo:
        fn = ("df_wino43" if (family or WINO3D_FAMILY) == "f224" else "df_wino") if algo == 3 else (
            "df_wino2d43" if (family or _w2fam(mode)) == "f24" else "df_wino2d")
        wp = torch.empty(query(fn + "_packed_elems", cin, cout, mode), dtype=torch.float32, device=w.device)
        call(fn + "_pack_weights", _ptr(w), _ptr(wp), cin, cout, mode, _stream())
        return wp
    sfx = "" if fp32 else _sfx(cin, cout)
    n = query("df_conv_packed_elems" + sfx, taps, cin, cout, mode)
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    call("df_conv_pack_weights" + sfx, _ptr(w), _ptr(wp), taps, cin, cout, mode, _stream())
    return wp


# Sign-bit masks (conv_wino.hip): the forward convs of a fused generator block whose outputs only serve, in the backward pass, as the
# lrelu mask of the next layer's dgrad also emit that mask as bit words (1/32 of the bytes); the masked dgrad then reads the words
# instead of the fp32 activation.  Same arithmetic, bit-identical results; only the 3-D Winograd kernels have the path.
SIGN_BIT_MASKS = _os.environ.get("DF_SIGN_BIT_MASKS", "1") != "0"


def _bits_kind(cin, cout, dims, kz):
    """Which sign-word layout a forward conv (cin -> cout on `dims`) can emit for the masked dgrad of the layer above: 0 none, 3 the 3-D Winograd
    kernels' bytes (both 3-D families read and write them), 2 the words of the 2-D F(2,3) x F(4,3) kernel -- only where the dgrads run on that
    kernel too (WINO2D_FAMILY "f24": forward and dgrad share the thread <-> output mapping the words are indexed by)."""
    if not SIGN_BIT_MASKS:
        return 0
    algo = _use_wino(cin, cout, dims, kz)
    if algo == 3:
        return 3
    return 2 if (algo == 2 and _w2fam(0) == "f24" and _w2fam(1) == "f24") else 0


def _new_bits(dims, c, like, kind=3):
    B, D, H, W = dims
    nbytes = query("df_wino_signbits_bytes", B, D, H, W, c) if kind == 3 else query("df_wino2d43_signbits_bytes", B, H, W, c)
    return torch.empty(nbytes // 8, dtype=torch.int64, device=like.device)


def sign_words2d_to_mask(bits, dims, c):
    """Decode the sign words of ``df_wino2d43_conv_bits`` (conv_wino2d43.hip) into a bool tensor ``[B, H, W, c]`` = (activation > 0).  One 32-bit word
    per (tile block of 16 x 32 pixels, 32-cout slice cs, thread); thread = (wave = 4 yh + rp, lane = 16 kq + 4 qm + qi); bit (4 nb + e) * 4 + cc =
    pixel (16 by + 4 rp + 2 (kq >> 1) + (qi >> 1), 32 bx + 16 (kq & 1) + 4 e + 2 yh + (qi & 1)), channel 32 cs + 16 nb + 4 qm + cc."""
    B, _, H, W = (int(v) for v in dims)
    nby, nbx, ncs = -(-H // 16), -(-W // 32), c // 32
    wd = bits.view(torch.int32)[:B * nby * nbx * ncs * 512].view(B, nby, nbx, ncs, 2, 4, 2, 2, 4, 2, 2)      # [B, by, bx, cs, yh, rp, etr, kql, qm, qih, qil]
    sh = torch.arange(32, device=bits.device, dtype=torch.int32).view(2, 4, 4)                                # [nb, e, cc]
    m = ((wd[..., None, None, None] >> sh) & 1).bool()
    #   0  1   2   3   4   5   6    7    8   9    10   11  12  13
    #  [B, by, bx, cs, yh, rp, etr, kql, qm, qih, qil, nb, e, cc]  ->  (B | by rp etr qih | bx kql e yh qil | cs nb qm cc)
    m = m.permute(0, 1, 5, 6, 9, 2, 7, 12, 4, 10, 3, 11, 8, 13).reshape(B, nby * 16, nbx * 32, c)
    return m[:, :H, :W].contiguous()


def sign_bits_to_mask(bits, dims, c):
    """Decode the sign words of ``df_wino_conv_fwd_bits`` / ``_addup_bits`` / ``df_wino_upconv_fwd_bits`` (conv_wino.hip, kSignBits) into
    a bool tensor ``[B, D, H, W, c]`` = (activation > 0).  Layout: one byte per (tile block, cout slice, wave = (z-row th, xi_z), cout
    16-block nb, lane = (kq, tl)); bit s = (dz, dy, dx) of the lane's 2x2x2 outputs at (4 bz + 2 th + dz, 8 by + 2 kq + dy,
    8 bx + 2 xi_z + dx), channel 32 cs + 16 nb + tl.  The fetch-list counterpart for layers whose fp32 activation is never written."""
    B, D, H, W = (int(v) for v in dims)
    nbz, nby, nbx, ncs = -(-D // 4), -(-H // 8), -(-W // 8), c // 32
    by = bits.view(torch.uint8)[:B * nbz * nby * nbx * ncs * 1024].view(B, nbz, nby, nbx, ncs, 2, 4, 2, 4, 16)
    sh = torch.arange(8, device=bits.device, dtype=torch.uint8).view(2, 2, 2)
    m = ((by[..., None, None, None] >> sh) & 1).bool()      # [B, bz, by, bx, cs, th, xz, nb, kq, tl, dz, dy, dx]
    m = m.permute(0, 1, 5, 10, 2, 8, 11, 3, 6, 12, 4, 7, 9).reshape(B, nbz * 4, nby * 8, nbx * 8, c)
    return m[:, :D, :H, :W].contiguous()


# Fetch of the block-tail sign words (see ACTIVATION_FETCH below): when set to a list, every up-sampling block on the production tail
# (df_wino_conv_fwd_addup_bits: the last conv's activation is never written) appends ``(bits, fdims, cout)``; tests decode them with
# sign_bits_to_mask and compare with the activations of the fp32-mask path.
SIGN_BITS_FETCH = None


def _conv_raw(x, wp, bias, residual, mask_src, dims, cin, cout, kz, flags, leak, sign_bits=None, mask_bits=None):
    """`wp` must come from ``_pack(..., dims)`` with the same dims AND mode (the two agree on the algorithm and the family: every forward
    call carries DF_CONV_BIAS, no dgrad call does -- that is how the pack mode is recognised here)."""
    mode = 0 if (flags & DF_CONV_BIAS) else 1
    B, D, H, W = dims
    y = torch.empty((B, D, H, W, cout), dtype=torch.float32, device=x.device)
    algo = _use_wino(cin, cout, dims, kz)
    if DISPATCH_COUNTS is not None:
        thin = min(cin, cout) <= 4
        _count("conv", ("winograd-f2x2x4" if WINO3D_FAMILY == "f224" else "winograd-f2x2x2") if algo == 3 else
               ("winograd-f2x4" if _w2fam(mode) == "f24" else "winograd-f2x2") if algo == 2 else
               ("thin" + ("-valu-forced" if THIN_VALU_ONLY else "")) if thin else ("direct-mfma" + _sfx(cin, cout)), dims, cin, cout)
    if algo == 3 and WINO3D_FAMILY == "f224":
        call("df_wino43_conv", _ptr(x), _ptr(wp), _ptr(bias), _ptr(residual), _ptr(mask_src), _ptr(mask_bits), _ptr(y), None, _ptr(sign_bits),
             B, D, H, W, cin, cout, flags, float(leak), _stream())
        return y
    if algo == 3:
        if sign_bits is not None or mask_bits is not None:
            call("df_wino_conv_fwd_bits", _ptr(x), _ptr(wp), _ptr(bias), _ptr(mask_bits), _ptr(y), _ptr(sign_bits), B, D, H, W, cin, cout,
                 flags, float(leak), _stream())
            return y
        call("df_wino_conv_fwd", _ptr(x), _ptr(wp), _ptr(bias), _ptr(residual), _ptr(mask_src), _ptr(y), B, D, H, W, cin, cout,
             flags, float(leak), _stream())
        return y
    if (sign_bits is not None or mask_bits is not None) and not (algo == 2 and _w2fam(mode) == "f24"):
        raise _lib.DeepFluidsHipError("sign-bit masks exist for the 3-D Winograd kernels and the 2-D F(2,3) x F(4,3) kernel only")
    if algo == 2 and _w2fam(mode) == "f24" and (sign_bits is not None or mask_bits is not None):
        call("df_wino2d43_conv_bits", _ptr(x), _ptr(wp), _ptr(bias), _ptr(mask_bits), _ptr(y), _ptr(sign_bits), B, H, W, cin, cout, flags, float(leak),
             _stream())
        return y
    if algo == 2 and _w2fam(mode) == "f24":
        call("df_wino2d43_conv", _ptr(x), _ptr(wp), _ptr(bias), _ptr(residual), _ptr(mask_src), _ptr(y), B, H, W, cin, cout, flags, float(leak),
             _stream())
        return y
    if algo == 2:
        call("df_wino2d_conv_fwd", _ptr(x), _ptr(wp), _ptr(bias), _ptr(residual), _ptr(mask_src), _ptr(y), B, H, W, cin, cout,
             flags, float(leak), _stream())
        return y
    if THIN_VALU_ONLY:
        flags |= _lib.DF_CONV_VALU_ONLY
    call("df_conv_fwd" + _sfx(cin, cout), _ptr(x), _ptr(wp), _ptr(bias), _ptr(residual), _ptr(mask_src), _ptr(y), B, D, H, W,
         cin, cout, kz, flags, float(leak), _stream())
    return y


class _ConvSame3(torch.autograd.Function):
    """k=3, stride-1, SAME conv (+bias, + optional fused lrelu) on the fp32 MFMA kernels."""

    @staticmethod
    def forward(ctx, x, w, b, leak):
        x = _prep(x, "x"); w = _prep(w, "weights"); b = _prep(b, "biases")
        nd = x.dim() - 2
        kz = 3 if nd == 3 else 1
        taps = 27 if nd == 3 else 9
        cin, cout = w.shape[-2], w.shape[-1]
        if tuple(w.shape[:-2]) != (3,) * nd or x.shape[-1] != cin:
            raise ValueError("conv: weights %s do not match input %s (k=3 only)" % (tuple(w.shape), tuple(x.shape)))
        dims = (x.shape[0], x.shape[1] if nd == 3 else 1, x.shape[-3], x.shape[-2])
        wp = _pack(w, taps, cin, cout, 0, dims)
        flags = DF_CONV_BIAS | (DF_CONV_LRELU if leak is not None else 0)
        y = _conv_raw(x, wp, b, None, None, dims, cin, cout, kz, flags, leak if leak is not None else 0.0)
        y = y.view(x.shape[:-1] + (cout,))
        if ACTIVATION_FETCH is not None and leak is not None:
            ACTIVATION_FETCH.append(y)
        ctx.save_for_backward(x, w, y if leak is not None else None)
        ctx.leak = leak
        ctx.geom = (dims, cin, cout, kz, taps)
        ctx.bptr = b.data_ptr()
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        dims, cin, cout, kz, taps = ctx.geom
        B, D, H, W = dims
        gy = _prep(gy, "grad")
        if ctx.leak is not None:
            dp = torch.empty_like(gy)
            call("df_lrelu_bwd", _ptr(gy), _ptr(y), _ptr(dp), float(ctx.leak), gy.numel(), _stream())
        else:
            dp = gy
        gw, rw = _grad_out(w.data_ptr(), w.shape, x.device)
        gb, rb = _grad_out(ctx.bptr, (cout,), x.device)
        _wgrad(x, dp, gw, gb, B, D, H, W, cin, cout, kz, _sfx(cin, cout))
        gx = None
        if ctx.needs_input_grad[0]:
            wpd = _pack(w, taps, cin, cout, 1, dims)
            gx = _conv_raw(dp, wpd, None, None, None, dims, cout, cin, kz, 0, 0.0).view(x.shape)
        return gx, rw, rb, None


class _GenBlock(torch.autograd.Function):
    """One generator block as a single autograd node: n x [conv k3 s1 + bias + lrelu] and the residual add of the
    block input (model.py:24-40 / 66-82).  Forward = the same kernels as the layer-by-layer path; the hand-written
    reverse chain uses the fused epilogues of ``df_conv_fwd``: each dgrad multiplies by the lrelu slope of the layer
    below (DF_CONV_MASK) and the first layer's dgrad adds the skip gradient (DF_CONV_RESIDUAL), which removes three
    element-wise passes and one gradient-accumulation pass over the block's activations per block."""

    @staticmethod
    def forward(ctx, x0, leak, *wb):
        x0 = _prep(x0, "x")
        n = len(wb) // 2
        nd = x0.dim() - 2
        kz = 3 if nd == 3 else 1
        taps = 27 if nd == 3 else 9
        dims = (x0.shape[0], x0.shape[1] if nd == 3 else 1, x0.shape[-3], x0.shape[-2])
        xs = [x0]
        bits = []
        x = x0
        for i in range(n):
            w = _prep(wb[2 * i], "weights"); b = _prep(wb[2 * i + 1], "biases")
            cin, cout = w.shape[-2], w.shape[-1]
            if tuple(w.shape[:-2]) != (3,) * nd or x.shape[-1] != cin:
                raise ValueError("gen_block: weights %s do not match input %s" % (tuple(w.shape), tuple(x.shape)))
            wp = _pack(w, taps, cin, cout, 0, dims)
            # outputs of convs 1 .. n-1 are the masks of the dgrads of convs 2 .. n
            bk = _bits_kind(cin, cout, dims, kz) if i < n - 1 else 0
            sb = _new_bits(dims, cout, x0, bk) if bk else None
            bits.append(sb)
            x = _conv_raw(x, wp, b, None, None, dims, cin, cout, kz, DF_CONV_BIAS | DF_CONV_LRELU, leak, sign_bits=sb).view(
                x0.shape[:-1] + (cout,))
            xs.append(x)
        if x.shape != x0.shape:
            raise ValueError("gen_block: residual add needs Cout == Cin of the block")
        y = torch.empty_like(x)
        call("df_add", _ptr(x), _ptr(x0), _ptr(y), x.numel(), _stream())
        if ACTIVATION_FETCH is not None:
            ACTIVATION_FETCH.extend(xs[1:])
        ctx.save_for_backward(*(xs + [wb[2 * i] for i in range(n)]))
        ctx.geom = (n, dims, kz, taps, float(leak))
        ctx.bits = bits
        ctx.bptrs = [wb[2 * i + 1].data_ptr() for i in range(n)]
        return y

    @staticmethod
    def backward(ctx, dy):
        n, dims, kz, taps, leak = ctx.geom
        saved = ctx.saved_tensors
        xs, ws = saved[:n + 1], saved[n + 1:]
        B, D, H, W = dims
        dy = _prep(dy, "grad")
        dp = torch.empty_like(dy)
        call("df_lrelu_bwd", _ptr(dy), _ptr(xs[n]), _ptr(dp), leak, dy.numel(), _stream())
        grads = [None] * (2 * n)
        dx0 = None
        lane = _WgradLane(dims, dy, taps)
        for i in range(n, 0, -1):
            w = ws[i - 1]
            cin, cout = w.shape[-2], w.shape[-1]
            gw, rw = _grad_out(w.data_ptr(), w.shape, dy.device)
            gb, rb = _grad_out(ctx.bptrs[i - 1], (cout,), dy.device)
            lane.run(lambda: _wgrad(xs[i - 1], dp, gw, gb, B, D, H, W, cin, cout, kz, _sfx(cin, cout)), dp, gw, gb)
            grads[2 * (i - 1)] = rw; grads[2 * (i - 1) + 1] = rb
            wpd = _pack(w, taps, cin, cout, 1, dims)
            if i > 1:      # dgrad, times the lrelu slope of the layer below: directly the next dp
                mb = ctx.bits[i - 2]      # sign bits of conv i-1's output, if its forward emitted them
                dp = _conv_raw(dp, wpd, None, None, None if mb is not None else xs[i - 1], dims, cout, cin, kz, DF_CONV_MASK, leak,
                               mask_bits=mb).view(xs[i - 1].shape)
            elif ctx.needs_input_grad[0]:   # dgrad of the first layer + the skip gradient
                dx0 = _conv_raw(dp, wpd, None, dy, None, dims, cout, cin, kz, DF_CONV_RESIDUAL, 0.0).view(xs[0].shape)
        lane.join()
        return (dx0, None) + tuple(grads)


class _ConvChain(torch.autograd.Function):
    """n x [conv k3 s1 + bias + lrelu] in a row as ONE autograd node -- the per-level conv stack of the encoder (model.py:131-136 / 167-172;
    the first conv may change the channel count).  Same forward kernels as the layer-by-layer path; the reverse chain applies the lrelu
    slope of the layer below in each dgrad's epilogue (DF_CONV_MASK, sign bits where the 3-D Winograd forward emits them) exactly as
    _GenBlock does, so only the LAST layer's element-wise lrelu-backward pass remains (its gradient arrives from the concat)."""

    @staticmethod
    def forward(ctx, x0, leak, *wb):
        x0 = _prep(x0, "x")
        n = len(wb) // 2
        nd = x0.dim() - 2
        kz = 3 if nd == 3 else 1
        taps = 27 if nd == 3 else 9
        dims = (x0.shape[0], x0.shape[1] if nd == 3 else 1, x0.shape[-3], x0.shape[-2])
        xs = [x0]
        bits = []
        x = x0
        for i in range(n):
            w = _prep(wb[2 * i], "weights"); b = _prep(wb[2 * i + 1], "biases")
            cin, cout = w.shape[-2], w.shape[-1]
            if tuple(w.shape[:-2]) != (3,) * nd or x.shape[-1] != cin:
                raise ValueError("conv_chain: weights %s do not match input %s" % (tuple(w.shape), tuple(x.shape)))
            wp = _pack(w, taps, cin, cout, 0, dims)
            bk = _bits_kind(cin, cout, dims, kz) if i < n - 1 else 0
            sb = _new_bits(dims, cout, x0, bk) if bk else None
            bits.append(sb)
            x = _conv_raw(x, wp, b, None, None, dims, cin, cout, kz, DF_CONV_BIAS | DF_CONV_LRELU, leak, sign_bits=sb).view(
                x0.shape[:-1] + (cout,))
            xs.append(x)
        if ACTIVATION_FETCH is not None:
            ACTIVATION_FETCH.extend(xs[1:])
        ctx.save_for_backward(*(xs + [wb[2 * i] for i in range(n)]))
        ctx.geom = (n, dims, kz, taps, float(leak))
        ctx.bits = bits
        ctx.bptrs = [wb[2 * i + 1].data_ptr() for i in range(n)]
        return x

    @staticmethod
    def backward(ctx, dy):
        n, dims, kz, taps, leak = ctx.geom
        saved = ctx.saved_tensors
        xs, ws = saved[:n + 1], saved[n + 1:]
        B, D, H, W = dims
        dy = _prep(dy, "grad")
        dp = torch.empty_like(dy)
        call("df_lrelu_bwd", _ptr(dy), _ptr(xs[n]), _ptr(dp), leak, dy.numel(), _stream())
        grads = [None] * (2 * n)
        dx0 = None
        lane = _WgradLane(dims, dy, taps)
        for i in range(n, 0, -1):
            w = ws[i - 1]
            cin, cout = w.shape[-2], w.shape[-1]
            gw, rw = _grad_out(w.data_ptr(), w.shape, dy.device)
            gb, rb = _grad_out(ctx.bptrs[i - 1], (cout,), dy.device)
            lane.run(lambda: _wgrad(xs[i - 1], dp, gw, gb, B, D, H, W, cin, cout, kz, _sfx(cin, cout)), dp, gw, gb)
            grads[2 * (i - 1)] = rw; grads[2 * (i - 1) + 1] = rb
            if i > 1 or ctx.needs_input_grad[0]:
                wpd = _pack(w, taps, cin, cout, 1, dims)
                if i > 1:      # dgrad, times the lrelu slope of the layer below: directly the next dp
                    mb = ctx.bits[i - 2]
                    dp = _conv_raw(dp, wpd, None, None, None if mb is not None else xs[i - 1], dims, cout, cin, kz, DF_CONV_MASK, leak,
                                   mask_bits=mb).view(xs[i - 1].shape)
                else:
                    dx0 = _conv_raw(dp, wpd, None, None, None, dims, cout, cin, kz, 0, 0.0).view(xs[0].shape)
        lane.join()
        return (dx0, None) + tuple(grads)


class _UpGenBlock(torch.autograd.Function):
    """``x0 = upscale(xc, 2)`` followed by one generator block (model.py:36-40 / 78-82) as a single autograd node that
    never materialises ``x0``: the block's first conv runs as parity-class 2x2(x2)-tap convs on the coarse input
    (``df_upconv_*``: 3.4x fewer FLOPs forward, dgrad and 2.25x fewer in wgrad), the block-end residual add reads the
    coarse tensor (``df_add_up2x``) and the skip gradient is the 2x2(x2) sum-pool of dy (``df_upsample2x_bwd``)."""

    @staticmethod
    def forward(ctx, xc, leak, *wb):
        xc = _prep(xc, "x")
        n = len(wb) // 2
        nd = xc.dim() - 2
        is3d = nd == 3
        kz = 3 if is3d else 1
        taps = 27 if is3d else 9
        cdims = (xc.shape[0], xc.shape[1] if is3d else 1, xc.shape[-3], xc.shape[-2])
        fdims = (cdims[0], 2 * cdims[1] if is3d else 1, 2 * cdims[2], 2 * cdims[3])
        C = int(xc.shape[-1])
        fshape = (xc.shape[0],) + tuple(2 * int(d) for d in xc.shape[1:-1]) + (C,)
        xs = []
        bits = []
        y = None
        tail_bits = None
        for i in range(n):
            w = _prep(wb[2 * i], "weights"); b = _prep(wb[2 * i + 1], "biases")
            cin, cout = w.shape[-2], w.shape[-1]
            if tuple(w.shape[:-2]) != (3,) * nd or cin != C or cout != C:
                raise ValueError("up_gen_block: weights %s do not match %d channels" % (tuple(w.shape), C))
            # (2-D: the block's first conv runs the 9-point F(2,3)^2 form, which has no sign words -- the dgrad above it reads its fp32 activation)
            bk = _bits_kind(cin, cout, fdims, kz) if (i < n - 1 and (is3d or i > 0)) else 0
            sb = _new_bits(fdims, cout, xc, bk) if bk else None
            bits.append(sb)
            if DISPATCH_COUNTS is not None and i == 0:
                _count("upconv", "winograd-27pt" if (is3d and _use_wino(cin, cout, fdims, kz) == 3) else
                       "winograd2d-9pt" if (not is3d and _use_wino(cin, cout, fdims, kz) == 2) else "parity-class" + _sfx(cin, cout), fdims, cin, cout)
            if i == 0 and is3d and _use_wino(cin, cout, fdims, kz) == 3:
                # 27-point up-sampling-aware Winograd form (conv_wino.hip, UP variant): the F(2,3)^3 operand of a plain conv
                wp = _pack(w, taps, cin, cout, 0, fdims, family="f222")
                x = torch.empty(fshape, dtype=torch.float32, device=xc.device)
                if sb is not None:
                    call("df_wino_upconv_fwd_bits", _ptr(xc), _ptr(wp), _ptr(b), _ptr(x), _ptr(sb), cdims[0], cdims[1], cdims[2], cdims[3],
                         cin, cout, float(leak), _stream())
                else:
                    call("df_wino_upconv_fwd", _ptr(xc), _ptr(wp), _ptr(b), _ptr(x), cdims[0], cdims[1], cdims[2], cdims[3], cin, cout,
                         DF_CONV_BIAS | DF_CONV_LRELU, float(leak), _stream())
            elif i == 0 and not is3d and _use_wino(cin, cout, fdims, kz) == 2:
                # 2-D twin: 9 of the 16 Winograd products (conv_wino2d.hip, UP variant)
                wp = _pack(w, taps, cin, cout, 0, fdims, family="f22")
                x = torch.empty(fshape, dtype=torch.float32, device=xc.device)
                call("df_wino2d_upconv_fwd", _ptr(xc), _ptr(wp), _ptr(b), _ptr(x), cdims[0], cdims[2], cdims[3], cin, cout,
                     DF_CONV_BIAS | DF_CONV_LRELU, float(leak), _stream())
            elif i == 0:
                sfx = _sfx(cin, cout)
                wp = torch.empty(query("df_upconv_packed_elems" + sfx, cin, cout, kz, 0), dtype=torch.float32,
                                 device=xc.device)
                call("df_upconv_pack_weights" + sfx, _ptr(w), _ptr(wp), cin, cout, kz, 0, _stream())
                x = torch.empty(fshape, dtype=torch.float32, device=xc.device)
                call("df_upconv_fwd" + sfx, _ptr(xc), _ptr(wp), _ptr(b), _ptr(x), cdims[0], cdims[1], cdims[2], cdims[3], cin, cout,
                     kz, DF_CONV_BIAS | DF_CONV_LRELU, float(leak), _stream())
            else:
                wp = _pack(w, taps, cin, cout, 0, fdims)
                if i == n - 1 and is3d and _use_wino(cin, cout, fdims, kz) == 3 and SIGN_BIT_MASKS and ACTIVATION_FETCH is None:
                    # block-end skip add fused into the last conv's epilogue, and of the conv's own activation only the sign bits are
                    # kept (all the backward tail needs of it): y = lrelu(conv(x)) + upscale(xc), tail_bits = (lrelu(conv(x)) > 0)
                    f224 = WINO3D_FAMILY == "f224"
                    _count("conv", "winograd-%s+addup+signwords" % ("f2x2x4" if f224 else "f2x2x2"), fdims, cin, cout)
                    tail_bits = _new_bits(fdims, cout, xc)
                    y = torch.empty(fshape, dtype=torch.float32, device=xc.device)
                    if f224:
                        call("df_wino43_conv", _ptr(x), _ptr(wp), _ptr(b), _ptr(xc), None, None, None, _ptr(y), _ptr(tail_bits), fdims[0], fdims[1],
                             fdims[2], fdims[3], cin, cout, DF_CONV_BIAS | DF_CONV_LRELU | _lib.DF_CONV_ADDUP, float(leak), _stream())
                    else:
                        call("df_wino_conv_fwd_addup_bits", _ptr(x), _ptr(wp), _ptr(b), _ptr(xc), _ptr(y), _ptr(tail_bits), fdims[0], fdims[1],
                             fdims[2], fdims[3], cin, cout, float(leak), _stream())
                    if SIGN_BITS_FETCH is not None:
                        SIGN_BITS_FETCH.append((tail_bits, fdims, cout))
                    x = None
                elif i == n - 1 and is3d and _use_wino(cin, cout, fdims, kz) == 3:
                    # (fp32 masks: the activation is a second output)
                    f224 = WINO3D_FAMILY == "f224"
                    _count("conv", "winograd-%s+addup" % ("f2x2x4" if f224 else "f2x2x2"), fdims, cin, cout)
                    x_in, x = x, torch.empty(fshape, dtype=torch.float32, device=xc.device)
                    y = torch.empty(fshape, dtype=torch.float32, device=xc.device)
                    if f224:
                        call("df_wino43_conv", _ptr(x_in), _ptr(wp), _ptr(b), _ptr(xc), None, None, _ptr(x), _ptr(y), None, fdims[0], fdims[1],
                             fdims[2], fdims[3], cin, cout, DF_CONV_BIAS | DF_CONV_LRELU | _lib.DF_CONV_ADDUP, float(leak), _stream())
                    else:
                        call("df_wino_conv_fwd_addup", _ptr(x_in), _ptr(wp), _ptr(b), _ptr(xc), _ptr(x), _ptr(y), fdims[0], fdims[1],
                             fdims[2], fdims[3], cin, cout, float(leak), _stream())
                elif i == n - 1 and not is3d and _bits_kind(cin, cout, fdims, kz) == 2 and ACTIVATION_FETCH is None:
                    # the 2-D twin of the fused block tail: y = lrelu(conv(x)) + upscale(xc) from the conv's epilogue, only the sign words of its activation kept
                    _count("conv", "winograd-f2x4+addup+signwords", fdims, cin, cout)
                    tail_bits = _new_bits(fdims, cout, xc, 2)
                    y = torch.empty(fshape, dtype=torch.float32, device=xc.device)
                    call("df_wino2d43_conv_addup_bits", _ptr(x), _ptr(wp), _ptr(b), _ptr(xc), _ptr(y), _ptr(tail_bits), fdims[0], fdims[2], fdims[3], cin, cout,
                         float(leak), _stream())
                    x = None
                else:
                    x = _conv_raw(x, wp, b, None, None, fdims, cin, cout, kz, DF_CONV_BIAS | DF_CONV_LRELU, leak, sign_bits=sb).view(fshape)
            xs.append(x)
        if y is None:
            y = torch.empty_like(x)
            call("df_add_up2x", _ptr(x), _ptr(xc), _ptr(y), cdims[0], cdims[1], cdims[2], cdims[3], C, int(is3d), _stream())
        if ACTIVATION_FETCH is not None:
            ACTIVATION_FETCH.extend(xs)
        ctx.save_for_backward(*([xc] + xs + [wb[2 * i] for i in range(n)]))
        ctx.geom = (n, cdims, fdims, kz, taps, float(leak), C, is3d)
        ctx.bits = bits
        ctx.tail_bits = tail_bits
        ctx.bptrs = [wb[2 * i + 1].data_ptr() for i in range(n)]
        return y

    @staticmethod
    def backward(ctx, dy):
        n, cdims, fdims, kz, taps, leak, C, is3d = ctx.geom
        saved = ctx.saved_tensors
        xc, xs, ws = saved[0], saved[1:n + 1], saved[n + 1:]
        B, D, H, W = fdims
        dy = _prep(dy, "grad")
        dp = torch.empty_like(dy)
        dxc = None
        if ctx.tail_bits is not None:
            # the lrelu mask of the last conv from its sign bits (its fp32 activation was never written): 6.9 GB moved at the cfg3 top
            # level instead of 12.9; the pooled skip gradient comes with it (dropped below if xc needs no gradient)
            dxc = torch.empty_like(xc)
            if is3d:
                call("df_lrelu_bits_bwd_pool2x", _ptr(dy), _ptr(ctx.tail_bits), _ptr(dp), _ptr(dxc), leak, cdims[0], cdims[1], cdims[2], cdims[3],
                     C, _stream())
            else:
                call("df_lrelu_words2d_bwd_pool2x", _ptr(dy), _ptr(ctx.tail_bits), _ptr(dp), _ptr(dxc), leak, cdims[0], cdims[2], cdims[3], C, _stream())
        elif ctx.needs_input_grad[0]:
            # both consumers of dy in one pass: the masked gradient entering the last conv and the skip path's 2x2(x2) sum-pool
            dxc = torch.empty_like(xc)
            call("df_lrelu_bwd_pool2x", _ptr(dy), _ptr(xs[n - 1]), _ptr(dp), _ptr(dxc), leak, cdims[0], cdims[1], cdims[2], cdims[3], C,
                 int(is3d), _stream())
        else:
            call("df_lrelu_bwd", _ptr(dy), _ptr(xs[n - 1]), _ptr(dp), leak, dy.numel(), _stream())
        grads = [None] * (2 * n)
        lane = _WgradLane(fdims, dy, taps)
        for i in range(n, 0, -1):
            w = ws[i - 1]
            gw, rw = _grad_out(w.data_ptr(), w.shape, dy.device)
            gb, rb = _grad_out(ctx.bptrs[i - 1], (C,), dy.device)
            if i > 1:
                lane.run(lambda: _wgrad(xs[i - 2], dp, gw, gb, B, D, H, W, C, C, kz, _sfx(C, C)), dp, gw, gb)
                wpd = _pack(w, taps, C, C, 1, fdims)
                mb = ctx.bits[i - 2]      # sign bits of conv i-1's output (xs[i-2]), if its forward emitted them
                dp = _conv_raw(dp, wpd, None, None, None if mb is not None else xs[i - 2], fdims, C, C, kz, DF_CONV_MASK, leak,
                               mask_bits=mb).view(dy.shape)
            else:
                nbytes = query("df_upconv_wgrad_workspace_bytes", cdims[0], cdims[1], cdims[2], cdims[3], C, C, kz)
                wsb = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dy.device)

                def up_wgrad():
                    if _sfx(C, C):
                        call("df_upconv_wgrad" + _sfx(C, C), _ptr(xc), _ptr(dp), _ptr(gw), _ptr(gb), cdims[0], cdims[1], cdims[2],
                             cdims[3], C, C, kz, _ptr(wsb), nbytes, _stream())
                    else:
                        if DISPATCH_COUNTS is not None:
                            f = query("df_upconv_wgrad_form", cdims[0], cdims[1], cdims[2], cdims[3], C, C, kz, int(WGRAD_ALGO))
                            _count("upconv-wgrad", "winograd-xyz-27pt" if f == 3 else "parity-class", fdims, C, C)
                        call("df_upconv_wgrad_algo", _ptr(xc), _ptr(dp), _ptr(gw), _ptr(gb), cdims[0], cdims[1], cdims[2],
                             cdims[3], C, C, kz, _ptr(wsb), nbytes, int(WGRAD_ALGO), _stream())
                lane.run(up_wgrad, dp, gw, gb, wsb)
                if ctx.needs_input_grad[0]:
                    if DISPATCH_COUNTS is not None:
                        _count("upconv-dgrad", "winograd-27pt-pooled" if (is3d and _use_wino(C, C, fdims, kz) == 3) else
                               "winograd2d-9pt-pooled" if (not is3d and _use_wino(C, C, fdims, kz) == 2) else "parity-class" + _sfx(C, C), fdims, C, C)
                    # dxc holds the skip path's sum-pool of dy (df_lrelu_bwd_pool2x above); += the conv path per parity class
                    if is3d and _use_wino(C, C, fdims, kz) == 3:
                        # pooled-output Winograd form (conv_wino.hip, POOL variant): 27 of the 64 products, coarse stores
                        wpd = _pack(w, taps, C, C, 1, fdims, family="f222")
                        call("df_wino_upconv_dgrad", _ptr(dp), _ptr(wpd), _ptr(dxc), cdims[0], cdims[1], cdims[2], cdims[3], C, C,
                             _stream())
                    elif not is3d and _use_wino(C, C, fdims, kz) == 2:
                        wpd = _pack(w, taps, C, C, 1, fdims, family="f22")
                        call("df_wino2d_upconv_dgrad", _ptr(dp), _ptr(wpd), _ptr(dxc), cdims[0], cdims[2], cdims[3], C, C, _stream())
                    else:
                        sfx = _sfx(C, C)
                        wpd = torch.empty(query("df_upconv_packed_elems" + sfx, C, C, kz, 1), dtype=torch.float32,
                                          device=dy.device)
                        call("df_upconv_pack_weights" + sfx, _ptr(w), _ptr(wpd), C, C, kz, 1, _stream())
                        call("df_upconv_dgrad" + sfx, _ptr(dp), _ptr(wpd), _ptr(dxc), cdims[0], cdims[1], cdims[2], cdims[3], C, C,
                             kz, _stream())
            grads[2 * (i - 1)] = rw; grads[2 * (i - 1) + 1] = rb
        lane.join()
        return (dxc if ctx.needs_input_grad[0] else None, None) + tuple(grads)


class _ConvSame3S2(torch.autograd.Function):
    """k=3, stride-2, TF-'SAME' conv on even extents (pad 0 before / 1 after): the encoder's
    down-sampling layers (model.py:141-143, 177-179).  Forward is a dedicated MFMA kernel; the backward re-uses the
    stride-1 dgrad / wgrad kernels on the zero-inserted gradient (out[2o+1] = g[o]), which reproduces the stride-2
    adjoints exactly (at 4x / 8x the minimal FLOPs -- four layers of the auto-encoder only)."""

    @staticmethod
    def forward(ctx, x, w, b, leak):
        x = _prep(x, "x"); w = _prep(w, "weights"); b = _prep(b, "biases")
        nd = x.dim() - 2
        kz = 3 if nd == 3 else 1
        taps = 27 if nd == 3 else 9
        cin, cout = w.shape[-2], w.shape[-1]
        if tuple(w.shape[:-2]) != (3,) * nd or x.shape[-1] != cin:
            raise ValueError("conv: weights %s do not match input %s (k=3 only)" % (tuple(w.shape), tuple(x.shape)))
        if any(int(d) % 2 for d in x.shape[1:-1]):
            raise NotImplementedError("stride-2 conv: even spatial extents only (the encoder asserts them, model.py:125,161)")
        idims = (x.shape[0], x.shape[1] if nd == 3 else 1, x.shape[-3], x.shape[-2])
        odims = (idims[0], idims[1] // 2 if nd == 3 else 1, idims[2] // 2, idims[3] // 2)
        wp = _pack(w, taps, cin, cout, 0, fp32=True)       # df_conv_s2_fwd has no bf16x3 variant: always the fp32 operand
        flags = DF_CONV_BIAS | (DF_CONV_LRELU if leak is not None else 0)
        y = torch.empty(odims + (cout,), dtype=torch.float32, device=x.device)
        call("df_conv_s2_fwd", _ptr(x), _ptr(wp), _ptr(b), _ptr(y), odims[0], odims[1], odims[2], odims[3], cin, cout, kz,
             flags, float(leak if leak is not None else 0.0), _stream())
        y = y.view((x.shape[0],) + tuple(int(d) // 2 for d in x.shape[1:-1]) + (cout,))
        if ACTIVATION_FETCH is not None and leak is not None:
            ACTIVATION_FETCH.append(y)
        ctx.save_for_backward(x, w, y if leak is not None else None)
        ctx.leak = leak
        ctx.geom = (idims, odims, cin, cout, kz, taps)
        ctx.bptr = b.data_ptr()
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        idims, odims, cin, cout, kz, taps = ctx.geom
        B, D, H, W = idims
        gy = _prep(gy, "grad")
        if ctx.leak is not None:
            dp = torch.empty_like(gy)
            call("df_lrelu_bwd", _ptr(gy), _ptr(y), _ptr(dp), float(ctx.leak), gy.numel(), _stream())
        else:
            dp = gy
        gw, rw = _grad_out(w.data_ptr(), w.shape, x.device)
        gb, rb = _grad_out(ctx.bptr, (cout,), x.device)
        up = None
        nbytes = query("df_conv_s2_wgrad_workspace_bytes", odims[0], odims[1], odims[2], odims[3], cin, cout, kz)
        if nbytes > 0 and WGRAD_ALGO == 0:
            # native form on the output grid: x[2o + t] * g[o] (conv_wgrad.hip::wgrad_s2_kernel) -- no zero-inserted gradient
            _count("wgrad-s2", "native-direct-mfma", odims, cin, cout)
            ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
            call("df_conv_s2_wgrad", _ptr(x), _ptr(dp), _ptr(gw), _ptr(gb), odims[0], odims[1], odims[2], odims[3], cin, cout, kz, _ptr(ws),
                 nbytes, _stream())
        else:
            # shapes the native kernel is not instantiated for: the stride-1 kernels on the zero-inserted gradient (out[2o+1] = g[o])
            up = torch.empty((B, D, H, W, cout), dtype=torch.float32, device=x.device)
            call("df_dilate2_odd", _ptr(dp), _ptr(up), odims[0], odims[1], odims[2], odims[3], cout, int(kz == 3), _stream())
            _wgrad(x, up, gw, gb, B, D, H, W, cin, cout, kz)
        gx = None
        if ctx.needs_input_grad[0]:
            if cin > 4 and cout > 4:
                # adjoint as 8 (4) parity-class 2x2(x2)-tap convs from the coarse gradient straight to the fine grid: the
                # up-sampling-aware forward kernel with the stride-2 dgrad operand (pack mode 2) -- 3.4x fewer FLOPs than the
                # stride-1 dgrad on the zero-inserted gradient, which stays as the thin-channel fallback below
                wpd = torch.empty(query("df_upconv_packed_elems", cin, cout, kz, 2), dtype=torch.float32, device=x.device)
                call("df_upconv_pack_weights", _ptr(w), _ptr(wpd), cin, cout, kz, 2, _stream())
                gx = torch.empty((B, D, H, W, cin), dtype=torch.float32, device=x.device)
                # [r5] df_conv_s2_dgrad: the same parity classes, each on a kernel specialised on its LIVE taps (27 of the 64 the generic
                # 2x2x2-tap parity-class kernel df_upconv_fwd multiplies -- half of the mode-2 operand is structural zeros)
                if DISPATCH_COUNTS is not None:      # (the library falls back to the generic class kernel on other channel counts / an unaligned gradient)
                    _count("dgrad-s2", "parity-class-live-taps" if query("df_conv_s2_dgrad_form", _ptr(dp), cin, cout) == 1 else
                           "parity-class-generic", odims, cin, cout)
                call("df_conv_s2_dgrad", _ptr(dp), _ptr(wpd), _ptr(gx), odims[0], odims[1], odims[2], odims[3], cin, cout, kz, _stream())
                gx = gx.view(x.shape)
            else:
                if up is None:
                    up = torch.empty((B, D, H, W, cout), dtype=torch.float32, device=x.device)
                    call("df_dilate2_odd", _ptr(dp), _ptr(up), odims[0], odims[1], odims[2], odims[3], cout, int(kz == 3), _stream())
                wpd = _pack(w, taps, cin, cout, 1, idims)
                gx = _conv_raw(up, wpd, None, None, None, idims, cout, cin, kz, 0, 0.0).view(x.shape)
        return gx, rw, rb, None


class _Concat2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = _prep(a, "a"); b = _prep(b, "b")
        if a.shape[:-1] != b.shape[:-1]:
            raise ValueError("concat: leading shapes differ %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        ca, cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // ca
        y = _empty(tuple(a.shape[:-1]) + (ca + cb,), a)
        call("df_concat2_fwd", _ptr(a), _ptr(b), _ptr(y), rows, ca, cb, _stream())
        ctx.geom = (rows, ca, cb, tuple(a.shape), tuple(b.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        rows, ca, cb, sa, sb = ctx.geom
        gy = _prep(gy, "grad")
        ga, gb = _empty(sa, gy), _empty(sb, gy)
        call("df_concat2_bwd", _ptr(gy), _ptr(ga), _ptr(gb), rows, ca, cb, _stream())
        return ga, gb


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _prep(x, "x")
        y = torch.empty_like(x)
        call("df_sigmoid_fwd", _ptr(x), _ptr(y), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = _prep(gy, "grad")
        gx = torch.empty_like(gy)
        call("df_sigmoid_bwd", _ptr(gy), _ptr(y), _ptr(gx), gy.numel(), _stream())
        return gx


class _KlBernoulli(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, n, rho):
        z = _prep(z, "z")
        if z.dim() != 2 or not 0 <= n <= z.shape[1]:
            raise ValueError("kl_bernoulli: z must be [B, ncol] with n <= ncol")
        out = torch.empty((), dtype=torch.float32, device=z.device)
        call("df_kl_bernoulli_fwd", _ptr(z), z.shape[0], z.shape[1], int(n), float(rho), _ptr(out), _stream())
        ctx.save_for_backward(z)
        ctx.geom = (int(n), float(rho))
        return out

    @staticmethod
    def backward(ctx, gout):
        (z,) = ctx.saved_tensors
        n, rho = ctx.geom
        gout = _prep(gout, "grad")
        gz = torch.empty_like(z)
        call("df_kl_bernoulli_bwd", _ptr(z), _ptr(gout), 1.0, _ptr(gz), z.shape[0], z.shape[1], n, rho, _stream())
        return gz, None, None


class _MseMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = _prep(a, "a"); b = _prep(b, "b")
        if a.shape != b.shape:
            raise ValueError("mse_mean: shapes differ %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        n = a.numel()
        out = _empty((), a)
        nbytes = query("df_l1_mean_workspace_bytes", n)
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=a.device)
        call("df_mse_mean_fwd", _ptr(a), _ptr(b), n, _ptr(out), _ptr(ws), nbytes, _stream())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        gout = _prep(gout, "grad")
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = torch.empty_like(a)
            call("df_mse_mean_bwd", _ptr(a), _ptr(b), _ptr(gout), 1.0, _ptr(ga), a.numel(), _stream())
        if ctx.needs_input_grad[1]:
            gb = torch.empty_like(b)
            call("df_mse_mean_bwd", _ptr(a), _ptr(b), _ptr(gout), -1.0, _ptr(gb), a.numel(), _stream())
        return ga, gb


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = _prep(x, "x"); w = _prep(w, "weights"); b = _prep(b, "biases")
        B, K = x.shape
        N = w.shape[1]
        if w.shape[0] != K:
            raise ValueError("linear: weights %s do not match input %s" % (tuple(w.shape), tuple(x.shape)))
        y = _empty((B, N), x)
        nbytes = query("df_linear_workspace_bytes", B, K, N)
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
        call("df_linear_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(y), B, K, N, _ptr(ws), nbytes, _stream())
        ctx.save_for_backward(x, w)
        ctx.bptr = b.data_ptr()
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _prep(gy, "grad")
        B, K = x.shape
        N = w.shape[1]
        gw, rw = _grad_out(w.data_ptr(), w.shape, x.device)
        gb, rb = _grad_out(ctx.bptr, (N,), x.device)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        call("df_linear_bwd", _ptr(x), _ptr(w), _ptr(gy), _ptr(gx), _ptr(gw), _ptr(gb), B, K, N, _stream())
        return gx, rw, rb


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _prep(x, "x")
        is3d = x.dim() == 5
        B = x.shape[0]
        D = x.shape[1] if is3d else 1
        H, W, C = x.shape[-3], x.shape[-2], x.shape[-1]
        out_shape = (B, 2 * D, 2 * H, 2 * W, C) if is3d else (B, 2 * H, 2 * W, C)
        y = _empty(out_shape, x)
        call("df_upsample2x_fwd", _ptr(x), _ptr(y), B, D, H, W, C, int(is3d), _stream())
        ctx.geom = (B, D, H, W, C, is3d, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        B, D, H, W, C, is3d, shp = ctx.geom
        gy = _prep(gy, "grad")
        gx = _empty(shp, gy)
        call("df_upsample2x_bwd", _ptr(gy), _ptr(gx), B, D, H, W, C, int(is3d), _stream())
        return gx


class _ConvGeneral(torch.autograd.Function):
    """slim.conv2d / conv3d with ANY cubic kernel and stride, TF 'SAME' on any extents (ops.py:12-16: the wrappers' own defaults are
    k=4, s=2) -- the general-shape vector-ALU kernels of conv_general.hip.  The call sites of the reference's trainers (k=3, s=1|2, even
    extents) never come here: they run on the matrix-core kernels (_ConvSame3 / _ConvSame3S2)."""

    @staticmethod
    def forward(ctx, x, w, b, leak, k, s):
        x = _prep(x, "x"); w = _prep(w, "weights"); b = _prep(b, "biases")
        nd = x.dim() - 2
        kz = k if nd == 3 else 1
        cin, cout = int(w.shape[-2]), int(w.shape[-1])
        if tuple(w.shape[:-2]) != (k,) * nd or x.shape[-1] != cin:
            raise ValueError("conv: weights %s do not match input %s / k=%d" % (tuple(w.shape), tuple(x.shape), k))
        B = int(x.shape[0])
        D = int(x.shape[1]) if nd == 3 else 1
        H, W = int(x.shape[-3]), int(x.shape[-2])
        od = [-(-D // s) if nd == 3 else 1, -(-H // s), -(-W // s)]
        y = torch.empty((B,) + tuple(od[3 - nd:]) + (cout,), dtype=torch.float32, device=x.device)
        flags = DF_CONV_BIAS | (DF_CONV_LRELU if leak is not None else 0)
        _count("conv", "general-valu k%d s%d" % (k, s), (B, D, H, W), cin, cout)
        call("df_conv_general_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(y), B, D, H, W, cin, cout, kz, k, s, flags,
             float(leak if leak is not None else 0.0), _stream())
        if ACTIVATION_FETCH is not None and leak is not None:
            ACTIVATION_FETCH.append(y)
        ctx.save_for_backward(x, w, y if leak is not None else None)
        ctx.leak = leak
        ctx.geom = (B, D, H, W, cin, cout, kz, k, s)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        B, D, H, W, cin, cout, kz, k, s = ctx.geom
        gy = _prep(gy, "grad")
        if ctx.leak is not None:
            dp = torch.empty_like(gy)
            call("df_lrelu_bwd", _ptr(gy), _ptr(y), _ptr(dp), float(ctx.leak), gy.numel(), _stream())
        else:
            dp = gy
        gw = torch.empty_like(w)
        gb = torch.empty(cout, dtype=torch.float32, device=x.device)
        call("df_conv_general_wgrad", _ptr(x), _ptr(dp), _ptr(gw), _ptr(gb), B, D, H, W, cin, cout, kz, k, s, _stream())
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            call("df_conv_general_dgrad", _ptr(dp), _ptr(w), _ptr(gx), B, D, H, W, cin, cout, kz, k, s, _stream())
        return gx, gw, gb, None, None, None


class _ResizeNN(torch.autograd.Function):
    """tf.image.resize_nearest_neighbor(align_corners=False) to ANY size per spatial axis (ops.py:66-73): src = min(floor(dst in / out), in - 1)."""

    @staticmethod
    def forward(ctx, x, new_size):
        x = _prep(x, "x")
        is3d = x.dim() == 5
        B, C = int(x.shape[0]), int(x.shape[-1])
        D = int(x.shape[1]) if is3d else 1
        H, W = int(x.shape[-3]), int(x.shape[-2])
        ns = [int(v) for v in new_size]
        if len(ns) != (3 if is3d else 2) or min(ns) <= 0:
            raise ValueError("resize: new_size %r does not match a %d-D tensor" % (new_size, x.dim()))
        Do, Ho, Wo = (ns[0], ns[1], ns[2]) if is3d else (1, ns[0], ns[1])
        y = _empty(((B, Do, Ho, Wo, C) if is3d else (B, Ho, Wo, C)), x)
        call("df_resize_nn_fwd", _ptr(x), _ptr(y), B, D, H, W, C, Do, Ho, Wo, _stream())
        ctx.geom = (B, D, H, W, C, Do, Ho, Wo, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        B, D, H, W, C, Do, Ho, Wo, shp = ctx.geom
        gy = _prep(gy, "grad")
        gx = _empty(shp, gy)
        call("df_resize_nn_bwd", _ptr(gy), _ptr(gx), B, D, H, W, C, Do, Ho, Wo, _stream())
        return gx, None


class _Curl2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, psi):
        psi = _prep(psi, "x")
        B, Y, X, C = psi.shape
        if C != 1:
            raise ValueError("curl kernel expects the 1-channel stream function [B,Y,X,1], got %s" % (tuple(psi.shape),))
        u = _empty((B, Y, X, 2), psi)
        call("df_curl2d_fwd", _ptr(psi), _ptr(u), B, Y, X, _stream())
        ctx.geom = (B, Y, X)
        return u

    @staticmethod
    def backward(ctx, gu):
        B, Y, X = ctx.geom
        gu = _prep(gu, "grad")
        g = _empty((B, Y, X, 1), gu)
        call("df_curl2d_bwd", _ptr(gu), _ptr(g), B, Y, X, _stream())
        return g


class _Jacobian2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _prep(x, "x")
        B, Y, X, C = x.shape
        if C != 2:
            raise ValueError("jacobian expects a 2-channel field [B,Y,X,2], got %s" % (tuple(x.shape),))
        j = _empty((B, Y, X, 4), x); w = _empty((B, Y, X, 1), x)
        call("df_jacobian2d_fwd", _ptr(x), _ptr(j), _ptr(w), B, Y, X, _stream())
        ctx.geom = (B, Y, X)
        ctx.set_materialize_grads(False)
        return j, w

    @staticmethod
    def backward(ctx, gj, gw):
        B, Y, X = ctx.geom
        if gj is None and gw is None:
            return None
        gj = None if gj is None else _prep(gj, "grad")
        gw = None if gw is None else _prep(gw, "grad")
        ref = gj if gj is not None else gw
        gx = _empty((B, Y, X, 2), ref)
        call("df_jacobian2d_bwd", _ptr(gj), _ptr(gw), _ptr(gx), B, Y, X, _stream())
        return gx


class _Jacobian3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, want_j, want_c):
        x = _prep(x, "x")
        B, Z, Y, X, C = x.shape
        if C != 3:
            raise ValueError("jacobian3 expects a 3-channel field [B,Z,Y,X,3], got %s" % (tuple(x.shape),))
        j = _empty((B, Z, Y, X, 9), x) if want_j else None
        c = _empty((B, Z, Y, X, 3), x) if want_c else None
        call("df_jacobian3d_fwd", _ptr(x), _ptr(j), _ptr(c), B, Z, Y, X, _stream())
        ctx.geom = (B, Z, Y, X)
        ctx.set_materialize_grads(False)
        if want_j and want_c:
            return j, c
        return j if want_j else c

    @staticmethod
    def backward(ctx, *grads):
        B, Z, Y, X = ctx.geom
        if len(grads) == 2:
            gj, gc = grads
        else:
            gj, gc = (grads[0], None) if grads[0] is not None and grads[0].shape[-1] == 9 else (None, grads[0])
        if gj is None and gc is None:
            return None, None, None
        gj = None if gj is None else _prep(gj, "grad")
        gc = None if gc is None else _prep(gc, "grad")
        ref = gj if gj is not None else gc
        gx = _empty((B, Z, Y, X, 3), ref)
        call("df_jacobian3d_bwd", _ptr(gj), _ptr(gc), _ptr(gx), B, Z, Y, X, _stream())
        return gx, None, None


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a = _prep(a, "a"); b = _prep(b, "b")
        if a.shape != b.shape:
            raise ValueError("l1_mean: shapes differ %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        n = a.numel()
        out = _empty((), a)
        nbytes = query("df_l1_mean_workspace_bytes", n)
        ws = torch.empty(nbytes // 8, dtype=torch.float64, device=a.device)
        call("df_l1_mean_fwd", _ptr(a), _ptr(b), n, _ptr(out), _ptr(ws), nbytes, _stream())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        gout = _prep(gout, "grad")
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = torch.empty_like(a)
            call("df_l1_mean_bwd", _ptr(a), _ptr(b), _ptr(gout), 1.0, _ptr(ga), a.numel(), _stream())
        if ctx.needs_input_grad[1]:
            gb = torch.empty_like(b)
            call("df_l1_mean_bwd", _ptr(a), _ptr(b), _ptr(gout), -1.0, _ptr(gb), a.numel(), _stream())
        return ga, gb


class _VelocityLoss(torch.autograd.Function):
    """Fused tail: (psi, x) -> (l1, jl1, u) with u = curl(psi) | jacobian3(psi)[1], l1 = mean|u - x|, jl1 = mean|J(u) - J(x)|
    (velocity_loss.hip).  x is data: no gradient."""

    @staticmethod
    def forward(ctx, psi, x):
        psi = _prep(psi, "psi"); x = _prep(x, "x")
        is3d = psi.dim() == 5
        if is3d:
            B, Z, Y, X, C = psi.shape
            if C != 3 or tuple(x.shape) != (B, Z, Y, X, 3):
                raise ValueError("velocity_loss: psi [B,Z,Y,X,3] and x [B,Z,Y,X,3] expected, got %s / %s" % (tuple(psi.shape), tuple(x.shape)))
            geom = (B, Z, Y, X)
            u = _empty((B, Z, Y, X, 3), psi)
        else:
            B, Y, X, C = psi.shape
            if C != 1 or tuple(x.shape) != (B, Y, X, 2):
                raise ValueError("velocity_loss: psi [B,Y,X,1] and x [B,Y,X,2] expected, got %s / %s" % (tuple(psi.shape), tuple(x.shape)))
            geom = (B, Y, X)
            u = _empty((B, Y, X, 2), psi)
        fn = "df_velocity_loss3d" if is3d else "df_velocity_loss2d"
        nbytes = query(fn + "_workspace_bytes", *geom)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=psi.device)
        l1, jl1 = _empty((), psi), _empty((), psi)
        call(fn + "_fwd", _ptr(psi), _ptr(x), _ptr(u), _ptr(l1), _ptr(jl1), *geom, _ptr(ws), nbytes, _stream())
        ctx.save_for_backward(u, x)
        ctx.geom = (fn, geom, nbytes, tuple(psi.shape))
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(u)          # u is handed out for logging / metrics; differentiate through curl()/curl3() instead
        return l1, jl1, u

    @staticmethod
    def backward(ctx, g1, g9, _gu):
        u, x = ctx.saved_tensors
        fn, geom, nbytes, pshape = ctx.geom
        if g1 is None and g9 is None:
            return None, None
        zero = None
        if g1 is None or g9 is None:
            zero = torch.zeros((), dtype=torch.float32, device=u.device)
        g1 = _prep(g1, "grad") if g1 is not None else zero
        g9 = _prep(g9, "grad") if g9 is not None else zero
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=u.device)
        gpsi = _empty(pshape, u)
        call(fn + "_bwd", _ptr(u), _ptr(x), _ptr(g1), _ptr(g9), _ptr(gpsi), *geom, _ptr(ws), nbytes, _stream())
        return gpsi, None


# --------------------------------------------------------------------------------------------------
# the reference's call surface
# --------------------------------------------------------------------------------------------------
def lrelu(x, leak=0.2):
    """ops.py:9-10  ``tf.maximum(x, leak*x)``."""
    return _Lrelu.apply(x, leak)


def add(a, b):
    """The residual ``x += x0`` of model.py:35,40,77,82 as a HIP kernel."""
    return _Add.apply(a, b)


def concat(values, axis=-1):
    """``tf.concat([x, x0], axis=-1)`` (model.py:138,174): channel concat of two channels-last tensors."""
    if len(values) != 2 or axis not in (-1, values[0].dim() - 1):
        raise NotImplementedError("concat: two tensors along the channel axis (the encoder's skip connection)")
    return _Concat2.apply(values[0], values[1])


def sigmoid(x):
    """``tf.sigmoid`` (model.py:196,210)."""
    return _Sigmoid.apply(x)


def kl_bernoulli(z, n, rho):
    """``tf.reduce_sum(ds.kl_divergence(ds.Bernoulli(probs=rho), ds.Bernoulli(probs=tf.reduce_mean(z[:, :n], axis=0))))``
    (trainer3.py:272-277)."""
    return _KlBernoulli.apply(z, n, rho)


def mse_mean(a, b):
    """``tf.reduce_mean(tf.squared_difference(a, b))`` (trainer3.py:270)."""
    return _MseMean.apply(a, b)


FUSED_BLOCKS = True     # GeneratorBE(3) uses one fused autograd node per block (same kernels, fused backward epilogues)

# Fetch of intermediate tensors (the counterpart of adding a tensor to ``sess.run``'s fetch list): when set to a list, every
# fused generator block and every layer-by-layer conv with an lrelu appends its post-lrelu conv outputs (execution order) to it.  Used by the full-size parity tests to hand the
# oracle the lrelu sign pattern the GPU actually took.  None (default) = no fetch, no cost.
ACTIVATION_FETCH = None


# The switches above are process-wide module attributes (like the reference's single global `config`).  `options` is the one supported way
# to change them temporarily: it sets them, restores the previous values on exit -- also when the body raises, so a failing test cannot
# leak its mode into the next one -- and holds a re-entrant lock for the duration of the block, so two threads cannot interleave
# different option sets (a second thread entering `options` waits until the first leaves).
_OPTION_ATTRS = {"conv_precision": "CONV_PRECISION", "conv_algo": "CONV_ALGO", "wino3d_family": "WINO3D_FAMILY", "wino2d_family": "WINO2D_FAMILY", "wgrad_algo": "WGRAD_ALGO",
                 "thin_valu_only": "THIN_VALU_ONLY", "sign_bit_masks": "SIGN_BIT_MASKS", "fused_blocks": "FUSED_BLOCKS",
                 "dispatch_counts": "DISPATCH_COUNTS", "concurrent_wgrad_work": "CONCURRENT_WGRAD_WORK", "activation_fetch": "ACTIVATION_FETCH", "sign_bits_fetch": "SIGN_BITS_FETCH"}
_OPTION_CHOICES = {"conv_precision": ("fp32", "bf16x3"), "conv_algo": ("auto", "direct", "winograd"), "wino3d_family": ("f224", "f222"), "wino2d_family": ("auto", "f24", "f22"), "wgrad_algo": (0, 1, 2, 3, 4)}
_OPTION_LOCK = _threading.RLock()


@_contextlib.contextmanager
def options(**kw):
    """``with ops.options(conv_precision="bf16x3", conv_algo="direct", wgrad_algo=1, activation_fetch=[]): ...``
    Keys: conv_precision, conv_algo, wino3d_family ("f224" | "f222"), wino2d_family ("auto" | "f24" | "f22"), wgrad_algo, thin_valu_only, sign_bit_masks, fused_blocks, dispatch_counts (a dict to count into),
    concurrent_wgrad_work (weight gradients of levels up to that many B x voxels x taps run on a second stream; 0 = serial),
    activation_fetch / sign_bits_fetch (a list to append to).  Unknown keys and out-of-range values raise before anything changes."""
    for k, v in kw.items():
        if k not in _OPTION_ATTRS:
            raise TypeError("ops.options: unknown option %r (known: %s)" % (k, ", ".join(sorted(_OPTION_ATTRS))))
        if k in _OPTION_CHOICES and v not in _OPTION_CHOICES[k]:
            raise ValueError("ops.options: %s=%r not in %r" % (k, v, _OPTION_CHOICES[k]))
    g = globals()
    with _OPTION_LOCK:
        old = {k: g[_OPTION_ATTRS[k]] for k in kw}
        try:
            for k, v in kw.items():
                g[_OPTION_ATTRS[k]] = v
            yield
        finally:
            for k, v in old.items():
                g[_OPTION_ATTRS[k]] = v


def gen_block(x, filters, names, nd, leak=0.2):
    """``num_conv`` x conv(k=3,s=1,act=lrelu) + ``x += x0`` (model.py:24-40 / 66-82) with slim variables ``names``."""
    wb = []
    cin = int(x.shape[-1])
    for name in names:
        wb.append(get_variable(name + "/weights", (3,) * nd + (cin, int(filters)), "xavier", x.device))
        wb.append(get_variable(name + "/biases", (int(filters),), "zeros", x.device))
        cin = int(filters)
    return _GenBlock.apply(x, leak, *wb)


def conv_chain(x, filters, names, nd, leak=0.2):
    """``len(names)`` x conv(k=3,s=1,act=lrelu) (the encoder's per-level stack, model.py:131-136 / 167-172) as one autograd node."""
    wb = []
    cin = int(x.shape[-1])
    for name in names:
        wb.append(get_variable(name + "/weights", (3,) * nd + (cin, int(filters)), "xavier", x.device))
        wb.append(get_variable(name + "/biases", (int(filters),), "zeros", x.device))
        cin = int(filters)
    return _ConvChain.apply(x, leak, *wb)


def up_gen_block(xc, filters, names, nd, leak=0.2):
    """``upscale(xc, 2)`` + one generator block on the up-sampled tensor (model.py:36-40 / 78-82), fused."""
    wb = []
    for name in names:
        wb.append(get_variable(name + "/weights", (3,) * nd + (int(filters), int(filters)), "xavier", xc.device))
        wb.append(get_variable(name + "/biases", (int(filters),), "zeros", xc.device))
    return _UpGenBlock.apply(xc, leak, *wb)


def nchw_to_nhwc(x):
    """ops.py:108-109."""
    return x.permute(0, 2, 3, 1).contiguous()


def nhwc_to_nchw(x):
    """ops.py:111-112."""
    return x.permute(0, 3, 1, 2).contiguous()


def int_shape(tensor):
    """ops.py:96-98."""
    return [int(s) for s in tensor.shape]


def get_conv_shape(tensor, data_format="NHWC"):
    """ops.py:100-106: always returns [N,H,W,C] (for 5-D NDHWC input: the shape itself)."""
    shape = int_shape(tensor)
    if data_format == "NCHW":
        return [shape[0], shape[2], shape[3], shape[1]]
    elif data_format == "NHWC":
        return shape


def reshape(x, h, w, c, data_format="NHWC"):
    """ops.py:198-203."""
    if data_format == "NCHW":
        return x.reshape(-1, c, h, w)
    return x.reshape(-1, h, w, c)


def _act_leak(act):
    """Map the reference's ``act`` argument onto the fused epilogue: None -> linear, lrelu -> leak 0.2."""
    if act is None:
        return None, None
    if act is lrelu:
        return 0.2, None
    return None, act          # any other callable is applied after the conv, un-fused


def _conv(x, o_dim, nd, data_format, name, k, s, act):
    k, s = int(k), int(s)
    if not (1 <= k <= 7 and 1 <= s <= 4):
        raise ValueError("conv: kernel size 1..7 and stride 1..4 (got k=%d s=%d)" % (k, s))
    if nd == 2 and data_format == "NCHW":
        x = nchw_to_nhwc(x)
    cin = int(x.shape[-1])
    lname = _layer_name(name, "Conv")
    w = get_variable(lname + "/weights", (k,) * nd + (cin, int(o_dim)), "xavier", x.device)
    b = get_variable(lname + "/biases", (int(o_dim),), "zeros", x.device)
    leak, post = _act_leak(act)
    # the reference's own call sites (k=3; s=1, or s=2 on even extents: model.py:26,42,68,84,127-143,163-179) run on the matrix cores;
    # every other (k, s, extent) of the wrapper (its defaults are k=4, s=2, ops.py:12-16) on the general-shape kernels
    fast = k == 3 and (s == 1 or (s == 2 and not any(int(d) % 2 for d in x.shape[1:-1])))
    if fast:
        y = (_ConvSame3 if s == 1 else _ConvSame3S2).apply(x, w, b, leak)
    else:
        y = _ConvGeneral.apply(x, w, b, leak, k, s)
    if post is not None:
        y = post(y)
    if nd == 2 and data_format == "NCHW":
        y = nhwc_to_nchw(y)
    return y


def conv2d(x, o_dim, data_format="NHWC", name=None, k=4, s=2, act=None):
    """ops.py:12-13 (slim.conv2d, SAME; the wrapper's defaults k=4, s=2 included)."""
    return _conv(x, o_dim, 2, data_format, name, k, s, act)


def conv3d(x, o_dim, data_format="NDHWC", name=None, k=4, s=2, act=None):
    """ops.py:15-16 (slim.conv3d, SAME)."""
    if data_format != "NDHWC":
        raise NotImplementedError("conv3d: NDHWC only (the reference never uses NCDHW)")
    return _conv(x, o_dim, 3, data_format, name, k, s, act)


def linear(x, o_dim, name=None, act=None):
    """ops.py:23-24 (slim.fully_connected)."""
    lname = _layer_name(name, "fully_connected")
    w = get_variable(lname + "/weights", (int(x.shape[-1]), int(o_dim)), "xavier", x.device)
    b = get_variable(lname + "/biases", (int(o_dim),), "zeros", x.device)
    y = _Linear.apply(x, w, b)
    return act(y) if act is not None else y


def resize_nearest_neighbor(x, new_size, data_format="NHWC"):
    """ops.py:66-73 (tf.image.resize_nearest_neighbor, align_corners=False): any target size; the exact 2x of the reference's call sites
    takes the vectorised kernel."""
    if data_format == "NCHW":
        x = nchw_to_nhwc(x)
    if tuple(int(v) for v in new_size) == (2 * x.shape[1], 2 * x.shape[2]) and x.shape[-1] % 4 == 0:
        y = _Upsample2x.apply(x)
    else:
        y = _ResizeNN.apply(x, tuple(int(v) for v in new_size))
    return nhwc_to_nchw(y) if data_format == "NCHW" else y


def upscale(x, scale, data_format="NHWC"):
    """ops.py:75-77."""
    _, h, w, _ = get_conv_shape(x, data_format)
    return resize_nearest_neighbor(x, (h * scale, w * scale), data_format)


def upscale3(x, scale):
    """ops.py:79-91: two 2-D nearest resizes == one 3-D nearest resize by `scale` (src = dst // scale); scale 2 (model.py:78) takes the
    vectorised kernel."""
    scale = int(scale)
    if scale == 2 and x.shape[-1] % 4 == 0:
        return _Upsample2x.apply(x)
    if scale < 1:
        raise ValueError("upscale3: scale must be a positive integer")
    return _ResizeNN.apply(x, tuple(int(d) * scale for d in x.shape[1:4]))


def jacobian(x, data_format="NHCW"):
    """ops.py:205-225 -> (j [..,4], w [..,1]).  (The reference default 'NHCW' is a typo that behaves as NHWC.)"""
    if data_format == "NCHW":
        x = nchw_to_nhwc(x)
    j, w = _Jacobian2d.apply(x)
    if data_format == "NCHW":
        j, w = nhwc_to_nchw(j), nhwc_to_nchw(w)
    return j, w


def jacobian3(x):
    """ops.py:227-262 -> (j [..,9], c [..,3]);  call-site idiom ``_, G_ = jacobian3(G_s)`` (trainer3.py:18)."""
    return _Jacobian3d.apply(x, True, True)


def curl3(x):
    """North-star alias for ``jacobian3(x)[1]`` that skips the unused 9-channel output (24 B/voxel, not 60)."""
    return _Jacobian3d.apply(x, False, True)


def curl(x, data_format="NHWC"):
    """ops.py:264-274."""
    if data_format == "NCHW":
        x = nchw_to_nhwc(x)
    if x.shape[-1] != 1:
        x = x[..., :1]          # the reference reads channel 0 only (x[:,1:,:,0]); e.g. the 2-channel AE output, trainer.py:361
    c = _Curl2d.apply(x)
    return nhwc_to_nchw(c) if data_format == "NCHW" else c


def divergence(x, data_format="NHWC"):
    """ops.py:276-284 (no gradient: used as a diagnostic only)."""
    if data_format == "NCHW":
        x = nchw_to_nhwc(x)
    x = _prep(x.detach(), "x")
    B, Y, X, _ = x.shape
    d = _empty((B, Y - 1, X - 1, 1), x)
    call("df_divergence2d", _ptr(x), _ptr(d), B, Y, X, _stream())
    return nhwc_to_nchw(d) if data_format == "NCHW" else d


def divergence3(x):
    """ops.py:286-290."""
    x = _prep(x.detach(), "x")
    B, Z, Y, X, _ = x.shape
    d = _empty((B, Z - 1, Y - 1, X - 1, 1), x)
    call("df_divergence3d", _ptr(x), _ptr(d), B, Z, Y, X, _stream())
    return d


def pgrad(x, data_format):
    """ops.py:292-303: pressure gradient ``(D_x p, D_y p)`` of channel 0 with the last difference replicated.  The two
    differences are exactly the curl kernel's outputs re-ordered -- ``curl(p) = (D_y p, -D_x p)`` (ops.py:267-271) and negation
    is exact -- so the stencil runs on ``df_curl2d_fwd`` / ``_bwd``; only the channel swap is a tensor view op."""
    if data_format == "NCHW":
        x = nchw_to_nhwc(x)
    c = _Curl2d.apply(x[..., :1].contiguous())
    g = torch.stack([-c[..., 1], c[..., 0]], dim=-1)
    return nhwc_to_nchw(g) if data_format == "NCHW" else g


def l1_mean(a, b):
    """``tf.reduce_mean(tf.abs(a - b))`` (trainer.py:170-171) as one fused reduction."""
    return _L1Mean.apply(a, b)


def velocity_loss(psi, x):
    """The tail of ``build_model`` as ONE fused op (SURVEY 8(b) ``velocity_loss2d/3d``; trainer.py:140-146,170-172 and the
    ground-truth Jacobian of trainer.py:29-32 / trainer3.py:18-24,49-51):  returns ``(l1, j_l1, u)`` with
    ``u = curl(psi) | jacobian3(psi)[1]``, ``l1 = reduce_mean(abs(u - x))``, ``j_l1 = reduce_mean(abs(jacobian(u)[0] - jacobian(x)[0]))``.
    Gradients flow to ``psi`` through ``l1`` and ``j_l1``; ``u`` is returned detached (for metrics / summaries)."""
    return _VelocityLoss.apply(psi, x)


# ---- NumPy-facing twins (ops.py:305-324, 344-374): ndarray in, ndarray out, computed on the GPU ----
def _np_in(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(_DEFAULT_DEVICE)


def vort_np(x):
    """ops.py:305-310."""
    return jacobian(_np_in(x), data_format="NHWC")[1].cpu().numpy()


def curl_np(x):
    """ops.py:312-317."""
    return curl(_np_in(x)).cpu().numpy()


def grad_np(x):
    """ops.py:319-324: (dp/dx, dp/dy) == (-curl_v, curl_u)."""
    c = curl(_np_in(x)).cpu().numpy()
    return np.stack([-c[..., 1], c[..., 0]], axis=-1)


def jacobian_np3(x):
    """ops.py:344-374."""
    j, c = jacobian3(_np_in(x))
    return j.cpu().numpy(), c.cpu().numpy()
