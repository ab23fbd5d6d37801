"""ctypes binding of libdeepfluids_hip.so (the C-ABI declared in include/deepfluids_hip.h).

There is NO fallback: if the HIP library is missing this module raises at import of the first
symbol, and every compute entry point of the package goes through it.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdeepfluids_hip.so")
TUNING_LIB_PATH = os.path.join(_HERE, "csrc", "libdeepfluids_hip_tuning.so")      # tools/ only: `use_tuning_library()`
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "deepfluids_hip.h")

P, I64, I32, F32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float

# name -> (restype, argtypes)
SIGNATURES = {
    "df_version": (I32, []),
    "df_last_error": (ctypes.c_char_p, []),
    "df_curl2d_fwd": (I32, [P, P, I64, I64, I64, P]),
    "df_curl2d_bwd": (I32, [P, P, I64, I64, I64, P]),
    "df_jacobian2d_fwd": (I32, [P, P, P, I64, I64, I64, P]),
    "df_jacobian2d_bwd": (I32, [P, P, P, I64, I64, I64, P]),
    "df_jacobian3d_fwd": (I32, [P, P, P, I64, I64, I64, I64, P]),
    "df_jacobian3d_bwd": (I32, [P, P, P, I64, I64, I64, I64, P]),
    "df_divergence2d": (I32, [P, P, I64, I64, I64, P]),
    "df_divergence3d": (I32, [P, P, I64, I64, I64, I64, P]),
    "df_velocity_loss3d_workspace_bytes": (I64, [I64, I64, I64, I64]),
    "df_velocity_loss3d_fwd": (I32, [P, P, P, P, P, I64, I64, I64, I64, P, I64, P]),
    "df_velocity_loss3d_bwd": (I32, [P, P, P, P, P, I64, I64, I64, I64, P, I64, P]),
    "df_velocity_loss2d_workspace_bytes": (I64, [I64, I64, I64]),
    "df_velocity_loss2d_fwd": (I32, [P, P, P, P, P, I64, I64, I64, P, I64, P]),
    "df_velocity_loss2d_bwd": (I32, [P, P, P, P, P, I64, I64, I64, P, I64, P]),
    "df_l1_mean_workspace_bytes": (I64, [I64]),
    "df_l1_mean_fwd": (I32, [P, P, I64, P, P, I64, P]),
    "df_l1_mean_bwd": (I32, [P, P, P, F32, P, I64, P]),
    "df_lrelu_fwd": (I32, [P, P, F32, I64, P]),
    "df_lrelu_bwd": (I32, [P, P, P, F32, I64, P]),
    "df_add": (I32, [P, P, P, I64, P]),
    "df_upsample2x_fwd": (I32, [P, P, I64, I64, I64, I64, I64, I32, P]),
    "df_conv_general_out_dims": (I32, [I64, I64, I64, I32, I32, I32, P, P, P]),
    "df_conv_general_fwd": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, I32, I32, F32, P]),
    "df_conv_general_dgrad": (I32, [P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, I32, P]),
    "df_conv_general_wgrad": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, I32, P]),
    "df_resize_nn_fwd": (I32, [P, P, I64, I64, I64, I64, I64, I64, I64, I64, P]),
    "df_resize_nn_bwd": (I32, [P, P, I64, I64, I64, I64, I64, I64, I64, I64, P]),
    "df_upsample2x_bwd": (I32, [P, P, I64, I64, I64, I64, I64, I32, P]),
    "df_lrelu_bwd_pool2x": (I32, [P, P, P, P, F32, I64, I64, I64, I64, I64, I32, P]),
    "df_linear_workspace_bytes": (I64, [I64, I64, I64]),
    "df_linear_fwd": (I32, [P, P, P, P, I64, I64, I64, P, I64, P]),
    "df_linear_bwd": (I32, [P, P, P, P, P, P, I64, I64, I64, P]),
    "df_concat2_fwd": (I32, [P, P, P, I64, I64, I64, P]),
    "df_concat2_bwd": (I32, [P, P, P, I64, I64, I64, P]),
    "df_dilate2_odd": (I32, [P, P, I64, I64, I64, I64, I64, I32, P]),
    "df_kl_bernoulli_fwd": (I32, [P, I64, I64, I64, F32, P, P]),
    "df_kl_bernoulli_bwd": (I32, [P, P, F32, P, I64, I64, I64, F32, P]),
    "df_sigmoid_fwd": (I32, [P, P, I64, P]),
    "df_sigmoid_bwd": (I32, [P, P, P, I64, P]),
    "df_mse_mean_fwd": (I32, [P, P, I64, P, P, I64, P]),
    "df_mse_mean_bwd": (I32, [P, P, P, F32, P, I64, P]),
    "df_colsum_workspace_bytes": (I64, [I64, I64]),
    "df_colsum": (I32, [P, P, I64, I64, P, I64, P]),
    "df_adam_tf1_step": (I32, [P, P, P, P, I64, F32, F32, F32, F32, F32, P]),
    "df_conv_packed_elems": (I64, [I64, I64, I64, I32]),
    "df_conv_pack_weights": (I32, [P, P, I64, I64, I64, I32, P]),
    "df_conv_fwd": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32, P]),
    "df_conv_s2_fwd": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32, P]),
    "df_conv_s2_wgrad_workspace_bytes": (I64, [I64, I64, I64, I64, I64, I64, I32]),
    "df_conv_s2_wgrad": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P]),
    "df_conv_s2_dgrad": (I32, [P, P, P, I64, I64, I64, I64, I64, I64, I32, P]),
    "df_conv_s2_dgrad_form": (I32, [P, I64, I64]),
    "df_upconv_packed_elems": (I64, [I64, I64, I32, I32]),
    "df_upconv_pack_weights": (I32, [P, P, I64, I64, I32, I32, P]),
    "df_upconv_fwd": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32, P]),
    "df_upconv_dgrad": (I32, [P, P, P, I64, I64, I64, I64, I64, I64, I32, P]),
    "df_upconv_wgrad_workspace_bytes": (I64, [I64, I64, I64, I64, I64, I64, I32]),
    "df_upconv_wgrad": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P]),
    "df_add_up2x": (I32, [P, P, P, I64, I64, I64, I64, I64, I32, P]),
    "df_wino_packed_elems": (I64, [I64, I64, I32]),
    "df_wino_pack_weights": (I32, [P, P, I64, I64, I32, P]),
    "df_wino_conv_fwd": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, F32, P]),
    "df_wino_conv_fwd_addup": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, F32, P]),
    "df_wino_conv_fwd_addup_bits": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, F32, P]),
    "df_lrelu_bits_bwd_pool2x": (I32, [P, P, P, P, F32, I64, I64, I64, I64, I64, P]),
    "df_wino_upconv_fwd": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, F32, P]),
    "df_wino_upconv_dgrad": (I32, [P, P, P, I64, I64, I64, I64, I64, I64, P]),
    "df_wino_signbits_bytes": (I64, [I64, I64, I64, I64, I64]),
    "df_wino_conv_fwd_bits": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, F32, P]),
    "df_wino_upconv_fwd_bits": (I32, [P, P, P, P, P, I64, I64, I64, I64, I64, I64, F32, P]),
    "df_wino43_packed_elems": (I64, [I64, I64, I32]),
    "df_wino43_pack_weights": (I32, [P, P, I64, I64, I32, P]),
    "df_wino43_conv": (I32, [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, F32, P]),
    "df_wino2d43_packed_elems": (I64, [I64, I64, I32]),
    "df_wino2d43_pack_weights": (I32, [P, P, I64, I64, I32, P]),
    "df_wino2d43_conv": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I32, F32, P]),
    "df_wino2d43_signbits_bytes": (I64, [I64, I64, I64, I64]),
    "df_wino2d43_conv_bits": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I32, F32, P]),
    "df_wino2d43_conv_addup_bits": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, F32, P]),
    "df_lrelu_words2d_bwd_pool2x": (I32, [P, P, P, P, F32, I64, I64, I64, I64, P]),
    "df_wino2d_packed_elems": (I64, [I64, I64, I32]),
    "df_wino2d_pack_weights": (I32, [P, P, I64, I64, I32, P]),
    "df_wino2d_conv_fwd": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I32, F32, P]),
    "df_wino2d_upconv_fwd": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I32, F32, P]),
    "df_wino2d_upconv_dgrad": (I32, [P, P, P, I64, I64, I64, I64, I64, P]),
    "df_conv_packed_elems_bf16x3": (I64, [I64, I64, I64, I32]),
    "df_conv_pack_weights_bf16x3": (I32, [P, P, I64, I64, I64, I32, P]),
    "df_conv_fwd_bf16x3": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32, P]),
    "df_upconv_packed_elems_bf16x3": (I64, [I64, I64, I32, I32]),
    "df_upconv_pack_weights_bf16x3": (I32, [P, P, I64, I64, I32, I32, P]),
    "df_upconv_fwd_bf16x3": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32, P]),
    "df_upconv_dgrad_bf16x3": (I32, [P, P, P, I64, I64, I64, I64, I64, I64, I32, P]),
    "df_conv_wgrad_bf16x3": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P]),
    "df_upconv_wgrad_bf16x3": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P]),
    "df_adam_tf1_step_dev": (I32, [P, P, P, P, I64, P, F32, F32, F32, P]),
    "df_gd_step": (I32, [P, P, I64, F32, F32, P]),
    "df_gd_step_dev": (I32, [P, P, I64, P, P]),
    "df_store_scalars": (I32, [P, I64, F32, F32, F32, F32, P]),
    "df_conv_wgrad_workspace_bytes": (I64, [I64, I64, I64, I64, I64, I64, I32]),
    "df_conv_wgrad": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P]),
    "df_conv_wgrad_algo": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, I32, P]),
    "df_upconv_wgrad_algo": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, I32, P]),
    "df_conv_wgrad_form": (I32, [I64, I64, I64, I64, I64, I64, I32, I32]),
    "df_upconv_wgrad_form": (I32, [I64, I64, I64, I64, I64, I64, I32, I32]),
}

DF_CONV_LRELU, DF_CONV_RESIDUAL, DF_CONV_MASK, DF_CONV_BIAS, DF_CONV_ADDUP, DF_CONV_VALU_ONLY = 1, 2, 4, 8, 16, 32

_lib = None


class DeepFluidsHipError(RuntimeError):
    pass


def declared_symbols():
    """Every function the public header declares (used by the export test)."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(df_[a-z0-9_]+)\s*\(", src)))


def use_tuning_library():
    """tools/ probes only: bind the -DDF_TUNING build (instrumented kernels + the knobs of deepfluids_hip_debug.h) instead of
    the release library.  Must be called before the first kernel call of the process."""
    global LIB_PATH, _lib
    if _lib is not None:
        raise DeepFluidsHipError("use_tuning_library() must precede the first call into the library")
    if not os.path.exists(TUNING_LIB_PATH):
        raise DeepFluidsHipError("%s not found -- `make -C deep_fluids_amd/csrc tuning`" % TUNING_LIB_PATH)
    LIB_PATH = TUNING_LIB_PATH


def lib():
    global _lib, LIB_PATH
    if _lib is None:
        if os.environ.get("DF_HIP_LIBRARY"):          # sanitizer runs only (tools/run_asan.sh): another BUILD of the same library
            LIB_PATH = os.environ["DF_HIP_LIBRARY"]
        if not os.path.exists(LIB_PATH):
            raise DeepFluidsHipError(
                "libdeepfluids_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C deep_fluids_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class KernelTimer(object):
    """Optional HIP-event timing of selected C-ABI calls (bench.py's live roofline measurement).

    ``select(name, args)`` returns None or ``(key, work)``; matching calls are bracketed by two events recorded
    on the CURRENT torch stream -- the stream the kernels are launched on -- and summarised after a sync."""

    def __init__(self, select):
        self.select = select
        self.records = []

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for key, work, e0, e1 in self.records:
            d = out.setdefault(key, {"launches": 0, "seconds": 0.0, "work": 0.0})
            d["launches"] += 1
            d["seconds"] += e0.elapsed_time(e1) * 1e-3
            d["work"] += work
        return out


TIMER = None   # set to a KernelTimer to enable


def call(name, *args):
    """Call an int-returning df_* entry point; raise with df_last_error() on failure."""
    h = lib()
    hit = TIMER.select(name, args) if TIMER is not None else None
    if hit is not None:
        import torch
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(h, name)(*args)
        e1.record()
        TIMER.records.append((hit[0], hit[1], e0, e1))
    else:
        rc = getattr(h, name)(*args)
    if rc != 0:
        msg = h.df_last_error()
        raise DeepFluidsHipError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else ""))


_QUERY_CACHE = {}


def query(name, *args):
    """Call a value-returning helper (workspace sizes, packed-operand sizes, algorithm forms, version).  Every one of them is a pure function of
    its integer arguments, so the answers are memoised: a 2-D train step at the reference's default batch asks ~150 of them, and its host issue
    time (2.6 ms) is within a millisecond of the device time (3.5 ms)."""
    key = (name,) + args
    try:
        return _QUERY_CACHE[key]
    except KeyError:
        v = _QUERY_CACHE[key] = getattr(lib(), name)(*args)
        return v
    except TypeError:      # an unhashable argument: not memoised
        return getattr(lib(), name)(*args)
