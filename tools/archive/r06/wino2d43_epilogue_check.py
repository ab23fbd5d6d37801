"""conv_wino2d43.hip epilogue A/B: sha-1 of every epilogue variant's output on full / ragged / sub-block images + time at cfg2's top level.
Run once per build (DF_HIP_LIBRARY=<other build>) and compare the lines."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream
from tools.gpu_probe import timeit
from deep_fluids_amd._lib import DF_CONV_BIAS, DF_CONV_LRELU, DF_CONV_RESIDUAL, DF_CONV_MASK
s = _stream()
tag = os.path.basename(os.environ.get("DF_HIP_LIBRARY", "release"))
def run(B, H, W, Ci, Co, fl, time_it=False):
    torch.manual_seed(B * 131 + H * 7 + W + Ci + fl)
    x = torch.rand((B, H, W, Ci), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, Ci, Co), device="cuda") * 2 - 1) * 0.05
    bias = torch.rand(Co, device="cuda") * 0.1
    aux = torch.rand((B, H, W, Co), device="cuda") * 2 - 1
    mode = 0 if (fl & DF_CONV_BIAS) else 1
    K, N = (Ci, Co) if mode == 0 else (Co, Ci)      # mode 1: the operand of the dgrad (input has Co channels)
    if mode == 1:
        x = torch.rand((B, H, W, Co), device="cuda") * 2 - 1
        aux = torch.rand((B, H, W, Ci), device="cuda") * 2 - 1
    w4 = torch.empty(query("df_wino2d43_packed_elems", Ci, Co, mode), device="cuda")
    call("df_wino2d43_pack_weights", _ptr(w), _ptr(w4), Ci, Co, mode, s)
    y = torch.full((B, H, W, N), float("nan"), device="cuda")
    f = lambda: call("df_wino2d43_conv", _ptr(x), _ptr(w4), _ptr(bias) if fl & DF_CONV_BIAS else None, _ptr(aux) if fl & DF_CONV_RESIDUAL else None, _ptr(aux) if fl & DF_CONV_MASK else None, _ptr(y),
                     B, H, W, K, N, fl, 0.2, s)
    f(); torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
    t = timeit(f, 6, 3) if time_it else 0.0
    print("%s  B%d %dx%d %d->%d flags %2d  sha %s%s" % (tag, B, H, W, K, N, fl, h, ("  %.3f ms" % (t * 1e3)) if time_it else ""), flush=True)
FW, MK, RS = DF_CONV_BIAS | DF_CONV_LRELU, DF_CONV_MASK, DF_CONV_RESIDUAL
for shp in ((2, 24, 40, 64, 32), (1, 16, 32, 32, 32), (3, 33, 47, 32, 64), (1, 10, 12, 32, 32), (2, 50, 70, 96, 96)):
    for fl in (FW, MK, RS, DF_CONV_BIAS, DF_CONV_BIAS | DF_CONV_RESIDUAL, DF_CONV_BIAS | DF_CONV_LRELU | DF_CONV_RESIDUAL):
        run(*shp, fl)
for fl in (FW, MK, RS):
    run(64, 128, 96, 128, 128, fl, True)
