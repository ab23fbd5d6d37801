"""conv_tiny2d_kernel (the lowest 2-D levels) vs the 32 x 32-block direct kernel of the previous build (DF_HIP_LIBRARY) and vs an fp64 reference: error + time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from deep_fluids_amd._lib import call, query, DF_CONV_BIAS, DF_CONV_LRELU, DF_CONV_MASK, DF_CONV_RESIDUAL
from deep_fluids_amd.ops import _ptr, _stream
from tools.gpu_probe import timeit
s = _stream(); tag = os.path.basename(os.environ.get("DF_HIP_LIBRARY", "release"))
for (B, H, W, Ci, Co) in ((8, 8, 6, 128, 128), (8, 16, 12, 128, 128), (8, 16, 12, 64, 64), (3, 7, 5, 32, 96), (64, 8, 6, 128, 128), (8, 16, 16, 128, 64)):
    torch.manual_seed(B + H + W + Ci + Co)
    x = torch.rand((B, H, W, Ci), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, Ci, Co), device="cuda") * 2 - 1) * (2.0 / (9 * Ci)) ** 0.5
    bias = torch.rand(Co, device="cuda") * 0.1
    aux = torch.rand((B, H, W, Co), device="cuda") * 2 - 1
    wp = torch.empty(query("df_conv_packed_elems", 9, Ci, Co, 0), device="cuda")
    call("df_conv_pack_weights", _ptr(w), _ptr(wp), 9, Ci, Co, 0, s)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().permute(3, 2, 0, 1).cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
    ref = torch.maximum(ref, 0.2 * ref)
    y = torch.full((B, H, W, Co), float("nan"), device="cuda")
    f = lambda: call("df_conv_fwd", _ptr(x), _ptr(wp), _ptr(bias), None, None, _ptr(y), B, 1, H, W, Ci, Co, 1, DF_CONV_BIAS | DF_CONV_LRELU, 0.2, s)
    f(); torch.cuda.synchronize()
    err = float((y.double().cpu() - ref).abs().sum() / ref.abs().sum())
    t = timeit(f, 20, 5)
    # masked + residual epilogues against the plain launch
    y2 = torch.empty_like(y); y3 = torch.empty_like(y)
    call("df_conv_fwd", _ptr(x), _ptr(wp), None, None, None, _ptr(y2), B, 1, H, W, Ci, Co, 1, 0, 0.2, s)
    call("df_conv_fwd", _ptr(x), _ptr(wp), None, _ptr(aux), _ptr(aux), _ptr(y3), B, 1, H, W, Ci, Co, 1, DF_CONV_MASK | DF_CONV_RESIDUAL, 0.2, s)
    exp = y2 + aux; exp = torch.where(aux > 0, exp, 0.2 * exp)
    print("%s  B%d %dx%d %d->%d: rel-L1 vs fp64 %.2e   epilogues equal %s   %.1f us" % (tag, B, H, W, Ci, Co, err, bool(torch.equal(y3, exp)), t * 1e6), flush=True)
