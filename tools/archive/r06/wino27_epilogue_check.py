"""conv_wino.hip epilogue A/B (run per build, DF_HIP_LIBRARY): sha-1 of df_wino_upconv_fwd / _bits and df_wino_conv_fwd / _bits outputs + sign bytes on full and
ragged shapes, and the top-level launch times of the 27-point forward."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream, _new_bits
from tools.gpu_probe import timeit
s = _stream(); tag = os.path.basename(os.environ.get("DF_HIP_LIBRARY", "release"))
sha = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:10]
def run(B, Dc, Hc, Wc, C, time_it=False):
    torch.manual_seed(B + Dc * 3 + Hc * 5 + Wc * 7 + C)
    xc = torch.rand((B, Dc, Hc, Wc, C), device="cuda") * 2 - 1
    xf = torch.rand((B, 2 * Dc, 2 * Hc, 2 * Wc, C), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
    bias = torch.rand(C, device="cuda") * 0.1 - 0.05
    wa = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda"); call("df_wino_pack_weights", _ptr(w), _ptr(wa), C, C, 0, s)
    y = torch.full_like(xf, float("nan")); bits = _new_bits((B, 2 * Dc, 2 * Hc, 2 * Wc), C, xf); bits.zero_()
    out = []
    def go(label, fn, *ts):
        fn(); torch.cuda.synchronize()
        t = timeit(fn, 4, 2) if time_it else 0.0
        out.append("%s %s%s" % (label, "/".join(sha(q) for q in ts), (" %.3f ms" % (t * 1e3)) if time_it else ""))
    go("up", lambda: call("df_wino_upconv_fwd", _ptr(xc), _ptr(wa), _ptr(bias), _ptr(y), B, Dc, Hc, Wc, C, C, 9, 0.2, s), y)
    go("up+bits", lambda: call("df_wino_upconv_fwd_bits", _ptr(xc), _ptr(wa), _ptr(bias), _ptr(y), _ptr(bits), B, Dc, Hc, Wc, C, C, 0.2, s), y, bits)
    go("plain", lambda: call("df_wino_conv_fwd", _ptr(xf), _ptr(wa), _ptr(bias), None, None, _ptr(y), B, 2 * Dc, 2 * Hc, 2 * Wc, C, C, 9, 0.2, s), y)
    go("plain+bits", lambda: call("df_wino_conv_fwd_bits", _ptr(xf), _ptr(wa), _ptr(bias), None, _ptr(y), _ptr(bits), B, 2 * Dc, 2 * Hc, 2 * Wc, C, C, 9, 0.2, s), y, bits)
    go("plain nobias", lambda: call("df_wino_conv_fwd", _ptr(xf), _ptr(wa), None, None, None, _ptr(y), B, 2 * Dc, 2 * Hc, 2 * Wc, C, C, 0, 0.2, s), y)
    go("masked", lambda: call("df_wino_conv_fwd", _ptr(xf), _ptr(wa), None, None, _ptr(xf), _ptr(y), B, 2 * Dc, 2 * Hc, 2 * Wc, C, C, 4, 0.2, s), y)
    print("%s  B%d coarse %dx%dx%d C%d:  %s" % (tag, B, Dc, Hc, Wc, C, "  ".join(out)), flush=True)
for shp in ((1, 2, 4, 4, 32), (2, 3, 5, 7, 64), (1, 7, 10, 7, 128), (1, 4, 8, 8, 64)):
    run(*shp)
run(16, 32, 48, 32, 128, True)
