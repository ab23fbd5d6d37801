#!/usr/bin/env python3
"""Where the time of the (x,y,z) Winograd weight gradient goes (tuning library, results wrong by construction for dbg != 0):
dbg 0 = production (one fused launch), 8 = four launches by (GZ, GY) class (the round-1 form), 9 = fused + cached loads (L1 hits),
10 = fused + every range reads the same 8 tile rows (L2-resident working set, L1 misses as in production: the bound of any L2 prefetch);
for the four-launch form: bit 1 = every operand load reads a cached zero row (no memory latency / bandwidth), 2 = no (z, y) operand combinations,
4 = no x transform either (MFMAs + loads only); W = 64 rows only."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deep_fluids_amd import _lib as _libmod  # noqa: E402
_libmod.use_tuning_library()
from deep_fluids_amd._lib import call, query, lib  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream  # noqa: E402
from tools.gpu_probe import timeit  # noqa: E402

s = _stream()
for (B, D, H, W) in ((16, 64, 96, 64),) if os.environ.get('DBGS') else ((16, 64, 96, 64), (4, 64, 96, 64)):
    C = 128
    x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    g = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
    nb = query("df_conv_wgrad_workspace_bytes", B, D, H, W, C, C, 3)
    ws = torch.empty((nb + 3) // 4, device="cuda")
    gw = torch.empty((27, C, C), device="cuda"); gb = torch.empty(C, device="cuda")
    fl = 2.0 * 27 * C * C * B * D * H * W / 3.375
    for dbg in [int(v) for v in os.environ.get('DBGS', '0,8,9,1,3,7').split(',')]:
        lib().df_debug_set_wgrad(ctypes.c_int(dbg))
        f = lambda: call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, 4, s)
        f(); torch.cuda.synchronize()
        t = timeit(f, 3, 1)
        print("B%d %dx%dx%d dbg %d: %.3f ms  executed %.1f TFLOP/s (%.2f of 157.3)" % (B, D, H, W, dbg, t * 1e3, fl / t / 1e12, fl / t / 157.3e12), flush=True)
    res = {}
    for dbg in (8, 0):
        lib().df_debug_set_wgrad(ctypes.c_int(dbg))
        call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, D, H, W, C, C, 3, _ptr(ws), nb, 4, s)
        res[dbg] = (gw.clone(), gb.clone())
    o = 8
    print("  one fused launch (0) vs variant %d: gw equal %s, gb equal %s" % (o, torch.equal(res[0][0], res[o][0]), torch.equal(res[0][1], res[o][1])))
    lib().df_debug_set_wgrad(ctypes.c_int(0))
    del x, g, ws
