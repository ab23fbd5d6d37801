"""2-D weight gradient at the default batch (B = 8, 128x96, 128 -> 128; the x form, 3 workgroup types): voxel ranges 64 (default: 192 workgroups) vs 85 (255)
vs 128 / 170 -- standalone, a generous workspace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream
from tools.gpu_probe import timeit
s = _stream(); C = 128
for (B, H, W) in ((8, 128, 96), (8, 64, 48), (8, 32, 24)):
    x = torch.rand((B, 1, H, W, C), device="cuda") - 0.5; g = torch.rand((B, 1, H, W, C), device="cuda") - 0.5
    gw = torch.empty((3, 3, C, C), device="cuda"); gb = torch.empty(C, device="cuda")
    nb = 1 << 30
    ws = torch.empty(nb // 4, device="cuda")
    ref = None
    for ranges in (0, 64, 85, 128, 170, 256):
        f = lambda: call("df_conv_wgrad_algo", _ptr(x), _ptr(g), _ptr(gw), _ptr(gb), B, 1, H, W, C, C, 1, _ptr(ws), nb, (ranges << 3) | 0, s)
        f(); torch.cuda.synchronize()
        t = timeit(f, 20, 5)
        print("B%d %dx%d ranges %3d: %.1f us" % (B, H, W, ranges, t * 1e6), flush=True)
