import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream
from tools.gpu_probe import timeit
s = _stream()
B, D, H, W, C = 16, 64, 96, 64, 128
x = torch.rand((B, D, H, W, C), device="cuda") * 2 - 1
w = (torch.rand((3, 3, 3, C, C), device="cuda") * 2 - 1) * (2.0 / (27 * C)) ** 0.5
bias = torch.rand(C, device="cuda") * 0.1
wa = torch.empty(query("df_wino_packed_elems", C, C, 0), device="cuda"); call("df_wino_pack_weights", _ptr(w), _ptr(wa), C, C, 0, s)
wad = torch.empty(query("df_wino_packed_elems", C, C, 1), device="cuda"); call("df_wino_pack_weights", _ptr(w), _ptr(wad), C, C, 1, s)
xc = torch.rand((B, D // 2, H // 2, W // 2, C), device="cuda") * 2 - 1
acc = torch.zeros_like(xc); y = torch.empty_like(x)
f27 = 2.0 * C * C * B * D * H * W * 27.0 / 8.0
for rep in range(3):
    tu = timeit(lambda: call("df_wino_upconv_fwd", _ptr(xc), _ptr(wa), _ptr(bias), _ptr(y), B, D // 2, H // 2, W // 2, C, C, 9, 0.2, s), 4, 2)
    tp = timeit(lambda: call("df_wino_upconv_dgrad", _ptr(x), _ptr(wad), _ptr(acc), B, D // 2, H // 2, W // 2, C, C, s), 4, 2)
    print(os.environ.get("DF_HIP_LIBRARY", "release"), "up27 %.3f ms (%.3f)  pool27 %.3f ms (%.3f)" % (tu * 1e3, f27 / tu / 157.3e12, tp * 1e3, f27 / tp / 157.3e12), flush=True)
