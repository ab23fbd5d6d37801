import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream
s = _stream(); torch.manual_seed(3)
for (B, Dc, Hc, Wc, C, N) in ((1, 2, 4, 4, 32, 32), (2, 3, 5, 7, 64, 32), (1, 7, 10, 7, 128, 128), (1, 8, 12, 8, 128, 128)):
    g = torch.rand((B, 2 * Dc, 2 * Hc, 2 * Wc, N), device="cuda") * 2 - 1
    w = (torch.rand((3, 3, 3, C, N), device="cuda") * 2 - 1) * 0.05
    wd = torch.empty(query("df_wino_packed_elems", C, N, 1), device="cuda")
    call("df_wino_pack_weights", _ptr(w), _ptr(wd), C, N, 1, s)
    acc = torch.rand((B, Dc, Hc, Wc, C), device="cuda")
    call("df_wino_upconv_dgrad", _ptr(g), _ptr(wd), _ptr(acc), B, Dc, Hc, Wc, C, N, s)
    torch.cuda.synchronize()
    print((B, Dc, Hc, Wc, C, N), hashlib.sha1(acc.cpu().numpy().tobytes()).hexdigest()[:16])
