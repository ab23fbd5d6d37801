"""Weight-pack kernels of the F(4,3) families: sha-1 of the packed operand (per build) and time."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from deep_fluids_amd._lib import call, query
from deep_fluids_amd.ops import _ptr, _stream
from tools.gpu_probe import timeit
s = _stream(); tag = os.path.basename(os.environ.get("DF_HIP_LIBRARY", "release"))
torch.manual_seed(1)
for C in (128, 64):
    w3 = torch.rand((3, 3, 3, C, C), device="cuda") - 0.5; w2 = torch.rand((3, 3, C, C), device="cuda") - 0.5
    for mode in (0, 1):
        p3 = torch.zeros(query("df_wino43_packed_elems", C, C, mode), device="cuda"); p2 = torch.zeros(query("df_wino2d43_packed_elems", C, C, mode), device="cuda")
        f3 = lambda: call("df_wino43_pack_weights", _ptr(w3), _ptr(p3), C, C, mode, s)
        f2 = lambda: call("df_wino2d43_pack_weights", _ptr(w2), _ptr(p2), C, C, mode, s)
        f3(); f2(); torch.cuda.synchronize()
        h3 = hashlib.sha1(p3.cpu().numpy().tobytes()).hexdigest()[:10]; h2 = hashlib.sha1(p2.cpu().numpy().tobytes()).hexdigest()[:10]
        t3 = timeit(f3, 30, 5); t2 = timeit(f2, 30, 5)
        print("%s C%d mode %d: wino43 pack %s %.1f us   wino2d43 pack %s %.1f us" % (tag, C, mode, h3, t3 * 1e6, h2, t2 * 1e6), flush=True)
