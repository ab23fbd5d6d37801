"""2-D default-batch step with the Winograd kernels forced at every level (conv_algo="winograd") vs the automatic thresholds."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import r06_smallbatch as sb
from deep_fluids_amd import ops
for case in sys.argv[1:] or ["2d_b8", "ae2d_b8", "3d_b1"]:
    out = {}
    for algo in ("auto", "winograd", "auto", "winograd"):
        with ops.options(conv_algo=algo):
            tr, x, y = sb.make(case, False)
            w, h = sb.wall(tr, x, y, 3, 20); w2, h2 = sb.wall(tr, x, y, 0, 20)
            out.setdefault(algo, []).append(round(min(w, w2), 3))
            del tr; torch.cuda.empty_cache()
    print(case, json.dumps(out), flush=True)
