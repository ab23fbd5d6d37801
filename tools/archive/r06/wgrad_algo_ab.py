import os, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import r06_smallbatch as sb
from deep_fluids_amd import ops
for case in sys.argv[1:]:
    out = {}
    for algo in (0, 3, 0, 3):
        with ops.options(wgrad_algo=algo):
            tr, x, y = sb.make(case, False)
            w, h = sb.wall(tr, x, y, 3, 20); w2, h2 = sb.wall(tr, x, y, 0, 20)
            out.setdefault("algo%d" % algo, []).append(round(min(w, w2), 3))
            del tr; torch.cuda.empty_cache()
    print(case, json.dumps(out), flush=True)
