import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch
import r06_smallbatch as sb
from deep_fluids_amd import trainer as T
for case in ["2d_b8", "ae2d_b8", "3d_b1", "2d_b64"]:
    out = {}
    for dg in (False, True):
        orig = T.default_config
        def dc(**kw):
            kw.setdefault("direct_grads", dg); return orig(**kw)
        T.default_config = dc
        try:
            for m in ("eager", "graph"):
                tr, x, y = sb.make(case, m == "graph")
                w, h = sb.wall(tr, x, y, 3, 20); w2, h2 = sb.wall(tr, x, y, 0, 20)
                out[f"dg{int(dg)}_{m}"] = round(min(w, w2), 3)
                del tr; torch.cuda.empty_cache()
        finally:
            T.default_config = orig
    print(case, json.dumps(out), flush=True)
