"""Host issue time vs wall time of the 2-D default-batch step, with and without the memoised size queries (same process, alternating)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import r06_smallbatch as sb
from deep_fluids_amd import ops, _lib
memo = _lib.query
raw = lambda name, *a: getattr(_lib.lib(), name)(*a)
for case in ("2d_b8", "ae2d_b8"):
    tr, x, y = sb.make(case, False)
    sb.wall(tr, x, y, 3, 5)
    for rep in range(3):
        r = {}
        for label, q in (("raw", raw), ("memo", memo)):
            ops.query = q
            # host-only cost: issue 20 steps with the device idle at the start; `h` = time until the host has issued them all
            w, h = sb.wall(tr, x, y, 0, 20)
            r[label] = (round(w, 3), round(h, 3))
        ops.query = memo
        print(case, json.dumps(r), flush=True)
    del tr; torch.cuda.empty_cache()
