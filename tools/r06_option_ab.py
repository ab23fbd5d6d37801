#!/usr/bin/env python3
"""A/B of one boolean trainer option (default_config key) on the small-batch cases of tools/r06_smallbatch.py, same process, eager steps.

  r06_option_ab.py direct_grads [case ...]      -> one line per case: ms/step with the option off / on
"""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import r06_smallbatch as sb  # noqa: E402
from deep_fluids_amd import trainer as T  # noqa: E402

opt = sys.argv[1]
for case in (sys.argv[2:] or ["2d_b8", "ae2d_b8", "dg2d_b8", "3d_b1", "2d_b64", "ae3d_b4"]):
    out = {}
    for on in (False, True):
        orig = T.default_config

        def dc(**kw):
            kw.setdefault(opt, on)
            return orig(**kw)
        T.default_config = dc
        try:
            tr, x, y = sb.make(case, False)
            w, h = sb.wall(tr, x, y, 3, 20)
            w2, h2 = sb.wall(tr, x, y, 0, 20)
            out["%s=%d" % (opt, on)] = round(min(w, w2), 3)
            del tr
            torch.cuda.empty_cache()
        finally:
            T.default_config = orig
    print(case, json.dumps(out), flush=True)
