import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import df_oracle as orc
from gpu_util import dev, host, rel_linf
from deep_fluids_amd import ops
from deep_fluids_amd.trainer import Trainer, default_config

is_3d = len(sys.argv) < 2 or sys.argv[1] == "3"
spatial, filters, batch = ((8, 16, 8), 16, 2) if is_3d else ((16, 8), 16, 3)
ops.reset_variables()
rng = np.random.RandomState(123)
oshape = list(spatial) + [3 if is_3d else 1]
p = orc.generator_init(rng, 3, oshape, filters)
x, y = orc.synthetic_batch(rng, batch, spatial)
cfg = default_config(is_3d=is_3d, res_x=spatial[-1], res_y=spatial[-2], res_z=spatial[0] if is_3d else 1,
                     filters=filters, batch_size=batch, num_samples=1000)
tr = Trainer(cfg); tr.load_variables(p)
tr.flat_g.zero_()
m = tr.build_model(dev(x), dev(y))
m.g_loss.backward()
torch.cuda.synchronize()
p64 = {k: v.astype(np.float64) for k, v in p.items()}
psi, cache = orc.generator_fwd(y.astype(np.float64), p64, oshape, filters, keep=True)
res = orc.velocity_loss(psi, x.astype(np.float64), is_3d)
grads = orc.generator_bwd(res["dpsi"], cache, p64)
gr = tr.grads_numpy()
for k in gr:
    print("%-22s shape %-20s |ref|max %.3e  |got|max %.3e  rel %.3e  nan %d" % (k, gr[k].shape, np.abs(grads[k]).max(), np.abs(gr[k]).max(), rel_linf(gr[k], grads[k]), np.isnan(gr[k]).sum()))
