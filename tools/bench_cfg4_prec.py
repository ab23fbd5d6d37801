import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from deep_fluids_amd import ops
from deep_fluids_amd.trainer import Trainer, default_config
B = 4
for prec in ("fp32", "bf16x3"):
    ops.CONV_PRECISION = prec
    ops.reset_variables()
    cfg = default_config(is_3d=True, res_x=112, res_y=160, res_z=112, filters=128, batch_size=B, num_samples=20000)
    tr = Trainer(cfg)
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.rand((B, 3), device="cuda", generator=g) * 2 - 1
    x = ops.curl3(torch.rand((B, 112, 160, 112, 3), device="cuda", generator=g) * 2 - 1)
    x = (x / x.abs().max()).contiguous()
    for _ in range(2): m = tr.train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m = tr.train_step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("cfg4 B=4 %s: %.1f ms/step loss %.5f" % (prec, dt * 1e3, float(m.g_loss.detach())))
    del tr, m; torch.cuda.empty_cache()
