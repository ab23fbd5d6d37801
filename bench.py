#!/usr/bin/env python3
"""bench.py -- the Deep Fluids velocity-field train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (config.workload): BASELINE cfg3 = 3-D 64x96x64 grid (Z,Y,X), 3-channel stream function / velocity,
GeneratorBE3 with filters=128, num_conv=4 (18 layers, 7,483,523 parameters), fp32 end to end, per-GPU batch 16,
one full step = generator fwd -> curl3 -> jacobian3 -> L1 + Jacobian-L1 -> backward -> (grad all-reduce) -> TF1 Adam
-> cosine LR.  Synthetic inputs resident in HBM, random-init (Xavier) weights.  Weak scaling: the global batch is
16 * N.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--res", type=int, nargs=3, default=[64, 96, 64], metavar=("Z", "Y", "X"))
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2"],
                    help="cfg3 = the BASELINE metric (3-D); cfg2 = 2-D 128x96 per-GPU batch 64 (diagnostic only)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="conv arithmetic: fp32 = exact fp32 MFMA (the BASELINE cfg3 dtype, default); bf16x3 = opt-in split-bf16 "
                         "MFMA mode (16-bit operand significands, fp32 accumulate)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra bf16x3-mode measurement appended at N=1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_inputs(batch, res, seed, ops):
    """y ~ U(-1,1) [B,3]; x = curl3(psi_gt) rescaled to max|x| = 1 (divergence-free, in [-1,1])."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    y = torch.rand((batch, 3), device="cuda", generator=g) * 2 - 1
    psi = torch.rand((batch, res[0], res[1], res[2], 3), device="cuda", generator=g) * 2 - 1
    x = ops.curl3(psi)
    x = x / x.abs().max()
    return x.contiguous(), y.contiguous()


def select_kernel(name, args):
    """KernelTimer filter: the three kernels whose rooflines are reported."""
    if name == "df_conv_fwd":
        B, D, H, W, cin, cout = args[6:12]
        if cin >= 64 and cout >= 64:
            return ("conv_mfma_kernel fwd/dgrad %dx%dx%d C%d->%d" % (D, H, W, cin, cout), 2.0 * 27 * cin * cout * B * D * H * W)
    if name == "df_wino_conv_fwd":
        B, D, H, W, cin, cout = args[6:12]
        # work = ALGORITHMIC flops of the convolution (what the direct kernel executes); the Winograd kernel executes 8/27 of it
        return ("wino3d_kernel fwd/dgrad %dx%dx%d C%d->%d" % (D, H, W, cin, cout), 2.0 * 27 * cin * cout * B * D * H * W)
    if name == "df_conv_wgrad":
        B, D, H, W, cin, cout = args[4:10]
        if cin >= 64 and cout >= 64:
            return ("wgrad_kernel %dx%dx%d C%dx%d" % (D, H, W, cin, cout), 2.0 * 27 * cin * cout * B * D * H * W)
    if name == "df_jacobian3d_fwd" and args[1] is not None and args[2] is not None:
        B, Z, Y, X = args[3:7]
        return ("jacobian3d_fwd_kernel<j,c>", 60.0 * B * Z * Y * X)
    return None


def cpu_baseline(res, filters, budget_s):
    """The oracle's PyTorch-CPU restatement of the SAME train step, timed on this node's host cores on a bounded
    sample (batch 1, grid halved per axis).  A reported baseline, not the optimisation target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import df_oracle as orc
    import df_oracle_torch as ort
    try:
        cores = len(os.sched_getaffinity(0))          # cores this process may actually use (cgroup / affinity)
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))                     # beyond one socket the small per-level convs only thrash
    torch.set_num_threads(cores)
    sres = [max(r // 4, 8) for r in res]
    rng = np.random.RandomState(123)
    oshape = sres + [3]
    p = ort.to_torch(orc.generator_init(rng, 3, oshape, filters))
    opt = ort.new_opt(p)
    x, y = orc.synthetic_batch(rng, 1, sres)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    t0 = time.time()
    ort.train_step(yt, xt, p, opt, oshape, filters, True)           # warm-up (thread pools, oneDNN primitives)
    warm = time.time() - t0
    n, t0, el = 0, time.time(), 0.0
    if warm < budget_s:
        while True:
            ort.train_step(yt, xt, p, opt, oshape, filters, True)
            n += 1
            el = time.time() - t0
            if el >= budget_s or n >= 50:
                break
    else:                                                            # pathologically slow host: keep the one step
        n, el = 1, warm
    vox = float(np.prod(sres))
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": vox * n / el, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": "PyTorch-CPU fp32 restatement of the reference graph (TF 1.15 unavailable): %d full train steps, "
                      "batch 1, grid %dx%dx%d, filters %d, %d threads, %.1f s" % (n, sres[0], sres[1], sres[2], filters,
                                                                                  cores, el),
            "cpu": model}


def l1_vs_oracle(filters, precision="fp32"):
    """Relative L1 of the velocity field vs the fp64 oracle on identical inputs/weights (reduced grid 16x24x16)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import df_oracle as orc
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, default_config
    ops.CONV_PRECISION = precision
    ops.reset_variables()
    rng = np.random.RandomState(123)
    spatial = (16, 24, 16)
    oshape = list(spatial) + [3]
    p = orc.generator_init(rng, 3, oshape, filters)
    x, y = orc.synthetic_batch(rng, 1, spatial)
    cfg = default_config(is_3d=True, res_x=16, res_y=24, res_z=16, filters=filters, batch_size=1, num_samples=100)
    tr = Trainer(cfg)
    tr.load_variables(p)
    u = tr.generate(torch.from_numpy(y).cuda()).cpu().numpy().astype(np.float64)
    psi = orc.generator_fwd(y.astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()}, oshape, filters)
    ref = orc.curl3(psi)
    ops.reset_variables()
    return float(np.abs(u - ref).sum() / np.abs(ref).sum())


def main():
    a = parse()
    from deep_fluids_amd import _lib, ops
    from deep_fluids_amd.dist import init_from_env
    from deep_fluids_amd.trainer import Trainer, default_config

    rank, local_rank, world = init_from_env()
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (a.gpus, a.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())

    rel_l1 = l1_vs_oracle(a.filters, a.precision) if rank == 0 else None
    ops.CONV_PRECISION = a.precision

    Z, Y, X = a.res
    cfg = default_config(is_3d=True, res_x=X, res_y=Y, res_z=Z, filters=a.filters, batch_size=a.batch * world,
                         num_samples=6600, random_seed=123)     # smoke3_obs_buo: 11*4*150 samples
    tr = Trainer(cfg)                                           # same seed on every rank -> identical init
    if world > 1:
        tr.enable_data_parallel()
    x, y = make_inputs(a.batch, a.res, 123 + rank, ops)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.train_step(x, y)
    timer = _lib.KernelTimer(select_kernel)
    sync_all()
    _lib.TIMER = timer
    t0 = time.perf_counter()
    last = None
    for _ in range(a.steps):
        last = tr.train_step(x, y)
    sync_all()
    elapsed = time.perf_counter() - t0
    _lib.TIMER = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(last.g_loss.detach())
    assert loss == loss, "Model diverged with loss = NaN"        # trainer.py:275

    if rank != 0:
        return
    vox_per_step = a.batch * Z * Y * X * world
    value = vox_per_step * a.steps / elapsed
    ks = timer.summary()

    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["kernels"]
    except (OSError, ValueError, KeyError):
        pmc = {}

    default_shape = list(a.res) == [64, 96, 64] and a.batch == 16 and a.filters == 128   # the shape the PMC passes ran on

    def roof(prefix, peak, unit, scale):
        sel = {k: v for k, v in ks.items() if k.startswith(prefix)}
        if not sel:
            return None
        k = max(sel, key=lambda q: sel[q]["seconds"])          # the dominant instance (top resolution)
        v = sel[k]
        ach = v["work"] / v["seconds"] / scale
        return {"kernel": k, "bound": "mfma" if unit == "TFLOP/s" else "hbm", "achieved": ach, "peak": peak, "unit": unit,
                "frac": ach / peak,
                # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE),
                # collected offline on the same kernels/shapes: profiles/pmc_latest.json
                "traffic": pmc.get(prefix, {}).get("traffic_bytes") if default_shape else None, "launches": v["launches"],
                "avg_launch_ms": v["seconds"] / v["launches"] * 1e3, "work_per_launch": v["work"] / v["launches"]}

    out = {
        "metric": "velocity-field voxels/sec (3D %dx%dx%d train step), whole job; per-GPU in `per_gpu`" % (Z, Y, X),
        "value": value, "unit": "voxels/s", "per_gpu": value / world,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if a.precision == "fp32" else "bf16x3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "cfg3: 3D smoke3 %dx%dx%d (Z,Y,X) fp32, GeneratorBE3 filters=%d num_conv=4, "
                               "per-GPU batch %d, full train step (fwd+curl3+jacobian3+L1 losses+bwd+Adam)" % (Z, Y, X, a.filters, a.batch),
                   "global_batch": a.batch * world, "grid": [Z, Y, X], "params": tr.n_params,
                   "parallelism": "dp%d" % world},
        "loss": loss,
        "l1_vs_ref": {"value": rel_l1, "tolerance": 1e-4,
                      "case": "relative L1 of the velocity field vs the fp64 oracle, grid 16x24x16, filters %d" % a.filters},
        "roofline": None,            # filled below: the kernel family with the largest share of the step
        "roofline_wgrad": roof("wgrad_kernel", PEAK_FP32_MFMA_TFLOPS, "TFLOP/s", 1e12),
        "roofline_conv": roof("conv_mfma_kernel", PEAK_FP32_MFMA_TFLOPS, "TFLOP/s", 1e12),
        "roofline_wino": roof("wino3d_kernel", PEAK_FP32_MFMA_TFLOPS, "TFLOP/s", 1e12),
        "roofline_stencil": roof("jacobian3d_fwd_kernel", PEAK_HBM_GBS, "GB/s", 1e9),
        "kernels": {k: {"launches": v["launches"], "ms_total": v["seconds"] * 1e3} for k, v in sorted(ks.items())},
    }
    if out["roofline_wgrad"]:
        # same convention for the weight gradient: Winograd in (x,y,z) (F(2x2x2,3x3x3)) executes 8/27 of the algorithmic flops at the
        # top levels (conv_wgrad.hip picks direct / x / (x,y) / (x,y,z) by size; the dominant instance is an (x,y,z) one)
        rg = out["roofline_wgrad"]
        rg["mfma_executed_tflops"] = rg["achieved"] * 8.0 / 27.0
        rg["mfma_executed_frac"] = rg["frac"] * 8.0 / 27.0
        rg["note"] = "achieved = direct-convolution flops / time (can exceed the fp32 MFMA peak); mfma_executed_* = flops the Winograd-(x,y,z) kernel issues"
    if out["roofline_wino"]:
        # `achieved` above counts the convolution's algorithmic flops; Winograd F(2x2x2,3x3x3) executes 8/27 of
        # them on the matrix pipe, so the matrix-pipe utilisation is frac * 8/27 -- both are reported
        rw = out["roofline_wino"]
        rw["mfma_executed_tflops"] = rw["achieved"] * 8.0 / 27.0
        rw["mfma_executed_frac"] = rw["frac"] * 8.0 / 27.0
        rw["note"] = "achieved = direct-convolution flops / time (can exceed the fp32 MFMA peak); mfma_executed_* = flops the kernel issues"
    fam = {}
    for k, v in ks.items():
        fam[k.split(" ")[0]] = fam.get(k.split(" ")[0], 0.0) + v["seconds"]
    dom = max((f for f in fam if f != "jacobian3d_fwd_kernel<j,c>"), key=lambda f: fam[f], default=None)
    out["roofline"] = {"wgrad_kernel": out["roofline_wgrad"], "conv_mfma_kernel": out["roofline_conv"],
                       "wino3d_kernel": out["roofline_wino"]}.get(dom)
    # extra, clearly separate from `value`: the same step in the opt-in bf16x3 conv mode (NOT the reported metric)
    out["alt_bf16x3_mode"] = None
    try:
        if world == 1 and a.precision == "fp32" and not a.no_alt:
            rel_alt = l1_vs_oracle(a.filters, "bf16x3")
            ops.CONV_PRECISION = "bf16x3"
            ops.reset_variables()
            tr2 = Trainer(cfg)
            for _ in range(2):
                tr2.train_step(x, y)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(3):
                tr2.train_step(x, y)
            torch.cuda.synchronize(); el2 = (time.perf_counter() - t1) / 3
            ops.CONV_PRECISION = "fp32"
            out["alt_bf16x3_mode"] = {"ms_per_step": el2 * 1e3, "value": vox_per_step / el2, "unit": "voxels/s",
                                      "l1_vs_ref": rel_alt, "note": "opt-in precision mode, not the BASELINE cfg3 dtype: conv operands "
                                      "split into bf16 hi/lo words, 3 bf16 MFMAs per product, fp32 accumulation"}
    except Exception as e:      # an extra must never take the metric line down with it
        out["alt_bf16x3_mode"] = {"error": repr(e)[:300]}
    ops.CONV_PRECISION = "fp32"
    # the north star also asks for the 2-D grid: same train step on 128x96, batch 64 (BASELINE cfg2's shape; fp32 here)
    out["extra_2d_128x96"] = None
    try:
        if world == 1 and not a.no_alt:
            ops.reset_variables()
            cfg2 = default_config(is_3d=False, res_x=96, res_y=128, filters=a.filters, batch_size=64, num_samples=21000)
            tr3 = Trainer(cfg2)
            g2 = torch.Generator(device="cuda").manual_seed(1)
            y2 = torch.rand((64, 3), device="cuda", generator=g2) * 2 - 1
            x2 = ops.curl(torch.rand((64, 128, 96, 1), device="cuda", generator=g2) * 2 - 1)
            x2 = (x2 / x2.abs().max()).contiguous()
            for _ in range(3):
                tr3.train_step(x2, y2)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            for _ in range(10):
                tr3.train_step(x2, y2)
            torch.cuda.synchronize(); el3 = (time.perf_counter() - t2) / 10
            out["extra_2d_128x96"] = {"ms_per_step": el3 * 1e3, "value": 64 * 128 * 96 / el3, "unit": "pixels/s", "batch": 64,
                                      "conv_tflops_reference_equivalent": 3.71e12 / el3 / 1e12, "dtype": "f32",
                                      "note": "2-D 128x96 train step (GeneratorBE filters=128), Winograd F(2x2,3x3) forward/dgrad + Winograd-(x,y) weight gradient at the top levels; not the reported metric"}
            ops.reset_variables()
    except Exception as e:      # an extra must never take the metric line down with it
        out["extra_2d_128x96"] = {"error": repr(e)[:300]}
    out["cpu_baseline"] = None
    try:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.res, a.filters, a.cpu_seconds)
    except Exception as e:      # an extra must never take the metric line down with it
        out["cpu_baseline"] = {"error": repr(e)[:300]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
