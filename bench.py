#!/usr/bin/env python3
"""bench.py -- the Deep Fluids velocity-field train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5] [--scaling weak|strong]

N > 1: one rank per GPU over RCCL.  Either launch it under ``python -m torch.distributed.run --nproc-per-node N ...`` (the
ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or run it as plain ``python bench.py --gpus N``:
without WORLD_SIZE in the environment the script re-executes itself under torch.distributed.run on 127.0.0.1.

Workload (config.workload): BASELINE cfg3 = 3-D 64x96x64 grid (Z,Y,X), 3-channel stream function / velocity,
GeneratorBE3 with filters=128, num_conv=4 (18 layers, 7,483,523 parameters), fp32 end to end, one full step = generator
fwd -> curl3 -> jacobian3 -> L1 + Jacobian-L1 -> backward -> (bucketed grad all-reduce, overlapped) -> TF1 Adam ->
cosine LR.  Synthetic inputs resident in HBM, random-init (Xavier) weights.
  --scaling weak   (default) per-GPU batch 16, global batch 16 N;
  --scaling strong global batch 16 split N ways (16 -> 2 per GPU at N = 8).
--config picks another BASELINE.json workload for the SAME contract line (default cfg3, the metric's own): cfg2 = 2-D 128x96 batch 64
per GPU (Trainer, GeneratorBE), cfg4 = 3-D 112x160x112 batch 4 per GPU (32 / 8 GPUs), cfg5 = AE3 128^3 filters 64 batch 4 per GPU
(AETrainer: encoder + decoder).  At N > 1 the cfg3 line additionally carries short cfg2 and cfg4 data-parallel legs (`extra_leg_cfg2`,
`extra_leg_cfg4`; --no-extra-legs skips them) and every leg a `cross_rank` check: the reduced flat gradient is bit-identical on every
rank and the mean of the shard losses equals ONE process's loss on the gathered global batch to 1e-6.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 peak
PEAK_HBM_GBS = 8000.0
# fraction of the convolution's algorithmic (direct-form) multiply-adds the Winograd kernels execute on the matrix pipe
EXEC_RATIO = {"wino43_kernel": 6.0 / 27.0, "wino3d_kernel": 8.0 / 27.0, "wino2d43_kernel": 3.0 / 9.0, "wino2d_kernel": 4.0 / 9.0, "conv_mfma_kernel": 1.0, "wgrad_kernel": None,
              "wino3d_27pt_kernel up": 1.0 / 8.0, "wino3d_27pt_kernel pooled": 1.0 / 8.0}      # 27 of the 64 points: 27/8 MACs per output voxel and channel pair


# BASELINE.json `configs` as bench workloads (cfg2 = trainer.py:136-184 on GeneratorBE; cfg3 / cfg4 = trainer3.py:14-63 on
# GeneratorBE3; cfg5 = build_model_ae, trainer3.py:240-309, on AE3).  `batch` = per-GPU batch of the weak leg = global batch of the strong leg.
WORKLOADS = {
    "cfg2": dict(kind="de", res=[128, 96], batch=64, filters=128, num_samples=21000, unit="pixels/s", model="GeneratorBE",
                 name="cfg2: 2D smoke_pos_size {grid} (Y,X)", step="fwd+curl+jacobian+L1 losses+bwd+Adam"),
    "cfg3": dict(kind="de", res=[64, 96, 64], batch=16, filters=128, num_samples=6600, unit="voxels/s", model="GeneratorBE3",
                 name="cfg3: 3D smoke3 {grid} (Z,Y,X)", step="fwd+curl3+jacobian3+L1 losses+bwd+Adam"),
    "cfg4": dict(kind="de", res=[112, 160, 112], batch=4, filters=128, num_samples=20000, unit="voxels/s", model="GeneratorBE3",
                 name="cfg4: 3D smoke3 {grid} (Z,Y,X) (BASELINE: global batch 32 = 4 per GPU x 8)", step="fwd+curl3+jacobian3+L1 losses+bwd+Adam"),
    "cfg5": dict(kind="ae", res=[128, 128, 128], batch=4, filters=64, num_samples=5000, unit="voxels/s", model="AE3 (EncoderBE3 + GeneratorBE3) z_num=16",
                 name="cfg5: 3D autoencoder liquid3 {grid}", step="encoder+decoder fwd+curl3+jacobian3+L1 losses+latent loss+bwd+Adam"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", choices=sorted(WORKLOADS),
                    help="BASELINE.json workload (default cfg3 = the metric's own; cfg2 2-D 128x96 B=64/GPU, cfg4 112x160x112 B=4/GPU, cfg5 AE3 128^3 F=64 B=4/GPU)")
    ap.add_argument("--batch", type=int, default=None, help="weak: per-GPU batch; strong: GLOBAL batch (split over the GPUs); default: the workload's (cfg3: 16)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--res", type=int, nargs="+", default=None, metavar="N", help="grid (Z Y X | Y X); default: the workload's (cfg3: 64 96 64)")
    ap.add_argument("--filters", type=int, default=None)
    ap.add_argument("--no-extra-legs", action="store_true", help="N > 1, --config cfg3: skip the short cfg2 and cfg4 data-parallel legs appended to the line")
    ap.add_argument("--extra-leg-steps", type=int, default=5)
    ap.add_argument("--no-cross-rank-check", action="store_true", help="N > 1: skip the gradient-checksum / global-batch-loss check of each leg")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="conv arithmetic: fp32 = exact fp32 MFMA (the BASELINE cfg3 dtype, default); bf16x3 = opt-in split-bf16 "
                         "MFMA mode (16-bit operand significands, fp32 accumulate)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra measurements appended at N=1 (bf16x3 mode, 2-D, cfg4, AE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget of the CPU-baseline sample (one warm-up + >= 5 timed full-grid steps at ~2.9 s each on the bench box)")
    ap.add_argument("--init-timeout", type=float, default=120.0,
                    help="N > 1: seconds the rendezvous / the RCCL communicator creation + first all-reduce may take before the watchdog "
                         "prints a diagnostic JSON line (rccl_ranks 0) and exits non-zero")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1: do not run the two rocprofv3 --pmc passes that measure `roofline.traffic` in this run (the committed offline table "
                         "profiles/pmc_latest.json is used instead)")
    ap.add_argument("--no-other-leg", action="store_true", help="N > 1: skip the short leg in the other scaling mode")
    ap.add_argument("--other-steps", type=int, default=5)
    ap.add_argument("--sidecar", default=None, help="where the full record goes (default gpurun_out/bench_full_n<N>.json)")
    ap.add_argument("--full-line", action="store_true", help="print the full record instead of the compact line")
    a = ap.parse_args()
    wl = WORKLOADS[a.config]
    a.batch = wl["batch"] if a.batch is None else a.batch
    a.res = list(wl["res"]) if a.res is None else list(a.res)
    a.filters = wl["filters"] if a.filters is None else a.filters
    if len(a.res) != len(wl["res"]):
        ap.error("--config %s takes a %d-D grid" % (a.config, len(wl["res"])))
    return a


def self_launch(a):
    """``python bench.py --gpus N`` without a launcher: re-execute under torch.distributed.run (one rank per GPU, rendezvous
    on 127.0.0.1 -- the container hostname may not resolve) and hand its exit code back."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_inputs(batch, res, seed, ops):
    """np.random.RandomState(seed) (123 + rank), never a framework RNG: y ~ U(-1,1) [B,3]; x = curl3(psi_gt) with
    psi_gt ~ U(-1,1), rescaled to max|x| = 1 (divergence-free, in [-1,1] like the reference's normalised data, data.py:87-88,329)."""
    import numpy as np
    import torch
    rng = np.random.RandomState(seed)
    y = torch.from_numpy(rng.uniform(-1, 1, (batch, 3)).astype(np.float32)).cuda()
    psi = torch.empty([batch] + list(res) + [3], dtype=torch.float32, device="cuda")
    for b in range(batch):
        psi[b].copy_(torch.from_numpy(rng.uniform(-1, 1, list(res) + [3]).astype(np.float32)))
    x = ops.curl3(psi) if len(res) == 3 else ops.curl(psi[..., :1].contiguous())
    x = x / x.abs().max()
    return x.contiguous(), y.contiguous()


def select_kernel(name, args):
    """KernelTimer filter: the kernels whose rooflines are reported.  work = ALGORITHMIC flops (direct form) / bytes."""
    if name == "df_conv_fwd":
        B, D, H, W, cin, cout, kz = args[6:13]
        if cin >= 64 and cout >= 64:
            return ("conv_mfma_kernel fwd/dgrad %dx%dx%d C%d->%d" % (D, H, W, cin, cout), 2.0 * (27 if kz == 3 else 9) * cin * cout * B * D * H * W)
    if name in ("df_wino_conv_fwd", "df_wino_conv_fwd_addup", "df_wino_conv_fwd_addup_bits", "df_wino_conv_fwd_bits"):
        B, D, H, W, cin, cout = args[6:12]
        return ("wino3d_kernel fwd/dgrad %dx%dx%d C%d->%d" % (D, H, W, cin, cout), 2.0 * 27 * cin * cout * B * D * H * W)
    if name == "df_wino43_conv":      # F(2,3) x F(2,3) x F(4,3): 6 of the 27 direct-form multiply-adds
        B, D, H, W, cin, cout = args[9:15]
        return ("wino43_kernel fwd/dgrad %dx%dx%d C%d->%d" % (D, H, W, cin, cout), 2.0 * 27 * cin * cout * B * D * H * W)
    if name in ("df_wino_upconv_fwd", "df_wino_upconv_fwd_bits", "df_wino_upconv_dgrad"):      # the 27-point forms (fine grid = 2 x coarse)
        o = {"df_wino_upconv_fwd": 4, "df_wino_upconv_fwd_bits": 5, "df_wino_upconv_dgrad": 3}[name]
        B, Dc, Hc, Wc, cin, cout = args[o:o + 6]
        kind = "pooled adjoint" if name.endswith("dgrad") else "up-sampling-aware forward"
        return ("wino3d_27pt_kernel %s -> %dx%dx%d C%d->%d" % (kind, 2 * Dc, 2 * Hc, 2 * Wc, cin, cout), 2.0 * 27 * cin * cout * B * 8 * Dc * Hc * Wc)
    if name == "df_wino2d43_conv":      # F(2,3) x F(4,3): 3 of the 9 direct-form multiply-adds
        B, H, W, cin, cout = args[6:11]
        return ("wino2d43_kernel fwd/dgrad %dx%d C%d->%d" % (H, W, cin, cout), 2.0 * 9 * cin * cout * B * H * W)
    if name == "df_wino2d_conv_fwd":
        B, H, W, cin, cout = args[6:11]
        return ("wino2d_kernel fwd/dgrad %dx%d C%d->%d" % (H, W, cin, cout), 2.0 * 9 * cin * cout * B * H * W)
    if name == "df_conv_wgrad_algo":
        B, D, H, W, cin, cout, kz = args[4:11]
        if cin >= 64 and cout >= 64:
            return ("wgrad_kernel %dx%dx%d C%dx%d B%d" % (D, H, W, cin, cout, B), 2.0 * (27 if kz == 3 else 9) * cin * cout * B * D * H * W)
    if name == "df_jacobian3d_fwd" and args[1] is not None and args[2] is not None:
        B, Z, Y, X = args[3:7]
        return ("jacobian3d_fwd_kernel<j,c>", 60.0 * B * Z * Y * X)
    if name == "df_jacobian3d_bwd":
        B, Z, Y, X = args[3:7]
        nb = (36.0 if args[0] is not None else 0.0) + (12.0 if args[1] is not None else 0.0) + 12.0
        return ("jacobian3d_bwd_kernel<%s>" % ("j" if args[0] is not None else "c"), nb * B * Z * Y * X)
    if name == "df_velocity_loss3d_fwd":
        B, Z, Y, X = args[5:9]
        return ("velocity_loss3d_fwd_kernel", 36.0 * B * Z * Y * X)          # psi 12 + x 12 read, u 12 written
    if name == "df_velocity_loss3d_bwd":
        B, Z, Y, X = args[5:9]
        return ("velocity_loss3d_bwd_kernel+curl_adjoint", 60.0 * B * Z * Y * X)   # (u, x) -> du 36, du -> dpsi 24
    if name == "df_jacobian2d_fwd":
        B, Y, X = args[3:6]
        return ("jacobian2d_fwd_kernel", 28.0 * B * Y * X)
    return None


def wgrad_exec_ratio(B, D, H, W, cin, cout):
    """Executed / algorithmic multiply-adds of the weight-gradient form the LIBRARY picks for this call (df_conv_wgrad_form = the
    predicate of conv_wgrad.hip::wgrad_algo): Winograd-(x,y,z) 8/27, (x,y) 4/9 (2-D) | 12/27 (3-D, direct in z), x 2/3, direct 1."""
    from deep_fluids_amd import _lib
    kz = 3 if D > 1 else 1
    form = _lib.query("df_conv_wgrad_form", B, D, H, W, cin, cout, kz, 0)
    return {3: 8.0 / 27.0, 2: 4.0 / 9.0, 1: 2.0 / 3.0}.get(form, 1.0), form


def cpu_baseline(res, filters, budget_s, fullsize_parity=None):
    """The oracle's PyTorch-CPU restatement of the SAME train step, timed on this node's host cores at the FULL grid of the
    workload, batch 1 (a bounded sample: one warm-up step + as many timed steps as fit the budget, at least one).  A reported
    baseline, not the optimisation target.  ``fullsize_parity`` (a dict): the warm-up step's velocity field and loss -- the oracle's
    own, at the benchmarked grid -- are compared with one GPU train step on the same weights / inputs and the figures stored in it."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import df_oracle as orc
    import df_oracle_torch as ort
    try:
        cores = len(os.sched_getaffinity(0))          # cores this process may actually use (cgroup / affinity)
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))                     # beyond one socket the per-level convs only thrash
    torch.set_num_threads(cores)
    sres = list(res)
    rng = np.random.RandomState(123)
    oshape = sres + [3]
    p0 = orc.generator_init(rng, 3, oshape, filters)
    p = ort.to_torch(p0)
    opt = ort.new_opt(p)
    x, y = orc.synthetic_batch(rng, 1, sres)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    t0 = time.time()
    first = ort.train_step(yt, xt, p, opt, oshape, filters, True)   # warm-up (thread pools, oneDNN primitives, page faults)
    warm = time.time() - t0
    if fullsize_parity is not None:
        try:
            fullsize_parity.update(gpu_vs_oracle_fullsize(sres, filters, p0, x, y, first["u"].numpy(), first["loss"]))
        except Exception as e:
            fullsize_parity.update(error=repr(e)[:300])
    del first
    n, t0, el = 0, time.time(), 0.0
    while True:
        ort.train_step(yt, xt, p, opt, oshape, filters, True)
        n += 1
        el = time.time() - t0
        if el + el / n > budget_s - warm or n >= 50:
            break
    vox = float(np.prod(sres))
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    tail = None
    try:
        tail = cpu_stencil_tail(sres, cores)
    except Exception as e:
        tail = {"error": repr(e)[:200]}
    return {"value": vox * n / el, "unit": "voxels/s", "cores": cores, "kind": "port", "stencil_tail": tail,
            "sample": "PyTorch-CPU fp32 restatement of the reference graph (TF 1.15 unavailable): %d full train step(s) after one "
                      "warm-up step (%.1f s), batch 1, FULL grid %dx%dx%d, filters %d, %d threads, %.1f s timed" % (
                          n, warm, sres[0], sres[1], sres[2], filters, cores, el),
            "ms_per_step": el / n * 1e3, "cpu": model}


def cpu_stencil_tail(res, cores, batch=2):
    """SURVEY 8(d) CPU baseline part (i): the stencil / loss tail of the reference graph (jacobian3 on the ground truth, curl3 of psi,
    jacobian3 of u, two L1 means: 240 B/voxel of algorithmic traffic) on the host, (a) oracle/df_oracle.c with OpenMP on `cores`
    threads, (b) the oracle's NumPy restatement in the style of the reference's own jacobian_np3 (ops.py:344-374), single thread."""
    import ctypes
    import numpy as np
    import df_oracle as orc
    Z, Y, X = res
    rng = np.random.RandomState(5)
    psi = rng.uniform(-1, 1, (batch, Z, Y, X, 3)).astype(np.float32)
    x = rng.uniform(-1, 1, (batch, Z, Y, X, 3)).astype(np.float32)
    nv = float(batch * Z * Y * X)
    out = {"batch": batch, "algorithmic_bytes_per_voxel": 240}
    so = os.path.join(ROOT, "oracle", "libdf_oracle.so")
    if os.path.exists(so):
        h = ctypes.CDLL(so)
        h.dfo_set_threads(int(cores))           # through the OpenMP runtime itself: OMP_NUM_THREADS is read once, at ITS initialisation
        threads = int(h.dfo_max_threads())
        I64 = ctypes.c_int64
        u = np.empty_like(psi); ju = np.empty((batch, Z, Y, X, 9), np.float32); jx = np.empty_like(ju)
        l1 = ctypes.c_double(); jl1 = ctypes.c_double()
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        args = (ptr(psi), ptr(x), ptr(u), ptr(ju), ptr(jx), ctypes.byref(l1), ctypes.byref(jl1), I64(batch), I64(Z), I64(Y), I64(X))
        h.dfo_velocity_tail3d(*args)
        n, t0 = 0, time.time()
        while n < 3 or (time.time() - t0 < 1.0 and n < 50):
            h.dfo_velocity_tail3d(*args); n += 1
        el = (time.time() - t0) / n
        out["c_openmp"] = {"ms": el * 1e3, "voxels_per_s": nv / el, "GBs": 240.0 * nv / el / 1e9, "threads": threads, "l1": l1.value}
    t0 = time.time()
    jx_ = orc.jacobian3(x)[0]; u_ = orc.jacobian3(psi)[1]; ju_ = orc.jacobian3(u_)[0]
    l1n = orc.l1_mean(u_, x); orc.l1_mean(ju_, jx_)
    el = time.time() - t0
    out["numpy"] = {"ms": el * 1e3, "voxels_per_s": nv / el, "GBs": 240.0 * nv / el / 1e9, "threads": 1, "l1": float(l1n)}
    return out


def gpu_vs_oracle_fullsize(res, filters, p0, x, y, u_ref, loss_ref):
    """One default-dispatch GPU train step at the BENCHMARKED grid (batch 1) on the weights / inputs the CPU baseline's first step
    used: relative L1 of the velocity field (north-star tolerance 1e-4) and relative loss difference vs the PyTorch-CPU oracle."""
    import numpy as np
    import torch
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, default_config
    ops.reset_variables()
    cfg = default_config(is_3d=True, res_x=res[2], res_y=res[1], res_z=res[0], filters=filters, batch_size=1, num_samples=6600)
    tr = Trainer(cfg)
    tr.load_variables(p0)
    m = tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    u = m.G_.detach().cpu().numpy().astype(np.float64)
    loss = float(m.g_loss.detach())
    del m, tr
    ops.reset_variables()
    torch.cuda.empty_cache()
    ref = np.asarray(u_ref, np.float64)
    return {"velocity_rel_l1": float(np.abs(u - ref).sum() / np.abs(ref).sum()), "loss_rel": abs(loss - loss_ref) / abs(loss_ref),
            "loss_gpu": loss, "loss_oracle": float(loss_ref), "tolerance": 1e-4,
            "case": "one train step at the benchmarked grid %dx%dx%d, filters %d, batch 1, same weights / inputs on the GPU (default "
                    "dispatch) and the PyTorch-CPU fp32 oracle" % (res[0], res[1], res[2], filters)}


def l1_vs_oracle(filters, precision="fp32", is_3d=True):
    """Relative L1 of the velocity field vs the fp64 oracle on identical inputs/weights (reduced grid 16x24x16 | 32x24)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import df_oracle as orc
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, default_config
    ops.reset_variables()
    rng = np.random.RandomState(123)
    spatial = (16, 24, 16) if is_3d else (32, 24)
    oshape = list(spatial) + [3 if is_3d else 1]
    p = orc.generator_init(rng, 3, oshape, filters)
    x, y = orc.synthetic_batch(rng, 1, spatial)
    cfg = default_config(is_3d=is_3d, res_x=spatial[-1], res_y=spatial[-2], res_z=spatial[0] if is_3d else 1, filters=filters,
                         batch_size=1, num_samples=100)
    with ops.options(conv_precision=precision):
        tr = Trainer(cfg)
        tr.load_variables(p)
        u = tr.generate(torch.from_numpy(y).cuda()).cpu().numpy().astype(np.float64)
    psi = orc.generator_fwd(y.astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()}, oshape, filters)
    ref = orc.curl3(psi) if is_3d else orc.curl(psi)
    ops.reset_variables()
    return float(np.abs(u - ref).sum() / np.abs(ref).sum())


def l1_vs_oracle_ae(filters, z_num=16, spatial=(16, 16, 16)):
    """cfg5's parity figure: relative L1 of the auto-encoder's reconstructed velocity field (AE3 forward + curl3) vs the fp64 oracle on
    identical inputs / weights at a reduced grid."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import df_oracle as orc
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import AETrainer, default_config
    ops.reset_variables()
    rng = np.random.RandomState(123)
    xshape = list(spatial) + [3]
    p = orc.ae_init(rng, xshape, filters, z_num)
    x, _ = orc.synthetic_batch(rng, 1, spatial)
    cfg = default_config(is_3d=True, res_x=spatial[2], res_y=spatial[1], res_z=spatial[0], filters=filters, batch_size=1, num_samples=100,
                         z_num=z_num, p_num=2)
    tr = AETrainer(cfg)
    tr.load_variables(p)
    y = torch.zeros((1, 2, 3), device="cuda")
    with torch.no_grad():
        u = tr.build_model(torch.from_numpy(x).cuda(), y).G_.cpu().numpy().astype(np.float64)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    z = orc.encoder_fwd(x.astype(np.float64), p64, filters, z_num, "AE/enc", 3, 0)
    psi = orc.generator_fwd(z, p64, xshape, filters, "AE/dec", 4, 0)
    ref = orc.curl3(psi)
    ops.reset_variables()
    return float(np.abs(u - ref).sum() / np.abs(ref).sum())


def live_pmc(timeout_s=240.0):
    """`roofline.traffic` measured IN THIS RUN on this box: two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE: one counter
    set per pass, no other trace domains -- MI355X_MICROARCH.md) over tools/pmc_target.py, the roofline kernels at the benchmark's shapes, as a
    child process; summarised by tools/pmc_summary.py (FETCH_SIZE doubled per the guide's gfx950 correction).  Returns the `kernels` table or
    None (rocprofv3 missing, a pass failed or timed out): the caller then falls back to the committed offline table and says so."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="dfpmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.time()
    try:
        dirs = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, c)
            left = timeout_s - (time.time() - t0)
            if left < 20:
                return None
            r = subprocess.run([exe, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "tools", "pmc_target.py")], cwd="/tmp", env=env, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=left)
            if r.returncode != 0:
                return None
            dirs[c] = d
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_summary
        tab = pmc_summary.summarize(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"])["kernels"]
        return tab or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_of(ks, prefix, pmc, with_traffic, pmc_source=None):
    """Roofline object of the dominant instance (largest total time) of a kernel family.
    MFMA-bound families: `achieved` = multiply-add flops the kernel EXECUTES on the matrix pipe / time (<= peak); the
    convolution's algorithmic (direct-form) rate is reported beside it as `algorithmic_tflops` (it exceeds the peak when a Winograd
    form removes multiplies) with `algorithmic_speedup` = algorithmic / executed flops."""
    sel = {k: v for k, v in ks.items() if k.startswith(prefix)}
    if not sel:
        return None
    k = max(sel, key=lambda q: sel[q]["seconds"])
    v = sel[k]
    hbm = prefix.startswith("jacobian") or prefix.startswith("velocity_loss")
    out = {"kernel": k, "bound": "hbm" if hbm else "mfma", "launches": v["launches"], "avg_launch_ms": v["seconds"] / v["launches"] * 1e3,
           "work_per_launch": v["work"] / v["launches"]}
    if hbm:
        ach = v["work"] / v["seconds"] / 1e9
        out.update(achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS)
    else:
        alg = v["work"] / v["seconds"] / 1e12
        ratio = EXEC_RATIO.get(prefix)
        if ratio is None:                                        # weight gradient: depends on the form conv_wgrad.hip picks
            f = k.split(" ")
            dims = [int(t) for t in f[1].split("x")]
            cc = [int(t) for t in f[2][1:].split("x")]
            ratio, form = wgrad_exec_ratio(int(f[3][1:]), dims[0], dims[1], dims[2], cc[0], cc[1])
            out["wgrad_form"] = {3: "winograd-xyz", 2: "winograd-xy", 1: "winograd-x", 0: "direct"}.get(form, str(form))
        ach = alg * ratio
        # algorithmic HBM bytes of one launch: every input and output element once (weights: < 0.1 %)
        abytes = None
        try:
            f = k.split(" ")
            cc = [int(t) for t in f[-2 if f[-1].startswith("B") else -1][1:].replace("->", "x").split("x")]
            dims = [int(t) for t in [q for q in f if "x" in q and q[0].isdigit()][-1].split("x")]
            taps = 27 if len(dims) == 3 and dims[0] > 1 else 9
            vox = out["work_per_launch"] / (2.0 * taps * cc[0] * cc[1])
            abytes = vox * (cc[0] + cc[1]) * 4.0
            if "up-sampling-aware forward" in k:      # reads the COARSE tensor (1/8 of the voxels), writes the fine one
                abytes = vox * (cc[0] / 8.0 + cc[1]) * 4.0
            elif "pooled adjoint" in k:               # reads the fine gradient, accumulates into the coarse tensor (read + write)
                abytes = vox * (cc[0] + 2.0 * cc[1] / 8.0) * 4.0
        except Exception:
            pass
        out.update(achieved=ach, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_FP32_MFMA_TFLOPS,
                   frac_definition="EXECUTED multiply-add flops on the fp32 matrix pipe / time / peak (this kernel executes %.3f of the "
                                   "direct-form flops the roofline numerator of SURVEY 8(d) counts: %s)" % (
                                       ratio, "Winograd F(2,3) x F(2,3) x F(4,3)" if abs(ratio - 6.0 / 27.0) < 1e-9 else "Winograd F(2,3) x F(4,3)" if abs(ratio - 3.0 / 9.0) < 1e-9 else "Winograd F(2,3) per axis" if ratio < 1 else "direct form"),
                   executed_over_algorithmic=ratio, algorithmic_bytes=abytes,
                   algorithmic_tflops=alg, algorithmic_speedup=1.0 / ratio,
                   note="achieved = multiply-add flops EXECUTED on the fp32 matrix pipe / time; algorithmic_tflops = direct-convolution "
                        "flops / time (the Winograd forms execute 1/algorithmic_speedup of them)")
    # HBM bytes per launch from rocprofv3 PMC passes (2 x FETCH_SIZE per the gfx950 correction + WRITE_SIZE), collected offline
    # on the same kernel at the default shape (B = 16, 64x96x64, F = 128)
    fam = prefix.split("<")[0]
    out["traffic"] = pmc.get(fam, {}).get("traffic_bytes") if with_traffic else None
    if out["traffic"] and out.get("algorithmic_bytes"):
        out["traffic_over_algorithmic"] = out["traffic"] / out["algorithmic_bytes"]
    out["traffic_source"] = (pmc_source or "profiles/pmc_latest.json (offline rocprofv3 --pmc passes, not measured in this run)") if out["traffic"] else None
    return out


def stencil_rooflines(B, Z, Y, X):
    """Standalone launches of the stencil kernels at the workload's shape, HIP-event timed on the launch stream, over ROTATING
    buffers (8 inputs x 75 MB: the 256 MB Infinity Cache cannot hold the working set, so this is the cold-HBM figure).
    Algorithmic bytes per voxel: jacobian3 60, curl3 24, their adjoints 48 / 24, fused tail 36 fwd / 60 bwd."""
    import torch
    from deep_fluids_amd import _lib
    from deep_fluids_amd.ops import _ptr, _stream
    call, query = _lib.call, _lib.query
    s = _stream()
    nv = B * Z * Y * X
    xs = [torch.rand((B, Z, Y, X, 3), device="cuda") * 2 - 1 for _ in range(8)]
    js = [torch.empty((B, Z, Y, X, 9), device="cuda") for _ in range(3)]
    cs = [torch.empty((B, Z, Y, X, 3), device="cuda") for _ in range(4)]
    nb = query("df_velocity_loss3d_workspace_bytes", B, Z, Y, X)
    ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device="cuda")
    l1 = torch.empty((), device="cuda"); jl1 = torch.empty((), device="cuda")
    k = [0]

    def nxt():
        k[0] += 1
        return k[0]
    cases = {
        "jacobian3d_fwd_kernel<j,c>": (60.0, lambda i: call("df_jacobian3d_fwd", _ptr(xs[i % 8]), _ptr(js[i % 3]), _ptr(cs[i % 4]), B, Z, Y, X, s)),
        "jacobian3d_fwd_kernel<c> (curl3)": (24.0, lambda i: call("df_jacobian3d_fwd", _ptr(xs[i % 8]), None, _ptr(cs[i % 4]), B, Z, Y, X, s)),
        "jacobian3d_bwd_kernel<j>": (48.0, lambda i: call("df_jacobian3d_bwd", _ptr(js[i % 3]), None, _ptr(cs[i % 4]), B, Z, Y, X, s)),
        "jacobian3d_bwd_kernel<c> (curl3 adjoint)": (24.0, lambda i: call("df_jacobian3d_bwd", None, _ptr(xs[i % 8]), _ptr(cs[i % 4]), B, Z, Y, X, s)),
        "velocity_loss3d_fwd_kernel": (36.0, lambda i: call("df_velocity_loss3d_fwd", _ptr(xs[i % 8]), _ptr(xs[(i + 3) % 8]), _ptr(cs[i % 4]),
                                                            _ptr(l1), _ptr(jl1), B, Z, Y, X, _ptr(ws), nb, s)),
        "velocity_loss3d_bwd_kernel+curl_adjoint": (60.0, lambda i: call("df_velocity_loss3d_bwd", _ptr(xs[i % 8]), _ptr(xs[(i + 3) % 8]), _ptr(l1),
                                                                         _ptr(jl1), _ptr(cs[i % 4]), B, Z, Y, X, _ptr(ws), nb, s)),
    }
    for j in js:
        j.uniform_(-1, 1)
    l1.fill_(1.0); jl1.fill_(1.0)
    # the copy rate this box reaches on the same rotating buffers (device-to-device copy of one 75 MB field: 12 B/voxel read + 12 written):
    # the practical ceiling of any streaming kernel here, reported beside the 8 TB/s spec
    for _ in range(3):
        cs[nxt() % 4].copy_(xs[nxt() % 8])
    c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(20):
        i = nxt()
        cs[i % 4].copy_(xs[i % 8])
    c1.record()
    torch.cuda.synchronize()
    copy_gbs = 24.0 * nv / (c0.elapsed_time(c1) * 1e-3 / 20) / 1e9
    out = {"copy_rate": {"achieved": copy_gbs, "unit": "GB/s", "frac": copy_gbs / PEAK_HBM_GBS,
                         "note": "torch device-to-device copy of one [B,Z,Y,X,3] field over the same rotating buffers"}}
    for name, (bpv, fn) in cases.items():
        for _ in range(3):
            fn(nxt())
        # ONE figure per kernel (round 5): the average DURATION of a launch, each launch bracketed by its own pair of HIP events on the launch
        # stream -- what rocprofv3's kernel trace reports (profiles/r0*_bench_kernel_by_grid.md) and what `achieved` is defined on.  The
        # back-to-back rate of 20 launches between ONE event pair is kept beside it (`back_to_back_us`): consecutive launches overlap their
        # ramp-down / ramp-up, which made rounds 2-4 print a figure 4-6 % above the per-kernel duration.
        n = 20
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            i = nxt()
            a.record()
            fn(i)
            b.record()
        torch.cuda.synchronize()
        t = sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / n
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn(nxt())
        e1.record()
        torch.cuda.synchronize()
        tb = e0.elapsed_time(e1) * 1e-3 / n
        ach = bpv * nv / t / 1e9
        out[name] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                     "frac_of_copy_rate": ach / copy_gbs, "avg_launch_us": t * 1e6, "back_to_back_us": tb * 1e6,
                     "algorithmic_bytes_per_voxel": bpv, "traffic": None}
    return out


def timed_steps(tr, x, y, warm, n, rooflines=None, families=("wino43_kernel", "wino3d_kernel", "wgrad_kernel", "conv_mfma_kernel")):
    """Mean wall time per step; ``rooflines`` (a dict) additionally receives the live HIP-event rooflines of the dominant conv /
    weight-gradient kernel families of these steps."""
    import torch
    from deep_fluids_amd import _lib
    for _ in range(warm):
        tr.train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        m = tr.train_step(x, y)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    if rooflines is not None:                 # a separate, untimed pass: the per-launch event records stay out of ms_per_step
        timer = _lib.KernelTimer(select_kernel)
        _lib.TIMER = timer
        try:
            for _ in range(2):
                m = tr.train_step(x, y)
        finally:
            _lib.TIMER = None
        ks = timer.summary()
        for fam in families:
            r = roofline_of(ks, fam, {}, False)
            if r is not None:
                rooflines["roofline_" + fam.split("_")[0]] = r
        if "roofline_wino43" in rooflines:      # the plain 3-D Winograd convs run on the F(2,2,4) family: it is "the" wino3d roofline of the extras
            rooflines["roofline_wino3d"] = rooflines["roofline_wino43"]
    return el, m


def extras(out, a, cfg, x, y, vox_per_step, pmc):
    """Appended at N = 1, clearly separate from `value`: the other BASELINE shapes / the opt-in bf16x3 mode.  An extra must never
    take the metric line down with it.  Precision modes are entered through ``ops.options`` (restored whatever happens)."""
    import torch
    from deep_fluids_amd import _lib, ops
    from deep_fluids_amd.trainer import Trainer, AETrainer, default_config

    def guarded(key, fn):
        try:
            out[key] = fn()
        except Exception as e:
            out[key] = {"error": repr(e)[:300]}
        _lib.TIMER = None
        ops.reset_variables()
        torch.cuda.empty_cache()

    def alt_bf16x3():
        rel_alt = l1_vs_oracle(a.filters, "bf16x3")
        with ops.options(conv_precision="bf16x3"):
            ops.reset_variables()
            el, _ = timed_steps(Trainer(cfg), x, y, 2, 3)
        return {"ms_per_step": el * 1e3, "value": vox_per_step / el, "unit": "voxels/s", "l1_vs_ref": rel_alt,
                "note": "opt-in precision mode, not the BASELINE cfg3 dtype: conv operands split into bf16 hi/lo words, 3 bf16 MFMAs per "
                        "product, fp32 accumulation"}

    def two_d():
        res = {}
        x2, y2 = make_inputs(64, [128, 96], 1, ops)
        cfg2 = default_config(is_3d=False, res_x=96, res_y=128, filters=a.filters, batch_size=64, num_samples=21000)
        for prec in ("fp32", "bf16x3"):
            rel = l1_vs_oracle(a.filters, prec, is_3d=False)
            with ops.options(conv_precision=prec):
                ops.reset_variables()
                tr = Trainer(cfg2)
                rf = {} if prec == "fp32" else None
                el, _ = timed_steps(tr, x2, y2, 3, 10, rf, families=("wino2d43_kernel", "wino2d_kernel", "wgrad_kernel", "jacobian2d_fwd_kernel"))
            r = {"ms_per_step": el * 1e3, "value": 64 * 128 * 96 / el, "unit": "pixels/s", "batch": 64, "l1_vs_ref": rel,
                 "conv_tflops_reference_equivalent": 3.71e12 / el / 1e12}
            if rf:
                # the F(2,3) x F(4,3) kernel (forward convs and dgrads since the 16-byte epilogue); ops.WINO2D_FAMILY "auto" / "f22": the F(2,3)^2 kernel
                r["roofline"] = rf.get("roofline_wino2d43") or rf.get("roofline_wino2d")
                r["roofline_fwd_f24"] = rf.get("roofline_wino2d43")
                r["roofline_wgrad"] = rf.get("roofline_wgrad")
                r["roofline_stencil"] = rf.get("roofline_jacobian2d")
            res[prec] = r
            del tr
        f = res["fp32"]
        f["bf16x3_mode"] = {k: res["bf16x3"][k] for k in ("ms_per_step", "value", "l1_vs_ref")}
        f["dtype"] = "f32"
        f["note"] = ("BASELINE cfg2's shape: 2-D 128x96 train step (GeneratorBE filters=128, batch 64); Winograd F(2x2,3x3) forward/dgrad + "
                     "Winograd-(x,y) weight gradient at the top levels.  cfg2 names bf16: plain bf16 operands miss the 1e-4 velocity "
                     "tolerance (tests/test_gpu_precision.py); the split-operand bf16x3 mode is kept as a tested option, not offered as the "
                     "reduced-precision path (DESIGN section 6)")
        return f

    def cfg4_slice():
        B4 = 4
        cfg4 = default_config(is_3d=True, res_x=112, res_y=160, res_z=112, filters=a.filters, batch_size=B4, num_samples=20000)
        x4, y4 = make_inputs(B4, [112, 160, 112], 7, ops)
        res = {}
        for prec in ("fp32", "bf16x3"):
            with ops.options(conv_precision=prec):
                ops.reset_variables()
                tr = Trainer(cfg4)
                rf = {} if prec == "fp32" else None
                el, m = timed_steps(tr, x4, y4, 2, 3, rf)
            res[prec] = {"ms_per_step": el * 1e3, "value": B4 * 112 * 160 * 112 / el}
            if rf:
                res[prec].update(rf)
            n_params = tr.n_params
            del tr, m
            torch.cuda.empty_cache()
        return {"grid": [112, 160, 112], "batch_per_gpu": B4, "params": n_params, "unit": "voxels/s", "dtype": "f32",
                "ms_per_step": res["fp32"]["ms_per_step"], "value": res["fp32"]["value"], "bf16x3_mode": res["bf16x3"],
                "roofline": res["fp32"].get("roofline_wino3d"), "roofline_wgrad": res["fp32"].get("roofline_wgrad"),
                "note": "BASELINE cfg4's grid and per-GPU batch (32 / 8 GPUs) on ONE GPU: the per-rank work of the 8-GPU batch-DP job; "
                        "parity: tests/test_gpu_fullsize.py::test_cfg4_*"}

    def ae_cfg5():
        import numpy as np
        B5, R = 4, 128
        cfg5 = default_config(is_3d=True, res_x=R, res_y=R, res_z=R, filters=64, batch_size=B5, num_samples=5000, z_num=16, p_num=2)
        tr = AETrainer(cfg5)
        x5, _ = make_inputs(B5, [R, R, R], 1, ops)
        y5 = torch.from_numpy(np.random.RandomState(2).uniform(-1, 1, (B5, 2, 10)).astype(np.float32)).cuda()
        rf = {}
        el, m = timed_steps(tr, x5, y5, 2, 3, rf)
        return {"grid": [R, R, R], "batch_per_gpu": B5, "filters": 64, "z_num": 16, "params": tr.n_params, "ms_per_step": el * 1e3,
                "roofline": rf.get("roofline_wino3d"), "roofline_wgrad": rf.get("roofline_wgrad"), "roofline_conv": rf.get("roofline_conv"),
                "value": B5 * R ** 3 / el, "unit": "voxels/s", "dtype": "f32",
                "note": "BASELINE cfg5's shape (AE3 encoder + decoder train step, 128^3, F = 64), fp32; parity at this shape: "
                        "tests/test_gpu_fullsize.py::test_cfg5_*"}

    def ref_grids():
        """The reference's OWN documented workloads (run.bat) at their native grids and batch sizes: ms/step of the full train step
        (parity at these grids: tests/test_gpu_refgrids.py)."""
        import numpy as np
        cases = [  # key, run.bat line, (Z,)Y,X, batch, filters, use_curl, kind
            ("smoke3_vel_buo_32x64x112", 21, [32, 64, 112], 4, 128, True, "de"),
            ("smoke3_obs_buo_64x96x64", 27, [64, 96, 64], 3, 128, True, "de"),
            ("liquid3_d_r_96x48x96", 37, [96, 48, 96], 3, 128, False, "de"),
            ("liquid3_vis_48x72x96", 42, [48, 72, 96], 3, 128, False, "de"),
            ("liquid_pos_size_2d_64x128", 32, [64, 128], 8, 128, False, "de"),
            ("smoke_pos_size_2d_128x96", 13, [128, 96], 8, 128, True, "de"),
            ("ae_smoke_mov_2d_128x96", 56, [128, 96], 8, 64, True, "ae"),
            ("ae3_smoke3_mov_48x72x48", 73, [48, 72, 48], 4, 64, True, "ae"),
            ("dg_smoke_pos_size_2d_128x96", 13, [128, 96], 8, 128, True, "dg"),      # the README's 2-D training command: --arch=dg (generator + PatchGAN)
        ]
        res = {}
        for key, line, grid, B, F, use_curl, kind in cases:
            ops.reset_variables()
            is3 = len(grid) == 3
            kw = dict(is_3d=is3, res_x=grid[-1], res_y=grid[-2], res_z=grid[0] if is3 else 1, filters=F, batch_size=B, num_samples=6000,
                      use_curl=use_curl)
            xr, yr = make_inputs(B, grid, 3, ops)
            if kind == "ae":
                tr = AETrainer(default_config(z_num=16, p_num=2 if is3 else 1, **kw))
                yr = torch.from_numpy(np.random.RandomState(4).uniform(-1, 1, (B, 2 if is3 else 1, 10)).astype(np.float32)).cuda()
            elif kind == "dg":
                from deep_fluids_amd.trainer import GANTrainer
                tr = GANTrainer(default_config(arch="dg", **kw))
            else:
                tr = Trainer(default_config(**kw))
            el, m = timed_steps(tr, xr, yr, 2, 5) if is3 else timed_steps(tr, xr, yr, 3, 20)      # (ms-scale 2-D steps: 5 would mostly time the host's lead-in)
            n = B
            for g in grid:
                n *= g
            res[key] = {"run_bat_line": line, "grid": grid, "batch": B, "filters": F, "use_curl": use_curl, "arch": kind, "ms_per_step": el * 1e3,
                        "value": n / el, "unit": "voxels/s" if is3 else "pixels/s", "params": tr.n_params}
            del tr, m, xr, yr
            torch.cuda.empty_cache()
        return res

    guarded("alt_bf16x3_mode", alt_bf16x3)
    guarded("extra_ref_grids", ref_grids)
    guarded("extra_2d_128x96", two_d)
    guarded("extra_cfg4_slice", cfg4_slice)
    guarded("extra_ae_cfg5", ae_cfg5)


class Watchdog(object):
    """A hung RCCL bring-up must not hang the lease: if `stage` is not disarmed within `seconds`, rank 0's line is a diagnostic JSON
    object (`rccl_ranks: 0`, the stage that hung) and every rank exits non-zero (os._exit: the main thread may sit inside a collective)."""

    def __init__(self, rank, world):
        import threading
        self.rank, self.world = rank, world
        self._lock = threading.Lock()
        self._deadline, self._stage, self._soft, self._fallback = None, None, False, None
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, stage, seconds, soft=False, fallback=None):
        """``soft`` (the optional extra legs that follow the metric's own leg): on expiry rank 0 prints ``fallback`` -- the contract line of the
        leg that DID complete, with the stage that hung under `extra_legs_error` -- and every rank exits 0: a hang in an appended leg must not
        cost the job its headline number."""
        with self._lock:
            self._stage, self._deadline, self._soft, self._fallback = stage, time.time() + seconds, soft, fallback

    def disarm(self):
        with self._lock:
            self._stage, self._deadline, self._soft, self._fallback = None, None, False, None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                stage, dl, soft, fallback = self._stage, self._deadline, self._soft, self._fallback
            if dl is not None and time.time() > dl and soft:
                err = "watchdog: stage %r did not complete in time" % stage
                if self.rank == 0 and fallback is not None:
                    print(json.dumps(dict(fallback, extra_legs_error=err, truncated=True)), flush=True)
                    os._exit(0)
                # no completed leg to fall back on (rank 0), or a peer of the hung leg: say so; only rank 0 WITHOUT a line to print fails the
                # job (exit 4 = soft expiry with nothing measured) -- a peer exiting non-zero would void rank 0's valid headline line
                print(json.dumps({"error": err, "rank": self.rank, "soft": True}), file=sys.stdout if self.rank == 0 else sys.stderr, flush=True)
                os._exit(4 if self.rank == 0 else 0)
            if dl is not None and time.time() > dl:
                msg = {"metric": "velocity-field voxels/sec (3D train step), whole job", "value": None, "unit": "voxels/s",
                       "n_gpus": self.world, "rccl_ranks": 0, "error": "watchdog: stage %r did not complete in time" % stage,
                       "rank": self.rank, "env": {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "LOCAL_RANK",
                                                                               "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")}}
                stream = sys.stdout if self.rank == 0 else sys.stderr
                print(json.dumps(msg), file=stream, flush=True)
                os._exit(3)


def run_leg(tr, sync, x, y, warmup, steps, world, dist, torch, timer=None):
    """warmup + `steps` timed train steps bracketed by barrier + device sync on both sides; MAX over ranks; all-reduce timing of the
    timed steps only.  Returns (elapsed seconds, per-step HIP-event times in ms, last step's graph, all-reduce timing)."""
    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        tr.train_step(x, y)
    if sync is not None:
        sync.timing(reset=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    from deep_fluids_amd import _lib
    sync_all()
    _lib.TIMER = timer            # live HIP-event rooflines of exactly the timed steps (the contract's `roofline.achieved`)
    t0 = time.perf_counter()
    last = None
    try:
        ev[0].record()
        for i in range(steps):
            last = tr.train_step(x, y)
            ev[i + 1].record()
        sync_all()
    finally:
        _lib.TIMER = None
    elapsed = time.perf_counter() - t0
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        if dist.get_backend() == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    comm = sync.timing() if sync is not None else None
    return elapsed, step_ms, last, comm


def compact(out):
    """The printed line: headline objects first, bulky diagnostics (per-kernel table, dispatch log, standalone stencil table, the
    extras' nested rooflines, long notes) only in the sidecar file."""
    def slim(r, keep=("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_of_copy_rate", "traffic", "traffic_source", "avg_launch_ms", "avg_launch_us",
                      "launches", "algorithmic_tflops", "algorithmic_speedup", "wgrad_form", "algorithmic_bytes_per_voxel", "frac_definition",
                      "executed_over_algorithmic", "algorithmic_bytes", "traffic_over_algorithmic")):
        return None if not isinstance(r, dict) else ({k: r[k] for k in keep if k in r} if "error" not in r else r)
    head = ["metric", "value", "unit", "per_gpu", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"]
    line = {k: out[k] for k in head}
    for k in ("roofline", "roofline_stencil", "roofline_wgrad", "roofline_tail_fwd", "roofline_tail_bwd"):
        line[k] = slim(out.get(k))
    for k in ("roofline_up27", "roofline_pool27"):          # fraction + launch time only
        if isinstance(out.get(k), dict):
            line[k] = {q: out[k][q] for q in ("frac", "avg_launch_ms") if q in out[k]}
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = None if not isinstance(cb, dict) else {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "ms_per_step", "cpu", "error") if k in cb}
    line["l1_vs_ref"] = out.get("l1_vs_ref")
    line["l1_vs_ref_fullsize"] = out.get("l1_vs_ref_fullsize")
    for k in ("step_ms", "value_median", "rccl_ranks", "counted_ranks", "dist_backend", "distinct_devices", "allreduce", "other_scaling_leg",
              "cross_rank", "loss"):
        line[k] = out.get(k)
    for k in ("extra_leg_cfg2", "extra_leg_cfg4"):          # N > 1: the 2-D workload and cfg4's grid as data-parallel legs of the same job
        e = out.get(k)
        if isinstance(e, dict) and "error" not in e:
            e = {q: e[q] for q in ("config", "grid", "scaling", "global_batch", "batch_per_gpu", "steps", "ms_per_step", "value", "unit", "allreduce",
                                   "loss", "other_scaling_leg", "cross_rank") if q in e}
            if isinstance(e.get("cross_rank"), dict):
                e["cross_rank"] = {q: e["cross_rank"][q] for q in ("ok", "grad_identical_on_all_ranks", "loss_rel_diff") if q in e["cross_rank"]}
        if e is not None:
            line[k] = e
    for k in ("alt_bf16x3_mode", "extra_2d_128x96", "extra_cfg4_slice", "extra_ae_cfg5"):
        e = out.get(k)
        if isinstance(e, dict) and "error" not in e:
            c = {q: e[q] for q in ("ms_per_step", "value", "unit", "l1_vs_ref", "batch", "batch_per_gpu", "grid") if q in e}
            if isinstance(e.get("bf16x3_mode"), dict):
                c["bf16x3_ms_per_step"] = e["bf16x3_mode"].get("ms_per_step")
            for q in ("roofline", "roofline_wgrad", "roofline_fwd_f24"):
                if isinstance(e.get(q), dict):
                    c[q + "_frac"] = e[q].get("frac")
            e = c
        line[k] = e
    rg = out.get("extra_ref_grids")      # the reference's own run.bat workloads: ms/step only
    if isinstance(rg, dict):
        line["extra_ref_grids_ms"] = rg if "error" in rg else {k: round(v["ms_per_step"], 2) for k, v in rg.items()}
    line["sidecar"] = out.get("sidecar")
    return line


def workload_job(name, res, filters, per_gpu, global_batch, world, rank, profile=True):
    """Trainer + resident inputs of one BASELINE workload on this rank: same seed on every rank -> identical initial variables
    (enable_data_parallel additionally broadcasts rank 0's state); inputs from RandomState(123 + rank).  Returns (trainer, GradSync | None,
    make(batch) -> (x, y))."""
    import numpy as np
    import torch
    from deep_fluids_amd import ops
    from deep_fluids_amd.trainer import Trainer, AETrainer, default_config
    wl = WORKLOADS[name]
    is_3d = len(res) == 3
    kw = dict(is_3d=is_3d, res_x=res[-1], res_y=res[-2], res_z=res[0] if is_3d else 1, filters=filters, batch_size=global_batch,
              num_samples=wl["num_samples"], random_seed=123)
    ops.reset_variables()
    if wl["kind"] == "ae":
        tr = AETrainer(default_config(z_num=16, p_num=2, **kw))
    else:
        tr = Trainer(default_config(**kw))
    sync = tr.enable_data_parallel(profile=profile) if world > 1 else None

    def make(batch):
        x, y = make_inputs(batch, res, 123 + rank, ops)
        if wl["kind"] == "ae":      # supervised latent part: y[:, :, -1] (trainer3.py:268); [B, dof = 2, frames]
            y = torch.from_numpy(np.random.RandomState(1000 + rank).uniform(-1, 1, (batch, 2, 10)).astype(np.float32)).cuda()
        return x, y
    return tr, sync, make


def cross_rank_check(tr, x, y, world, rank, dist, torch, max_samples=2):
    """One extra, untimed forward + backward + exchange (no optimizer step) on <= `max_samples` samples per rank:
    (1) the reduced flat gradient must be BIT-identical on every rank after GradSync.finish() (exact integer checksum of its bit
        pattern + float64 sum, MIN == MAX over the ranks);
    (2) the mean of the ranks' shard losses must equal the loss ONE process computes on the gathered global batch (rank 0, forward only,
        same parameters) to 1e-6 relative -- reduce_mean over the global batch == mean of equal-sized shard means."""
    from deep_fluids_amd.dist import all_equal_across_ranks
    n = min(int(x.shape[0]), max_samples)
    xs, ys = x[:n].contiguous(), y[:n].contiguous()
    m, gscale = tr.forward_backward(xs, ys)
    g = tr.flat_g
    bits = int(g.view(torch.int32).to(torch.int64).sum().item())
    gsum = float(g.double().sum().item())
    same = all_equal_across_ranks([float(bits & 0xFFFFFFFF), float((bits >> 32) & 0xFFFFFFFF), gsum])
    host = dist.get_backend() == "gloo"
    lt = m.g_loss.detach().double().reshape(1)
    lt = lt.cpu() if host else lt
    dist.all_reduce(lt)
    shard_mean = float(lt.item()) / world

    def gather(t):
        src = t.cpu() if host else t
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        return torch.cat(parts, 0).to(t.device)
    xg, yg = gather(xs), gather(ys)
    out = None
    if rank == 0:
        with torch.no_grad():
            mg = tr.build_model(xg, yg)
        glob = float(mg.g_loss.detach().double().item())
        rel = abs(shard_mean - glob) / max(abs(glob), 1e-30)
        out = {"grad_identical_on_all_ranks": bool(same), "grad_sum": gsum, "grad_scale": gscale, "samples_per_rank": n,
               "loss_shard_mean": shard_mean, "loss_global_batch_one_process": glob, "loss_rel_diff": rel, "tolerance": 1e-6,
               "ok": bool(same and rel <= 1e-6)}
        del mg
    del m, xg, yg
    return out


def dp_leg(name, a, world, rank, dist, torch, steps, warmup, scaling="weak", res=None, batch=None, filters=None, timer=None, other=True):
    """One data-parallel leg of workload `name`: warm-up + `steps` timed steps (barrier + device sync on both sides, MAX over ranks),
    all-reduce timing, optionally the short leg in the other scaling mode, and the cross-rank check.  Returns (record, state) where
    state = (trainer, x, y, last graph, per-step HIP-event times) for the caller's own diagnostics."""
    wl = WORKLOADS[name]
    res = list(res or wl["res"]); batch = batch or wl["batch"]; filters = filters or wl["filters"]
    if scaling == "strong":
        if batch % world:
            raise SystemExit("--scaling strong: the global batch %d is not divisible by %d GPUs" % (batch, world))
        per_gpu, global_batch = batch // world, batch
    else:
        per_gpu, global_batch = batch, batch * world
    tr, sync, make = workload_job(name, res, filters, per_gpu, global_batch, world, rank)
    x, y = make(batch if world > 1 else per_gpu)          # N > 1: enough samples for either leg
    xm, ym = x[:per_gpu].contiguous(), y[:per_gpu].contiguous()
    units = 1
    for r in res:
        units *= r
    for _ in range(min(warmup, 1)):
        tr.train_step(xm, ym)
    torch.cuda.synchronize()
    elapsed, step_ms, last, comm = run_leg(tr, sync, xm, ym, max(warmup - 1, 0), steps, world, dist, torch, timer)
    loss = float(last.g_loss.detach())
    assert loss == loss, "Model diverged with loss = NaN"        # trainer.py:275
    rec = {"config": name, "workload": "%s fp32, %s filters=%d num_conv=4, per-GPU batch %d, full train step (%s)" % (
               wl["name"].format(grid="x".join(str(r) for r in res)), wl["model"], filters, per_gpu, wl["step"]),
           "grid": res, "scaling": scaling, "global_batch": global_batch, "batch_per_gpu": per_gpu, "params": tr.n_params, "steps": steps,
           "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "value": global_batch * units * steps / elapsed, "unit": wl["unit"],
           "units_per_step": global_batch * units, "allreduce": comm, "loss": loss, "other_scaling_leg": None, "cross_rank": None}
    if world > 1 and other:
        o_mode = "strong" if scaling == "weak" else "weak"
        o_per = (batch // world) if o_mode == "strong" else batch
        if o_per >= 1 and (o_mode == "weak" or batch % world == 0):
            xo, yo = x[:o_per].contiguous(), y[:o_per].contiguous()
            o_el, o_ms, o_last, o_comm = run_leg(tr, sync, xo, yo, 2, a.other_steps, world, dist, torch)
            o_glob = o_per * world
            rec["other_scaling_leg"] = {"scaling": o_mode, "global_batch": o_glob, "batch_per_gpu": o_per, "steps": a.other_steps, "warmup": 2,
                                        "ms_per_step": o_el / a.other_steps * 1e3, "value": o_glob * units * a.other_steps / o_el,
                                        "unit": wl["unit"], "allreduce": o_comm, "loss": float(o_last.g_loss.detach())}
            del o_last, xo, yo
    if world > 1 and not a.no_cross_rank_check:
        rec["cross_rank"] = cross_rank_check(tr, xm, ym, world, rank, dist, torch)
    return rec, (tr, xm, ym, last, step_ms, elapsed)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    import torch
    import torch.distributed as dist
    from deep_fluids_amd import _lib, ops
    from deep_fluids_amd.dist import init_from_env, verify_world

    env_rank, env_world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dog = Watchdog(env_rank, env_world) if env_world > 1 else None
    if dog:
        dog.arm("torch.distributed rendezvous (init_process_group)", a.init_timeout)
    rank, local_rank, world = init_from_env()
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    # what the job REALLY runs on, asked before anything is timed: ranks counted by an on-device all-reduce of ones over the production
    # backend (for RCCL this is also the communicator creation), physical devices gathered
    if dog:
        dog.arm("RCCL communicator creation + first all-reduce (verify_world)", a.init_timeout)
    wv = verify_world()
    if dog:
        dog.disarm()

    wl = WORKLOADS[a.config]
    is_3d = len(a.res) == 3
    with ops.options(conv_precision=a.precision):
        rel_l1 = None
        if rank == 0:
            rel_l1 = l1_vs_oracle_ae(a.filters) if wl["kind"] == "ae" else l1_vs_oracle(a.filters, a.precision, is_3d=is_3d)
        if dog:
            dog.arm("first leg (first gradient all-reduces)", a.init_timeout + 60.0 * (a.warmup + a.steps))
        timer = _lib.KernelTimer(select_kernel)
        rec, (tr, xm, ym, last, step_ms, elapsed) = dp_leg(a.config, a, world, rank, dist, torch, a.steps, a.warmup, a.scaling, a.res, a.batch,
                                                           a.filters, timer, other=not a.no_other_leg)
        if dog:
            dog.disarm()
        per_gpu, global_batch = rec["batch_per_gpu"], rec["global_batch"]
        ks = timer.summary()
        pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * (len(step_ms) - 1) + 0.5))]
        step_stats = {"median_ms": pct(0.5), "p10_ms": pct(0.1), "p90_ms": pct(0.9), "min_ms": step_ms[0], "max_ms": step_ms[-1],
                      "source": "HIP events on the launch stream, one per step"}
        loss, comm, other = rec["loss"], rec["allreduce"], rec["other_scaling_leg"]

        # which algorithm every conv / weight-gradient call of a step takes (one extra, untimed step)
        dispatch = {}
        with ops.options(dispatch_counts=dispatch):
            tr.train_step(xm, ym)
        torch.cuda.synchronize()
        n_params = tr.n_params

        # N > 1, the metric's own config: short data-parallel legs of the 2-D workload and of cfg4's grid in the SAME job, so that one
        # driver invocation per N yields the 2-D and the 3-D numbers north_star asks for
        extra_legs = {}
        if world > 1 and a.config == "cfg3" and not a.no_extra_legs and a.precision == "fp32":
            del tr, last, xm, ym
            torch.cuda.empty_cache()
            units0 = 1
            for r in a.res:
                units0 *= r
            fallback = {"metric": "velocity-field voxels/sec (3D %s train step), whole job; per-GPU in `per_gpu`" % "x".join(str(r) for r in a.res),
                        "value": global_batch * units0 * a.steps / elapsed, "unit": wl["unit"], "per_gpu": global_batch * units0 * a.steps / elapsed / world,
                        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
                        "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                        "config": {"workload": rec["workload"], "global_batch": global_batch, "batch_per_gpu": per_gpu, "grid": list(a.res),
                                   "params": n_params, "parallelism": "dp%d" % world},
                        "rccl_ranks": wv["ranks"] if wv["backend"] == "nccl" else 0, "counted_ranks": wv["ranks"], "dist_backend": wv["backend"],
                        "distinct_devices": wv["distinct_devices"], "allreduce": comm, "other_scaling_leg": other, "cross_rank": rec["cross_rank"],
                        "loss": loss, "l1_vs_ref": {"value": rel_l1, "tolerance": 1e-4}} if rank == 0 else None
            for name in ("cfg2", "cfg4"):
                if dog:
                    dog.arm("extra leg %s" % name, a.init_timeout + 120.0 * a.extra_leg_steps, soft=True, fallback=fallback)
                try:
                    r, st = dp_leg(name, a, world, rank, dist, torch, a.extra_leg_steps, 2, "weak", other=not a.no_other_leg)
                    del st
                    extra_legs["extra_leg_" + name] = r
                except Exception as e:      # every rank takes the same path (shape / memory errors are deterministic); never the metric line
                    extra_legs["extra_leg_" + name] = {"error": repr(e)[:300]}
                if dog:
                    dog.disarm()
                ops.reset_variables()
                torch.cuda.empty_cache()
            tr = last = xm = ym = None

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    units = 1
    for r in a.res:
        units *= r
    vox_per_step = global_batch * units
    value = vox_per_step * a.steps / elapsed
    grid_s = "x".join(str(r) for r in a.res)

    default_shape = a.config == "cfg3" and list(a.res) == [64, 96, 64] and per_gpu == 16 and a.filters == 128 and a.precision == "fp32"   # the PMC passes' shape
    pmc, pmc_source = None, None
    if world == 1 and default_shape and not a.no_live_pmc:
        pmc = live_pmc()
        if pmc:
            pmc_source = "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE passes over tools/pmc_target.py (2 x FETCH_SIZE + WRITE_SIZE)"
    if not pmc:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["kernels"]
        except (OSError, ValueError, KeyError):
            pmc = {}

    what = "pixels" if not is_3d else "voxels"
    out = {
        "metric": "velocity-field %s/sec (%s %s train step), whole job; per-GPU in `per_gpu`" % (what, "3D" if is_3d else "2D", grid_s),
        "value": value, "unit": wl["unit"], "per_gpu": value / world,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": "f32" if a.precision == "fp32" else "bf16x3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": rec["workload"],
                   "global_batch": global_batch, "batch_per_gpu": per_gpu, "grid": list(a.res), "params": n_params,
                   "parallelism": "dp%d" % world},
        "roofline": None,            # filled below: the kernel family with the largest share of the step
        "roofline_stencil": None,    # the standalone jacobian3 kernel (the >= 70 % target); the step itself runs the fused tail
        "cpu_baseline": None,
        "l1_vs_ref": {"value": rel_l1, "tolerance": 1e-4,
                      "case": ("relative L1 of the auto-encoder's reconstructed velocity field vs the fp64 oracle, grid 16x16x16, filters %d" % a.filters) if wl["kind"] == "ae"
                              else "relative L1 of the velocity field vs the fp64 oracle, grid %s, filters %d" % ("16x24x16" if is_3d else "32x24", a.filters)},
        "l1_vs_ref_fullsize": None,
        "step_ms": step_stats,
        "value_median": vox_per_step / (step_stats["median_ms"] * 1e-3) if world == 1 else None,
        "rccl_ranks": wv["ranks"] if wv["backend"] == "nccl" else 0,      # counted by an on-device all-reduce over RCCL; 0 = not an RCCL job
        "counted_ranks": wv["ranks"], "dist_backend": wv["backend"], "devices": wv["devices"], "distinct_devices": wv["distinct_devices"],
        "allreduce": comm,           # per step: bytes, buckets, comm_span_ms, exposed_ms (after backward), hidden_ms (under backward)
        "other_scaling_leg": other,
        "cross_rank": rec["cross_rank"],
        "loss": loss,
        "roofline_wgrad": roofline_of(ks, "wgrad_kernel", pmc, default_shape, pmc_source),
        "roofline_conv": roofline_of(ks, "conv_mfma_kernel", pmc, default_shape, pmc_source),
        "roofline_wino": (roofline_of(ks, "wino43_kernel", pmc, default_shape, pmc_source) or roofline_of(ks, "wino3d_kernel", pmc, default_shape, pmc_source))
                         if is_3d else roofline_of(ks, "wino2d_kernel", pmc, default_shape, pmc_source),
        "roofline_wino2d_fwd_f24": None if is_3d else roofline_of(ks, "wino2d43_kernel", {}, False),
        "roofline_up27": roofline_of(ks, "wino3d_27pt_kernel up", {}, False),          # the 27-point forms (verdict r4 item 3a)
        "roofline_pool27": roofline_of(ks, "wino3d_27pt_kernel pooled", {}, False),
        "roofline_tail_fwd": roofline_of(ks, "velocity_loss3d_fwd_kernel", {}, False),
        "roofline_tail_bwd": roofline_of(ks, "velocity_loss3d_bwd_kernel", {}, False),
        "stencils_standalone": None,
        "dispatch": dispatch,
        "kernels": {k: {"launches": v["launches"], "ms_total": v["seconds"] * 1e3} for k, v in sorted(ks.items())},
    }
    out.update(extra_legs)
    fam = {}
    for k, v in ks.items():
        fam[k.split(" ")[0]] = fam.get(k.split(" ")[0], 0.0) + v["seconds"]
    dom = max((f for f in fam if not (f.startswith("jacobian") or f.startswith("velocity_loss"))), key=lambda f: fam[f], default=None)
    out["roofline"] = {"wgrad_kernel": out["roofline_wgrad"], "conv_mfma_kernel": out["roofline_conv"],
                       "wino43_kernel": out["roofline_wino"], "wino3d_kernel": out["roofline_wino"], "wino2d_kernel": out["roofline_wino"],
                       "wino2d43_kernel": out["roofline_wino2d_fwd_f24"]}.get(dom)
    for key in ("alt_bf16x3_mode", "extra_ref_grids", "extra_2d_128x96", "extra_cfg4_slice", "extra_ae_cfg5"):
        out[key] = None
    if world == 1 and is_3d:
        try:
            out["stencils_standalone"] = stencil_rooflines(per_gpu, *a.res)
            out["roofline_stencil"] = dict(out["stencils_standalone"]["jacobian3d_fwd_kernel<j,c>"], kernel="jacobian3d_fwd_kernel<j,c>",
                                           traffic=pmc.get("jacobian3d_fwd_kernel", {}).get("traffic_bytes") if default_shape else None,
                                           traffic_source=(pmc_source or "profiles/pmc_latest.json") if default_shape else None,
                                           note="standalone launches over rotating buffers (cold HBM); the train step itself runs the fused "
                                                "tail (roofline_tail_fwd / _bwd)")
            copy_gbs = out["stencils_standalone"]["copy_rate"]["achieved"]
            for k in ("roofline_tail_fwd", "roofline_tail_bwd"):       # the in-step fused tail beside the copy rate of the same run
                if out[k]:
                    out[k]["frac_of_copy_rate"] = out[k]["achieved"] / copy_gbs
                    out[k]["avg_launch_us"] = out[k]["avg_launch_ms"] * 1e3
        except Exception as e:
            out["stencils_standalone"] = {"error": repr(e)[:300]}
    elif world == 1:
        out["roofline_stencil"] = roofline_of(ks, "jacobian2d_fwd_kernel", {}, False)
    if world == 1 and not a.no_alt and a.precision == "fp32" and a.config == "cfg3":
        from deep_fluids_amd.trainer import default_config
        cfg = default_config(is_3d=True, res_x=a.res[2], res_y=a.res[1], res_z=a.res[0], filters=a.filters, batch_size=global_batch,
                             num_samples=wl["num_samples"], random_seed=123)
        del tr, last
        extras(out, a, cfg, xm, ym, vox_per_step, pmc)
    try:
        if world == 1 and not a.no_cpu_baseline and wl["kind"] == "de":
            par = {} if (a.precision == "fp32" and is_3d) else None
            out["cpu_baseline"] = cpu_baseline(a.res, a.filters, a.cpu_seconds, par)
            out["l1_vs_ref_fullsize"] = par
    except Exception as e:      # an extra must never take the metric line down with it
        out["cpu_baseline"] = {"error": repr(e)[:300]}
    # the full record goes to a sidecar file, the printed line stays short (headline objects first)
    try:
        side = a.sidecar or os.path.join(ROOT, "gpurun_out", "bench_full_n%d%s.json" % (world, "" if a.config == "cfg3" else "_" + a.config))
        os.makedirs(os.path.dirname(side), exist_ok=True)
        out["sidecar"] = os.path.relpath(side, ROOT)
        with open(side, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        out["sidecar"] = "not written: %r" % (e,)
    print(json.dumps(out if a.full_line else compact(out)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
