"""CPU-side checks of bench.py's contract pieces that need no GPU: argument defaults, the self-launch command line, the kernel
selector's algorithmic work figures and the shape of the `roofline` object (executed-MFMA fraction <= 1, algorithmic rate separate)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_defaults_and_self_launch_command(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.scaling, a.batch, a.res, a.filters, a.precision) == (1, "weak", 16, [64, 96, 64], 128, "fp32")
    assert a.steps > 0 and a.warmup >= 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--scaling", "strong"])
    a = b.parse()
    seen = {}
    monkeypatch.setattr(b.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    assert b.self_launch(a) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == ["--gpus", "8", "--steps", "3", "--warmup", "1", "--scaling", "strong"][-7:]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_kernel_selector_and_roofline_object():
    b = _bench()
    B, D, H, W, C = 16, 64, 96, 64, 128
    # df_wino_conv_fwd(x, wp, bias, residual, mask, y, B, D, H, W, Cin, Cout, flags, leak, stream)
    key, work = b.select_kernel("df_wino_conv_fwd", (1, 2, 3, None, None, 4, B, D, H, W, C, C, 9, 0.2, None))
    assert key == "wino3d_kernel fwd/dgrad 64x96x64 C128->128" and work == 2.0 * 27 * C * C * B * D * H * W      # 5.566 TFLOP
    key, work = b.select_kernel("df_jacobian3d_fwd", (1, 2, 3, B, D, H, W, None))
    assert key == "jacobian3d_fwd_kernel<j,c>" and work == 60.0 * B * D * H * W                                    # 60 B/voxel
    assert b.select_kernel("df_jacobian3d_fwd", (1, None, 3, B, D, H, W, None)) is None
    key, work = b.select_kernel("df_conv_wgrad_algo", (1, 2, 3, 4, B, D, H, W, C, C, 3, 5, 6, 0, None))
    assert key == "wgrad_kernel 64x96x64 C128x128 B16"
    ks = {"wino3d_kernel fwd/dgrad 64x96x64 C128->128": {"launches": 10, "seconds": 0.170, "work": 10 * 5.566277615616e12},
          "wino3d_kernel fwd/dgrad 8x12x8 C128->128": {"launches": 10, "seconds": 0.001, "work": 10 * 1.0872e10},
          "wgrad_kernel 64x96x64 C128x128 B16": {"launches": 5, "seconds": 0.075, "work": 5 * 5.566277615616e12}}
    r = b.roofline_of(ks, "wino3d_kernel", {"wino3d_kernel": {"traffic_bytes": 3.2e10}}, True)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and r["kernel"].startswith("wino3d_kernel fwd/dgrad 64x96x64")
    assert abs(r["algorithmic_tflops"] - 327.43) < 0.1 and abs(r["algorithmic_speedup"] - 3.375) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_tflops"] * 8 / 27) < 1e-9 and 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / 157.3) < 1e-12
    assert r["traffic"] == 3.2e10 and r["traffic_source"]
    w = b.roofline_of(ks, "wgrad_kernel", {}, False)
    assert abs(w["algorithmic_speedup"] - 3.375) < 1e-12 and w["traffic"] is None and w["frac"] <= 1.0 and w["wgrad_form"] == "winograd-xyz"
    # the executed / algorithmic ratio follows the LIBRARY's own choice (df_conv_wgrad_form == conv_wgrad.hip::wgrad_algo), incl. its
    # row-length and channel-count conditions: (x,y,z) exists for 128 -> 128 and 64 -> 64 at the instantiated row lengths (any batch since
    # round 3), elsewhere (x,y) from 4096 image rows | x | direct
    assert b.wgrad_exec_ratio(64, 1, 128, 96, 128, 128) == (4.0 / 9.0, 2) and b.wgrad_exec_ratio(16, 64, 96, 64, 128, 128) == (8.0 / 27.0, 3)
    assert b.wgrad_exec_ratio(16, 7, 10, 7, 128, 128) == (1.0, 0) and b.wgrad_exec_ratio(2, 32, 48, 32, 128, 128) == (8.0 / 27.0, 3)
    assert b.wgrad_exec_ratio(2, 8, 12, 8, 128, 128)[1] in (0, 1) and b.wgrad_exec_ratio(16, 64, 96, 64, 64, 64) == (8.0 / 27.0, 3)
    assert b.wgrad_exec_ratio(16, 64, 96, 64, 96, 96) == (4.0 / 9.0, 2) and b.wgrad_exec_ratio(1, 8, 8, 16, 96, 96) == (2.0 / 3.0, 1)
    json.dumps(r)      # the object must be JSON-serialisable as is


def test_compact_line_puts_headline_objects_first_and_drops_bulk():
    b = _bench()
    rf = {"kernel": "wino3d_kernel fwd/dgrad 64x96x64 C128->128", "bound": "mfma", "achieved": 99.0, "peak": 157.3, "unit": "TFLOP/s",
          "frac": 0.63, "traffic": 3.0e10, "note": "x" * 500, "work_per_launch": 1.0, "avg_launch_ms": 16.6, "launches": 3}
    out = {"metric": "m", "value": 1.0, "unit": "voxels/s", "per_gpu": 1.0, "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 198.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "cfg3"},
           "roofline": rf, "roofline_stencil": dict(rf, bound="hbm"), "roofline_wgrad": rf, "roofline_tail_fwd": None, "roofline_tail_bwd": None,
           "cpu_baseline": {"value": 1.2e5, "unit": "voxels/s", "cores": 64, "kind": "port", "sample": "s", "stencil_tail": {"big": "x" * 900}},
           "l1_vs_ref": {"value": 1e-6}, "l1_vs_ref_fullsize": {"velocity_rel_l1": 2e-6, "loss_rel": 1e-7},
           "kernels": {"k%d" % i: {"launches": 1, "ms_total": 1.0} for i in range(60)}, "dispatch": {"d%d" % i: 1 for i in range(60)},
           "stencils_standalone": {"x": "y" * 2000}, "extra_cfg4_slice": {"ms_per_step": 258.0, "value": 3e7, "roofline": rf, "note": "n" * 400,
                                                                           "bf16x3_mode": {"ms_per_step": 238.0}},
           "extra_ae_cfg5": {"error": "boom"}, "sidecar": "gpurun_out/bench_full_n1.json", "loss": 0.5}
    line = b.compact(out)
    keys = list(line)
    assert keys[:14] == ["metric", "value", "unit", "per_gpu", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                         "vs_baseline", "dtype", "data", "config"]
    assert keys.index("roofline") < keys.index("cpu_baseline") < keys.index("l1_vs_ref") < keys.index("l1_vs_ref_fullsize") < keys.index("step_ms")
    assert "kernels" not in line and "dispatch" not in line and "stencils_standalone" not in line
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"]) and "note" not in line["roofline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"]) and "stencil_tail" not in line["cpu_baseline"]
    assert line["extra_cfg4_slice"] == {"ms_per_step": 258.0, "value": 3e7, "bf16x3_ms_per_step": 238.0, "roofline_frac": 0.63}
    assert line["extra_ae_cfg5"] == {"error": "boom"}
    assert len(json.dumps(line)) < 3000


def test_watchdog_prints_a_diagnostic_line_and_exits_nonzero(tmp_path):
    """N > 1 bring-up that hangs (rendezvous / RCCL communicator / first all-reduce): one JSON line with rccl_ranks 0 on rank 0's stdout,
    exit code 3 -- never a hung lease."""
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r); import importlib.util as u; s = u.spec_from_file_location('b', %r); "
            "m = u.module_from_spec(s); s.loader.exec_module(m); d = m.Watchdog(0, 8); d.arm('first all-reduce', 0.6); time.sleep(30)"
            % (ROOT, os.path.join(ROOT, "bench.py")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3
    msg = json.loads(r.stdout.strip().splitlines()[-1])
    assert msg["rccl_ranks"] == 0 and msg["value"] is None and msg["n_gpus"] == 8 and "first all-reduce" in msg["error"]
    # disarmed in time: nothing happens
    code2 = code.replace("time.sleep(30)", "d.disarm(); time.sleep(1.5); print('alive')")
    r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=60)
    assert r2.returncode == 0 and r2.stdout.strip().endswith("alive")


import pytest  # noqa: E402


@pytest.mark.parametrize("config,res,batch,filters,kind,unit", [("cfg2", [128, 96], 64, 128, "de", "pixels/s"), ("cfg3", [64, 96, 64], 16, 128, "de", "voxels/s"),
                                                                ("cfg4", [112, 160, 112], 4, 128, "de", "voxels/s"), ("cfg5", [128, 128, 128], 4, 64, "ae", "voxels/s")])
def test_config_switch_resolves_the_baseline_workloads(monkeypatch, config, res, batch, filters, kind, unit):
    """--config {cfg2,cfg3,cfg4,cfg5}: BASELINE.json's grids / per-GPU batches / widths; explicit --res / --batch / --filters still win;
    the N > 1 command line carries the switch through the self-launch."""
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", config])
    a = b.parse()
    assert (a.config, a.res, a.batch, a.filters) == (config, res, batch, filters)
    wl = b.WORKLOADS[config]
    assert wl["kind"] == kind and wl["unit"] == unit and config in wl["name"].format(grid="x")
    small = ["16", "24"] if len(res) == 2 else ["16", "24", "16"]
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", config, "--gpus", "8", "--res"] + small + ["--batch", "2", "--filters", "32"])
    a = b.parse()
    assert (a.res, a.batch, a.filters) == ([int(v) for v in small], 2, 32)
    seen = {}
    monkeypatch.setattr(b.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd) or 0)
    assert b.self_launch(a) == 0
    assert seen["cmd"][seen["cmd"].index("--config") + 1] == config
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", config, "--res", "8"] + (["8", "8", "8"] if len(res) == 3 else []))
    with pytest.raises(SystemExit):       # a grid of the wrong dimensionality for the workload
        b.parse()


def test_compact_line_carries_extra_legs_and_cross_rank():
    b = _bench()
    leg = {"config": "cfg2", "grid": [128, 96], "scaling": "weak", "global_batch": 128, "batch_per_gpu": 64, "steps": 5, "ms_per_step": 21.0,
           "value": 7.4e7, "unit": "pixels/s", "allreduce": {"comm_span_ms": 1.0, "exposed_ms": 0.1, "hidden_ms": 0.9}, "loss": 0.5,
           "other_scaling_leg": None, "workload": "w" * 300,
           "cross_rank": {"ok": True, "grad_identical_on_all_ranks": True, "loss_rel_diff": 2e-8, "grad_sum": 1.0, "loss_shard_mean": 0.5}}
    out = {"metric": "m", "value": 1.0, "unit": "voxels/s", "per_gpu": 1.0, "n_gpus": 2, "steps": 2, "warmup": 1, "ms_per_step": 198.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "cfg3"},
           "cross_rank": leg["cross_rank"], "extra_leg_cfg2": leg, "extra_leg_cfg4": {"error": "boom"}}
    line = b.compact(out)
    assert line["cross_rank"]["ok"] is True
    assert line["extra_leg_cfg2"]["cross_rank"] == {"ok": True, "grad_identical_on_all_ranks": True, "loss_rel_diff": 2e-8}
    assert "workload" not in line["extra_leg_cfg2"] and line["extra_leg_cfg2"]["allreduce"]["hidden_ms"] == 0.9
    assert line["extra_leg_cfg4"] == {"error": "boom"}
    json.dumps(line)


def test_watchdog_soft_stage_prints_the_fallback_line_and_exits_zero():
    """A hang in one of the OPTIONAL extra legs (cfg2 / cfg4 appended to the cfg3 line at N > 1) must not cost the job its headline number:
    rank 0 prints the already-measured contract line with `extra_legs_error`, every rank exits 0."""
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r); import importlib.util as u; s = u.spec_from_file_location('b', %r); "
            "m = u.module_from_spec(s); s.loader.exec_module(m); d = m.Watchdog(0, 8); "
            "d.arm('extra leg cfg4', 0.6, soft=True, fallback={'metric': 'm', 'value': 1.5e8, 'n_gpus': 8}); time.sleep(30)"
            % (ROOT, os.path.join(ROOT, "bench.py")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    msg = json.loads(r.stdout.strip().splitlines()[-1])
    assert msg["value"] == 1.5e8 and msg["n_gpus"] == 8 and "extra leg cfg4" in msg["extra_legs_error"]
    code1 = code.replace("d = m.Watchdog(0, 8)", "d = m.Watchdog(3, 8)").replace(", fallback={'metric': 'm', 'value': 1.5e8, 'n_gpus': 8}", "")
    r1 = subprocess.run([sys.executable, "-c", code1], capture_output=True, text=True, timeout=60)
    assert r1.returncode == 0 and r1.stdout.strip() == ""
