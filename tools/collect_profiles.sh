#!/bin/bash
# Round-end evidence on the GPU box (gpurun): bench lines, rocprofv3 kernel tables (B = 16 and B = 2), PMC passes (one counter set per
# run, --kernel-trace only: MI355X_MICROARCH.md).  Everything lands in gpurun_out/${RND}_* (RND=r05 by default); the summaries are then copied to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${RND:-r05}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --sidecar $O/${RND}_bench_full_n1.json > $O/${RND}_bench_n1.json 2> $O/${RND}_bench_n1.err
python $R/bench.py --batch 2 --steps 50 --warmup 10 --no-alt --no-cpu-baseline --no-live-pmc --sidecar $O/${RND}_bench_full_b2.json > $O/${RND}_bench_b2.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b16 -- python $R/bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-live-pmc --sidecar /tmp/side_prof.json > $O/${RND}_bench_prof.json 2> /dev/null
python $R/tools/summarize_trace.py $(find $O/prof_b16 -name "*kernel_trace.csv" | head -1) > $O/${RND}_bench_kernel_by_grid.md; rm -rf $O/prof_b16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b2 -- python $R/bench.py --batch 2 --steps 10 --warmup 3 --no-alt --no-cpu-baseline --no-live-pmc --sidecar /tmp/side_prof2.json > /dev/null 2>&1
python $R/tools/summarize_trace.py $(find $O/prof_b2 -name "*kernel_trace.csv" | head -1) > $O/${RND}_bench_b2_kernel_by_grid.md; rm -rf $O/prof_b2
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -- python $R/tools/pmc_target.py > $O/pmc_$C.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/${RND}_pmc.json
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python $R/tools/pmc_target.py > $O/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python $R/tools/pmc_target.py > $O/pmc_sq.log 2>&1
python - <<PY > $O/${RND}_sq_counters.md
import csv, glob, collections, re
def read(d):
    out = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[(r["Kernel_Name"], int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
    return out
def dur(d):
    out = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[(r["Kernel_Name"], int(r["Dispatch_Id"]))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return out
sq, tcc, du = read("$O/pmc_sq"), read("$O/pmc_tcc"), dur("$O/pmc_sq")
fam = [("wino3d_kernel<0,9,0> forward 64x96x64 B=16", r"wino3d_kernel<0, 9, 0[,>]"), ("wino3d_kernel<0,4,0> dgrad + lrelu mask", r"wino3d_kernel<0, 4, 0[,>]"),
       ("wino3d_kernel<0,9,3> 27-point up-sampling-aware forward (coarse staging) -> 64x96x64", r"wino3d_kernel<0, 9, 3[,>]"),
       ("wino3d_kernel<0,0,2> 27-point pooled adjoint", r"wino3d_kernel<0, 0, 2[,>]"),
       ("wgrad_wxyz_fused_kernel<8,128>", r"wgrad_wxyz_fused_kernel<8, 128"), ("jacobian3d_fwd_vec_kernel<j,c>", r"jacobian3d_fwd_vec_kernel<true, true"),
       ("velocity_loss3d_tile_kernel (one-kernel tail forward)", r"velocity_loss3d_tile_kernel"), ("velocity_du3d_vec_kernel (tail backward)", r"velocity_du3d_vec_kernel")]
print("| kernel | duration ms | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | mfma busy cycles / SIMD | of kernel time @2.4 GHz | shader clock GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration) | of kernel time at that clock | L2 hit rate |\n|---|---|---|---|---|---|---|---|---|---|")
for name, rx in fam:
    ks = sorted(k for k in sq if re.search(rx, k[0]))
    if not ks: continue
    k = ks[-1]; c = sq[k]; w = c.get("SQ_WAVE_CYCLES", 0) or 1
    kt = sorted(q for q in tcc if re.search(rx, q[0]))
    hit = "-"
    if kt:
        t = tcc[kt[-1]]; hit = "%.3f" % (t.get("TCC_HIT_sum", 0) / max(t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0), 1))
    d = du.get(k, 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0
    ghz = c.get("GRBM_GUI_ACTIVE", 0) / 8.0 / (d * 1e-3) / 1e9 if d else 0.0
    print("| \`%s\` | %.3f | %.3f | %.3f | %.3f | %.3e | %s | %.3f | %s | %s |" % (name, d, c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w,
          busy, ("%.3f" % (busy / (d * 1e-3 * 2.4e9))) if d and busy else "-", ghz, ("%.3f" % (busy / (d * 1e-3 * ghz * 1e9))) if d and busy and ghz else "-", hit))
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_tcc $O/pmc_sq
cd $R
python -c "
import json
d=json.load(open('gpurun_out/${RND}_bench_n1.json')); print('B16', d['ms_per_step'], d['step_ms'], d['roofline']['frac'], d['roofline_wgrad']['frac'], d['roofline_tail_fwd'], d['roofline_tail_bwd']['frac'])
d=json.load(open('gpurun_out/${RND}_bench_b2.json')); print('B2', d['ms_per_step'], d['step_ms']['median_ms'])
"
cat gpurun_out/${RND}_sq_counters.md; python -c "
import json; d=json.load(open('gpurun_out/${RND}_pmc.json'))['kernels']
for k,v in d.items(): print(k, round(v['traffic_bytes']/1e6,1),'MB', round(v['traffic_over_algorithmic'],2))"
