#!/bin/bash
# Round 6: SQ / TCC counters of the roofline kernels incl. the F(2,3)xF(2,3)xF(4,3) family (tools/pmc_target.py), one counter set per pass,
# --kernel-trace only -> gpurun_out/r06_sq_counters.md (an independent check of the bench line's executed-MFMA fractions).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python $R/tools/pmc_target.py > $O/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python $R/tools/pmc_target.py > $O/pmc_sq.log 2>&1
python - <<PY > $O/r06_sq_counters.md
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
fam = [("wino43_kernel<9> F(2,2,4) forward 64x96x64 B=16 (96 points: 6 of 27 direct-form products)", r"wino43_kernel<9,"),
       ("wino43_kernel<4> F(2,2,4) dgrad + fp32 lrelu mask", r"wino43_kernel<4,"),
       ("wino3d_kernel<0,9,0> F(2,3)^3 forward (round 3-5 family)", r"wino3d_kernel<0, 9, 0[,>]"),
       ("wino3d_kernel<0,9,3> 27-point up-sampling-aware forward", r"wino3d_kernel<0, 9, 3[,>]"),
       ("wino3d_kernel<0,0,2> 27-point pooled adjoint", r"wino3d_kernel<0, 0, 2[,>]"),
       ("wgrad_wxyz_fused_kernel<8,128>", r"wgrad_wxyz_fused_kernel<8, 128"), ("jacobian3d_fwd_vec_kernel<j,c>", r"jacobian3d_fwd_vec_kernel<true, true"),
       ("velocity_loss3d_tile_kernel (one-kernel tail forward)", r"velocity_loss3d_tile_kernel"), ("velocity_du3d_vec_kernel (tail backward)", r"velocity_du3d_vec_kernel")]
print("# SQ / TCC counters of the roofline kernels at B = 16, 64x96x64, F = 128 (tools/pmc_target.py; rocprofv3 --kernel-trace --pmc, one counter set per pass)\n")
print("\`mfma busy cycles / SIMD\` = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs; the fp32-MFMA utilisation is that over the kernel's cycles at 2.4 GHz (the peak's clock) and at the shader clock measured under the kernel.\n")
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
rm -rf $O/pmc_tcc $O/pmc_sq
cat $O/r06_sq_counters.md
