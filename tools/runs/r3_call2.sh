set -x
mkdir -p gpurun_out
./tools/ubench/mfma_overlap > gpurun_out/r3_mfma_overlap.log 2>&1
./tools/ubench/wino43_probe 4 > gpurun_out/r3_wino43_probe.log 2>&1
./tools/ubench/wino43_probe 16 >> gpurun_out/r3_wino43_probe.log 2>&1
VARIANTS=100,131072,0 python tools/wino_diag.py 16 > gpurun_out/r3_wino_diag.log 2>&1
SPX=4 VARIANTS=100 python tools/wino_diag.py 16 >> gpurun_out/r3_wino_diag.log 2>&1
SPX=1 VARIANTS=100 python tools/wino_diag.py 16 >> gpurun_out/r3_wino_diag.log 2>&1
python -m pytest tests/test_gpu_fullsize.py -q -s -m gpu -k "cfg5" > gpurun_out/r3_t2.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b2 -- python $GRAFT_REPO_ROOT/bench.py --batch 2 --steps 10 --warmup 3 --no-alt --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3_bench_b2.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3_bench_b2.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_b2 -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $f > gpurun_out/r3_b2_by_grid.md
rm -rf gpurun_out/prof_b2
tail -3 gpurun_out/r3_t2.log; cat gpurun_out/r3_wino43_probe.log; cat gpurun_out/r3_wino_diag.log | grep variant
