mkdir -p gpurun_out
python -m pytest tests/test_gpu_layers.py tests/test_gpu_ae.py -q -m gpu -x -k "wgrad or upconv or ae_train or stride2" > gpurun_out/r3_t5.log 2>&1; tail -3 gpurun_out/r3_t5.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ae -- python $GRAFT_REPO_ROOT/tools/ae_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r3_ae_bench2.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_ae -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $f > gpurun_out/r3_ae_by_grid.md; rm -rf gpurun_out/prof_ae
grep "ms/step" gpurun_out/r3_ae_bench2.log; head -34 gpurun_out/r3_ae_by_grid.md | cut -c1-170
