mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tail -- python $GRAFT_REPO_ROOT/tools/tail_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_tail -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $f > gpurun_out/r3_tail_by_grid.md; rm -rf gpurun_out/prof_tail
head -16 gpurun_out/r3_tail_by_grid.md | cut -c1-200
