#!/bin/bash
# Round-5 evidence on the GPU box: rocprofv3 kernel tables of the cfg3 / cfg2 / cfg5 steps (bench.py --config), written to gpurun_out/r05_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in ${CONFIGS:-cfg3 cfg2 cfg5}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$C -- python $R/bench.py --config $C --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-live-pmc --sidecar /tmp/side_$C.json > $O/r05_bench_prof_$C.json 2> /dev/null
  python $R/tools/summarize_trace.py $(find $O/prof_$C -name "*kernel_trace.csv" | head -1) > $O/r05_${C}_kernel_by_grid.md
  rm -rf $O/prof_$C
done
