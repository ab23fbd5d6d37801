#!/bin/bash
# Round-6 evidence on the GPU box: the driver's own command (python bench.py) + rocprofv3 kernel tables of the cfg3 / cfg2 / cfg4 / cfg5 steps
# (bench.py --config) + the PMC passes bench.py runs itself (live_pmc), written to gpurun_out/r06_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -z "$SKIP_BENCH" ]; then
  python $R/bench.py --sidecar $O/r06_bench_full_n1.json > $O/r06_bench_n1.json 2> $O/r06_bench_n1.err
fi
for C in ${CONFIGS:-cfg3 cfg2 cfg4 cfg5}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$C -- python $R/bench.py --config $C --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-live-pmc --sidecar /tmp/side_$C.json > $O/r06_bench_prof_$C.json 2> /dev/null
  python $R/tools/summarize_trace.py $(find $O/prof_$C -name "*kernel_trace.csv" | head -1) > $O/r06_${C}_kernel_by_grid.md
  rm -rf $O/prof_$C
done
if [ -z "$SKIP_PMC" ]; then
  for CN in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $CN --output-format csv -d $O/pmc_$CN -- python $R/tools/pmc_target.py > /dev/null 2>&1
  done
  python $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/r06_pmc.json
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
fi
