#!/bin/bash
# Round 6: small-batch regime.  (1) wall ms/step eager vs graph; (2) rocprofv3 kernel traces of both modes -> launches and kernel-time sum per step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/r06_smallbatch.py time ${TIME_CASES:-2d_b8 ae2d_b8 dg2d_b8 3d_b1 3d_b2 2d_b64} > $O/r06_smallbatch_wall.jsonl 2> $O/r06_smallbatch_wall.err
for C in ${TRACE_CASES:-2d_b8 ae2d_b8 3d_b1}; do
  for M in eager graph; do
    rocprofv3 --kernel-trace --output-format csv -d $O/prof_${C}_$M -- python $R/tools/r06_smallbatch.py trace $C $M > $O/r06_${C}_${M}_trace.json 2> /dev/null
    python $R/tools/summarize_trace.py $(find $O/prof_${C}_$M -name "*kernel_trace.csv" | head -1) --steps 23 > $O/r06_${C}_${M}_kernel_by_grid.md
    rm -rf $O/prof_${C}_$M
  done
done
