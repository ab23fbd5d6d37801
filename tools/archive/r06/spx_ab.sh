cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in spx2 spx4; do
  if [ $V = spx4 ]; then export DF_HIP_LIBRARY=$R/deep_fluids_amd/csrc/libdf_spx4.so.keep; else unset DF_HIP_LIBRARY; fi
  for CN in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $CN --output-format csv -d /tmp/pmc_${V}_$CN -- python $R/tools/pmc_target.py > /dev/null 2>&1
  done
  python $R/tools/pmc_summary.py /tmp/pmc_${V}_FETCH_SIZE /tmp/pmc_${V}_WRITE_SIZE | python -c "
import sys,json
d=json.load(sys.stdin)['kernels']
for k in ('wino43_kernel','wino43_kernel_dgrad_mask'): print('$V', k, round(d[k]['traffic_bytes']/1e9,2), 'GB x', round(d[k]['traffic_over_algorithmic'],2))"
  python $R/bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$V step ms', round(d['ms_per_step'],2), 'launch ms', round(d['roofline']['avg_launch_ms'],3))"
done
