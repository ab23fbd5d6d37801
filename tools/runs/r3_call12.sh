mkdir -p gpurun_out
DF_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --scaling strong --no-alt --no-cpu-baseline > gpurun_out/r03a_bench_2rank_gloo_one_gpu.json 2> gpurun_out/r03a_bench_2rank.err
tail -3 gpurun_out/r03a_bench_2rank.err; python -c "
import json; d=json.load(open('gpurun_out/r03a_bench_2rank_gloo_one_gpu.json')); print({k:d[k] for k in ('value','ms_per_step','n_gpus','rccl_ranks','counted_ranks','dist_backend','devices','distinct_devices','scaling','allreduce')})"
# RCCL with two ranks on ONE device must be refused by verify_world (not a multi-GPU measurement)
timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --batch 2 --no-alt --no-cpu-baseline > gpurun_out/r03a_rccl_shared.out 2> gpurun_out/r03a_rccl_shared.err; echo "rc=$?"; grep -h "one rank per GPU\|Duplicate\|RuntimeError\|Error" gpurun_out/r03a_rccl_shared.err | head -5
