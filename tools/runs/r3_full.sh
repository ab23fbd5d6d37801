mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r3_full_tests.log 2>&1; tail -4 gpurun_out/r3_full_tests.log
python bench.py > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err; tail -c 300 gpurun_out/r3_bench2.json
python bench.py --batch 2 --steps 20 --warmup 5 --no-alt --no-cpu-baseline > gpurun_out/r3_bench_b2b.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/r3_bench2.json')); print('B16', d['ms_per_step'], d['step_ms']['median_ms'], d['extra_cfg4_slice'].get('ms_per_step'), d['extra_ae_cfg5'].get('ms_per_step'), d['extra_2d_128x96'].get('ms_per_step'), d['alt_bf16x3_mode'].get('ms_per_step'))
d=json.load(open('gpurun_out/r3_bench_b2b.json')); print('B2', d['ms_per_step'], d['step_ms'])
"
