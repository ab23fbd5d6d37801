mkdir -p gpurun_out
python -m pytest tests/test_gpu_stencils.py -q -m gpu -x > gpurun_out/r3_t6.log 2>&1; tail -4 gpurun_out/r3_t6.log
python tools/tail_probe.py > gpurun_out/r3_tail_probe.log 2>&1; cat gpurun_out/r3_tail_probe.log | grep variant
