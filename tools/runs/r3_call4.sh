mkdir -p gpurun_out
python tools/wgrad_b2_probe.py 2 > gpurun_out/r3_wgrad_b2_after.log 2>&1
for R in 32 64 256; do echo "RANGES $R"; WGRAD_TOP=1 python tools/wgrad_ranges_probe.py $R; done > gpurun_out/r3_wgrad_ranges.log 2>&1
python -m pytest tests/test_gpu_layers.py tests/test_gpu_train_step.py -q -m gpu -x > gpurun_out/r3_t3.log 2>&1
tail -3 gpurun_out/r3_t3.log; cat gpurun_out/r3_wgrad_b2_after.log gpurun_out/r3_wgrad_ranges.log
