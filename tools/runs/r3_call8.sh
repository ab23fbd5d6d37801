mkdir -p gpurun_out
./tools/ubench/wino43b_probe 16 > gpurun_out/r3_wino43b_probe.log 2>&1
cat gpurun_out/r3_wino43b_probe.log
