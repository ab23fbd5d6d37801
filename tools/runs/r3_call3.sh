set -x
mkdir -p gpurun_out
./tools/ubench/wino43_probe 16 > gpurun_out/r3_wino43_probe2.log 2>&1
python tools/wgrad_b2_probe.py 2 4 16 > gpurun_out/r3_wgrad_b2.log 2>&1
cat gpurun_out/r3_wino43_probe2.log gpurun_out/r3_wgrad_b2.log
