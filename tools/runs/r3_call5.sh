mkdir -p gpurun_out
DBGS=0,11,12 python tools/wgrad_diag.py > gpurun_out/r3_wgrad_diag.log 2>&1
cat gpurun_out/r3_wgrad_diag.log
