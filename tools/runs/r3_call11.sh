mkdir -p gpurun_out
VARIANTS=100,262144,100,262144 python tools/wino_diag.py 16 > gpurun_out/r3_wino_asym.log 2>&1; grep variant gpurun_out/r3_wino_asym.log
