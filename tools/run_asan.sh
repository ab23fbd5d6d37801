#!/bin/bash
# Host-side AddressSanitizer run: builds libdeepfluids_hip_asan.so (host code instrumented, device code unchanged) and the
# C oracle with -fsanitize=address, then runs the tests that exercise the host side without a GPU (C-ABI exports, status-code paths,
# workspace queries, the C oracle against the golden vectors) with the ASAN runtime preloaded.  On a GPU box add `-m gpu` tests to taste.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C "$R/deep_fluids_amd/csrc" asan
gcc -O1 -g -std=c99 -fPIC -shared -fopenmp -fsanitize=address -fno-omit-frame-pointer -o "$R/oracle/libdf_oracle_asan.so" "$R/oracle/df_oracle.c" -lm
export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
export DF_HIP_LIBRARY="$R/deep_fluids_amd/csrc/libdeepfluids_hip_asan.so"
export DF_ORACLE_LIBRARY="$R/oracle/libdf_oracle_asan.so"
cd "$R" && python -m pytest tests/test_cabi.py tests/test_oracle_c.py -q "$@"
