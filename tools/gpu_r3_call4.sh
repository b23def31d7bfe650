#!/bin/bash
set -u
mkdir -p gpurun_out
bash tools/gpu_ablate.sh 2>&1 | tee gpurun_out/ablate_call4.log
bash tools/gpu_tests.sh "tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_aux.py" || exit $?
grep -n "bf16 depth-50" gpurun_out/test_gpu_model.log | cut -c1-900
