#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "batchnorm or stem_tail" > gpurun_out/k7.log 2>&1
echo "bn tests exit $?: $(tail -n 1 gpurun_out/k7.log)"
for cfg in "3211264 8" "3211264 8 16" "1605632 16" "802816 32" "401408 64" "100352 256" "802816 64" "802816 256"; do
  timeout -k 10 120 python tools/bench_bn.py $cfg 2>&1 | grep "^px"
done | tee gpurun_out/bench_bn.log
bash tools/gpu_ab_env.sh "" "" 2>&1 | tee gpurun_out/ab_call7.log
