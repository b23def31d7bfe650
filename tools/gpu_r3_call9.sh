#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "batchnorm or stem_tail or tail" > gpurun_out/k9.log 2>&1
echo "bn tests exit $?: $(tail -n 1 gpurun_out/k9.log)"
for cfg in "3211264 8" "802816 32" "401408 64" "100352 256"; do timeout -k 10 120 python tools/bench_bn.py $cfg 2>&1 | grep "bwd_reduce"; done
bash tools/gpu_ab_env.sh "" "" 2>&1 | tee gpurun_out/ab_call9.log
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer.log 2>&1
python tools/per_layer_report.py gpurun_out/per_layer.json gpurun_out/per_layer.txt && sed -n 3,14p gpurun_out/per_layer.txt
