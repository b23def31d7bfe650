#!/bin/bash
set -u
mkdir -p gpurun_out
bash tools/gpu_tests.sh || exit $?
bash tools/gpu_ab_env.sh "" "" 2>&1 | tee gpurun_out/ab_call8.log
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer.log 2>&1
echo "per-layer exit $?"
python tools/per_layer_report.py gpurun_out/per_layer.json gpurun_out/per_layer.txt && head -n 30 gpurun_out/per_layer.txt
