#!/bin/bash
set -u
mkdir -p gpurun_out
SFK_FUSE_BNB=1 SFK_PER_LAYER=gpurun_out/per_layer_bnb.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer_bnb.log 2>&1
rc=$?; echo "per-layer bnb exit $rc"; tail -n 1 gpurun_out/bench_per_layer_bnb.log | cut -c1-200
