#!/bin/bash
# per-layer report (serial + production lanes) and the lane timeline of the current tree
set -u
mkdir -p gpurun_out
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer.log 2>&1
rc=$?; echo "per-layer exit $rc"; tail -n 1 gpurun_out/bench_per_layer.log | cut -c1-200
if [ $rc -ne 0 ]; then tail -n 20 gpurun_out/bench_per_layer.log; exit $rc; fi
python tools/per_layer_report.py gpurun_out/per_layer.json gpurun_out/per_layer.txt && head -n 24 gpurun_out/per_layer.txt
python tools/lane_timeline.py gpurun_out/per_layer.json > gpurun_out/lane_timeline.txt 2>&1; head -n 8 gpurun_out/lane_timeline.txt
