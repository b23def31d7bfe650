#!/bin/bash
# rocprofv3 kernel trace of the bench command (eager launches so every kernel is its own dispatch) + per-layer event timings
set -u
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_eager.log 2>&1
echo "bench eager exit $?"; tail -n 2 gpurun_out/bench_eager.log | cut -c1-400
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?"; tail -n 3 gpurun_out/rocprof.log | cut -c1-300
find gpurun_out/prof -name "*stats*" | head; find gpurun_out/prof -name "*kernel_stats*" -exec head -n 40 {} \;
# keep the merge small: drop the raw per-dispatch trace if it is big
find gpurun_out/prof -name "*kernel_trace.csv" -size +40M -delete
