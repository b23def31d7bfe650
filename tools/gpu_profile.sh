#!/bin/bash
# The round's evidence in one GPU call: default bench line -> rocprofv3 kernel trace of the SAME command (+ per-class
# summary and the roofline fraction recomputed from the trace) -> optional PMC traffic passes (TRAFFIC=1).
# Outputs under gpurun_out/; copy what is to be judged into profiles/rNN_*.
set -u
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
STEPS=${STEPS:-10}; WARM=${WARM:-3}
timeout -k 10 600 python bench.py --gpus 1 --steps $STEPS --warmup $WARM ${BENCH_ARGS:-} > gpurun_out/bench_default.log 2>&1
rc=$?; echo "bench exit $rc"; tail -n 1 gpurun_out/bench_default.log | cut -c1-600
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
# per-layer view of the same tree (serial + production lanes): profiles/rNN_per_layer.txt (MFMA utilisation of the MFMA-bound
# conv layers, HBM GB/s of the BatchNorm / pool stages, lane sums)
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer.log 2>&1
rc=$?; echo "per-layer exit $rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
python tools/per_layer_report.py gpurun_out/per_layer.json gpurun_out/per_layer.txt && head -n 22 gpurun_out/per_layer.txt
python tools/lane_timeline.py gpurun_out/per_layer.json.lanes gpurun_out/per_layer.json 300 > gpurun_out/lane_timeline.txt 2>&1; head -n 6 gpurun_out/lane_timeline.txt
rm -rf gpurun_out/prof/*
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline > gpurun_out/bench_traced.log 2>&1
rc=$?; echo "rocprof exit $rc"; tail -n 1 gpurun_out/bench_traced.log | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
KS=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -n 1)
cp "$KS" gpurun_out/kernel_stats.csv
# steps the traced command ran: warm-up + timed + 3 instrumented
python tools/trace_classes.py gpurun_out/kernel_stats.csv $((STEPS+WARM+3)) gpurun_out/bench_traced.log > gpurun_out/class_stats.json
python - <<'PY'
import json
c = json.load(open("gpurun_out/class_stats.json"))
for k, v in list(c["classes"].items())[:12]: print(f"{k:16s} {v}")
print(c.get("roofline_check"))
PY
KT=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -n 1)
python tools/trace_outliers.py "$KT" > gpurun_out/trace_outliers.txt 2>&1; head -n 25 gpurun_out/trace_outliers.txt
find gpurun_out/prof -name "*kernel_trace.csv" -size +40M -delete
if [ "${TRAFFIC:-0}" = "1" ]; then bash tools/gpu_traffic.sh; fi
