#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or bnb or dgrad or tail" > gpurun_out/k18.log 2>&1; rc=$?
echo "conv tests exit $rc: $(tail -n 1 gpurun_out/k18.log)"
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/k18.log; exit $rc; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/abg_$i.log 2>&1; rc=$?
if [ $rc -ne 0 ]; then tail -n 5 gpurun_out/abg_$i.log; exit $rc; fi
tail -n 1 gpurun_out/abg_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer.log 2>&1
rc=$?; echo "per-layer exit $rc"
python tools/per_layer_report.py gpurun_out/per_layer.json gpurun_out/per_layer.txt && sed -n 3,15p gpurun_out/per_layer.txt
