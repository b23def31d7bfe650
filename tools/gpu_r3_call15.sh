#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "batchnorm or bn_ or fold or tail" > gpurun_out/k15.log 2>&1; rc=$?
echo "bn tests exit $rc: $(tail -n 1 gpurun_out/k15.log)"
if [ $rc -ne 0 ]; then tail -n 30 gpurun_out/k15.log; exit $rc; fi
timeout -k 10 120 python tools/bench_finalize.py 2>&1 | grep rows
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/abf_$i.log 2>&1; rc=$?
if [ $rc -ne 0 ]; then tail -n 5 gpurun_out/abf_$i.log; exit $rc; fi
tail -n 1 gpurun_out/abf_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
