#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "dual or tail" > gpurun_out/k25.log 2>&1; rc=$?
echo "kernel tests exit $rc: $(tail -n 1 gpurun_out/k25.log)"
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/k25.log; exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "two_steps_bf16 or metric_geometry_train_step" > gpurun_out/m25.log 2>&1; rc=$?
echo "model tests exit $rc: $(tail -n 1 gpurun_out/m25.log)"
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/m25.log; exit $rc; fi
run() { name=$1; shift
  env SFK_X=1 "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ab_$name.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "$name failed $rc"; tail -n 5 gpurun_out/ab_$name.log; return $rc; fi
  echo "$name: $(tail -n 1 gpurun_out/ab_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
}
run twopass SFK_TAIL_DUAL=0 || exit 1
run dual || exit 1
run twopass2 SFK_TAIL_DUAL=0 || exit 1
run dual2 || exit 1
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer.log 2>&1
echo "per-layer exit $?"
