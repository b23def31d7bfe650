#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or bnb or dgrad or tail" > gpurun_out/k22.log 2>&1; rc=$?
echo "conv tests exit $rc: $(tail -n 1 gpurun_out/k22.log)"
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/k22.log; exit $rc; fi
SFK_FUSE_BNB=1 timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "train_step or mini or bf16" > gpurun_out/m22.log 2>&1; rc=$?
echo "model tests (FUSE_BNB=1) exit $rc: $(tail -n 1 gpurun_out/m22.log)"
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/m22.log; exit $rc; fi
run() { name=$1; shift
  env SFK_X=1 "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ab_$name.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "$name failed $rc"; tail -n 5 gpurun_out/ab_$name.log; return $rc; fi
  echo "$name: $(tail -n 1 gpurun_out/ab_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
}
run base || exit 1
run fused SFK_FUSE_BNB=1 || exit 1
run base2 || exit 1
run fused2 SFK_FUSE_BNB=1 || exit 1
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer.log 2>&1
echo "per-layer exit $?"
SFK_FUSE_BNB=1 SFK_PER_LAYER=gpurun_out/per_layer_bnb.json timeout -k 10 600 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_per_layer_bnb.log 2>&1
echo "per-layer bnb exit $?"
