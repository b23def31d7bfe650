#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "k_steps_fp32 or two_steps_bf16 or bit_reproducible" > gpurun_out/m27.log 2>&1; rc=$?
echo "model tests exit $rc: $(tail -n 1 gpurun_out/m27.log)"
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/m27.log; exit $rc; fi
run() { name=$1; shift
  env SFK_X=1 "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ab_$name.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "$name failed $rc"; tail -n 5 gpurun_out/ab_$name.log; return $rc; fi
  echo "$name: $(tail -n 1 gpurun_out/ab_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['loss_after'])")"
}
run one SFK_SPLIT_ADAM=0 || exit 1
run split || exit 1
run one2 SFK_SPLIT_ADAM=0 || exit 1
run split2 || exit 1
