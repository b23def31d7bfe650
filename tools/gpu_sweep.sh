#!/bin/bash
# A/B sweep of tuning knobs on ONE box: every configuration = one default bench run (10 steps); baseline first, last and in the middle.
# usage: bash tools/gpu_sweep.sh "name VAR=val [VAR=val]" ...   (a baseline run is interleaved every four configurations)
set -u
mkdir -p gpurun_out
run() { name=$1; shift
  env SFK_X=1 "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/sw_$name.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "$name failed $rc"; tail -n 3 gpurun_out/sw_$name.log; return 0; fi
  echo "$name: $(tail -n 1 gpurun_out/sw_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
}
i=0
run base_0
for cfg in "$@"; do
  run $cfg
  i=$((i+1))
  if [ $((i % 4)) -eq 0 ]; then run base_$i; fi
done
run base_end
