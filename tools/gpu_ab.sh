#!/bin/bash
# A/B of an environment knob on the eager bench: usage gpu_ab.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
  echo "$VAR=$v: $(tail -n 1 gpurun_out/ab_$v.log | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['stages'].items()})")"
done
