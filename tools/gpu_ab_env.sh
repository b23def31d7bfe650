#!/bin/bash
# A/B of several environment settings on the eager bench: tools/gpu_ab_env.sh "A=1 B=2" "A=3" "" ...
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/abe_$i.log 2>&1
  echo "[$v]: $(tail -n 1 gpurun_out/abe_$i.log | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['stages'].items()})")"
done
