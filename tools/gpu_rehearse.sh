#!/bin/bash
# what the comm stream of an N > 1 run costs the step, rehearsed on one GPU: tools/gpu_rehearse.sh "ENV" ...
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --rehearse-comm > gpurun_out/reh_$i.log 2>&1
  echo "[$v] rehearse: $(grep '^{' gpurun_out/reh_$i.log | tail -n 1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'])")"
done
