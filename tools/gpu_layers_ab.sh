#!/bin/bash
# per-layer serial timings (one kernel at a time on one stream) for two environment settings: tools/gpu_layers_ab.sh "ENV_A" "ENV_B"
i=0
for v in "$@"; do
  i=$((i+1))
  env $v SFK_PER_LAYER=gpurun_out/pl_$i.json timeout -k 10 300 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pl_$i.log 2>&1
  echo "[$v] exit $?: $(grep '^{' gpurun_out/pl_$i.log | tail -n 1 | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'], l.get('stages_serial_ms'))")"
done
