#!/bin/bash
set -u
mkdir -p gpurun_out
for v in "" "lane1,lane3" "lane2,lane3" "lane1,lane2,lane3"; do
  SFK_ABLATE="$v" timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ablate.log 2>&1
  echo "[skip: $v]: $(tail -n 1 gpurun_out/ablate.log | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['ms_per_step'], 'ms/step')")"
done 2>&1 | tee gpurun_out/ablate_lanes.log
bash tools/gpu_ab_env.sh "" "SFK_WG_CUS=128" "SFK_WG_CUS=192" "SFK_WG_CUS=128 SFK_WGT256=256" "SFK_WG_CUS=64" "" 2>&1 | tee gpurun_out/ab_call6.log
