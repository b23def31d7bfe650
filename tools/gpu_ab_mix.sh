#!/bin/bash
# A/B of (library variant, environment) pairs on the eager bench: tools/gpu_ab_mix.sh "lib|ENV=..." ...  (lib = base or a build_variant name)
i=0
for spec in "$@"; do
  i=$((i+1))
  lib=${spec%%|*}; envs=${spec#*|}
  if [ "$lib" = base ]; then L=""; else L="SFK_LIB=$PWD/video-classification_amd/libsfk_$lib.so"; fi
  env $L $envs timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/abm_$i.log 2>&1
  echo "[$spec]: $(tail -n 1 gpurun_out/abm_$i.log | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'])")"
done
