#!/bin/bash
# A/B of library builds on the eager bench: tools/gpu_ab_lib.sh base nt1 nt2 ...  ("base" = the in-tree libsfk.so)
for v in "$@"; do
  if [ "$v" = base ]; then unset SFK_LIB; else export SFK_LIB=$PWD/video-classification_amd/libsfk_$v.so; fi
  timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/abl_$v.log 2>&1
  echo "$v: $(tail -n 1 gpurun_out/abl_$v.log | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['stages'].items()})")"
done
