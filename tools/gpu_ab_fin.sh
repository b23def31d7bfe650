#!/bin/bash
# fused BatchNorm finalize A/B (short timeouts: the first build of the prologue deadlocked under the four-lane schedule)
i=0
for v in "SFK_FUSE_FIN=0" "SFK_FUSE_FIN_MAXC=64" "SFK_FUSE_FIN_MAXC=128" "SFK_FUSE_FIN_MAXC=512" "SFK_FUSE_FIN=0" "SFK_FUSE_FIN_MAXC=64"; do
  i=$((i+1))
  env $v timeout -k 10 150 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/abf_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "[$v]: exit $rc"; tail -n 3 gpurun_out/abf_$i.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; continue; fi
  echo "[$v]: $(tail -n 1 gpurun_out/abf_$i.log | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['stages'].items()})")"
done
