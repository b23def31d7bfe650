#!/bin/bash
# kernel parity tests + eager bench with per-layer timings (no CPU baseline)
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 180 -p no:cacheprovider -x > gpurun_out/kernels.log 2>&1
rc=$?; echo "kernels exit $rc"; tail -n 5 gpurun_out/kernels.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
if [ $rc -ne 0 ]; then grep -E "^E|FAILED" gpurun_out/kernels.log | head -20; exit $rc; fi
SFK_PER_LAYER=gpurun_out/per_layer.json timeout -k 10 600 python bench.py --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1
echo "bench exit $?"; tail -n 1 gpurun_out/bench_quick.log | python -c "
import sys, json
l = json.loads(sys.stdin.readline())
print('clips/s', l['value'], 'ms/step', l['ms_per_step'], 'loss', l['loss_after'])
for k, v in l.get('stages', {}).items(): print(' ', k, v)
print(' roofline', l.get('roofline'))
"
