#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "fused_bn_backward or masked_store or relu_bitmap or tail" > gpurun_out/k5.log 2>&1
rc=$?; echo "kernel tests exit $rc: $(tail -n 1 gpurun_out/k5.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k5.log | head
if [ $rc -ne 0 ]; then exit 0; fi
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 600 -p no:cacheprovider -k "schedule_variants" > gpurun_out/m5.log 2>&1
echo "variant tests exit $?: $(tail -n 1 gpurun_out/m5.log)"
bash tools/gpu_ab_env.sh "" "SFK_FUSE_BNB=1" "" "SFK_FUSE_BNB=1" 2>&1 | tee gpurun_out/ab_call5.log
