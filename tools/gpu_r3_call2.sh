#!/bin/bash
# round 3, second GPU call: the new kernels (256-column filter-gradient tile, pipelined 256 x 256 implicit-GEMM loop) against the
# CPU restatement, their layer timings A/B in one box, the bf16 parity probe, the step
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/k2.log 2>&1
rc=$?; echo "kernel tests exit $rc: $(tail -n 1 gpurun_out/k2.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k2.log | head
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
L="fwd_a4 dgrad_a4 dgradacc_a4 fwd_b4 dgrad_b4 wgrad_a4 wgrad_b4 wgrad_b3 wgrad_a5 wgrad_b5"
echo "== new kernels"; bash tools/gpu_layers.sh $L 2>&1 | tee gpurun_out/layers_new.log
echo "== SFK_PIPE=0 SFK_WGT256=0"; SFK_PIPE=0 SFK_WGT256=0 bash tools/gpu_layers.sh $L 2>&1 | tee gpurun_out/layers_old.log
if [ $rc -ne 0 ]; then echo "kernel tests failed: no step runs"; exit 0; fi
bash tools/gpu_ab_env.sh "" "SFK_PIPE=0" "SFK_WGT256=0" "SFK_PIPE=0 SFK_WGT256=0" "" 2>&1 | tee gpurun_out/ab_call2.log
timeout -k 10 600 python tools/probe/bf16_parity.py 2 > gpurun_out/bf16_parity.log 2>&1; echo "parity probe exit $?"; grep "^\[" gpurun_out/bf16_parity.log
