#!/bin/bash
# round 3, third GPU call: kernel parity of the 256-column filter-gradient tile, its trace, the step A/B, the bf16 parity probe
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/k3.log 2>&1
rc=$?; echo "kernel tests exit $rc: $(tail -n 1 gpurun_out/k3.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k3.log | head
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
rm -rf gpurun_out/prof_wg; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wg -o t -- python tools/bench_layer.py wgrad_a4 20 > gpurun_out/wg_trace.log 2>&1
KS=$(find gpurun_out/prof_wg -name "*kernel_stats.csv" | head -n 1); [ -n "$KS" ] && head -n 6 "$KS" | cut -c1-220
find gpurun_out/prof_wg -name "*kernel_trace.csv" -delete
if [ $rc -ne 0 ]; then echo "kernel tests failed: no step runs"; exit 0; fi
bash tools/gpu_ab_env.sh "" "SFK_WGT256=0" "" "SFK_WGT256=0" 2>&1 | tee gpurun_out/ab_call3.log
timeout -k 10 600 python tools/probe/bf16_parity.py 2 > gpurun_out/bf16_parity.log 2>&1; echo "parity probe exit $?"; grep "^\[" gpurun_out/bf16_parity.log
