#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite on the new ABI, the default bench line, scheduling A/Bs, layer micro-benchmarks
set -u
mkdir -p gpurun_out
bash tools/gpu_tests.sh || exit $?
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r03_base.log 2>&1
rc=$?; echo "bench exit $rc"; tail -n 1 gpurun_out/bench_r03_base.log | cut -c1-400
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
bash tools/gpu_ab_env.sh "" "SFK_WGRAD_LANES=2" "SFK_WG_DEFER=1" "SFK_WG_DEFER=2" "SFK_WG_DEFER=2 SFK_WGRAD_LANES=2" "" 2>&1 | tee gpurun_out/ab_call1.log
bash tools/gpu_layers.sh fwd_a4 dgrad_a4 dgradacc_a4 wgrad_a4 fwd_b4 dgrad_b4 wgrad_b4 fwd_c4 wgrad_c4 fwd_b3 wgrad_b3 fwd_b2 wgrad_b2 2>&1 | tee gpurun_out/layers_call1.log
