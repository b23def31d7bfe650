#!/bin/bash
# the deep-pipelined filter-gradient tile: parity cases, then layer micro-benchmarks with and without it (SFK_WGP8)
set -u
mkdir -p gpurun_out
SFK_WGP8=2 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "filter_gradient_p8" > gpurun_out/wp8_tests.log 2>&1
rc=$?; echo "wp8 tests exit $rc: $(tail -n 1 gpurun_out/wp8_tests.log)"
if [ $rc -ne 0 ]; then grep -E "^E|FAILED|Error" gpurun_out/wp8_tests.log | head -30; exit $rc; fi
for k in ${LAYERS:-wgrad_a4 wgrad_b4 wgrad_c4 wgrad_a5 wgrad_b5}; do
  for v in 0 ${WP8V:-8}; do
    echo "WGP8=$v $(SFK_WGP8=$v timeout -k 10 120 python tools/bench_layer.py $k 30 2>&1 | tail -n 1)"
  done
done
