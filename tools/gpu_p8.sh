#!/bin/bash
# the deep-pipelined conv tile: parity cases, then layer micro-benchmarks with and without it (SFK_P8)
set -u
mkdir -p gpurun_out
SFK_P8=3 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "p8 or (conv_forward and c256-256) or (conv_forward and c512-256) or (data_gradient and c256-256) or (data_gradient and c512-256)" > gpurun_out/p8_tests.log 2>&1
rc=$?; echo "p8 tests exit $rc: $(tail -n 1 gpurun_out/p8_tests.log)"
if [ $rc -ne 0 ]; then grep -E "^E|FAILED|Error" gpurun_out/p8_tests.log | head -30; exit $rc; fi
for k in ${LAYERS:-fwd_a4 dgrad_a4 fwd_b4 dgrad_b4 fwd_a5 dgrad_a5 fwd_b5 dgrad_b5 fwd_c4}; do
  for v in 0 ${P8V:-3}; do
    echo "P8=$v $(SFK_P8=$v timeout -k 10 120 python tools/bench_layer.py $k 30 2>&1 | tail -n 1)"
  done
done
