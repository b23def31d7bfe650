#!/bin/bash
# the LDS-band 3x3 kernel: parity cases, then layer micro-benchmarks with and without it (SFK_HALO)
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "halo" > gpurun_out/halo_tests.log 2>&1
rc=$?; echo "halo tests exit $rc: $(tail -n 1 gpurun_out/halo_tests.log)"
if [ $rc -ne 0 ]; then grep -E "^E|FAILED|Error" gpurun_out/halo_tests.log | head -30; exit $rc; fi
for k in ${LAYERS:-fwd_b2 dgrad_b2 fwd_b3 dgrad_b3}; do
  for v in 0 1; do
    echo "HALO=$v $(SFK_HALO=$v timeout -k 10 120 python tools/bench_layer.py $k 30 2>&1 | tail -n 1)"
  done
done
