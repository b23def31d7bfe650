#!/bin/bash
# the stem kernels: parity cases, then the metric-geometry micro-benchmarks with / without the v3 forward (SFK_STEM3)
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "stem_conv" > gpurun_out/stem_tests.log 2>&1
rc=$?; echo "stem tests exit $rc: $(tail -n 1 gpurun_out/stem_tests.log)"
if [ $rc -ne 0 ]; then grep -E "^E|FAILED|Error" gpurun_out/stem_tests.log | head -30; exit $rc; fi
SFK_STEM3=${ALT:-0} timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "stem_conv" > gpurun_out/stem_tests_alt.log 2>&1
rc=$?; echo "stem tests (SFK_STEM3=${ALT:-0}) exit $rc: $(tail -n 1 gpurun_out/stem_tests_alt.log)"
if [ $rc -ne 0 ]; then grep -E "^E|FAILED|Error" gpurun_out/stem_tests_alt.log | head -30; exit $rc; fi
for k in ${KINDS:-fwd_fast fwd_slow wgrad_fast wgrad_slow}; do
  for v in ${VARIANTS:-0 3}; do
    echo "STEM3=$v $(SFK_STEM3=$v timeout -k 10 120 python tools/bench_stem.py $k 10 2>&1 | tail -n 1)"
  done
done
