#!/bin/bash
# the LDS-band filter-gradient kernel: parity cases, then the layer micro-benchmark with and without it (SFK_WGBAND)
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "filter_gradient_band" > gpurun_out/wband_tests.log 2>&1
rc=$?; echo "wband tests exit $rc: $(tail -n 1 gpurun_out/wband_tests.log)"
if [ $rc -ne 0 ]; then grep -E "^E|FAILED|Error" gpurun_out/wband_tests.log | head -30; exit $rc; fi
for k in ${LAYERS:-wgrad_b2 wgrad_b3}; do
  for v in 0 3; do
    echo "WGBAND=$v $(SFK_WGBAND=$v timeout -k 10 120 python tools/bench_layer.py $k 30 2>&1 | tail -n 1)"
  done
done
for e in ${EXPS:-}; do      # timing experiments (wrong results): libsfk_bexp<e>.so = tools/build_variant.sh bexp<e> -DSFK_BAND_EXP=<e>
  for k in ${EXPLAYERS:-wgrad_b2}; do
    echo "EXP=$e $(SFK_LIB=$PWD/video-classification_amd/libsfk_bexp$e.so timeout -k 10 120 python tools/bench_layer.py $k 30 2>&1 | tail -n 1)"
  done
done
if [ "${TRACE:-0}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wband_prof -o wband -- python $GRAFT_REPO_ROOT/tools/bench_layer.py ${TRACELAYER:-wgrad_b2} 20 > $GRAFT_REPO_ROOT/gpurun_out/wband_prof.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find gpurun_out/wband_prof -name "*kernel_stats.csv" | head -n 1)
  if [ -n "$f" ]; then head -n 8 "$f" | cut -c1-220; fi
fi
