#!/bin/bash
# All -m gpu tests (K='expr' = pytest -k), one pytest process per file (one at a time), logs under gpurun_out/; a killed step ends the script.
set -u
mkdir -p gpurun_out
: > gpurun_out/tests_summary.log
FILES=${1:-"tests/test_gpu_kernels.py tests/test_gpu_aux.py tests/test_gpu_dist.py tests/test_gpu_model.py"}
for f in $FILES; do
  name=$(basename $f .py)
  timeout -k 10 ${TEST_SECS:-1000} python -m pytest $f -m gpu -q --timeout 900 -p no:cacheprovider -s ${K:+-k "$K"} > gpurun_out/$name.log 2>&1
  rc=$?
  echo "$name exit $rc: $(tail -n 1 gpurun_out/$name.log)" | tee -a gpurun_out/tests_summary.log
  grep -E "^(FAILED|ERROR)|worst" gpurun_out/$name.log | head -40 | tee -a gpurun_out/tests_summary.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/tests_summary.log; exit $rc; fi
done
# the deep-pipelined tiles with every case eligible: 224-row conv tiles (SFK_P8=3), filter-gradient tile from 2 K-tiles per workgroup (SFK_WGP8=2)
SFK_P8=3 SFK_WGP8=2 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "p8 or conv_forward or data_gradient or filter_gradient" > gpurun_out/test_gpu_kernels_p8.log 2>&1
echo "test_gpu_kernels (SFK_P8=3 SFK_WGP8=2) exit $?: $(tail -n 1 gpurun_out/test_gpu_kernels_p8.log)" | tee -a gpurun_out/tests_summary.log
# the kernels the new paths replace stay callable: band filter gradient off, v2 stem forward / generic slow stem, band conv off
SFK_WGBAND=0 SFK_STEM3=0 SFK_HALO=0 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "stem or filter_gradient or conv_forward or data_gradient" > gpurun_out/test_gpu_kernels_old.log 2>&1
echo "test_gpu_kernels (SFK_WGBAND=0 SFK_STEM3=0 SFK_HALO=0) exit $?: $(tail -n 1 gpurun_out/test_gpu_kernels_old.log)" | tee -a gpurun_out/tests_summary.log
exit 0
