#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in none ab abc abcd; do
  mkdir -p gpurun_out/lc_$v; rm -rf gpurun_out/lc_$v/*
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lc_$v -o t -- python tools/probe/lane_coupling.py $v > gpurun_out/lc_$v.log 2>&1
  rc=$?; echo "$v exit $rc"; tail -n 2 gpurun_out/lc_$v.log | cut -c1-200
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
