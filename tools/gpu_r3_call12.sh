#!/bin/bash
set -u
mkdir -p gpurun_out/prof2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/prof2/*
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof2 -o b2 -- python bench.py --batch 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_b2_traced.log 2>&1
rc=$?; echo "rocprof exit $rc"; tail -n 1 gpurun_out/bench_b2_traced.log | cut -c1-300
