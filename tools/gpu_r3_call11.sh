#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python tools/probe/host_issue.py 32 2 > gpurun_out/host_issue.log 2>&1; rc=$?; echo "host_issue exit $rc"; cat gpurun_out/host_issue.log | tail -5
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --graph --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.log 2>&1; rc=$?; echo "graph exit $rc"; tail -n 1 gpurun_out/bench_graph.log | cut -c1-300
