#!/bin/bash
# SFK_EXP sweep on layer micro-benchmarks: usage gpu_exp.sh "0 1 2" kind1 kind2 ...
VALS=$1; shift
for k in "$@"; do for v in $VALS; do echo -n "SFK_EXP=$v "; SFK_EXP=$v timeout -k 10 120 python tools/bench_layer.py $k 30 2>&1 | tail -n 1; done; done
