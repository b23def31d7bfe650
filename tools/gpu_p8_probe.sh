#!/bin/bash
# main-loop rate of the deep-pipelined conv tile on full-chip synthetic shapes + PMC counters
set -u
mkdir -p gpurun_out
for k in fwd_x1024,256,3,1,1,32,8,16,16 fwd_x4096,256,3,1,1,32,8,16,16 fwdns_x1024,1024,1,1,1,32,8,16,16 dgrad_x1024,256,3,1,1,32,8,16,16; do
  for v in 0 ${P8V:-3}; do
    echo "P8=$v $(SFK_P8=$v SFK_LIB=${SFK_LIB:-} timeout -k 10 120 python tools/bench_layer.py $k 30 2>&1 | tail -n 1)"
  done
done
if [ "${PMC:-1}" = "1" ]; then SFK_P8=${P8V:-3} bash tools/gpu_pmc.sh fwd_x1024,256,3,1,1,32,8,16,16 2>&1 | tail -n 40; fi
