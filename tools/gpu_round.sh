#!/bin/bash
# one GPU round trip: kernel parity tests -> quick eager bench with per-layer timings -> (optional) PMC traffic passes
set -u
mkdir -p gpurun_out
bash tools/gpu_quick.sh
rc=$?
if [ $rc -ne 0 ]; then exit $rc; fi
if [ "${TRAFFIC:-0}" = "1" ]; then bash tools/gpu_traffic.sh; fi
