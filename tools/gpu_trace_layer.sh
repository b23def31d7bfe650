#!/bin/bash
# per-kernel durations of one layer micro-benchmark: tools/gpu_trace_layer.sh KIND [env...]
set -u
KIND=$1; shift
mkdir -p gpurun_out/trl
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/trl/*
env "$@" true
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trl -o t -- python tools/bench_layer.py $KIND 10 > gpurun_out/trl/run.log 2>&1
python - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob('gpurun_out/trl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:70s} n={len(v):3d} median {v2[len(v2)//2]:8.1f} us  min {v2[0]:8.1f}")
PY
