#!/bin/bash
# HBM traffic of the bench's kernels from the PMC counters, collected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only), unit = KiB, and FETCH_SIZE doubled
# (gfx950 tallies a 128-B read request as 64 B for wide coalesced streams).  Writes gpurun_out/pmc_traffic.json;
# copy it to profiles/rNN_pmc_traffic.json (bench.py reports `roofline.traffic` from the newest one).
set -u
OUT=gpurun_out/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
STEPS=${STEPS:-2}; WARM=${WARM:-1}
i=0
for ctr in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT -o pass_$ctr -- \
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline > $OUT/run_$ctr.log 2>&1
  rc=$?
  echo "pass $ctr exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit $rc; fi
done
python tools/traffic_agg.py $OUT $(python tools/tree_hash.py) > gpurun_out/pmc_traffic.json
rc=$?
head -c 1500 gpurun_out/pmc_traffic.json; echo
# the raw per-dispatch tables are large; keep only the aggregate
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
exit $rc
