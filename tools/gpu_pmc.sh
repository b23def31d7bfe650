#!/bin/bash
# PMC counters for one layer micro-benchmark (separate passes, no tracing domains besides kernel-trace)
# BENCH=tools/bench_stem.py bash tools/gpu_pmc.sh fwd_fast   for the stems; MATCH=substring of the kernel names to print
set -u
KIND=${1:-fwd_a4}
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH=${BENCH:-tools/bench_layer.py}
python $BENCH $KIND 50
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc -o p$i -- python $BENCH $KIND 5 > gpurun_out/pmc/run$i.log 2>&1
  echo "pass $i exit $?"
done
MATCH=${MATCH:-conv} python - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
for k, d in agg.items():
    if os.environ['MATCH'] not in k: continue
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v:.4g}")
PY
