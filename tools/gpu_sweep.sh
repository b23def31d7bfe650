#!/bin/bash
# A/B sweep of tuning knobs on ONE box: every configuration = one default bench run (10 steps); baseline first, last and in the middle.
set -u
mkdir -p gpurun_out
run() { name=$1; shift
  env SFK_X=1 "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/sw_$name.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "$name failed $rc"; tail -n 3 gpurun_out/sw_$name.log; return 0; fi
  echo "$name: $(tail -n 1 gpurun_out/sw_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
}
run base
run wgtg_256 SFK_WGTG=256
run wgtg_384 SFK_WGTG=384
run wgtg_512 SFK_WGTG=512
run wgtg_640 SFK_WGTG=640
run base_mid
run bnparts_2048 SFK_BN_PARTS=2048
run bnparts_4096 SFK_BN_PARTS=4096
run wgtg512_bn2048 SFK_WGTG=512 SFK_BN_PARTS=2048
run wgtg384_bn2048 SFK_WGTG=384 SFK_BN_PARTS=2048
run wgtg512_b SFK_WGTG=512
run base_end
