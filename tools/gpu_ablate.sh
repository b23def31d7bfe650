#!/bin/bash
# What does the step time owe to each kernel class?  Runs the bench with one class SKIPPED by the lane scheduler (bench.py --ablate;
# the step's results are garbage, only the time means something) and prints ms/step beside the full step.
for v in "" "conv_wgrad" "bn_bwd_reduce,bn_bwd_apply" "bn_apply" "conv_dgrad" "conv_fwd" "stem_fwd,stem_wgrad" ""; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline ${v:+--ablate "$v"} > gpurun_out/ablate.log 2>&1
  echo "[skip: $v]: $(tail -n 1 gpurun_out/ablate.log | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['ms_per_step'], 'ms/step')")"
done
