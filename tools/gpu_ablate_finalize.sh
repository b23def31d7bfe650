for v in "" "bn_finalize" "" "bn_finalize"; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline ${v:+--ablate "$v"} > gpurun_out/ablate.log 2>&1
  echo "[skip: $v]: $(tail -n 1 gpurun_out/ablate.log | python -c "import sys,json; l=json.loads(sys.stdin.readline()); print(l['ms_per_step'], 'ms/step')")"
done
