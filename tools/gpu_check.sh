#!/bin/bash
# Run on the GPU box (through gpurun): parity tests -> smoke -> short bench, logs under gpurun_out/.
# A step that is killed by its timeout (124/137) ends the script: no further GPU work after a hang.
set -u
mkdir -p gpurun_out
run() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/summary.log
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name exit $rc" | tee -a gpurun_out/summary.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/summary.log; exit $rc; fi
  return 0
}
: > gpurun_out/summary.log
STEPS=${1:-kernels,model,smoke,bench}
case ",$STEPS," in *,kernels,*) run kernels 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 180 -p no:cacheprovider ;; esac
case ",$STEPS," in *,model,*)   run model 900 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 600 -p no:cacheprovider ;; esac
case ",$STEPS," in *,smoke,*)   run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;; esac
case ",$STEPS," in *,bench,*)   run bench 900 python bench.py --gpus 1 --steps 10 --warmup 3 ;; esac
cat gpurun_out/summary.log
