#!/bin/bash
# usage: bash tools/gpu_session.sh <step> [<step> ...]   -- runs the named steps in order on the GPU box; a step that was killed
# by its timeout ends the session (no further GPU step after a hang).  Logs go to gpurun_out/<step>.log.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {  # name, timeout seconds, command...
  local name=$1 lim=$2; shift 2
  echo "== $name"
  timeout -k 10 $lim "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "== $name rc=$rc"
  tail -4 gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
}
for step in "$@"; do
  case $step in
    newtests)   run newtests 900 python -m pytest tests/test_assemble_gpu.py tests/test_configs45_gpu.py -m gpu -x -q -k "assemble or robot or overlapping or config4" ;;
    alltests)   run alltests 1100 python -m pytest tests -m gpu -x -q ;;
    benchshort) run benchshort 600 python bench.py --steps 20 --warmup 5 ;;
    bench)      run bench 900 python bench.py ;;
    latency)    run latency 120 ./jrl-walkgen_amd/bin/latency_b1 ;;
    *) echo "unknown step $step" ;;
  esac
done
