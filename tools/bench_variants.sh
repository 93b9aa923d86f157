#!/bin/bash
# bench.py over experiment builds / launch shapes of the tick kernel (diagnostic; results under gpurun_out/)
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
mkdir -p gpurun_out
out=gpurun_out/variants.txt
: > $out
run() {  # label, lib, waves_per_cu
  echo "== $1 lib=$2 wpc=$3" >> $out
  WG_TICK_LDS_PAD=${4:-0} WG_LIB_PATH=$PWD/jrl-walkgen_amd/lib/$2 python bench.py --per-tick-launch --no-per-tick-leg --no-cpu-baseline --steps 100 --warmup 20 2>&1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])" >> $out 2>&1
}
for spec in "$@"; do IFS=: read lib wpc pad <<< "$spec"; run "$spec" "$lib" "$wpc" "$pad"; done
cat $out
