#!/bin/bash
# usage (on the GPU box):  bash tools/pmc_traffic.sh <tag> <program> [args...]
# The two HBM-traffic passes only (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with tracing), summarised
# by tools/prof_summary.py into gpurun_out/prof_<tag>/summary.{txt,json}.  <program> is python3 or a binary: it is put
# directly after `--` (no env / bash -c hop under rocprofv3).
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=$1; shift
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- "$@" > $OUT/pmc4.log 2>&1
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
grep -h "ticks/s\|GB/s" $OUT/pmc3.log | tail -20
grep -A3 "^counters" $OUT/summary.txt | grep -v "^--"
