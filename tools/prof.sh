#!/bin/bash
# usage (on the GPU box, via gpurun):  bash tools/prof.sh <tag> <script.py> [args...]
# 1) kernel trace + stats, 2) PMC passes -- counters are collected in their own runs, never with tracing.
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=$1; shift
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
SCRIPT=$R/$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $SCRIPT "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc1 -- python3 $SCRIPT "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/pmc2 -- python3 $SCRIPT "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $SCRIPT "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 $SCRIPT "$@" > $OUT/pmc4.log 2>&1
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
