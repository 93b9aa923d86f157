#!/bin/bash
# diagnostic: instruction-cache counters of the tick kernels at a given horizon (PN) and batch (PB): one --pmc pass
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_icache; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $OUT -- python3 $R/tools/probe_run.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
cd "$R"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:44]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    if "mpc" in k: print(k, {a: "%.4g" % b for a, b in c.items()})
PY
