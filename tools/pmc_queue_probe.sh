#!/bin/bash
# diagnostic: wave-lifetime counters of the multi-tick kernel for a given library (WG_LIB_PATH), one --pmc pass
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
export TMPDIR=/tmp
for v in "$@"; do
  export WG_LIB_PATH=$R/jrl-walkgen_amd/lib/$v
  OUT=$R/gpurun_out/pmcq_$v; rm -rf "$OUT"; mkdir -p "$OUT"
  cd /tmp
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $OUT -- python3 $R/tools/probe_run.py > $OUT/log.txt 2>&1 || exit 1
  cd "$R"
  python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    if "run" in k: print(k, {a: "%.3g" % b for a, b in c.items()})
PY
done
