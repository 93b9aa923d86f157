#!/bin/bash
# per gait-tick instruction counts of the multi-tick kernel at HEAD (N = 16, B = 4096, 10 + 3 x 100 ticks): one rocprofv3 --pmc pass
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
export TMPDIR=/tmp
O="$R/gpurun_out/pmcq"; rm -rf "$O"; mkdir -p "$O"
export PN=${PN:-16} PB=${PB:-4096} PT=${PT:-100} PR=${PR:-3}
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O -- python3 $R/tools/probe_elem.py > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/pmcq"
tot=collections.Counter()
for f in glob.glob(O+"/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wg_mpc_run_xcd_kernel" in r["Kernel_Name"]: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
gt=int(os.environ["PB"])*(10+int(os.environ["PR"])*int(os.environ["PT"]))   # probe_elem.py: one 10-tick warm-up launch + PR launches of PT ticks
print({k: round(v/gt,1) for k,v in tot.items()})
PY
