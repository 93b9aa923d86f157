#!/bin/bash
# Round profiles (run on the GPU box via gpurun): tick kernel with PMC passes, then kernel-trace stats of the other kernels.
# gpurun MERGES gpurun_out/ back: remove the local gpurun_out/prof_* first, or tools/prof_summary.py averages old runs in.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/prof.sh tick bench.py --steps 200 --warmup 50 --no-cpu-baseline > $R/gpurun_out/prof_tick.log 2>&1
for spec in "dimitrov tools/probe_dimitrov.py" "pldp tools/probe_pldp.py" "preview tools/probe_preview.py" "zmpdisc tools/probe_zmpdisc.py"; do
  set -- $spec
  bash $R/tools/prof.sh $1 $2 > $R/gpurun_out/prof_$1.log 2>&1
  cp $R/gpurun_out/prof_$1/trace.log $R/gpurun_out/prof_$1/run.log 2>/dev/null
  head -3 $R/gpurun_out/prof_$1.log | cut -c1-300
done
PN=32 PB=8192 PT=30 python3 $R/tools/probe_tick.py 2>&1 | grep -v amdgpu.ids | head -3
tail -30 $R/gpurun_out/prof_tick.log
