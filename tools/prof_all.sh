#!/bin/bash
# Round profiles (run on the GPU box via gpurun): tick kernel with PMC passes, then kernel-trace stats of the other kernels.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/prof.sh tick bench.py --steps 200 --warmup 50 --no-cpu-baseline > $R/gpurun_out/prof_tick.log 2>&1
cd /tmp
for spec in "dimitrov tools/probe_dimitrov.py" "pldp tools/probe_pldp.py" "preview tools/probe_preview.py"; do
  set -- $spec
  OUT=$R/gpurun_out/prof_$1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/$2 > $OUT/run.log 2>&1
  python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
  tail -3 $OUT/run.log; cat $OUT/summary.txt | head -4
done
PN=32 PB=8192 PT=30 python3 $R/tools/probe_tick.py 2>&1 | grep -v amdgpu.ids | head -3
tail -30 $R/gpurun_out/prof_tick.log
