#!/bin/bash
# the N = 32 passes of tools/prof_round.sh alone (element-view kernels changed, the rest of the set stands)
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
export TMPDIR=/tmp
export PN=32 PB=8192 PT=50
bash $R/tools/prof.sh config5 tools/probe_run.py > $R/gpurun_out/prof_config5.log 2>&1
echo "config5 done"
export PR=3
bash $R/tools/prof.sh elem tools/probe_elem.py > $R/gpurun_out/prof_elem.log 2>&1
unset PN PB PT PR
echo "elem done"
PN=32 PB=3072 python3 $R/tools/probe_tick_phases.py 2>&1 | { grep -v amdgpu.ids || true; } > $R/gpurun_out/phases_tick32.txt
echo "phases done"
