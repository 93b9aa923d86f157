#!/bin/bash
# N = 16 multi-tick kernel: ticks/s against the batch size (2048 = one gait per resident wave) and the launch length
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
export PN=16 PR=3 PMAXW=8
for spec in "2048 50" "4096 50" "4096 200" "6144 50" "8192 50" "8192 200" "16384 50"; do
  set -- $spec
  PB=$1 PT=$2 python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-175
done
