#!/bin/bash
# the multi-tick kernel's hand-over policy: a wave keeps a gait that is behind its XCD's mean progress (default, WG_RUN_KEEP=0)
# against the plain first-in first-out ring (WG_RUN_KEEP=off); N = 16 and N = 32, several batch sizes.  Same state checksum.
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
export PR=3
for spec in "16 2048 50 8" "16 4096 50 8" "16 4096 200 8" "16 8192 50 8" "32 8192 50 12" "32 3072 50 12"; do
  set -- $spec
  for keep in off 0 1; do
    echo "== N=$1 B=$2 T=$3 WG_RUN_KEEP=$keep"
    WG_RUN_KEEP=$keep PN=$1 PB=$2 PT=$3 PMAXW=$4 python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-230
  done
done
