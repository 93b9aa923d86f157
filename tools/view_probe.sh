#!/bin/bash
# dense view (G and A as LDS matrices; WG_TICK_DENSE=1) against the element view (the default for every model but the
# benchmark's) at several horizons; B = 4096, multi-tick launches of 50 ticks.  Different views, same bits (state checksum).
# Horizons below 16 keep the 1.6 s preview window (T = 1.6 / N): with T = 0.1 they do not see the next step and their QPs fail.
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
export PB=4096 PT=50 PR=2 PMAXW=12
for spec in "4 0.4" "8 0.2" "12 0.125" "20 0.1" "24 0.1" "28 0.1"; do
  set -- $spec
  echo "== N=$1 T=$2 dense"; WG_TICK_DENSE=1 PN=$1 PQT=$2 python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-200
  echo "== N=$1 T=$2 element"; PN=$1 PQT=$2 python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-200
done
