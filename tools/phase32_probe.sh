#!/bin/bash
# N = 32 (element view) at the benchmark's residency: what running phase k of every active-set iteration twice costs
# (lib/libwg_mpc_xr<k>.so = -DWG_REPEAT_PHASE=k; 1 scan, 2 Z^T a, 3 norm chain, 4 back substitution, 5 route sums, 6 xmag, 7 pick_drop)
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
export PN=32 PB=8192 PT=30 PR=2 PMAXW=12
for k in 0 1 2 3 4 5 6 7; do
  export WG_LIB_PATH=$R/jrl-walkgen_amd/lib/libwg_mpc_xr$k.so
  echo "phase $k: $(python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-200)"
done
