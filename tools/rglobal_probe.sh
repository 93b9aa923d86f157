#!/bin/bash
# What would R outside the LDS cost / buy at N = 32?  (build first: make -C jrl-walkgen_amd lib/libwg_mpc_x3.so EXTRA=-DWG_TICK32_WPE=3;
# measured in round 3 BEFORE the hand-over became a continuation: WG_ELEM_ABORT_AT then meant "repeat from scratch")  (A) default build, every solve repeated with R in the global slot after one
# iteration (WG_ELEM_ABORT_AT=1): the price of R (and the working column) in global memory at unchanged residency;
# (B) 168-register build with a small LDS part of R (WG_ELEM_NACT_CAP): twelve gaits per CU, most solves aborted and repeated.
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
export PN=32 PB=8192 PT=50 PR=2 PMAXW=12
echo "== default"; python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-200
echo "== default, WG_ELEM_ABORT_AT=1"; WG_ELEM_ABORT_AT=1 python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-200
export WG_LIB_PATH=$R/jrl-walkgen_amd/lib/libwg_mpc_x3.so
for cap in 54 48 40 30; do
  echo "== x3, WG_ELEM_NACT_CAP=$cap"; WG_ELEM_NACT_CAP=$cap python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-200
done
