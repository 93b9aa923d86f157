#!/bin/bash
# Does the element view keep scaling past eight gaits per CU?  N = 20 forced onto the element view (n <= 48: 13 KB of LDS per gait)
# with the 168-register build (since round 3 the default: WG_TICK32_WPE=3, WG_ELEM_GRP=4; lib/libwg_mpc_x3.so may be any build of it:
#   make -C jrl-walkgen_amd lib/libwg_mpc_x3.so EXTRA=-DWG_TICK32_WPE=3), residency lowered by LDS padding.
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
export WG_LIB_PATH=$R/jrl-walkgen_amd/lib/libwg_mpc_x3.so WG_TICK_VIEW=e PN=20 PB=8192 PT=50 PR=2 PMAXW=12
for pad in 0 1280 2560 3840 5120 7680; do
  WG_TICK_LDS_PAD=$pad python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-200
done
