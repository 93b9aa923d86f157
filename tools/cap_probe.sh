#!/bin/bash
# N = 32 at more than eight gaits per CU: a 168-register build (the default since round 3; lib/libwg_mpc_x3.so:
#   make -C jrl-walkgen_amd lib/libwg_mpc_x3.so EXTRA=-DWG_TICK32_WPE=3)
# with the LDS part of R capped at fewer columns (WG_ELEM_NACT_CAP); a solve that outgrows them moves R to the global slot and
# goes on where it stopped (no repeat).  Same state checksum = same bits.
set -u
cd $GRAFT_REPO_ROOT
export PN=32 PB=8192 PT=50 PR=2 PMAXW=12
echo "== default build"; python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-190
export WG_LIB_PATH=$GRAFT_REPO_ROOT/jrl-walkgen_amd/lib/libwg_mpc_x3.so
for cap in 60 54 48 44 41 36 30; do
  echo "== x3, WG_ELEM_NACT_CAP=$cap"; WG_ELEM_NACT_CAP=$cap python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-190
done
