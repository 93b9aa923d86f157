#!/bin/bash
# usage (GPU box): bash tools/ab_n16.sh <lib-suffix> [...]  -- N = 16 multi-tick kernel, B = 4096, launches of 100 ticks: the default
# library against experiment builds lib/libwg_mpc_<suffix>.so (same state checksum = same bits); then N = 32 at B = 8192
set -u
cd $GRAFT_REPO_ROOT
for sfx in "" "$@"; do
  if [ -n "$sfx" ]; then export WG_LIB_PATH=$GRAFT_REPO_ROOT/jrl-walkgen_amd/lib/libwg_mpc_$sfx.so; else unset WG_LIB_PATH; fi
  echo "== lib ${sfx:-default}"
  PN=16 PB=4096 PT=100 PR=3 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
  PN=32 PB=8192 PT=30 PR=2 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done
