#!/bin/bash
# usage (GPU box): bash tools/ab_libs.sh <lib-suffix> [...]   -- the default library against experiment builds lib/libwg_mpc_<suffix>.so
# on the multi-tick kernels: N = 16 (B = 4096, 100 ticks per launch) and N = 32 (B = 8192, 50 ticks); same state checksum = same bits
set -u
cd $GRAFT_REPO_ROOT
for sfx in "" "$@"; do
  if [ -n "$sfx" ]; then export WG_LIB_PATH=$GRAFT_REPO_ROOT/jrl-walkgen_amd/lib/libwg_mpc_$sfx.so; else unset WG_LIB_PATH; fi
  echo "== lib ${sfx:-default}"
  PN=16 PB=4096 PT=100 PR=3 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids
  PN=32 PB=8192 PT=50 PR=3 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids
done
