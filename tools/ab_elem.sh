#!/bin/bash
# usage (GPU box): bash tools/ab_elem.sh <lib-suffix> [...]  -- N = 32 multi-tick kernel only: the default library against the
# experiment builds lib/libwg_mpc_<suffix>.so (same state checksum = same bits)
set -u
cd $GRAFT_REPO_ROOT
for sfx in "" "$@"; do
  if [ -n "$sfx" ]; then export WG_LIB_PATH=$GRAFT_REPO_ROOT/jrl-walkgen_amd/lib/libwg_mpc_$sfx.so; else unset WG_LIB_PATH; fi
  echo "== lib ${sfx:-default}"
  PN=32 PB=8192 PT=50 PR=3 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1
done
