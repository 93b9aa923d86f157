#!/bin/bash
# usage (GPU box): [LIB=x3] bash tools/cap_probe.sh [caps...]
# N = 32: the LDS part of R capped at c columns (WG_ELEM_NACT_CAP; 0 = the host's own choice): residency against the share of
# solves that continue in the global slot (a solve that outgrows the columns moves R there and goes on where it stopped).
# LIB selects an experiment build lib/libwg_mpc_<LIB>.so (e.g. make -C jrl-walkgen_amd lib/libwg_mpc_x3.so EXTRA=-DWG_TICK32_WPE=3).
# Same state checksum = same bits.  (round 4's cap_probe.sh and cap_probe2.sh in one)
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
export PN=32 PB=8192 PT=50 PR=2
if [ -n "${LIB:-}" ]; then export WG_LIB_PATH="$R/jrl-walkgen_amd/lib/libwg_mpc_$LIB.so"; fi
[ $# -ge 1 ] || set -- 0 60 54 48 44 41 36 30
for cap in "$@"; do
  echo -n "cap $cap: "; WG_ELEM_NACT_CAP=$cap timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150
done
