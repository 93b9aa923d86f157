#!/bin/bash
# N = 16 (the benchmark horizon): the compact view (rows in registers, Z in LDS, 256 registers, six gaits per CU) against the
# element view (WG_TICK_VIEW=e: Z in the global slot, 168 registers, twelve per CU).  Same bits (state checksum).
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
export PN=16 PB=8192 PT=50 PR=3 PMAXW=12
echo "== N=16 compact";  python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-220
echo "== N=16 element";  WG_TICK_VIEW=e python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-220
export PB=32768
echo "== N=16 compact B=32768";  python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-220
echo "== N=16 element B=32768";  WG_TICK_VIEW=e python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-220
