#!/bin/bash
# dense view (G and A as LDS matrices: the default for horizons other than 16 while they fit 160 KiB) against the element view
# (WG_TICK_VIEW=e) at several horizons; B = 4096, multi-tick launches of 50 ticks.  Different views, same bits (state checksum).
set -u
cd $GRAFT_REPO_ROOT
export PB=4096 PT=50 PR=2 PMAXW=12
for N in 8 12 20 24 28; do
  echo "== N=$N dense"; PN=$N python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
  echo "== N=$N element"; WG_TICK_VIEW=e PN=$N python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done
