# N = 32: the LDS part of R capped at c columns (WG_ELEM_NACT_CAP): residency against the share of solves that continue in the global slot
cd $GRAFT_REPO_ROOT
export PN=32 PB=8192 PT=50 PR=2
for cap in 0 60 54 48 44 41 36 30; do
  echo -n "cap $cap: "; WG_ELEM_NACT_CAP=$cap timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150
done
