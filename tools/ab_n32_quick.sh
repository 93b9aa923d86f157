cd $GRAFT_REPO_ROOT
for r in 1 2; do for L in "$@"; do
  echo -n "$L: "; PN=32 PB=8192 PT=50 PR=2 WG_LIB_PATH=$PWD/jrl-walkgen_amd/lib/$L timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | grep -o "\-> [0-9]* ticks/s.*checksum [0-9a-f]*"
done; done
