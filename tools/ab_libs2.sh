# usage: bash tools/ab_libs2.sh libA.so libB.so   -- interleaved same-box A/B of the multi-tick kernel (N = 16, B = 4096, 100-tick launches), state checksums printed
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for L in "$@"; do
    echo -n "$L: "; PN=16 PB=4096 PT=100 PR=3 WG_LIB_PATH=$PWD/jrl-walkgen_amd/lib/$L timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | grep -o "\-> [0-9]* ticks/s.*checksum [0-9a-f]*" 
  done
done
