cd $GRAFT_REPO_ROOT
export PB=4096 PT=50 PR=2
for spec in "4 0.4" "8 0.2" "12 0.125"; do set -- $spec
  for L in libwg_mpc_old.so libwg_mpc.so; do
    echo -n "N=$1 $L: "; PN=$1 PQT=$2 WG_LIB_PATH=$PWD/jrl-walkgen_amd/lib/$L timeout -k 10 200 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/launches.*-> //' | cut -c1-150
  done
done
