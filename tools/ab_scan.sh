cd $GRAFT_REPO_ROOT
for L in libwg_mpc.so libwg_mpc_xsc8.so libwg_mpc_xsc16.so libwg_mpc.so libwg_mpc_xsc8.so; do
  echo -n "$L: "; PN=32 PB=8192 PT=50 WG_LIB_PATH=$PWD/jrl-walkgen_amd/lib/$L timeout -k 10 300 python3 tools/probe_run.py 2>&1 | grep -v amdgpu.ids | tail -1
done
