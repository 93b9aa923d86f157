cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PT=20 PSAVE=$PWD/gpurun_out/xrun_4096_20.npz WG_LIB_PATH=$PWD/jrl-walkgen_amd/lib/libwg_mpc_xs.so timeout -k 10 200 python3 tools/xrun_stats.py 2>&1 | grep -v amdgpu.ids | tail -3
