cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python3 tools/probe_launch_fit.py 2>&1 | grep -v amdgpu.ids
PT=20 WG_LIB_PATH=$PWD/jrl-walkgen_amd/lib/libwg_mpc_xs.so timeout -k 10 200 python3 tools/xrun_stats.py 2>&1 | grep -v amdgpu.ids
PTS=1,2,4,8,12,20 PN=32 PB=8192 timeout -k 10 300 python3 tools/probe_launch_fit.py 2>&1 | grep -v amdgpu.ids
