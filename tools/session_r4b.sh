cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== new lib"; timeout -k 10 200 python3 tools/debug_two_streams.py 2>&1 | grep -v amdgpu.ids
echo "== old lib"; WG_LIB_PATH=$GRAFT_REPO_ROOT/jrl-walkgen_amd/lib/libwg_mpc_old.so timeout -k 10 200 python3 tools/debug_two_streams.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_ql_gpu.py::test_two_streams_share_one_context > gpurun_out/alltests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/alltests.log | cut -c1-300
