cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do timeout -k 10 200 python -m pytest tests/test_ql_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|differ" | cut -c1-200; done
echo "== old lib"
for i in 1 2 3; do WG_LIB_PATH=$GRAFT_REPO_ROOT/jrl-walkgen_amd/lib/libwg_mpc_old.so timeout -k 10 200 python -m pytest tests/test_ql_gpu.py -m gpu -q -x --deselect tests/test_ql_gpu.py::test_two_host_threads_share_one_context 2>&1 | grep -E "passed|failed|differ" | cut -c1-200; done
