cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ql_gpu.py tests/test_assemble_gpu.py -m gpu -q -x > gpurun_out/tql.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/tql.log | cut -c1-250
for v in 1 0 1 0; do WG_QL_FIXED=$v timeout -k 10 200 python3 tools/probe_dense.py 2>&1 | grep -v amdgpu.ids | tail -1; done
