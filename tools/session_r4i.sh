cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_dimitrov_gpu.py tests/test_ql_gpu.py -m gpu -q -x -s > gpurun_out/dim.log 2>&1; echo "dim rc=$?"; grep -E "passed|failed|differ|^FAILED|PLDP vs QLD|Error|assert" gpurun_out/dim.log | cut -c1-300 | head -20
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/alltests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/alltests.log | cut -c1-250
for v in 1 0; do WG_QL_FIXED=$v timeout -k 10 200 python3 tools/probe_dense.py 2>&1 | grep -v amdgpu.ids | tail -1; done
for v in 1 0; do WG_QL_FIXED=$v timeout -k 10 200 python3 tools/probe_dense.py 2>&1 | grep -v amdgpu.ids | tail -1; done
