cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/alltests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|Error" gpurun_out/alltests.log | cut -c1-250
for v in 1 0 1; do WG_QL_FIXED=$v timeout -k 10 200 python3 tools/probe_dense.py 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout -k 10 200 python3 tools/probe_dimitrov.py 2>&1 | grep -v amdgpu.ids | tail -3
