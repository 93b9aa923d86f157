cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_dimitrov_gpu.py -m gpu -q -x -s > gpurun_out/dim.log 2>&1; echo "dim rc=$?"; grep -E "passed|failed|^FAILED|PLDP vs|Error|assert " gpurun_out/dim.log | cut -c1-400 | head -20
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/alltests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/alltests.log | cut -c1-250
