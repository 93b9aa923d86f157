cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/alltests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|differ|^FAILED" gpurun_out/alltests.log | cut -c1-250
timeout -k 10 600 bash tools/ab_n16.sh old > gpurun_out/ab_n16.log 2>&1; cut -c1-230 gpurun_out/ab_n16.log
timeout -k 10 600 bash tools/ab_n16.sh old > gpurun_out/ab_n16b.log 2>&1; cut -c1-230 gpurun_out/ab_n16b.log
PB=4096 timeout -k 10 300 python3 tools/probe_tick_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/phases_tick.txt; cat gpurun_out/phases_tick.txt
