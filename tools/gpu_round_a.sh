cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/alltests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|Error" gpurun_out/alltests.log | cut -c1-250
bash tools/prof_round.sh > gpurun_out/prof_round.log 2>&1; tail -3 gpurun_out/prof_round.log
