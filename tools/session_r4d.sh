cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/alltests_$i.log 2>&1; echo "run $i rc=$?"; grep -E "passed|failed|differ|^FAILED" gpurun_out/alltests_$i.log | cut -c1-250; done
