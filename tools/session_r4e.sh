cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
WG_DEBUG_QP=1 timeout -k 10 600 python -m pytest tests -m gpu -q -s -x > gpurun_out/alltests_dbg.log 2>&1; echo "rc=$?"
grep -n "wg qp\|differ\|passed\|failed" gpurun_out/alltests_dbg.log | tail -30 | cut -c1-200
