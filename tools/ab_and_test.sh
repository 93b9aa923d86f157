cd $GRAFT_REPO_ROOT
bash tools/ab_libs2.sh "$@"
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ql_gpu.py tests/test_tick_gpu.py tests/test_run_gpu.py tests/test_assemble_gpu.py -m gpu -q -x 2>&1 | tail -3
