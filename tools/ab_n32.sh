cd $GRAFT_REPO_ROOT
bash tools/ab_n32_quick.sh "$@"
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_configs45_gpu.py tests/test_run_gpu.py -m gpu -q -x 2>&1 | tail -3
