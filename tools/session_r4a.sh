cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/alltests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/alltests.log
timeout -k 10 600 bash tools/ab_elem.sh old xg6 > gpurun_out/ab_elem.log 2>&1; echo "ab rc=$?"; cat gpurun_out/ab_elem.log | cut -c1-260
WG_TICK_ELEM_GENERIC=1 PN=32 PB=8192 PT=50 PR=3 timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/benchshort.log 2>&1; echo "bench rc=$?"; tail -c 1500 gpurun_out/benchshort.log
