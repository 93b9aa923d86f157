cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py tests/test_configs45_gpu.py tests/test_run_gpu.py -m gpu -q -x > gpurun_out/t32.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/t32.log | cut -c1-250
timeout -k 10 600 bash tools/ab_elem.sh old > gpurun_out/ab_elem.log 2>&1; cut -c1-260 gpurun_out/ab_elem.log
PN=32 PB=8192 PT=50 PR=3 bash tools/pmc_traffic.sh elem2 python3 tools/probe_elem.py > gpurun_out/pmc_elem2.log 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/prof_elem2/summary.json'))
for k,v in d['counters'].items():
    if 'run_xcd' in k:
        n=v['FETCH_SIZE']['launches']; gt=8192*160
        rd=v['FETCH_SIZE']['mean_per_launch']*n*2048; wr=v['WRITE_SIZE']['mean_per_launch']*n*1024
        print(k,'read %.3f MB write %.3f MB per gait-tick'%(rd/gt/1e6,wr/gt/1e6))
PY
