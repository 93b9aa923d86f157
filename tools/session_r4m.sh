cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
WG_TICK_REGZ=1 timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "config5 and not generic" > gpurun_out/t33.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|Error|assert" gpurun_out/t33.log | cut -c1-250 | head
echo "== regz"; WG_TICK_REGZ=1 PMAXW=4 PN=32 PB=8192 PT=50 PR=3 timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260
echo "== default"; PN=32 PB=8192 PT=50 PR=3 timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260
