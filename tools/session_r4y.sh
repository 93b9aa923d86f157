cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/probe_outs.py 2>&1 | grep -v amdgpu.ids
PT=50 timeout -k 10 300 python3 tools/probe_outs.py 2>&1 | grep -v amdgpu.ids
