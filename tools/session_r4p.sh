cd $GRAFT_REPO_ROOT
PB=2048 timeout -k 10 300 python3 tools/probe_phases.py 2>&1 | grep -v amdgpu.ids
