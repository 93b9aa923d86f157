#!/bin/bash
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
timeout -k 10 500 python bench.py --no-cpu-baseline --steps 1000 --warmup 50 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1000 steps', d['value'], d['solver']['failed_qps'], d['per_tick_launch']['value'])"
PB=100 PT=400 timeout -k 10 300 python tools/probe_run.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1
PB=1793 PT=60 timeout -k 10 300 python tools/probe_run.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1
PB=20000 PT=15 timeout -k 10 300 python tools/probe_run.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1
