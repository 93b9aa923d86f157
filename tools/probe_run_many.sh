#!/bin/bash
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
for spec in "4096 50" "4096 50" "1000 30" "7000 20" "300 100"; do set -- $spec; PB=$1 PT=$2 timeout -k 10 200 python tools/probe_run.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1; done
