#!/bin/bash
# round profile set, GPU session b: both phase attributions (N = 16, N = 32; needs lib/libwg_mpc_xr<k>.so), both bench lines, both soaks
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
mkdir -p gpurun_out
bash tools/phase_attribution.sh | cut -c1-120
ATTR_DIR=attr32 PN=32 PB=8192 PT=50 PR=2 bash tools/phase_attribution.sh | cut -c1-120
bash tools/bench_both.sh
timeout -k 10 600 python3 tools/soak_parity.py > gpurun_out/soak_parity.txt 2>&1 || { tail -5 gpurun_out/soak_parity.txt; exit 1; }
tail -1 gpurun_out/soak_parity.txt
SOAK_LONG=1 timeout -k 10 900 python3 tools/soak_parity.py > gpurun_out/soak_parity_long.txt 2>&1 || { tail -5 gpurun_out/soak_parity_long.txt; exit 1; }
tail -1 gpurun_out/soak_parity_long.txt
