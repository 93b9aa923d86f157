#!/bin/bash
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
mkdir -p gpurun_out
timeout -k 10 900 python3 bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "bench rc=$?"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_window.json 2> gpurun_out/bench_window.err; echo "window rc=$?"
