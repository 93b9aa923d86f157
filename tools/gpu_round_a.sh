#!/bin/bash
# round profile set, GPU session a: every GPU test, then tools/prof_round.sh (kernel traces, PMC passes, timers, launch fit)
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/alltests.log 2>&1 || { tail -30 gpurun_out/alltests.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/alltests.log
bash tools/prof_round.sh > gpurun_out/prof_round.log 2>&1 || { tail -20 gpurun_out/prof_round.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/prof_round.log | cut -c1-250
