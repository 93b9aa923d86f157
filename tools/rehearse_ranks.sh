#!/bin/bash
# Rehearsal of bench.py's multi-rank bookkeeping on a ONE-GPU box: N ranks (at most 4), all on card 0, gloo instead of
# RCCL (RCCL refuses two ranks on one device).  Checks rank-offset workloads, barriers, max/sum over ranks and the single
# JSON line; it says nothing about xGMI.  usage: bash tools/rehearse_ranks.sh [N]
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
N=${1:-2}
export WG_BENCH_BACKEND=gloo WG_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus $N --steps 100 --warmup 50
