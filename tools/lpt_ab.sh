#!/bin/bash
# longest-solve-first start order of the dense QP boundary and of the Dimitrov tick's QL back-end: the new tests, then
# same-box A/B against index order (WG_QL_LPT=0)
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ql_gpu.py tests/test_dimitrov_gpu.py tests/test_assemble_gpu.py -m gpu -q -x > gpurun_out/lpt_tests.log 2>&1 || { tail -30 gpurun_out/lpt_tests.log | cut -c1-250; exit 1; }
tail -2 gpurun_out/lpt_tests.log
{
for r in 1 2; do
  for lpt in 0 1; do
    WG_QL_LPT=$lpt timeout -k 10 200 python3 tools/probe_dense.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1
    PSAME=1 WG_QL_LPT=$lpt timeout -k 10 200 python3 tools/probe_dense.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1
    echo -n "dimitrov QLDANDLQ LPT=$lpt: "; PSOLVER=2 WG_QL_LPT=$lpt timeout -k 10 300 python3 tools/probe_dimitrov.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1
  done
done
} | tee gpurun_out/lpt_ab.txt
