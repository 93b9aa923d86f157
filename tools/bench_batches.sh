#!/bin/bash
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
for b in 1792 3584 4096 5376 7168 16384; do python bench.py --no-cpu-baseline --steps 100 --warmup 20 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($b, round(d['value']), d['roofline']['kernel_ms'])"; done
