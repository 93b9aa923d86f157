#!/bin/bash
# round 5, session a: new tests (stream-scoped host calls, m = 0, padded outs), full GPU suite, barrier micro, short bench
set -eu
R=${GRAFT_REPO_ROOT:?}
cd "$R"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/alltests.log 2>&1 || { tail -30 gpurun_out/alltests.log; exit 1; }
tail -3 gpurun_out/alltests.log
timeout -k 10 120 tools/micro/barrier > gpurun_out/barrier.txt 2>&1 || { tail gpurun_out/barrier.txt; exit 1; }
cat gpurun_out/barrier.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/benchshort.log 2>&1 || { tail -5 gpurun_out/benchshort.log | cut -c1-600; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/benchshort.log') if l.startswith('{')][-1])
print('value',d['value'],'outs_on',d['outs_on']['value'],d['outs_on']['delta_vs_value'],'config5',d['config5']['default']['value'])
print({k:v.get('value') for k,v in d['kernels'].items() if isinstance(v,dict) and 'value' in v})
PY
