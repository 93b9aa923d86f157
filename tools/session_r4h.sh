cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PB=4096 timeout -k 10 300 python3 tools/probe_tick_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/phases_tick.txt; grep -A8 "^2[123] " gpurun_out/phases_tick.txt | cut -c1-120
timeout -k 10 200 ./jrl-walkgen_amd/bin/latency_b1 > gpurun_out/latency_b1.json 2> gpurun_out/latency_b1.err; cat gpurun_out/latency_b1.json | cut -c1-1500
timeout -k 10 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print('value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
print('outs_on', d.get('outs_on'))
print('per_tick', d.get('per_tick_launch'))
for k,v in d.get('kernels',{}).items():
    print(k, {kk:vv for kk,vv in v.items() if isinstance(vv,(int,float))} if isinstance(v,dict) else v)
print('config5', {k:(v.get('value'), v.get('roofline',{}).get('frac')) for k,v in d['config5'].items() if isinstance(v,dict)})
print('parity', {k:v for k,v in d['parity'].items() if not isinstance(v,(dict,str))}, d['parity'].get('b1_tick_latency_us'))
PY
