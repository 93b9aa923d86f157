cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-config5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window', d['value'], d['outs_on']['value'], d['outs_on']['delta_vs_value'], d['outs_on']['launches'], d['outs_on']['first_tick'])"
done
timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-kernels --no-config5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('200', d['value'], d['outs_on']['value'], d['outs_on']['delta_vs_value'], d['outs_on']['launches'], d['outs_on']['first_tick'])"
