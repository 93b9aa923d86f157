cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_dimitrov_gpu.py tests/test_pldp_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print({n:(v['value'] if isinstance(v,dict) and 'value' in v else None) for n,v in k.items()})"
