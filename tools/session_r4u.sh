cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in libwg_mpc.so libwg_mpc_xw2.so libwg_mpc.so; do
  PN=32 PB=8192 PT=50 WG_LIB_PATH=$PWD/jrl-walkgen_amd/lib/$L timeout -k 10 300 python3 tools/probe_run.py 2>&1 | grep -v amdgpu.ids | tail -1
done
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-config5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window', d['value'], d.get('outs_on',{}).get('value'))"
done
