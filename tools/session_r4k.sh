cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
BENCH="bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-parity --no-per-tick-leg --no-config5 --no-kernels --no-outs-leg"
bash tools/prof.sh tick $BENCH > gpurun_out/prof_tick.log 2>&1; echo "tick done"
bash tools/prof.sh ticko $BENCH --outs-on > gpurun_out/prof_ticko.log 2>&1; echo "ticko done"
export PN=32 PB=8192 PT=50 PR=3
bash tools/prof.sh elem tools/probe_elem.py > gpurun_out/prof_elem.log 2>&1; echo "elem done"
unset PN PB PT PR
bash tools/prof.sh b1 tools/probe_b1.py > gpurun_out/prof_b1.log 2>&1; echo "b1 done"
grep -h "^hbm\|^kernel wg_mpc" gpurun_out/prof_tick/summary.txt gpurun_out/prof_ticko/summary.txt gpurun_out/prof_elem/summary.txt | cut -c1-400
cat gpurun_out/prof_b1/summary.txt | cut -c1-200
