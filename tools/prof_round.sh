#!/bin/bash
# Round profile set (run on the GPU box via gpurun; the PMC passes are separate runs, never combined with tracing):
#   tick      python3 bench.py --steps 200 --warmup 50 with nothing but the timed workload (no CPU baseline, no parity replay,
#             no per-tick leg, no config-5 leg): launches of wg_mpc_run_xcd_kernel<16> only, B = 4096, 48 + 4 x 50 ticks
#   tickg     the same with the device-wide queue (WG_RUN_QUEUE=global): the write-back traffic the XCD-local hand-over removed
#   pertick   the same workload with one launch per tick (wg_mpc_tick_kernel<16>)
#   config5   N = 32, B = 8192 (tools/probe_run.py)
#   gramian   tools/probe_gramian.py with the MFMA counters
#   dimitrov / pldp / preview / zmpdisc   kernel-trace stats + the standard passes
# gpurun MERGES gpurun_out/ back: remove the local gpurun_out/prof_* first.
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
export TMPDIR=/tmp
python3 "$R/tools/csrc_hash.py" > "$R/gpurun_out/csrc_hash.txt"       # the sources these profiles describe (filed with them)
BENCH="bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-parity --no-per-tick-leg --no-config5 --no-kernels --no-outs-leg"
bash $R/tools/prof.sh tick $BENCH > $R/gpurun_out/prof_tick.log 2>&1
echo "tick done"
# the same with the tick's deliverable stored in the timed launch (wg_tick_out_t per gait-tick): the WRITE_SIZE of the outs-on leg
bash $R/tools/prof.sh ticko $BENCH --outs-on > $R/gpurun_out/prof_ticko.log 2>&1
echo "ticko done"
export WG_RUN_QUEUE=global
bash $R/tools/prof.sh tickg $BENCH > $R/gpurun_out/prof_tickg.log 2>&1
unset WG_RUN_QUEUE
echo "tickg done"
bash $R/tools/prof.sh pertick $BENCH --per-tick-launch > $R/gpurun_out/prof_pertick.log 2>&1
echo "pertick done"
export PN=32 PB=8192 PT=50
bash $R/tools/prof.sh config5 tools/probe_run.py > $R/gpurun_out/prof_config5.log 2>&1
unset PN PB PT
echo "config5 done"
# the element-view run kernel alone (N = 32, B = 8192, launches of 50 ticks): the traffic bench.py's config5 leg reports
export PN=32 PB=8192 PT=50 PR=3
bash $R/tools/prof.sh elem tools/probe_elem.py > $R/gpurun_out/prof_elem.log 2>&1
unset PN PB PT PR
echo "elem done"
# FETCH_SIZE / WRITE_SIZE on known byte counts in the access shapes of that kernel
$R/tools/micro/fetchcal > $R/gpurun_out/fetchcal.txt 2>&1
bash $R/tools/pmc_traffic.sh fetchcal $R/tools/micro/fetchcal > $R/gpurun_out/fetchcal_pmc.txt 2>&1
echo "calibration done"
# one robot: what one wave alone on a CU spends its cycles on
bash $R/tools/prof.sh b1 tools/probe_b1.py > $R/gpurun_out/prof_b1.log 2>&1
echo "b1 done"
# one robot: the tick's latency split
$R/jrl-walkgen_amd/bin/latency_b1 > $R/gpurun_out/latency_b1.json 2> $R/gpurun_out/latency_b1.err
echo "latency done"
# the Gramian kernel with the matrix-core counters (its own passes)
OUT=$R/gpurun_out/prof_gramian; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/probe_gramian.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $R/tools/probe_gramian.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 $R/tools/probe_gramian.py > $OUT/pmc4.log 2>&1
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
echo "gramian done"
cd "$R"
for spec in "dimitrov tools/probe_dimitrov.py" "pldp tools/probe_pldp.py" "preview tools/probe_preview.py" "zmpdisc tools/probe_zmpdisc.py"; do
  set -- $spec
  bash $R/tools/prof.sh $1 $2 > $R/gpurun_out/prof_$1.log 2>&1
  echo "$1 done"
done
# in-kernel phase timers of the tick (diagnostic build lib/libwg_mpc_prof.so, one launch per tick)
PB=4096 python3 $R/tools/probe_tick_phases.py 2>&1 | { grep -v amdgpu.ids || true; } > $R/gpurun_out/phases_tick.txt
PN=32 PB=3072 python3 $R/tools/probe_tick_phases.py 2>&1 | { grep -v amdgpu.ids || true; } > $R/gpurun_out/phases_tick32.txt
echo "phases done"
# duration of a multi-tick launch against its length: the steady rate and what every launch pays once (ramp + idle tail)
python3 $R/tools/probe_launch_fit.py 2>&1 | { grep -v amdgpu.ids || true; } > $R/gpurun_out/launch_fit.txt
PTS=1,2,4,8,12,20 PN=32 PB=8192 python3 $R/tools/probe_launch_fit.py 2>&1 | { grep -v amdgpu.ids || true; } >> $R/gpurun_out/launch_fit.txt
echo "launch fit done"
tail -25 $R/gpurun_out/prof_tick.log
