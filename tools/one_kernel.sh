#!/bin/bash
# usage: bash tools/one_kernel.sh <kernel> <NH> [extra hipcc flags...]    e.g.  tools/one_kernel.sh wg_mpc_run_xcd_kernel 32
# Compiles ONE instantiation of a tick kernel (csrc/wg_tick_kernels.hpp) for gfx950 in a scratch translation unit: its resource
# usage (registers, spills, scratch) on stdout, its ISA in /tmp/wg_one/<kernel>_<NH>.s -- seconds instead of the whole library.
# tools/isa_audit.py --file <that .s> places the spill code by loop depth.
set -eu
K=$1; NH=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=/tmp/wg_one; mkdir -p $O
T=$O/${K}_${NH}.hip
case $K in
  wg_mpc_tick_kernel) ARGS='int, wg_model_t, const wg::TickTables *, wg_gait_state_t *, wg_tick_out_t *, int *, int, int *, int, int *, unsigned, double *, unsigned, int, wg_gait_state_t *, int *, const int *, int *' ;;
  wg_mpc_run_kernel) ARGS='int, int, const wg_model_t *, const wg::TickTables *, wg_gait_state_t *, wg_tick_out_t *, int *, int, wg_run_queue *, int *, int *, unsigned, double *, unsigned, int' ;;
  wg_mpc_run_xcd_kernel) ARGS='int, int, const wg_model_t *, const wg::TickTables *, wg_gait_state_t *, wg_tick_out_t *, int *, int, wg_xrun_ctl *, unsigned long long *, int, int *, unsigned, double *, unsigned, const double *, int, int, int' ;;
  *) echo "unknown kernel $K"; exit 2 ;;
esac
cat > $T <<SRC
#include "$ROOT/jrl-walkgen_amd/csrc/wg_tick_kernels.hpp"
template __global__ void $K<$NH>($ARGS);
SRC
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed --cuda-device-only"
/opt/rocm/bin/hipcc $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -S -o $O/${K}_${NH}.s $T 2>&1 | grep -E "Function Name|VGPRs:|SGPRs|Spill|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
echo "ISA: $O/${K}_${NH}.s ($(wc -l < $O/${K}_${NH}.s) lines)"
