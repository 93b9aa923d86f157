#!/bin/bash
# Per-phase attribution of the N = 16 tick's instruction streams and wait cycles (run on the GPU box via gpurun).
# Builds lib/libwg_mpc_xr<k>.so = -DWG_REPEAT_PHASE=k (csrc/wg_ql_device.hpp: the idempotent phase k of every active-set
# iteration is executed twice; k = 0: no phase repeated, same compiler barriers) must exist:
#   for k in 0 1 2 3 4 5 6 7 8 9 11; do make -C jrl-walkgen_amd lib/libwg_mpc_xr$k.so EXTRA=-DWG_REPEAT_PHASE=$k; done
# (8: factor(), 9: the tick's QP assembly, 11: gradient + residuals of the residual refresh -- idempotent as well)
# For each build: the multi-tick kernel's rate (no profiler), then ONE rocprofv3 --pmc pass of eight SQ counters.
# tools/phase_attribution.py turns the differences against k = 0 into the per-phase table.
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
export TMPDIR=/tmp
O=$R/gpurun_out/${ATTR_DIR:-attr}
rm -rf "$O"; mkdir -p "$O"
# ATTR_DIR=attr32 PN=32 PB=8192 PT=50 PR=2 bash tools/phase_attribution.sh  does the same for the N = 32 kernel
export PN=${PN:-16} PB=${PB:-4096} PT=${PT:-100} PR=${PR:-3}
cd /tmp
for k in 0 1 2 3 4 5 6 7 8 9 11; do
  export WG_LIB_PATH=$R/jrl-walkgen_amd/lib/libwg_mpc_xr$k.so
  python3 $R/tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } > $O/time_$k.txt
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d $O/pmc_$k -- python3 $R/tools/probe_elem.py > $O/pmc_$k.log 2>&1
  echo "phase $k done: $(grep ticks/s $O/time_$k.txt | cut -c1-160)"
done
