#!/bin/bash
# Round-3 experiments on the N = 32 element-view kernel (run on the GPU box via gpurun):
#   1. FETCH_SIZE / WRITE_SIZE calibrated on known byte counts in this kernel's access shapes (tools/micro/fetchcal)
#   2. residency sweep 8 / 6 / 5 / 4 / 3 gaits per CU (WG_TICK_LDS_PAD): ticks/s, and the traffic per gait-tick at 8, 4, 3
#   3. the Z stream marked non-temporal (lib/libwg_mpc_x1.so = -DWG_Z_NT=1; build it first:
#      make -C jrl-walkgen_amd lib/libwg_mpc_x1.so EXTRA=-DWG_Z_NT=1).  These rows were measured with round 2's 256-register
#      build of the element view (make ... EXTRA="-DWG_TICK32_WPE=2 -DWG_ELEM_GRP=8" reproduces it).
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
mkdir -p $R/gpurun_out
cd "$R"
./tools/micro/fetchcal > gpurun_out/fetchcal.txt 2>&1
bash tools/pmc_traffic.sh fetchcal $R/tools/micro/fetchcal > gpurun_out/fetchcal_pmc.txt 2>&1
echo "calibration done"
export PN=32 PB=8192 PT=50 PR=3
for pad in 0 6400 11000 20000 33000; do
  WG_TICK_LDS_PAD=$pad python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; }
done > gpurun_out/elem_residency.txt
echo "residency sweep done"
for pad in 0 20000 33000; do
  export WG_TICK_LDS_PAD=$pad
  bash tools/pmc_traffic.sh elem_pad$pad python3 $R/tools/probe_elem.py > gpurun_out/elem_pad${pad}_pmc.txt 2>&1
done
unset WG_TICK_LDS_PAD
echo "traffic passes done"
export WG_LIB_PATH=$R/jrl-walkgen_amd/lib/libwg_mpc_x1.so
python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } > gpurun_out/elem_nt.txt
bash tools/pmc_traffic.sh elem_nt python3 $R/tools/probe_elem.py > gpurun_out/elem_nt_pmc.txt 2>&1
unset WG_LIB_PATH
echo "nt done"
cat gpurun_out/fetchcal.txt gpurun_out/elem_residency.txt gpurun_out/elem_nt.txt
