#!/bin/bash
# The "Z on chip" experiment of docs/HISTORY.md 3.2 (N = 32, Z's rows in registers, mpc_tick<33>): parity tests, rate and counters of the
# experiment build against the default library.  Build first (CPU container):
#   make -C jrl-walkgen_amd lib/libwg_mpc_xregz.so EXTRA=-DWG_WITH_REGZ
#   make -C jrl-walkgen_amd lib/libwg_mpc_xregz2.so EXTRA="-DWG_WITH_REGZ -DWG_ZR_TILE=6 -DWG_ZR_WPS=2"      (two waves per SIMD)
# then on the GPU box:  bash tools/regz_probe.sh
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=$R/jrl-walkgen_amd/lib
echo "== parity: config 5 through the register-Z kernels (every tick against the oracle, cap hand-overs included)"
WG_LIB_PATH=$L/libwg_mpc_xregz.so WG_TICK_REGZ=1 timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py tests/test_run_gpu.py -m gpu -q -k "config5 or 32" 2>&1 | grep -E "passed|failed" | cut -c1-200
echo "== default library (Z through the fabric, twelve gaits per CU)"
PN=32 PB=8192 PT=50 PR=3 timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-260
echo "== Z in registers, one wave per SIMD (four gaits per CU)"
WG_LIB_PATH=$L/libwg_mpc_xregz.so WG_TICK_REGZ=1 PMAXW=4 PN=32 PB=8192 PT=50 PR=3 timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-260
if [ -f $L/libwg_mpc_xregz2.so ]; then
  echo "== Z in registers, two waves per SIMD (256 registers: spills), tile of 6 columns"
  WG_LIB_PATH=$L/libwg_mpc_xregz2.so WG_TICK_REGZ=1 PMAXW=8 PN=32 PB=8192 PT=50 PR=3 timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | { grep -v amdgpu.ids || true; } | tail -1 | cut -c1-260
fi
export WG_LIB_PATH=$L/libwg_mpc_xregz.so WG_TICK_REGZ=1 PMAXW=4 PN=32 PB=8192 PT=50 PR=3
bash tools/prof.sh regz tools/probe_elem.py > gpurun_out/prof_regz.log 2>&1
grep -A22 "^counters wg_mpc_run_xcd_kernel<33>" gpurun_out/prof_regz/summary.txt
