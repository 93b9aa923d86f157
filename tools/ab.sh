#!/bin/bash
# usage (GPU box): bash tools/ab.sh [-n "16 32"] [-r ROUNDS] [-s "N T ..."] [-t] LIB [LIB ...]
# Interleaved same-box A/B of library builds on the multi-tick kernel (tools/probe_elem.py): N = 16 at B = 4096 with 100-tick
# launches, N = 32 at B = 8192 with 50-tick launches; every line ends in the state checksum (same checksum = same bits).
#   LIB       a file under jrl-walkgen_amd/lib/ (libwg_mpc.so, libwg_mpc_x1.so ...) or a bare suffix (x1 -> libwg_mpc_x1.so); "default" =
#             libwg_mpc.so.  Experiment builds: make -C jrl-walkgen_amd lib/libwg_mpc_x1.so EXTRA="-D..."
#   -n LIST   horizons to run (default "16 32"; "" for none)
#   -s SPEC   other horizons as "N T" pairs, e.g. -s "4 0.4 8 0.2 12 0.125" (B = 4096, 50-tick launches)
#   -r K      rounds of the interleaved series (default 2)
#   -t        the element-view / full-size GPU tests afterwards
# (replaces round 4's ab_both / ab_elem / ab_libs / ab_libs2 / ab_n16 / ab_n32 / ab_n32_quick / ab_small_n)
set -eu
R=${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
cd "$R"
HORIZONS="16 32"; ROUNDS=2; SMALL=""; TESTS=0
while getopts "n:r:s:t" o; do
  case $o in n) HORIZONS=$OPTARG ;; r) ROUNDS=$OPTARG ;; s) SMALL=$OPTARG ;; t) TESTS=1 ;; *) exit 2 ;; esac
done
shift $((OPTIND - 1))
[ $# -ge 1 ] || { echo "no library given"; exit 2; }
libpath() { case $1 in default) echo "$R/jrl-walkgen_amd/lib/libwg_mpc.so" ;; *.so) echo "$R/jrl-walkgen_amd/lib/$1" ;; *) echo "$R/jrl-walkgen_amd/lib/libwg_mpc_$1.so" ;; esac; }
probe() {  # label, then env assignments for probe_elem.py
  local label=$1; shift
  echo -n "$label: "
  env "$@" timeout -k 10 300 python3 tools/probe_elem.py 2>&1 | grep -v amdgpu.ids | grep -o "\-> [0-9]* ticks/s.*checksum [0-9a-f]*" || echo "FAILED"
}
read -r -a small <<< "$SMALL"
for r in $(seq 1 "$ROUNDS"); do
  for L in "$@"; do
    P=$(libpath "$L"); [ -f "$P" ] || { echo "$P missing"; exit 1; }
    for N in $HORIZONS; do
      case $N in
        16) probe "N=16 $L" PN=16 PB=4096 PT=100 PR=3 WG_LIB_PATH="$P" ;;
        32) probe "N=32 $L" PN=32 PB=8192 PT=50 PR=2 WG_LIB_PATH="$P" ;;
        *)  probe "N=$N $L" PN="$N" PB=4096 PT=50 PR=2 WG_LIB_PATH="$P" ;;
      esac
    done
    for ((i = 0; i + 1 < ${#small[@]}; i += 2)); do
      probe "N=${small[i]} T=${small[i+1]} $L" PN="${small[i]}" PQT="${small[i+1]}" PB=4096 PT=50 PR=2 WG_LIB_PATH="$P"
    done
  done
done
if [ "$TESTS" = 1 ]; then
  timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_configs45_gpu.py tests/test_run_gpu.py -m gpu -q -x 2>&1 | tail -3
fi
