cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/phase_attribution.sh | cut -c1-120
cd $GRAFT_REPO_ROOT
ATTR_DIR=attr32 PN=32 PB=8192 PT=50 PR=2 bash tools/phase_attribution.sh | cut -c1-120
cd $GRAFT_REPO_ROOT
bash tools/bench_both.sh
timeout -k 10 600 python3 tools/soak_parity.py > gpurun_out/soak_parity.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/soak_parity.txt
SOAK_LONG=1 timeout -k 10 900 python3 tools/soak_parity.py > gpurun_out/soak_parity_long.txt 2>&1; echo "long soak rc=$?"; tail -1 gpurun_out/soak_parity_long.txt
