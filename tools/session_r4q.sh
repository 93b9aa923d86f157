cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/phase_attribution.sh > gpurun_out/attr.log 2>&1; tail -12 gpurun_out/attr.log | cut -c1-200
python3 tools/phase_attribution.py > gpurun_out/phase_attribution.txt 2>&1; cat gpurun_out/phase_attribution.txt | cut -c1-220
