#!/bin/bash
# copy the summaries of the last tools/prof_all.sh run (merged back under gpurun_out/) into profiles/ as <tag>_*
TAG=${1:-round1_s3}
for k in tick dimitrov pldp preview zmpdisc; do
  python tools/prof_summary.py gpurun_out/prof_$k > gpurun_out/prof_$k/summary.txt
  cp gpurun_out/prof_$k/summary.txt profiles/${TAG}_${k}_rocprofv3_summary.txt
  cp gpurun_out/prof_$k/summary.json profiles/${TAG}_${k}_rocprofv3_summary.json
  cp $(ls -t gpurun_out/prof_$k/trace/runc/*_kernel_stats.csv | head -1) profiles/${TAG}_${k}_kernel_stats.csv
done
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
s = json.load(open(f'profiles/{tag}_tick_rocprofv3_summary.json'))
k = 'wg_mpc_run_kernel<16>'
h = s['hbm'][k]
out = {"kernel": k, "hbm_bytes_per_launch": h['hbm_bytes_per_launch'],
       "read_bytes_per_launch_corrected": h['read_bytes_per_launch_corrected'],
       "write_bytes_per_launch": h['write_bytes_per_launch'],
       "ticks_per_launch": 49.6,
       "known_bytes_per_tick": {"state_read": 2 * 4947968, "state_write": 2 * 4947968, "diag_write": 98304},
       "source": f"profiles/{tag}_tick_rocprofv3_summary.json: separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of "
                 "`python bench.py --steps 200 --warmup 50 --no-cpu-baseline` (B=4096 gaits, 50 ticks per launch of the multi-tick "
                 "kernel; the mean is over one 48-tick and four 50-tick launches), tools/prof.sh. Units and "
                 "correction per MI355X_MICROARCH.md: KiB x1024; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B). The "
                 "state is written twice and read twice per tick (it is parked in its HBM slot during the solve to free LDS); "
                 "WRITE_SIZE equals those bytes + diagnostics (no scratch: the kernel has no spills).",
       "avg_kernel_ns_rocprofv3": s['kernels'][k]['avg_ns']}
json.dump(out, open('profiles/round1_pmc_summary.json', 'w'), indent=1)
print(out['hbm_bytes_per_launch'], out['read_bytes_per_launch_corrected'], out['write_bytes_per_launch'], out['avg_kernel_ns_rocprofv3'])
PY
