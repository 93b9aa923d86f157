cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export WG_TICK_REGZ=1 PMAXW=4 PN=32 PB=8192 PT=50 PR=3
bash tools/prof.sh regz tools/probe_elem.py > gpurun_out/prof_regz.log 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/prof_regz/summary.json'))
for k,c in d['counters'].items():
    if 'run_xcd' in k:
        n=c['FETCH_SIZE']['launches']; gt=8192*160
        rd=c['FETCH_SIZE']['mean_per_launch']*n*2048; wr=c['WRITE_SIZE']['mean_per_launch']*n*1024
        print(k,'read %.3f MB write %.3f MB per gait-tick'%(rd/gt/1e6,wr/gt/1e6))
        for cn in sorted(c):
            print('   %-28s %.5g per gait-tick'%(cn,c[cn]['mean_per_launch']*c[cn]['launches']/gt))
PY
