"""Where a kernel saves / restores the exec mask (lane-dependent branches), by source line and loop depth: each site costs a
scalar save, a branch and a restore on the spot (~27 cycles measured on the back substitution's path).
usage: hipcc ... -gline-tables-only -S --cuda-device-only -o /tmp/k.s csrc/wg_capi.hip; python tools/exec_sites.py /tmp/k.s <mangled-kernel-prefix> [min-depth]"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]; mind = int(sys.argv[3]) if len(sys.argv) > 3 else 1
start = next(i for i, l in enumerate(L) if l.startswith(pref))
end = next(i for i in range(start, len(L)) if 's_endpgm' in L[i])
files = {}
for l in L:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
depth = 0; loc = None; hits = collections.Counter()
for l in L[start:end]:
    m = re.search(r'Depth[= ](\d+)', l)
    if re.match(r'^\.LBB', l) or l.startswith('; %bb.'): depth = int(m.group(1)) if m else 0
    m2 = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m2: loc = (files.get(int(m2.group(1)), '?').split('/')[-1], int(m2.group(2)))
    if 's_and_saveexec' in l and depth >= mind: hits[(loc, depth)] += 1
for (lc, d), c in sorted(hits.items(), key=lambda x: (x[0][0] or ('', 0), x[0][1])):
    print("%-22s line %5d  depth %d  x%d" % (lc[0], lc[1], d, c))
