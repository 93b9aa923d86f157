#!/usr/bin/env python3
"""Where a kernel's VGPR spill code sits, by source line: python tools/scratch_sites.py /tmp/wg_one/<kernel>.s  (an ISA file from
tools/one_kernel.sh <kernel> <NH> -gline-tables-only).  Each scratch_ instruction with the loop depth of its block (the compiler's
own annotations) and the last .loc in front of it."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().splitlines()
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
cur = None; depth = 0; out = []
i = 0
while i < len(lines):
    s = lines[i].strip()
    m = re.match(r'^(\.LBB\d+_\d+|; %bb\.\d+):(.*)$', s)
    if m:
        cmt = m.group(2); j = i + 1
        while j < len(lines) and lines[j].strip().startswith(';'):
            cmt += ' ' + lines[j].strip(); j += 1
        ds = [int(x) for x in re.findall(r'Depth=(\d+)', cmt)]
        depth = max(ds) if ds else 0
        i = j; continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)\s+(\d+)', s)
    if m:
        cur = (files.get(int(m.group(1)), '?').split('/')[-1], int(m.group(2)))
    if s.startswith('scratch_'):
        off = re.search(r'offset:(\d+)', s)
        out.append((depth, 'load' if 'load' in s.split()[0] else 'store', cur, int(off.group(1)) if off else 0))
    i += 1
c = Counter((d, k, loc) for d, k, loc, _ in out)
for k, v in sorted(c.items()):
    print("depth %d %-5s %-24s line %-5d x%d" % (k[0], k[1], k[2][0], k[2][1], v))
print("slots:", sorted(Counter(o for _, _, _, o in out).items()))
