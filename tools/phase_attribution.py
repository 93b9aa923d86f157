#!/usr/bin/env python3
"""Per-phase shares of the N = 16 tick kernel, measured: reads gpurun_out/attr/ (tools/phase_attribution.sh) and prints, per
gait-tick, what executing each idempotent phase of the active-set iteration ONCE MORE adds to the hardware counters -- i.e.
that phase's own VALU / SALU / LDS instruction counts, its parked cycles (SQ_WAIT_ANY: s_waitcnt) and issue stalls
(SQ_WAIT_INST_ANY), its wave cycles and its share of the tick time.     python tools/phase_attribution.py [> profiles/...]"""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "gpurun_out", os.environ.get("ATTR_DIR", "attr"))
NH = int(os.environ.get("PN", "16"))
B = int(os.environ.get("PB", "4096"))
TICKS = 10 + int(os.environ.get("PR", "3")) * int(os.environ.get("PT", "100"))   # probe_elem.py: one 10-tick warm-up launch + PR launches of PT ticks
NAMES = {0: "(plain: nothing repeated)", 1: "violation scan, qld.cpp:1255-1331", 2: "Z^T a of the new normal, :1421-1470",
         3: "sweep: chain of rotation norms, :1992-2030 (phase 1)", 4: "back substitution of the step, :1824-1851",
         5: "linear-dependence sums, :1491-1532", 6: "xmag ordered sums (both sites), :2039-2058", 7: "pick_drop, :1861-1889",
         8: "factor(): constant blocks, border rows of R and Z (once per tick)", 9: "QP assembly of the tick (S c, gradient, tables, rhs)",
         11: "residual refresh: gradient and residuals, :1031-1099 (once per tick)"}
PHASES = [k for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 11) if os.path.isdir(os.path.join(D, "pmc_%d" % k))]
CNT = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"]


def load(k):
    tot = collections.Counter()
    files = sorted(glob.glob(os.path.join(D, "pmc_%d" % k, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                            # gpurun merges runs: the newest pass only
        for r in csv.DictReader(open(f)):
            if "wg_mpc_run_xcd_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    txt = open(os.path.join(D, "time_%d.txt" % k)).read()
    m = re.search(r"-> (\d+) ticks/s .*state checksum (\w+)", txt)
    return {c: tot[c] / (B * TICKS) for c in CNT}, float(m.group(1)), m.group(2)


base, rate0, sum0 = load(0)
print("# N = %d tick, B = %d, multi-tick launches (wg_mpc_run_xcd_kernel<%d>): per gait-tick, measured by rocprofv3 --pmc." % (NH, B, NH))
print("# Row k = counters of the build that runs phase k twice MINUS the plain build = what one execution of that phase costs per")
print("# gait-tick (%.1f active-set iterations on average).  Same state checksum in every build (the repeated phases are idempotent)." % (22.5 if NH == 16 else 44.2))
print("# plain build: %.0f ticks/s; VALU %.0f, SALU %.0f, LDS %.0f, VMEM %.0f instructions, %.0f wave cycles per gait-tick of which parked"
      " (s_waitcnt) %.0f = %.1f %%, issue-stalled %.0f = %.1f %%" % (rate0, base["SQ_INSTS_VALU"], base["SQ_INSTS_SALU"], base["SQ_INSTS_LDS"],
                                                                   base["SQ_INSTS_VMEM"], base["SQ_WAVE_CYCLES"], base["SQ_WAIT_ANY"],
                                                                   100 * base["SQ_WAIT_ANY"] / base["SQ_WAVE_CYCLES"], base["SQ_WAIT_INST_ANY"],
                                                                   100 * base["SQ_WAIT_INST_ANY"] / base["SQ_WAVE_CYCLES"]))
print("%-58s %8s %8s %7s %10s %9s %9s %8s %7s" % ("phase", "VALU", "SALU", "LDS", "wave cyc", "parked", "stalled", "time %", "SALU/VALU"))
acc = collections.Counter()
for k in PHASES:
    c, rate, chk = load(k)
    assert chk == sum0, (k, chk, sum0)
    d = {n: c[n] - base[n] for n in CNT}
    share = 100 * (rate0 / rate - 1.0)
    for n in CNT:
        acc[n] += d[n]
    acc["share"] += share
    acc["share_%d" % k] = share
    print("%-58s %8.0f %8.0f %7.0f %10.0f %9.0f %9.0f %7.1f%% %7.2f" % (NAMES[k][:58], d["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], d["SQ_INSTS_LDS"],
                                                                      d["SQ_WAVE_CYCLES"], d["SQ_WAIT_ANY"], d["SQ_WAIT_INST_ANY"], share,
                                                                      d["SQ_INSTS_SALU"] / max(1.0, d["SQ_INSTS_VALU"])))
rest = {n: base[n] - acc[n] for n in CNT}
print("%-58s %8.0f %8.0f %7.0f %10.0f %9.0f %9.0f %7.1f%% %7.2f" % ("everything else (see the timer table below for its split):", rest["SQ_INSTS_VALU"], rest["SQ_INSTS_SALU"],
                                                                  rest["SQ_INSTS_LDS"], rest["SQ_WAVE_CYCLES"], rest["SQ_WAIT_ANY"], rest["SQ_WAIT_INST_ANY"],
                                                                  100 - acc["share"], rest["SQ_INSTS_SALU"] / max(1.0, rest["SQ_INSTS_VALU"])))
print("%-58s" % "   sweep phases 2-3, drops, x / lambda updates, the refresh's substitutions, lane-0 bookkeeping, feet, state in / out")

# ---- the lump, split by the shader-clock timers (tools/probe_tick_phases.py on lib/libwg_mpc_prof.so, one launch per tick) ----
# The phases below cannot be executed twice (they rotate Z, move x, advance the state), so their cost comes from s_memtime
# brackets instead.  Timer cycles are not the counters' wave cycles (another build, one launch per tick, the reads themselves cost):
# each row's share = its timer cycles / the timer cycles of ALL rows listed here x the share the counters leave for the lump.
TIMERS = os.path.join(ROOT, "gpurun_out", "phases_tick.txt" if NH == 16 else "phases_tick32.txt")
if os.path.exists(TIMERS):
    top, sub = {}, {}
    for ln in open(TIMERS):
        m = re.match(r"^\s*(\d+) (.*?)\s+(\d+) cyc/tick", ln)
        if m:
            top[int(m.group(1))] = float(m.group(3)); continue
        m = re.match(r"^\s+(sweep|pre|post) (.*?)\s+(\d+) cyc/tick", ln)
        if m:
            sub[(m.group(1), m.group(2).strip())] = float(m.group(3))
    sw = {k[1][:7]: v for k, v in sub.items() if k[0] == "sweep"}
    rows = [("sweep phase 2: (ga, gb) of every rotation, one lane each, :2011-2014", sw.get("phase 2", 0.0)),
            ("sweep phase 3: lane i carries row i of Z through the rotations, :2015-2029", sw.get("phase 3", 0.0)),
            ("step direction and lengths before the back substitution (z, ww := Z s, ratios), :1715-1823", top.get(14, 0.0)),
            ("route of the new normal besides the dependence sums (counter row above), :1471-1560", max(0.0, top.get(13, 0.0) - acc["share_5"] / 100.0 * sum(top.get(k, 0.0) for k in range(24)))),
            ("x / lambda update, constraint drop (:1903-1982), bookkeeping of the iteration", top.get(17, 0.0) + top.get(18, 0.0) + top.get(10, 0.0) + top.get(20, 0.0)),
            ("residual refresh besides its gradient (counter row above): reset, Z^T ww, x shift, substitutions, lambda", max(0.0, top.get(4, 0.0) - top.get(25, 0.0) - top.get(27, 0.0)) + top.get(5, 0.0) + top.get(6, 0.0) + top.get(7, 0.0)),
            ("start of the solve: norms of the rows, diagonal test", top.get(0, 0.0) + top.get(1, 0.0))]
    for (grp, name), v in sub.items():
        if grp in ("pre", "post"):
            rows.append(("tick, %s the solve: %s" % ("before" if grp == "pre" else "after", name), v))
    if top.get(23) and not any(g == "post" for g, _ in sub):
        rows.append(("tick, after the solve", top[23]))
    tot_t = sum(top.get(k, 0.0) for k in range(24))
    lump = 100 - acc["share"]
    tsum = sum(v for _, v in rows)
    print()
    print("# the lump by shader-clock timers (%s; %.0f timer cycles per gait-tick in all, %.0f of them in the rows below = %.1f %% there," %
          (os.path.relpath(TIMERS, ROOT), tot_t, tsum, 100 * tsum / tot_t))
    print("# against %.1f %% by the counters: the timers' own reads weigh on short phases); share = row / sum of these rows x %.1f %%" % (lump, lump))
    print("%-112s %10s %8s" % ("phase (not repeatable: timed)", "timer cyc", "time %"))
    for name, v in sorted(rows, key=lambda r: -r[1]):
        print("%-112s %10.0f %7.1f%%" % (name[:112], v, lump * v / tsum))
    big = max(lump * v / tsum for _, v in rows)
    print("# rows above (counters) + rows here (timers) = 100 %% of the tick; largest row of the former lump: %.1f %%" % big)
