#!/usr/bin/env python3
"""Register / spill audit of the kernels in csrc/wg_capi.hip, from the compiler's own output (no GPU needed).

    python tools/isa_audit.py [--out profiles/round3_resource_usage.txt] [--isa /tmp/wg_capi.s]

1. `-Rpass-analysis=kernel-resource-usage` table: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, occupancy.
2. For the kernels that spill: where the spill code sits.  The gfx950 assembly (--cuda-device-only -S) of each kernel is cut into
   its loops -- a backward branch to a label opens a loop body [label, branch] -- and every `scratch_` instruction (VGPR spill
   traffic), `v_writelane` / `v_readlane` into the spill VGPRs (SGPR spills live in lanes of reserved VGPRs) is placed by its
   loop-nesting depth.  In the multi-tick kernels depth 1 is the persistent per-tick loop, depth 2 the solver's active-set
   iteration (its state machine `while (st != ST_FINISH)`), depth >= 3 the inner loops of the phases (sweep, back substitution,
   scan, Z^T a): spill code at depth >= 3 runs per rotation / per row and would be a performance bug; at depth <= 2 it runs a
   few times per iteration at most.
3. Per kernel, the instruction mix of the INNER loops (depth >= 3) by class: VALU, SALU (incl. s_waitcnt / branches), LDS,
   VMEM -- the static side of the SALU question: what the scalar instructions of the hot loops are.
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "jrl-walkgen_amd", "csrc", "wg_capi.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
         "-Wno-pass-failed", "-w"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    out = p.stdout.splitlines() if p.returncode == 0 else names
    return dict(zip(names, [re.sub(r"\(.*", "", o).replace("void ", "").replace("wg::", "") for o in out]))


def resource_table(extra):
    p = subprocess.run([HIPCC] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", SRC],
                       capture_output=True, text=True, cwd=os.path.join(ROOT, "jrl-walkgen_amd"))
    txt = p.stderr + p.stdout
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split()[0]
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1)) if re.search(k + r": (\d+)", b) else -1  # noqa: E731
        rows.append(dict(name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("TotalSGPRs"), vspill=g("VGPRs Spill"), sspill=g("SGPRs Spill"),
                         scratch=g(r"ScratchSize \[bytes/lane\]"), occ=g(r"Occupancy \[waves/SIMD\]")))
    return rows


def klass(op):
    if op.startswith(("scratch_",)):
        return "SCRATCH"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def audit_kernel(lines):
    """lines: the assembly of one kernel.  -> (depth histograms, inner-loop instruction mix, salu opcode histogram).
    Loop depth comes from the compiler's own block annotations (`; in Loop: Header=BB5_12 Depth=3`, `; =>This Inner Loop
    Header: Depth=2`, preceded by `Parent Loop` lines): every instruction takes the depth of the block it sits in."""
    insts, depth = [], []
    cur = 0
    n_loops = 0
    i = 0
    while i < len(lines):
        s = lines[i].strip()
        m = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):(.*)$", s)          # a labelled block, or a fall-through block (comment only)
        if m:
            cmt = m.group(2)
            j = i + 1
            while j < len(lines) and lines[j].strip().startswith(";"):      # continuation lines of the block comment
                cmt += " " + lines[j].strip(); j += 1
            ds = [int(x) for x in re.findall(r"Depth=(\d+)", cmt)]
            cur = max(ds) if ds else 0
            if "Loop Header" in cmt:
                n_loops += 1
            i = j
            continue
        i += 1
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        args = parts[1] if len(parts) > 1 else ""
        args = args.split(";")[0].strip()
        insts.append((parts[0], args)); depth.append(cur)
    loops = [None] * n_loops
    # SGPR spills live in lanes of VGPRs the compiler reserves for them: `v_writelane_b32 v237, s12, 5` stores, `v_readlane_b32
    # s12, v237, 5` reloads -- immediate lanes, and the SAME few VGPRs on both sides.  An immediate lane alone does not mark spill
    # code: the solver broadcasts with constant lanes too (the register Cholesky's rl(r[k], i): 1 335 v_readlane of the
    # Herdt-sized dense kernel, which rounds 3 - 4 counted as reloads).  So: the spill registers are the VGPRs that some
    # v_writelane with an SGPR source and an immediate lane writes; only reads from THOSE are reloads; the other constant-lane
    # readlanes are listed as what they are.
    hist = {k: collections.Counter() for k in ("scratch", "sgpr_spill_write", "sgpr_spill_read", "readlane_const_lane_other")}
    spill_vgprs = set()
    for op, args in insts:
        m = re.match(r"^(v\d+), s\d+, \d+\s*$", args) if op == "v_writelane_b32" else None
        if m:
            spill_vgprs.add(m.group(1))
    mix = collections.Counter(); salu = collections.Counter(); total = collections.Counter()
    for i, (op, args) in enumerate(insts):
        k = klass(op)
        total[k] += 1
        if k == "SCRATCH":
            hist["scratch"][depth[i]] += 1
        if op == "v_writelane_b32" and re.search(r",\s*\d+\s*$", args):
            hist["sgpr_spill_write"][depth[i]] += 1
        if op == "v_readlane_b32" and re.search(r",\s*\d+\s*$", args) and re.match(r"s\d+|s\[", args.strip()):
            src = re.match(r"^s\d+, (v\d+),", args)
            hist["sgpr_spill_read" if (src and src.group(1) in spill_vgprs) else "readlane_const_lane_other"][depth[i]] += 1
        if depth[i] >= 3:
            mix[k] += 1
            if k == "SALU":
                salu[re.sub(r"_(b|i|u)(32|64)$", "", op)] += 1
    return dict(n_insts=len(insts), n_loops=len(loops), max_depth=max(depth) if depth else 0, hist=hist, inner_mix=mix, inner_salu=salu, total=total)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--isa", default="/tmp/wg_capi_audit.s")
    ap.add_argument("--extra", default="", help="extra compiler flags (experiment builds)")
    ap.add_argument("--reuse", action="store_true", help="reuse the assembly file of an earlier run (skip both compilations)")
    ap.add_argument("--kernels", default="wg_mpc_run_xcd_kernel,wg_mpc_tick_kernel,wg_ql_dense_kernel,wg_zmpdisc_kernel,wg_mpc_assemble_kernel")
    args = ap.parse_args()
    extra = args.extra.split() if args.extra else []
    out = []
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from csrc_hash import csrc_hash
    out.append("# csrc sha256: %s   (tools/csrc_hash.py: the device sources this audit was made from)" % csrc_hash())
    rows = [] if args.reuse else resource_table(extra)
    names = demangle([r["name"] for r in rows])
    out.append("# kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950%s)" % (", " + args.extra if args.extra else ""))
    out.append("%-58s %5s %5s %5s %7s %7s %8s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "V-spill", "S-spill", "scratch", "occ"))
    for r in rows:
        out.append("%-58s %5d %5d %5d %7d %7d %8d %4d" % (names[r["name"]][:58], r["vgpr"], r["agpr"], r["sgpr"], r["vspill"], r["sspill"], r["scratch"], r["occ"]))
    if not args.reuse:
        subprocess.check_call([HIPCC] + FLAGS + extra + ["--cuda-device-only", "-S", "-o", args.isa, SRC], cwd=os.path.join(ROOT, "jrl-walkgen_amd"))
    txt = open(args.isa).read().splitlines()
    starts = [(i, re.match(r"^(_Z\w+):", ln).group(1)) for i, ln in enumerate(txt) if re.match(r"^_Z\w+:", ln)]
    want = args.kernels.split(",")
    dm = demangle([n for _, n in starts])
    out.append("")
    out.append("# where the spill code sits: instructions by loop-nesting depth (0 = straight-line, 1 = outermost loop, ...)")
    out.append("#   multi-tick kernels: depth 1 = per tick, depth 2 = per active-set iteration, depth >= 3 = inner loops of the phases")
    for k, (i0, name) in enumerate(starts):
        pretty = dm[name]
        if not any(w in pretty for w in want):
            continue
        i1 = starts[k + 1][0] if k + 1 < len(starts) else len(txt)
        end = next((j for j in range(i0, i1) if txt[j].startswith(".Lfunc_end")), i1)
        a = audit_kernel(txt[i0 + 1:end])
        out.append("")
        out.append("%s: %d instructions, %d loops, deepest nesting %d" % (pretty, a["n_insts"], a["n_loops"], a["max_depth"]))
        out.append("   totals: " + ", ".join("%s %d" % kv for kv in sorted(a["total"].items())))
        for what in ("scratch", "sgpr_spill_write", "sgpr_spill_read", "readlane_const_lane_other"):
            h = a["hist"][what]
            out.append("   %-25s total %4d   by depth: %s" % (what, sum(h.values()), ", ".join("d%d: %d" % kv for kv in sorted(h.items())) or "-"))
        inner = a["inner_mix"]
        out.append("   inner loops (depth >= 3), static mix: " + ", ".join("%s %d" % kv for kv in sorted(inner.items())))
        if a["inner_salu"]:
            out.append("   their SALU instructions: " + ", ".join("%s %d" % kv for kv in a["inner_salu"].most_common(14)))
    text = "\n".join(out) + "\n"
    sys.stdout.write(text)
    if args.out:
        open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w").write(text)


if __name__ == "__main__":
    main()
