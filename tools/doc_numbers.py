#!/usr/bin/env python3
"""The figures DESIGN.md / README.md / INTEGRATION.md quote, generated from profiles/<tag>_* (one truth: nothing typed by hand).

    python tools/doc_numbers.py round4            # rewrites the blocks between  <!-- BEGIN GENERATED name -->  and  <!-- END GENERATED name -->
    python tools/doc_numbers.py round4 --check    # exit 1 if a block in a document differs from what the profiles say

Blocks: `current` (every current figure in one place), `attribution` (the per-phase table), `headline` (README), `latency`."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "round4"
check = "--check" in sys.argv
P = lambda n: os.path.join(ROOT, "profiles", n)                                      # noqa: E731
J = lambda n: json.load(open(P(n)))                                                  # noqa: E731

bench, win, pmc, lat = J(f"{tag}_bench.json"), J(f"{tag}_bench_driver_window.json"), J("current_tick_pmc.json"), J(f"{tag}_latency_b1.json")
M = lambda v: "%.2f M" % (v / 1e6)                                                    # noqa: E731


def resources():
    """kernel -> (VGPR, AGPR, SGPR, V-spill, S-spill, scratch, occupancy) and kernel -> {'scratch': {...by depth}, ...}"""
    tab, depth, cur = {}, {}, None
    for ln in open(P(f"{tag}_resource_usage.txt")):
        m = re.match(r"^(wg_\S+(?:<[^>]*>)?)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s*$", ln)
        if m:
            tab[m.group(1)] = tuple(int(x) for x in m.groups()[1:]); continue
        m = re.match(r"^(wg_\S+(?:<[^>]*>)?): (\d+) instructions", ln)
        if m:
            cur = m.group(1); depth[cur] = {"instructions": int(m.group(2))}; continue
        m = re.match(r"^\s+(scratch|sgpr_spill_write|sgpr_spill_read)\s+total\s+(\d+)\s+by depth: (.*)$", ln)
        if m and cur:
            by = {int(a): int(b) for a, b in re.findall(r"d(\d+): (\d+)", m.group(3))}
            depth[cur][m.group(1)] = (int(m.group(2)), by)
    return tab, depth


RES, DEPTH = resources()


def res_row(k):
    v, a, s, vs, ss, sc, occ = RES[k]
    d = DEPTH.get(k, {})
    def by(name, lo):                                                                 # instructions at loop depth >= lo
        return sum(n for dd, n in d.get(name, (0, {}))[1].items() if dd >= lo)
    return (f"| `{k}` | {v} | {vs} | {ss} | {sc} B | {occ} | {d.get('scratch', (0, {}))[0]} ({by('scratch', 2)} at depth ≥ 2) | "
            f"{d.get('sgpr_spill_read', (0, {}))[0]} ({by('sgpr_spill_read', 2)} at depth 2, {by('sgpr_spill_read', 3)} at depth ≥ 3) |")


def launch_fit():
    out = []
    for ln in open(P(f"{tag}_launch_fit.txt")):
        m = re.match(r"B=(\d+) N=(\d+): duration = ([\d.]+) ms \+ ([\d.]+) ms per tick .*steady (\d+) ticks/s, the per-launch part = ([\d.]+) ticks", ln)
        if m:
            out.append(dict(B=int(m.group(1)), N=int(m.group(2)), a=float(m.group(3)), b=float(m.group(4)), steady=float(m.group(5)), ticks=float(m.group(6))))
    return out


def attribution():
    """counter rows (name, VALU, SALU, LDS, wave cycles, parked, stalled, share), timer rows (name, share), the plain build's line"""
    rows, lump, plain = [], [], ""
    mode = 0
    for ln in open(P(f"{tag}_phase_attribution.txt")):
        if ln.startswith("# plain build"):
            plain = ln[2:].strip()
        if ln.startswith("phase (not repeatable"):
            mode = 2; continue
        if ln.startswith("phase "):
            mode = 1; continue
        m = re.match(r"^(.*?)\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+([\d.]+)%", ln)
        if mode == 1 and m:
            r = (m.group(1).strip(), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)), float(m.group(8)))
            if ln.startswith("everything else"):
                rows.append(("__lump__",) + r[1:])
            else:
                rows.append(r)
        m2 = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)%\s*$", ln)
        if mode == 2 and m2:
            lump.append((m2.group(1).strip(), float(m2.group(3))))
    return plain, rows, lump


def block_attribution():
    plain, rows, lump = attribution()
    k = lambda v: "%.1f k" % (v / 1e3)                                                # noqa: E731
    L = [f"*Generated from `profiles/{tag}_phase_attribution.txt` ({plain.split(';')[0]}).*", "",
         "| phase (qld.cpp) | VALU | SALU | LDS | wave cycles | parked | stalled | share of the tick |", "|---|---|---|---|---|---|---|---|"]
    tot = [0] * 6
    for r in rows:
        if r[0] == "__lump__":
            continue
        L.append(f"| {r[0]} — *counters* | {r[1]:,} | {max(0, r[2]):,} | {r[3]:,} | {k(r[4])} | {k(r[5])} | {k(r[6])} | {r[7]:.1f} % |".replace(",", " "))
    lp = [r for r in rows if r[0] == "__lump__"]
    if lp:
        r = lp[0]
        L.append(f"| everything the counters cannot repeat (split by timers below) | {r[1]:,} | {r[2]:,} | {r[3]:,} | {k(r[4])} | {k(r[5])} | {k(r[6])} | {r[7]:.1f} % |".replace(",", " "))
    for n, sh in lump:
        L.append(f"| &nbsp;&nbsp;{n} — *timers* | | | | | | | {sh:.1f} % |")
    m = re.search(r"VALU (\d+), SALU (\d+), LDS (\d+), VMEM (\d+) instructions, (\d+) wave cycles per gait-tick of which parked \(s_waitcnt\) (\d+) = ([\d.]+) %, issue-stalled (\d+) = ([\d.]+) %", plain)
    if m:
        L.append(f"| **whole tick** (plain build) | {int(m.group(1)):,} | {int(m.group(2)):,} | {int(m.group(3)):,} | {k(int(m.group(5)))} | {k(int(m.group(6)))} ({m.group(7)} %) | {k(int(m.group(8)))} ({m.group(9)} %) | 100 % |".replace(",", " "))
    return "\n".join(L)


def block_current():
    r, o, e, b1 = pmc["run_kernel"], pmc["run_kernel_outs"], pmc["elem_run_kernel"], pmc["one_robot_kernel"]
    c5, c5w = bench["config5"]["default"], win["config5"]["default"]
    k, kw = bench["kernels"], win["kernels"]
    rf, rfw = bench["roofline"], win["roofline"]
    fit = launch_fit()
    plain, rows, lump = attribution()
    L = []
    L.append(f"*Generated by `python tools/doc_numbers.py {tag}` from `profiles/{tag}_*` and `profiles/current_tick_pmc.json` — figures elsewhere in this "
             "document that disagree with this block are history (the tables of steps say what was measured when); this block is HEAD.*")
    L.append("")
    L.append("**Bench lines** (`bench.py`, one MI355X, B = 4096, N = 16, fp64, inputs resident; builder's boxes — the driver's clock decides):")
    L.append("")
    L.append(f"| leg | default run (`--steps 200 --warmup 50`, `{tag}_bench.json`) | the driver's window (`--steps 20 --warmup 5`, `{tag}_bench_driver_window.json`) |")
    L.append("|---|---|---|")
    L.append(f"| **`value`: MPC ticks/s, one multi-tick launch** | **{M(bench['value'])}** ({bench['ms_per_step']:.4f} ms per step) | **{M(win['value'])}** ({win['ms_per_step']:.4f} ms per step) |")
    L.append(f"| `outs_on`: the same launch storing every gait-tick's `wg_tick_out_t` | {M(bench['outs_on']['value'])} ({100 * bench['outs_on']['delta_vs_value']:+.1f} %) | {M(win['outs_on']['value'])} ({100 * win['outs_on']['delta_vs_value']:+.1f} %) |")
    L.append(f"| `per_tick_launch`: one launch per tick | {M(bench['per_tick_launch']['value'])} | {M(win['per_tick_launch']['value'])} |")
    L.append(f"| `config5`: N = 32, B = 8192 (`{c5['roofline']['kernel']}`) | {M(c5['value'])} | {M(c5w['value'])} |")
    L.append(f"| `kernels.ql0001_dense`: the `ql0001_` boundary on the workload's real QPs | {M(k['ql0001_dense']['value'])} QPs/s | {M(kw['ql0001_dense']['value'])} QPs/s |")
    L.append(f"| `kernels.dimitrov_tick` (PLDP) / `dimitrov_tick_qldandlq` (in-wave `ql0001_`) | {M(k['dimitrov_tick']['value'])} / {M(k['dimitrov_tick_qldandlq']['value'])} ticks/s | {M(kw['dimitrov_tick']['value'])} / {M(kw['dimitrov_tick_qldandlq']['value'])} |")
    L.append(f"| `kernels.pldp` | {M(k['pldp']['value'])} solves/s | {M(kw['pldp']['value'])} |")
    L.append(f"| `kernels.zmpdisc` / `preview` / `steps_to_com` | {k['zmpdisc']['value'] / 1e9:.2f} G gait-samples/s / {k['preview']['value'] / 1e9:.2f} G gait-steps/s / {k['steps_to_com']['value'] / 1e6:.2f} M walks/s | {kw['zmpdisc']['value'] / 1e9:.2f} / {kw['preview']['value'] / 1e9:.2f} / {kw['steps_to_com']['value'] / 1e6:.2f} |")
    cb = bench["cpu_baseline"]
    ac = cb["all_cores"]
    L.append(f"| `cpu_baseline` (`kind: {cb['kind']}`): one host core / the {ac.get('cores_used', ac['cores'])} usable cores ({ac.get('cores_visible', '?')} visible; parallel efficiency {ac.get('parallel_efficiency', float('nan')):.2f}) | {cb['value'] / 1e3:.1f} k / {ac['value'] / 1e3:.0f} k ticks/s | — |")
    pr = bench["parity"]
    L.append(f"| `parity`: CoM vs the golden file / vs the libm oracle driven by the compiled `ql0001_` | RMSE {pr['com_rmse_m']:.2e} m (the file's print precision) / max {pr['com_max_abs_vs_libm_oracle_m']:.1e} m | same |")
    L.append("")
    L.append(f"`roofline` of the line: {rf['algorithmic_bytes_per_launch'] / 1e9:.3f} GB algorithmic per {rf['ticks_per_launch']}-tick launch ÷ {rf['kernel_ms']:.2f} ms = "
             f"{rf['achieved']:.0f} GB/s = **frac {rf['frac']:.4f}** of 8 TB/s ({rfw['achieved']:.0f} GB/s = {rfw['frac']:.4f} in the 20-step window); measured `traffic` "
             f"{r['hbm_bytes_per_gait_tick'] / 1e3:.2f} KB per gait-tick = {r['hbm_bytes_per_gait_tick'] * 4096 * rf['ticks_per_launch'] / rf['algorithmic_bytes_per_launch']:.3f} × the algorithmic bytes; "
             f"second axis: {rf['second_axis']['useful_fp64_tflops']:.2f} useful fp64 TFLOP/s = {100 * rf['second_axis']['useful_flop_frac']:.1f} % of the vector peak, VALU busy {rf['second_axis']['valu_busy']:.2f}.")
    L.append("")
    if fit:
        f16 = [f for f in fit if f["N"] == 16][0]
        L.append(f"**Launch length** (`{tag}_launch_fit.txt`): a B = 4096 launch of T ticks takes {f16['a']:.2f} ms + {f16['b']:.4f} ms × T — steady "
                 f"{M(f16['steady'])} ticks/s, and {f16['ticks']:.2f} tick-times once per launch (ramp, and the tail in which the last gait-ticks finish on a chip "
                 "running empty: a gait's ticks cannot run in parallel).  That, not another kernel, is the difference between the two columns above.")
        f32 = [f for f in fit if f["N"] == 32]
        if f32:
            L.append(f"N = 32, B = 8192: {f32[0]['a']:.2f} ms + {f32[0]['b']:.3f} ms × T, steady {M(f32[0]['steady'])}.")
        L.append("")
    L.append(f"**Resources** (`{tag}_resource_usage.txt`: the compiler's table, spill code placed by loop depth — 1 = per tick, 2 = per active-set iteration, ≥ 3 = inner loops):")
    L.append("")
    L.append("| kernel | VGPR | spilled VGPR | spilled SGPR | scratch per lane | waves / SIMD | `scratch_` instructions | SGPR-spill reloads |")
    L.append("|---|---|---|---|---|---|---|---|")
    for kk in ("wg_mpc_run_xcd_kernel<16>", "wg_mpc_tick_kernel<16>", "wg_mpc_run_xcd_kernel<32>", "wg_mpc_run_xcd_kernel<-1>", "wg_mpc_run_xcd_kernel<0>",
               "wg_ql_dense_kernel<false, false, false, 36, 76>", "wg_ql_dense_kernel<false, false, true, 0, 0>", "wg_dimitrov_qld_tick_kernel<true>",
               "wg_dimitrov_tick_kernel", "wg_pldp_kernel<true>", "wg_zmpdisc_kernel"):
        if kk in RES:
            L.append(res_row(kk))
    L.append("")
    L.append("**Counters per gait-tick** (`current_tick_pmc.json`; rocprofv3 `--pmc`, separate passes, FETCH_SIZE × 2, KiB units):")
    L.append("")
    L.append("| kernel | HBM read | HBM written | VALU | SALU | LDS | VMEM instr. | VALU busy | parked in `s_waitcnt` |")
    L.append("|---|---|---|---|---|---|---|---|---|")
    L.append(f"| `{r['kernel']}` (bench workload, launches of {r['ticks_per_launch']} ticks) | {r['hbm_read_bytes_per_gait_tick']:.0f} B | {r['hbm_write_bytes_per_gait_tick']:.0f} B | {r['valu_insts_per_gait_tick'] / 1e3:.1f} k | {r['salu_insts_per_gait_tick'] / 1e3:.1f} k | {r['lds_insts_per_gait_tick'] / 1e3:.1f} k | {r['vmem_insts_per_gait_tick']:.0f} | {r['valu_busy']:.2f} | {100 * r['wait_any_frac_per_wave']:.0f} % |")
    L.append(f"| the same, outs on (timed launch stores `wg_tick_out_t`) | {o['hbm_read_bytes_per_gait_tick']:.0f} B | {o['hbm_write_bytes_per_gait_tick']:.0f} B (+{o['extra_write_bytes_per_stored_gait_tick'] / 1e3:.1f} KB per *stored* gait-tick for a {bench['outs_on']['out_bytes_per_gait_tick']}-B struct) | {o['valu_insts_per_gait_tick'] / 1e3:.1f} k | {o['salu_insts_per_gait_tick'] / 1e3:.1f} k | {o['lds_insts_per_gait_tick'] / 1e3:.1f} k | {o['vmem_insts_per_gait_tick']:.0f} | {o['valu_busy']:.2f} | {100 * o['wait_any_frac_per_wave']:.0f} % |")
    L.append(f"| `{e['kernel']}` (N = 32, B = 8192) | {e['hbm_read_bytes_per_gait_tick'] / 1e6:.2f} MB | {e['hbm_write_bytes_per_gait_tick'] / 1e6:.2f} MB | {e['valu_insts_per_gait_tick'] / 1e3:.0f} k | {e['salu_insts_per_gait_tick'] / 1e3:.0f} k | {e['lds_insts_per_gait_tick'] / 1e3:.1f} k | {e['vmem_insts_per_gait_tick'] / 1e3:.1f} k | {e['valu_busy']:.2f} | {100 * e['wait_any_frac_per_wave']:.0f} % |")
    L.append("")
    c5r = c5["roofline"]
    L.append(f"N = 32: {e['hbm_bytes_per_gait_tick'] / 1e6:.2f} MB per gait-tick against {c5r['algorithmic_bytes_per_step'] / 8192 / 1e3:.1f} KB algorithmic "
             f"(× {e['hbm_bytes_per_gait_tick'] / (c5r['algorithmic_bytes_per_step'] / 8192):.0f}), moved at {c5r['traffic_gbs'] / 1e3:.2f} TB/s "
             f"({100 * c5r['traffic_frac_of_peak']:.0f} % of the HBM figure; the slots total 3072 × 41.5 KB = 127 MB, so most of it is L2 ↔ Infinity Cache, not DRAM); "
             f"`frac` on algorithmic bytes {c5r['frac']:.4f}.")
    L.append("")
    top = sorted([r for r in rows if r[0] != "__lump__"], key=lambda r: -r[7])[:4]
    L.append(f"**Where the tick's time goes**: §4.2's table (`{tag}_phase_attribution.txt`, generated too).  Largest rows: "
             + "; ".join(f"{r[0].split(',')[0]} {r[7]:.1f} %" for r in top) + f"; largest of the {len(lump)} timer rows {max(s_ for _, s_ in lump):.1f} %.")
    L.append("")
    s = lat["split_us"]
    L.append(f"**One robot** (`{tag}_latency_b1.json`, `current_tick_pmc.json → one_robot_kernel`): host-pointer call {lat['host_pointer_call_us']['median']:.0f} µs median "
             f"(copy in {s['copy_in']:.0f} + launch {s['launch']:.0f} + kernel {s['wait_for_kernel']:.0f} + copy out {s['copy_out']:.0f}), host-mapped call "
             f"(`wg_mpc_tick_pinned`) {lat['host_mapped_call_us']['median']:.0f} µs, the kernel alone {s['kernel_hip_events']:.0f} µs (HIP events; "
             f"{b1['avg_kernel_ns_rocprofv3'] / 1e3:.0f} µs under rocprofv3) = {b1['wave_cycles_per_tick'] / 1e3:.0f} k cycles of ONE wave alone on a CU: executing "
             f"{100 * b1['executing_frac']:.0f} % (VALU {100 * b1['valu_executing_frac']:.0f} %, LDS {100 * b1['lds_executing_frac']:.0f} %), parked in `s_waitcnt` "
             f"{100 * b1['parked_in_waitcnt_frac']:.0f} %, issue-stalled {100 * b1['issue_stalled_frac']:.1f} %; {b1['valu_insts_per_tick'] / 1e3:.1f} k VALU + "
             f"{b1['salu_insts_per_tick'] / 1e3:.1f} k SALU + {b1['lds_insts_per_tick'] / 1e3:.1f} k LDS instructions.  The reference tick on one host core: "
             f"{bench['parity']['b1_tick_latency_us']['cpu_reference_us_per_tick_one_core']:.0f} µs.")
    return "\n".join(L)


def block_headline():
    r, e = pmc["run_kernel"], pmc["elem_run_kernel"]
    k = win["kernels"]
    c5w = win["config5"]["default"]
    cb = bench["cpu_baseline"]
    return (f"Measured on one MI355X (B = 4096 gaits, N = 16, fp64; `profiles/{tag}_bench_driver_window.json`, `profiles/{tag}_bench.json`): "
            f"**{M(win['value'])} MPC ticks/s in the driver's window** (`bench.py --steps 20 --warmup 5`: one launch of 20 ticks; the driver's own clock on its own box is "
            f"the number of record — round 3: driver 5.87 M against the builder's 5.98 M); {M(bench['value'])} over the builder's default 200-step run (one launch of 200 "
            f"ticks — the same kernel; a launch pays about half a tick-time once for its ramp and drain, `profiles/{tag}_launch_fit.txt`).  With the tick's deliverable stored "
            f"(20 CoM / ZMP / feet samples per gait-tick, {bench['outs_on']['out_bytes_per_gait_tick'] / 1e3:.1f} KB): {M(win['outs_on']['value'])} / {M(bench['outs_on']['value'])}.  One launch per tick: "
            f"{M(win['per_tick_launch']['value'])}.  The reference's compiled solver on one host core: {cb['value'] / 1e3:.1f} k ticks/s, on the {cb['all_cores'].get('cores_used', cb['all_cores']['cores'])} cores the box grants: "
            f"{cb['all_cores']['value'] / 1e3:.0f} k.  {r['hbm_bytes_per_gait_tick'] / 1e3:.1f} KB of HBM traffic per gait-tick; CoM RMSE "
            f"{bench['parity']['com_rmse_m']:.0e} m against the reference's golden file (its print precision), max {bench['parity']['com_max_abs_vs_libm_oracle_m']:.0e} m against "
            f"the libm oracle driven by the compiled `ql0001_`.  N = 32 with foot-placement variables at B = 8192: {M(c5w['value'])} ticks/s "
            f"({e['hbm_bytes_per_gait_tick'] / 1e6:.1f} MB per gait-tick through the fabric, `DESIGN.md` §4.3).  The other kernels of the path on the same line (`kernels`): "
            f"dense `ql0001_` boundary {M(k['ql0001_dense']['value'])} QPs/s on the workload's real QPs, Dimitrov tick {M(k['dimitrov_tick']['value'])} ticks/s with PLDP and "
            f"{M(k['dimitrov_tick_qldandlq']['value'])} with the in-wave `ql0001_`, PLDP {M(k['pldp']['value'])} hot-started solves/s, step sequences → ZMP queue "
            f"{k['zmpdisc']['value'] / 1e9:.1f} G gait-samples/s → preview control {k['preview']['value'] / 1e9:.1f} G gait-steps/s, {k['steps_to_com']['value'] / 1e6:.2f} M whole "
            f"walks/s without leaving the device.  One robot (B = 1): {lat['host_mapped_call_us']['median']:.0f} µs per tick through `wg_mpc_tick_pinned` against the reference's "
            f"{bench['parity']['b1_tick_latency_us']['cpu_reference_us_per_tick_one_core']:.0f} µs on a host core — the path is built for fleets.  RCCL with ≥ 2 ranks is "
            "unmeasured on hardware (every shard of the 8-GPU configuration runs at full size on one GPU as a test).  See `DESIGN.md` §4 and `profiles/`.")


def block_latency():
    s = lat["split_us"]
    return (f"Measured (`jrl-walkgen_amd/bin/latency_b1`, `profiles/{tag}_latency_b1.json`): `wg_mpc_tick_batch(B = 1)` on host pointers "
            f"{lat['host_pointer_call_us']['median']:.0f} µs median (p90 {lat['host_pointer_call_us']['p90']:.0f}): copy in {s['copy_in']:.0f} µs, launch {s['launch']:.0f} µs, "
            f"kernel {s['wait_for_kernel']:.0f} µs, copy out {s['copy_out']:.0f} µs; `wg_mpc_tick_pinned` on host-mapped memory {lat['host_mapped_call_us']['median']:.0f} µs "
            f"(p90 {lat['host_mapped_call_us']['p90']:.0f}), same bytes; the kernel alone {s['kernel_hip_events']:.0f} µs.  The reference's tick takes "
            f"{bench['parity']['b1_tick_latency_us']['cpu_reference_us_per_tick_one_core']:.0f} µs on one host core: for ONE robot the CPU is the faster device, and DESIGN §4 "
            "(\"one robot\") says with counters why a second wave would not change that.")


def wrap_md(text, width=120):
    """prose at `width` columns; table rows, headings and blank lines as they are"""
    import textwrap
    out = []
    for ln in text.split("\n"):
        if len(ln) <= width or ln.startswith(("|", "#", "<!--")):
            out.append(ln)
        else:
            out.extend(textwrap.wrap(ln, width, break_long_words=False, break_on_hyphens=False))
    return "\n".join(out)


BLOCKS = {"current": ("DESIGN.md", block_current), "attribution": ("DESIGN.md", block_attribution), "headline": ("README.md", block_headline), "latency": ("INTEGRATION.md", block_latency)}
rc = 0
for name, (doc, fn) in BLOCKS.items():
    path = os.path.join(ROOT, doc)
    txt = open(path).read()
    a, b = f"<!-- BEGIN GENERATED {name} -->", f"<!-- END GENERATED {name} -->"
    if a not in txt or b not in txt:
        print(f"{doc}: no {name} block"); rc = 1; continue
    new = txt[:txt.index(a) + len(a)] + "\n" + wrap_md(fn()) + "\n" + txt[txt.index(b):]
    if new != txt:
        if check:
            print(f"{doc}: block `{name}` is stale"); rc = 1
        else:
            open(path, "w").write(new); print(f"{doc}: block `{name}` rewritten")
    else:
        print(f"{doc}: block `{name}` up to date")
sys.exit(rc)
