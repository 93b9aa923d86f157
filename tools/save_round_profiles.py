"""Files the summaries of the last tools/prof_round.sh run (merged back under gpurun_out/) into profiles/ as <tag>_*, writes
profiles/current_tick_pmc.json (what bench.py scales roofline.traffic and the second roofline axis from) and
profiles/README.md (which set describes HEAD).     python tools/save_round_profiles.py round2"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "round2"

B, W, K, REDRAW = 4096, 150, 200, 50          # W = bench.py's pre-roll (100) + warm-up (50): all untimed


def launch_plan(t0, t1, per_tick=False, redraw=REDRAW):            # bench.launch_plan (staged references), kept in step by hand
    out, t = [], t0
    while t < t1:
        n = 1 if (t < 2 or per_tick) else (t1 - t if t % redraw == 0 else min(t1, (t // redraw + 1) * redraw) - t)
        out.append((t, n)); t += n
    return out


names = ["tick", "ticko", "tickg", "pertick", "config5", "elem", "b1", "gramian", "dimitrov", "pldp", "preview", "zmpdisc"]
summ = {}
for k in names:
    d = os.path.join(ROOT, "gpurun_out", "prof_" + k)
    if not os.path.isdir(d):
        print("missing", d); continue
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), d], capture_output=True, text=True).stdout
    open(os.path.join(d, "summary.txt"), "w").write(txt)
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", f"{tag}_{k}_rocprofv3_summary.txt"))
    shutil.copy(os.path.join(d, "summary.json"), os.path.join(ROOT, "profiles", f"{tag}_{k}_rocprofv3_summary.json"))
    stats = sorted(glob.glob(os.path.join(d, "trace", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(ROOT, "profiles", f"{tag}_{k}_kernel_stats.csv"))
    summ[k] = json.load(open(os.path.join(d, "summary.json")))
    log = os.path.join(d, "trace.log")
    if k in ("gramian", "config5", "elem") and os.path.exists(log):
        keep = [ln for ln in open(log) if ("TFLOP" in ln or "ticks/s" in ln or "same" in ln or "run" in ln) and "amdgpu.ids" not in ln]
        open(os.path.join(ROOT, "profiles", f"{tag}_{k}_probe_output.txt"), "w").writelines(keep)


def per_gait_tick(s, kernel, gait_ticks):
    c = s["counters"][kernel]
    tot = lambda n: c[n]["mean_per_launch"] * c[n]["launches"]       # noqa: E731
    rd = tot("FETCH_SIZE") * 1024 * 2                               # MI355X_MICROARCH.md: KiB units; gfx950 reports half the read bytes
    wr = tot("WRITE_SIZE") * 1024
    waves_per_simd = 2
    r = {"kernel": kernel, "launches": int(c["FETCH_SIZE"]["launches"]), "gait_ticks": gait_ticks,
         "hbm_read_bytes_per_gait_tick": rd / gait_ticks, "hbm_write_bytes_per_gait_tick": wr / gait_ticks,
         "hbm_bytes_per_gait_tick": (rd + wr) / gait_ticks,
         "valu_insts_per_gait_tick": tot("SQ_INSTS_VALU") / gait_ticks, "salu_insts_per_gait_tick": tot("SQ_INSTS_SALU") / gait_ticks,
         "lds_insts_per_gait_tick": tot("SQ_INSTS_LDS") / gait_ticks, "vmem_insts_per_gait_tick": tot("SQ_INSTS_VMEM") / gait_ticks,
         "valu_active_frac_per_wave": tot("SQ_ACTIVE_INST_VALU") / tot("SQ_WAVE_CYCLES"),
         "valu_busy": waves_per_simd * tot("SQ_ACTIVE_INST_VALU") / tot("SQ_WAVE_CYCLES"),
         "wait_any_frac_per_wave": tot("SQ_WAIT_ANY") / tot("SQ_WAVE_CYCLES"),
         "wait_inst_any_frac_per_wave": tot("SQ_WAIT_INST_ANY") / tot("SQ_WAVE_CYCLES"),
         "mfma_f64_mops": tot("SQ_INSTS_VALU_MFMA_MOPS_F64"),
         "avg_kernel_ns_rocprofv3": s["kernels"][kernel]["avg_ns"], "kernel_calls_rocprofv3": s["kernels"][kernel]["calls"]}
    return r


PLAN = launch_plan(0, W) + launch_plan(W, W + K)              # bench.py: the untimed ticks, then the timed region
run_ticks = sum(n for _, n in PLAN if n > 1)
n_run = sum(1 for _, n in PLAN if n > 1)
plan_txt = ", ".join(str(n) for _, n in PLAN if n > 1)
out = {"tag": tag,
       "how": "rocprofv3 passes of `python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-parity --no-per-tick-leg "
              "--no-config5 --no-kernels --no-outs-leg` (tools/prof_round.sh): kernel trace + stats, then the counters in separate --pmc passes (FETCH_SIZE and "
              "WRITE_SIZE each in its own).  Units and gfx950 correction per MI355X_MICROARCH.md (HBM / rocprofv3): KiB x 1024, "
              "FETCH_SIZE doubled.  Totals over ALL launches of the kernel divided by the gait-ticks those launches ran "
              "(B = 4096; bench.py runs 100 pre-roll + 50 warm-up ticks untimed, then the timed 200; multi-tick launches: " + plan_txt + " ticks, "
              "the references of later stretches staged on the device; the device-wide queue takes one launch per stretch; per-tick: 350 launches).  valu_busy = "
              "2 waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES."}
if "tick" in summ:
    out["run_kernel"] = per_gait_tick(summ["tick"], "wg_mpc_run_xcd_kernel<16>", B * run_ticks)
    out["run_kernel"]["ticks_per_launch"] = plan_txt
    assert out["run_kernel"]["launches"] == n_run, (out["run_kernel"]["launches"], n_run)
if "ticko" in summ:
    # the same launches with the tick's deliverable stored (bench.py --outs-on): every gait-tick of the TIMED launch writes its
    # wg_tick_out_t; the untimed launches before it run without (so the per-gait-tick figures below are over the timed launch's
    # gait-ticks for the stores and over all launches for the rest: the stores are reported separately)
    o = per_gait_tick(summ["ticko"], "wg_mpc_run_xcd_kernel<16>", B * run_ticks)
    o["ticks_per_launch"] = plan_txt
    base = out.get("run_kernel")
    if base:
        # extra bytes of the outs-on run, all of them in the timed launch (K ticks)
        o["extra_write_bytes_per_stored_gait_tick"] = (o["hbm_write_bytes_per_gait_tick"] - base["hbm_write_bytes_per_gait_tick"]) * run_ticks / K
    out["run_kernel_outs"] = o
if "tickg" in summ:
    out["run_kernel_device_wide_queue"] = per_gait_tick(summ["tickg"], "wg_mpc_run_kernel<16>", B * run_ticks)
    out["run_kernel_device_wide_queue"]["ticks_per_launch"] = "48, 50 x 6"
if "pertick" in summ:
    out["per_tick_kernel"] = per_gait_tick(summ["pertick"], "wg_mpc_tick_kernel<16>", B * (W + K))
    out["per_tick_kernel"]["ticks_per_launch"] = "1"
if "elem" in summ:
    # the N = 32 element-view run kernel: probe_elem.py runs one 10-tick warm-up launch and PR = 3 launches of PT = 50 ticks
    EB, ET, ER, EW = 8192, 50, 3, 10
    kern = "wg_mpc_run_xcd_kernel<32>"                     # N = 32 has its own instantiation since round 4 (<-1>: any horizon)
    c = summ["elem"]["counters"][kern]
    gt = EB * (ET * ER + EW)
    tot = lambda n: c[n]["mean_per_launch"] * c[n]["launches"]       # noqa: E731
    assert int(c["FETCH_SIZE"]["launches"]) == ER + 1, c["FETCH_SIZE"]["launches"]
    rd_raw = tot("FETCH_SIZE") * 1024; wr = tot("WRITE_SIZE") * 1024
    cal = ""
    calf = os.path.join(ROOT, "gpurun_out", "prof_fetchcal", "summary.json")
    if os.path.exists(calf):
        cc = json.load(open(calf))["counters"]
        f8 = cc["wg_cal_rd<double>"]["FETCH_SIZE"]["mean_per_launch"] * 1024 / float(2 << 30)
        fc = cc["wg_cal_rd8_cols"]["FETCH_SIZE"]["mean_per_launch"] * 1024 / (51072 * 72 * 72 * 8.0)
        w8 = cc["wg_cal_wr<double>"]["WRITE_SIZE"]["mean_per_launch"] * 1024 / float(2 << 30)
        cal = ("tools/micro/fetchcal on this box: FETCH_SIZE reports %.3f of the bytes of an 8-B-per-lane coalesced stream, %.3f of the "
               "useful bytes of the Z^T a column walk (x 2 = %.2f: the walk over-fetches its 584-B-strided rows), WRITE_SIZE %.3f of "
               "8-B-per-lane stores" % (f8, fc, 2 * fc, w8))
        shutil.copy(os.path.join(ROOT, "gpurun_out", "fetchcal.txt"), os.path.join(ROOT, "profiles", f"{tag}_fetchcal_rates.txt"))
        shutil.copy(os.path.join(ROOT, "gpurun_out", "prof_fetchcal", "summary.txt"), os.path.join(ROOT, "profiles", f"{tag}_fetchcal_counters.txt"))
    out["elem_run_kernel"] = {
        "kernel": kern, "command": "PN=32 PB=8192 PT=50 PR=3 python3 tools/probe_elem.py", "batch": EB, "ticks_per_launch": "10, 50, 50, 50",
        "launches": ER + 1, "gait_ticks": gt,
        "hbm_read_bytes_per_gait_tick_uncorrected": rd_raw / gt, "hbm_read_bytes_per_gait_tick": 2 * rd_raw / gt,
        "hbm_write_bytes_per_gait_tick": wr / gt, "hbm_bytes_per_gait_tick": (2 * rd_raw + wr) / gt,
        "calibration": cal,
        "valu_insts_per_gait_tick": tot("SQ_INSTS_VALU") / gt, "salu_insts_per_gait_tick": tot("SQ_INSTS_SALU") / gt,
        "lds_insts_per_gait_tick": tot("SQ_INSTS_LDS") / gt, "vmem_insts_per_gait_tick": tot("SQ_INSTS_VMEM") / gt,
        "valu_busy": 3 * tot("SQ_ACTIVE_INST_VALU") / tot("SQ_WAVE_CYCLES"),        # three waves per SIMD
        "wait_any_frac_per_wave": tot("SQ_WAIT_ANY") / tot("SQ_WAVE_CYCLES"),
        "avg_kernel_ns_rocprofv3": summ["elem"]["kernels"][kern]["avg_ns"], "kernel_calls_rocprofv3": summ["elem"]["kernels"][kern]["calls"]}
if "b1" in summ:
    # one robot: one wave alone on a CU, one launch per tick (tools/probe_b1.py): what the wave's cycles are spent on
    kern = "wg_mpc_tick_kernel<16>"
    c = summ["b1"]["counters"][kern]
    m_ = lambda n: c[n]["mean_per_launch"]                           # noqa: E731
    out["one_robot_kernel"] = {
        "kernel": kern, "command": "python3 tools/probe_b1.py (B = 1, 200 launches of one tick)", "avg_kernel_ns_rocprofv3": summ["b1"]["kernels"][kern]["avg_ns"],
        "wave_cycles_per_tick": 4 * m_("SQ_WAVE_CYCLES"), "valu_insts_per_tick": m_("SQ_INSTS_VALU"), "salu_insts_per_tick": m_("SQ_INSTS_SALU"),
        "lds_insts_per_tick": m_("SQ_INSTS_LDS"), "vmem_insts_per_tick": m_("SQ_INSTS_VMEM"),
        "executing_frac": m_("SQ_ACTIVE_INST_ANY") / m_("SQ_WAVE_CYCLES"), "valu_executing_frac": m_("SQ_ACTIVE_INST_VALU") / m_("SQ_WAVE_CYCLES"),
        "lds_executing_frac": m_("SQ_ACTIVE_INST_LDS") / m_("SQ_WAVE_CYCLES"), "parked_in_waitcnt_frac": m_("SQ_WAIT_ANY") / m_("SQ_WAVE_CYCLES"),
        "issue_stalled_frac": m_("SQ_WAIT_INST_ANY") / m_("SQ_WAVE_CYCLES"),
        "note": "SQ_WAVE_CYCLES and the SQ_ACTIVE / SQ_WAIT counters are in quad-cycles; fractions are of the wave's lifetime"}
# the device sources the profiled run was built from: written on the GPU box by tools/prof_round.sh (its copy of the tree)
hp = os.path.join(ROOT, "gpurun_out", "csrc_hash.txt")
out["csrc_sha256"] = open(hp).read().split()[0] if os.path.exists(hp) else None
json.dump(out, open(os.path.join(ROOT, "profiles", "current_tick_pmc.json"), "w"), indent=1)
lat = os.path.join(ROOT, "gpurun_out", "latency_b1.json")
if os.path.exists(lat) and os.path.getsize(lat) > 0:
    shutil.copy(lat, os.path.join(ROOT, "profiles", f"{tag}_latency_b1.json"))
for src, dst in (("phases_tick.txt", "tick_phase_timers.txt"), ("phases_tick32.txt", "tick32_phase_timers.txt"), ("launch_fit.txt", "launch_fit.txt")):
    ph = os.path.join(ROOT, "gpurun_out", src)
    if os.path.exists(ph):
        shutil.copy(ph, os.path.join(ROOT, "profiles", f"{tag}_{dst}"))
agree = ""
if "run_kernel" in out:
    r = out["run_kernel"]
    tot_ms = r["avg_kernel_ns_rocprofv3"] * r["kernel_calls_rocprofv3"] / 1e6
    agree = (f"How the run kernel's table lines up with `bench.py`: rocprofv3 saw {r['kernel_calls_rocprofv3']} launches of "
             f"`wg_mpc_run_xcd_kernel<16>` ({r['ticks_per_launch']} ticks: the untimed launches and the timed one), {tot_ms:.1f} ms in "
             f"all = {tot_ms / run_ticks:.4f} ms per tick of the batch; the table's average ({r['avg_kernel_ns_rocprofv3'] / 1e6:.1f} ms) is "
             f"over launches of different lengths, its MaxNs is the timed launch that `bench.py` brackets with HIP events "
             f"(`roofline.kernel_ms` in `{tag}_bench.json`, / `ticks_per_launch` = ms per tick).")
readme = f"""# profiles/ -- one set, describing HEAD

Everything named `{tag}_*` was produced by ONE run of `tools/prof_round.sh` on an MI355X (gpurun) and filed by
`tools/save_round_profiles.py {tag}`.  Older sets (`round4_*`: what `docs/HISTORY.md` cites) stay as the record of their round; they
do not describe HEAD.  `current_tick_pmc.json` and `{tag}_resource_usage.txt` carry the hash of the device sources they were made
from (`tools/csrc_hash.py`); `tests/test_docs_numbers.py` fails when that is not HEAD's.

| files | command profiled | what to read there |
|---|---|---|
| `{tag}_tick_*` | `python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-parity --no-per-tick-leg --no-config5 --no-kernels --no-outs-leg` | the benchmarked kernel `wg_mpc_run_xcd_kernel<16>` alone: B = 4096, the untimed launches (pre-roll + warm-up: 48 and 100 ticks) and the timed one of 200 ticks with the velocity references of its four stretches staged on the device (plus the two single-tick launches of the control loop's first ticks under their own kernel name); kernel-trace stats and the PMC passes |
| `{tag}_tickg_*` | the same with `WG_RUN_QUEUE=global` | the device-wide queue of round 1 (`wg_mpc_run_kernel<16>`): the L2 write-back traffic the XCD-local hand-over removed |
| `{tag}_pertick_*` | the same with `--per-tick-launch` | `wg_mpc_tick_kernel<16>`, 350 launches of one tick |
| `{tag}_config5_*` | `PN=32 PB=8192 PT=50 python3 tools/probe_run.py` | BASELINE configs[4]'s size: N = 32, B = 8192 (element view), per-tick and multi-tick launches |
| `{tag}_elem_*` | `PN=32 PB=8192 PT=50 PR=3 python3 tools/probe_elem.py` | the element-view run kernel `wg_mpc_run_xcd_kernel<32>` alone (what `bench.py`'s `config5` leg times): kernel trace and all PMC passes; `current_tick_pmc.json` -> `elem_run_kernel` holds its HBM-side traffic per gait-tick (FETCH_SIZE x 2 and uncorrected, WRITE_SIZE) |
| `{tag}_fetchcal_*` | `tools/micro/fetchcal` (plain, then `tools/pmc_traffic.sh`) | FETCH_SIZE / WRITE_SIZE against known byte counts: 4 / 8 / 16 B per lane coalesced, the Z^T a column walk and the sweep's row walk of a 72 x 73 slot |
| `{tag}_ticko_*` | the same with `--outs-on` | the timed launch stores every gait-tick's `wg_tick_out_t`: WRITE_SIZE of the `outs_on` leg (`current_tick_pmc.json` -> `run_kernel_outs`) |
| `{tag}_b1_*` | `python3 tools/probe_b1.py` | one robot, one wave alone on a CU: the counters behind DESIGN 4.4 / 4.0 ("one robot") (`current_tick_pmc.json` -> `one_robot_kernel`) |
| `round4_regz_*` | `bash tools/regz_probe.sh` (experiment builds, `-DWG_WITH_REGZ`; round 4) | the "Z on chip" experiment at N = 32 (docs/HISTORY.md 3.2): parity, rate at four and eight gaits per CU, all counters of the four-per-CU build |
| `round5_mw_barrier.txt`, `round5_mw_iter.txt` | `tools/micro/barrier`, `tools/micro/mw_iter` (round 5) | the multi-wave "Z in LDS" layout at N = 32 measured instead of built (DESIGN 4.3): cost of an `s_barrier` hand-over at three workgroups per CU; cycles per active-set iteration of a W-wave workgroup, per phase, against what the shipped kernel needs |
| `{tag}_phase_attribution.txt` | `bash tools/phase_attribution.sh` + `python tools/phase_attribution.py`, then the timer table of `{tag}_tick_phase_timers.txt` | per-phase counters of the N = 16 run kernel (phases executed twice, differences against the plain build) and the shader-clock split of everything the counters cannot repeat |
| `{tag}_phase_attribution_n32.txt` | `ATTR_DIR=attr32 PN=32 PB=8192 PT=50 PR=2 bash tools/phase_attribution.sh` + `... python tools/phase_attribution.py` | the same counter attribution for the N = 32 kernel `wg_mpc_run_xcd_kernel<32>` at the benchmark's residency (back substitution 12.1 %, norm chain 11.6 %, scan 11.0 %, Z^T a 10.4 %) |
| `{tag}_latency_b1.json` | `jrl-walkgen_amd/bin/latency_b1` | one robot (B = 1): host-pointer call, its split (copy in / launch / kernel / copy out) and the host-mapped call |
| `{tag}_resource_usage.txt` | `python tools/isa_audit.py` (no GPU) | registers, spills, scratch, occupancy of every kernel; where the spill code sits by loop depth; instruction mix of the inner loops |
| `{tag}_gramian_*` | `python3 tools/probe_gramian.py` | `wg_gramian_kernel`: SQ_INSTS_VALU_MFMA_MOPS_F64 / _F32, SQ_VALU_MFMA_BUSY_CYCLES, duration against the dense MFMA peak (`*_probe_output.txt`) |
| `{tag}_dimitrov_*`, `{tag}_pldp_*`, `{tag}_preview_*`, `{tag}_zmpdisc_*` | `tools/probe_<name>.py` | the other kernels of the path |
| `{tag}_tick_phase_timers.txt`, `{tag}_tick32_phase_timers.txt` | `PB=4096 python3 tools/probe_tick_phases.py`, `PN=32 PB=3072 ...` (diagnostic build `lib/libwg_mpc_prof.so`) | in-kernel phase timers of the tick at N = 16 and N = 32 (shader cycles per gait-tick, one launch per tick) |
| `{tag}_launch_fit.txt` | `python3 tools/probe_launch_fit.py` (N = 16, B = 4096; then N = 32, B = 8192) | duration of a multi-tick launch against its length, least squares: the steady rate and what every launch pays once (ramp and the idle tail in which the last gait-ticks finish) -- the difference between the 200-step figure and the driver's 20-step window |
| `{tag}_soak_parity.txt` | `python tools/soak_parity.py` | every gait of the benchmark workload (4096 x 250 ticks at N = 16, 8192 x 50 at N = 32) advanced as `bench.py` does it, final states byte for byte against the CPU checker on the host cores |
| `{tag}_soak_parity_long.txt` | `SOAK_LONG=1 python tools/soak_parity.py` | the same at four times the length (4096 x 1000 ticks at N = 16) and through the element view at N = 20, 24, 28, 32: 5.0 M MPC ticks byte for byte |
| `{tag}_soak_vscale.txt` | `SOAK_VSCALE=3 python tools/soak_parity.py` | the same workload with its velocity references times three: failed QPs, states beyond 1e140, NaN iterates -- every gait against the oracle to its last tick (DESIGN 3.3) |
| `{tag}_fuzz_ql.txt` | `python tools/fuzz_ql.py 20000` | the dense boundary against the oracle: 322 000 QPs of seventeen families, ifail / iterations / history / x bit for bit |
| `{tag}_fuzz_oracle_vs_reference.txt` | `python tools/fuzz_oracle_vs_reference.py 20000` (development container, no GPU) | the oracle's QL restatement against the COMPILED reference qld.cpp on the same 322 000 QPs, histories included |
| `current_tick_pmc.json` | derived from `{tag}_tick_*`, `{tag}_tickg_*`, `{tag}_pertick_*` | per gait-tick: HBM bytes read / written (FETCH_SIZE x 2 and WRITE_SIZE, KiB units, separate passes), VALU / SALU / LDS / VMEM instructions, VALU busy; `bench.py` scales `roofline.traffic` and its second axis from this file |

{agree}

How the set is regenerated (≈ 25 minutes, 8 of them on the GPU): build `lib/libwg_mpc_prof.so` and the eleven
`lib/libwg_mpc_xr<k>.so` (`EXTRA=-DWG_REPEAT_PHASE=k`); `gpurun -- bash tools/gpu_round_a.sh` (GPU tests, then `tools/prof_round.sh`);
`python tools/save_round_profiles.py {tag}`; `gpurun -- bash tools/gpu_round_b.sh` (both attributions, both bench lines, both soaks);
`python tools/phase_attribution.py > profiles/{tag}_phase_attribution.txt` (and with `ATTR_DIR=attr32 PN=32 PB=8192 PT=50 PR=2` for
`_n32`), copy the soaks, `python tools/file_bench.py {tag}`, `python tools/isa_audit.py --out profiles/{tag}_resource_usage.txt`,
`python tools/doc_numbers.py {tag}` (the documents' generated blocks; `tests/test_docs_numbers.py` fails while they disagree).

Each `*_rocprofv3_summary.txt/json` = per-kernel averages of the trace (`kernels`) and per-launch means of every counter
(`counters`); `*_kernel_stats.csv` = rocprofv3's own `--stats` table of the same run.
"""
open(os.path.join(ROOT, "profiles", "README.md"), "w").write(readme)
for k in ("run_kernel", "run_kernel_device_wide_queue", "per_tick_kernel"):
    if k in out:
        r = out[k]
        print("%-30s %7.0f B read + %7.0f B written per gait-tick, %6.0f VALU, valu_busy %.2f, avg %.3f ms"
              % (k, r["hbm_read_bytes_per_gait_tick"], r["hbm_write_bytes_per_gait_tick"], r["valu_insts_per_gait_tick"], r["valu_busy"],
                 r["avg_kernel_ns_rocprofv3"] / 1e6))
