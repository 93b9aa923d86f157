#!/usr/bin/env python3
"""bench.py -- MPC ticks/s of the MI355X-native Herdt-2010 hot path.

Metric (BASELINE.json): QP-MPC ticks/sec at batch = 4096 gaits per GPU, horizon N = 16, fp64.
One *step* = one MPC tick for every gait of the batch (support-state preview, orientation preview, QP assembly, QL dual
active-set solve, LIPM update, feet).  Steps between two changes of the velocity references go into ONE launch of the
multi-tick kernel (wg_mpc_run_batch_dev): a gait's tick t+1 depends only on its own tick t, so the batch does not drain
between ticks; results are identical to one launch per tick (tests/test_run_gpu.py).  The JSON line also carries that
mode's rate, measured in the same process on the following ticks ("per_tick_launch"); --per-tick-launch makes it the
primary figure.
Workload (SURVEY.md section 8d, config[2]): B independent gaits, common start state, per-gait piecewise-constant
velocity references vx~U[-0.1,0.3], vy~U[-0.1,0.1], w~U[-0.2,0.2] redrawn every 5 s (50 ticks) from
MT19937-64 seeded 20100 + global gait index.  All inputs are resident in HBM before the timed region.

    python bench.py                       # 1 GPU, K = 200 steps, W = 50 warm-up steps
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W      # weak scaling: 4096 gaits per GPU, one RCCL broadcast
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd")
shard = importlib.import_module("jrl-walkgen_amd.shard")
import bench_kernels  # noqa: E402

BATCH_PER_GPU = 4096
REDRAW_TICKS = 50            # 5 s of walking
PREROLL_TICKS = 100          # untimed ticks before the warm-up: every timed window, however short, sees the de-synchronised batch
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (spec): 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
BOX_CPU_SHARE = 16           # host cores that go with one GPU of the pool's boxes (used only when the OS shows no limit)
CONFIG5_BATCH = 8192         # BASELINE configs[4]: N = 32 with foot-placement variables, batch = 8192
# counters of the timed kernel measured by rocprofv3 --pmc passes of this very command (tools/prof_round.sh files them)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "current_tick_pmc.json")


def pmc_file_is_current(pmc):
    """profiles/current_tick_pmc.json carries the hash of the device sources it was measured on (tools/csrc_hash.py): counters
    of another kernel than the one being timed are reported as such, not silently scaled"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from csrc_hash import csrc_hash
        return pmc.get("csrc_sha256") == csrc_hash()
    except Exception:                                                      # noqa: BLE001
        return False


def velocity_table(lo, hi, n_seg):
    """[n_seg, B, 3] references; gait g uses MT19937-64 seeded 20100 + g (global index)."""
    tab = np.empty((n_seg, hi - lo, 3))
    for k, g in enumerate(range(lo, hi)):
        r = np.random.Generator(np.random.MT19937(20100 + g))
        tab[:, k, 0] = r.uniform(-0.1, 0.3, n_seg)
        tab[:, k, 1] = r.uniform(-0.1, 0.1, n_seg)
        tab[:, k, 2] = r.uniform(-0.2, 0.2, n_seg)
    return tab


def start_states(model, B):
    s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
    s0.nb_steps_left = 2                                    # ":numberstepsbeforestop 2"
    one = bytes(memoryview(s0).cast("B"))
    return torch.frombuffer(bytearray(one * B), dtype=torch.uint8)


def launch_plan(t0, t1, per_tick=False, redraw=REDRAW_TICKS, staged=True):
    """[(first tick, number of ticks)] covering ticks [t0, t1): the control loop's first two ticks advance the clock by 1
    and 19 control periods and go alone (or every tick does, per_tick); a launch that starts where the velocity references
    change runs to t1 with the references of all its stretches staged on the device (wg_mpc_run_sched_dev), any other
    launch ends where they change next (staged = False: every launch does -- round 1's plan)."""
    out = []
    t = t0
    while t < t1:
        if t < 2 or per_tick:
            n = 1
        elif staged and t % redraw == 0:
            n = t1 - t
        else:
            n = min(t1, (t // redraw + 1) * redraw) - t
        out.append((t, n))
        t += n
    return out


def algorithmic_bytes(n, m):
    """Bytes the reference moves per tick at its own solver boundary (ql0001_ arguments):
    read 8*(n^2 + n + mmax*n + mmax + 2n), written 8*(n + m + 2n), with mmax = m + 1."""
    mmax = m + 1
    return 8.0 * (n * n + n + mmax * n + mmax + 2 * n) + 8.0 * (n + m + 2 * n)


def _cpu_run(args):
    """one process = one core: gaits [g0, g0+ng) of the benchmark workload through the CPU checker, loop in C"""
    g0, ng, n_ticks, use_ref = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as ol
    lib = ol.oracle()
    kind = "port"
    lib.wgo_set_reference_ql(None)
    if use_ref and ol.have_ref():
        lib.wgo_set_reference_ql(C.cast(getattr(ol.ref(), ol.REF_SYM), C.c_void_p))
        kind = "reference"
    model = wg.Model()
    lib.wgo_model_defaults(C.byref(model))
    tab = np.ascontiguousarray(velocity_table(g0, g0 + ng, (n_ticks + REDRAW_TICKS - 1) // REDRAW_TICKS))
    states = (wg.GaitState * ng)()
    s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
    s0.nb_steps_left = 2
    for g in range(ng):
        C.memmove(C.byref(states[g]), C.byref(s0), C.sizeof(wg.GaitState))
    lib.wgo_solve_timer(1, None, None)
    t0 = time.perf_counter()
    rc = lib.wgo_mpc_run(C.byref(model), states, ng, n_ticks, tab.ctypes.data_as(C.c_void_p), REDRAW_TICKS)
    dt = time.perf_counter() - t0
    sec, cnt = C.c_double(), C.c_long()
    lib.wgo_solve_timer(0, C.byref(sec), C.byref(cnt))
    assert rc == 0
    return ng * n_ticks, dt, kind, (sec.value if kind == "reference" else None), cnt.value


def usable_cores():
    """(cores this process may really use, what the OS shows, why): the affinity mask cut down by the cgroup CPU quota.
    os.cpu_count() is the host's core count -- on a shared GPU box most of them belong to somebody else."""
    visible = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = visible
    quota = None
    dirs = ["/sys/fs/cgroup"]
    try:                                                           # cgroup v2: this process's own group and every ancestor
        for l in open("/proc/self/cgroup"):
            if l.startswith("0::"):
                rel = l.strip()[3:].strip("/")
                while rel:
                    dirs.append(os.path.join("/sys/fs/cgroup", rel))
                    rel = os.path.dirname(rel)
    except OSError:
        pass
    for d in dirs:
        try:                                                       # "<quota|max> <period>"
            q, per = open(os.path.join(d, "cpu.max")).read().split()
            if q != "max":
                quota = min(quota or 1e9, float(q) / float(per))
        except (OSError, ValueError):
            pass
    if quota is None:
        try:                                                       # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    used = affinity if quota is None else max(1, min(affinity, int(quota + 1e-9)))
    note = None
    if quota is None and affinity == visible and visible > BOX_CPU_SHARE:
        # nothing the process can read limits it, yet a one-GPU lease of a many-core host owns a share of it
        used = BOX_CPU_SHARE
        note = "no affinity mask / cgroup quota visible on a %d-core host: the one-GPU lease's share (%d) assumed" % (visible, BOX_CPU_SHARE)
    cap = os.environ.get("WG_BENCH_CPU_CORES")                     # explicit override for boxes whose limits are not visible
    if cap:
        used = max(1, min(used, int(cap)))
    return used, {"os_cpu_count": visible, "affinity": affinity, "cgroup_quota": quota, "note": note}


def _cpu_worker(w, per, n_ticks, barrier, q):
    try:
        barrier.wait(120)                                          # every worker starts its timed loop together
        q.put((w,) + tuple(_cpu_run((w * per, per, n_ticks, True))))
    except Exception as e:                                         # noqa: BLE001
        q.put((w, 0, 1.0, "error: %s" % e, None, 0))


def cpu_model_name():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(n_gaits, n_ticks):
    """The CPU checker on the first `n_gaits` gaits of the same workload: tick assembly = oracle/ C restatement,
    QP solve = the reference's own qld.cpp compiled -O3 -DNDEBUG (oracle/_ref, ~all of the CPU time) when it was built,
    else the restated solver.  Timed on one core, then on every core this process may use (affinity mask and cgroup quota, not
    os.cpu_count(); one process per core, QLD keeps statics; >= 2 s of work per worker, all started behind one barrier).
    Checker code timed as a baseline -- never part of the measured GPU path."""
    import multiprocessing as mp
    ticks, dt, kind, solve_s, solves = _cpu_run((0, n_gaits, n_ticks, True))
    one = dict(value=ticks / dt, seconds=dt, kind=kind,
               solve_only=(solves / solve_s if solve_s else None))
    cores, seen = usable_cores()
    # >= 2 s per worker at the one-core rate just measured (more when the cores are slower under load), <= ~6 s
    per = int(min(max(8, np.ceil(2.5 * one["value"] / n_ticks)), 4 * n_gaits))
    try:
        ctx = mp.get_context("fork")
        barrier, q = ctx.Barrier(cores), ctx.Queue()
        procs = [ctx.Process(target=_cpu_worker, args=(w, per, n_ticks, barrier, q)) for w in range(cores)]
        t0 = time.perf_counter()
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in procs]
        for p in procs:
            p.join()
        wall = time.perf_counter() - t0
        res = [r[1:] for r in sorted(res)]
        bad = [r[2] for r in res if isinstance(r[2], str) and r[2].startswith("error")]
        if bad:
            raise RuntimeError(bad[0])
        rates = [r[0] / r[1] for r in res]
        value = sum(r[0] for r in res) / max(r[1] for r in res)
        eff = value / (cores * one["value"])
        allc = dict(value=value, cores=cores, cores_used=cores, cores_visible=seen["os_cpu_count"], cores_affinity=seen["affinity"],
                    cgroup_cpu_quota=seen["cgroup_quota"], cores_note=seen["note"], wall_seconds=wall,
                    worker_seconds_min=min(r[1] for r in res), worker_seconds_max=max(r[1] for r in res),
                    worker_rate_min=min(rates), worker_rate_max=max(rates), parallel_efficiency=eff,
                    cpu_model=cpu_model_name(),
                    sample="%d gaits x %d ticks per core, %d workers started together" % (per, n_ticks, cores))
        if eff < 0.5:
            allc["warning"] = ("parallel efficiency %.2f: %d workers deliver %.1f x one core -- the cores this process was told it may "
                               "use are shared or throttled; WG_BENCH_CPU_CORES sets the worker count by hand" % (eff, cores, value / one["value"]))
        if all(r[3] for r in res):
            # every core's solves at that core's own solve-only rate
            allc["solve_only"] = sum(r[4] / r[3] for r in res)
    except Exception as e:                                        # noqa: BLE001 -- a baseline, never fatal
        allc = dict(value=None, cores=cores, cores_visible=seen["os_cpu_count"], error=str(e))
    return one, allc


def oracle_golden_rows():
    """cpu_baseline leg (runs before this process touches the GPU): the EmergencyStop scenario through the CPU checker -- libm
    trigonometry, every QP solved by the reference's compiled ql0001_ when oracle/_ref was built -- as the 38-column rows of
    tests/TestObject.cpp.  golden_parity() compares the GPU's replay with it: the file is printed to 1e-7, this is not."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import herdt_replay as hr
    import oraclelib as ol
    lib = ol.oracle()
    kind = "port"
    lib.wgo_set_reference_ql(None)
    if ol.have_ref():
        lib.wgo_set_reference_ql(C.cast(getattr(ol.ref(), ol.REF_SYM), C.c_void_p))
        kind = "reference"
    datref = np.load(os.path.join(ROOT, "tests", "golden", "herdt_emergency_stop_datref.npz"))["datref"]
    model, state, events = hr.emergency_stop_setup(datref)
    rows = hr.replay(model, state, events, 6000, tick=hr.oracle_tick, legacy_running=True)
    lib.wgo_set_reference_ql(None)
    return rows, kind


def golden_parity(oracle_rows=None):
    """The second half of BASELINE's metric ("CoM RMSE vs ref"): the reference's own golden file
    TestHerdt2010EmergencyStopTestFGPI.datref (config 2: one gait, 22.5 s) replayed with every MPC tick on the GPU through
    the C ABI (tests/herdt_replay.py is the 5 ms control loop around the tick; no oracle call on this path)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import herdt_replay as hr
    datref = np.load(os.path.join(ROOT, "tests", "golden", "herdt_emergency_stop_datref.npz"))["datref"]
    m = wg.model_defaults()
    m.flags = 1                                          # WG_FLAG_NO_STOP_CENTERING: the file predates that branch
    s = wg.gait_init(m, [datref[0, 1], datref[0, 2], datref[0, 3]], [datref[0, 10], datref[0, 11], 0.0],
                     [datref[0, 22], datref[0, 23], 0.0])
    s.nb_steps_left = 2; s.nb_steps_ssds = 2; s.sup_y = 0.1
    wg.mpc_configure(m)

    lat = []

    def tick(model, state, want_dump):
        arr = (wg.GaitState * 1)()
        C.memmove(C.byref(arr[0]), C.byref(state), C.sizeof(wg.GaitState))
        t0 = time.perf_counter()
        outs, _, _, _ = wg.mpc_tick_batch(arr, want_out=True)
        lat.append(time.perf_counter() - t0)
        C.memmove(C.byref(state), C.byref(arr[0]), C.sizeof(wg.GaitState))
        out = wg.TickOut()
        C.memmove(C.byref(out), C.byref(outs[0]), C.sizeof(wg.TickOut))
        return out, None

    rows = hr.replay(m, s, hr.emergency_stop_events(), 6000, tick=tick, legacy_running=True)
    if rows.shape != datref.shape:
        return {"error": "replay produced %s rows, the golden file has %s" % (rows.shape, datref.shape)}
    # the same robot through the one-robot entry point: state and outputs in host-mapped memory (wg_mpc_tick_pinned), no copies
    lat_hm = []
    try:
        hm = wg.HostMapped()
        s1 = wg.gait_init(m, [datref[0, 1], datref[0, 2], datref[0, 3]], [datref[0, 10], datref[0, 11], 0.0], [datref[0, 22], datref[0, 23], 0.0])
        s1.nb_steps_left = 2; s1.vref[0] = 0.2
        C.memmove(C.addressof(hm.state), C.byref(s1), C.sizeof(wg.GaitState))
        for k in range(120):
            t0 = time.perf_counter()
            hm.tick(1 if k == 0 else (19 if k == 1 else 20))
            lat_hm.append(time.perf_counter() - t0)
        hm.close()
    except Exception:                                                      # noqa: BLE001 -- reported as missing
        lat_hm = []
    d = rows - datref
    vs_oracle = {}
    if oracle_rows is not None and oracle_rows[0].shape == rows.shape:
        do = rows - oracle_rows[0]
        vs_oracle = {"com_max_abs_vs_libm_oracle_m": float(np.abs(do[:, 1:3]).max()),
                     "com_rmse_vs_libm_oracle_m": float(np.sqrt((do[:, 1:3] ** 2).mean())),
                     "max_abs_vs_libm_oracle_all_columns": float(np.abs(do).max()),
                     "libm_oracle": "oracle/herdt_oracle.c with libm trigonometry, QP solve by %s, same scenario on the host "
                                    "(cpu_baseline leg); BASELINE's bar is CoM <= 1e-9 m" %
                                    ("the reference's compiled ql0001_ (oracle/_ref)" if oracle_rows[1] == "reference" else "the restated QL")}
    return {"com_rmse_m": float(np.sqrt((d[:, 1:3] ** 2).mean())), "max_abs_err": float(np.abs(d).max()), **vs_oracle,
            "rows": int(datref.shape[0]), "columns": int(datref.shape[1]), "tolerance": 1e-6,
            # what a drop-in user with ONE robot sees per MPC tick (every 0.1 s of walking): wg_mpc_tick_batch(B = 1) with
            # host pointers = copy in, launch, synchronise, copy out
            "b1_tick_latency_us": {"median": float(np.median(lat) * 1e6), "p90": float(np.quantile(lat, 0.9) * 1e6),
                                   "ticks": len(lat),
                                   "host_mapped_median": float(np.median(lat_hm[20:]) * 1e6) if lat_hm else None,
                                   "host_mapped_note": "wg_mpc_tick_pinned: state / outputs in host-mapped memory, the call spins on "
                                                       "the kernel's completion counter; what is left is the kernel itself (one wave "
                                                       "alone on a CU; jrl-walkgen_amd/bin/latency_b1 has the split)"},
            "reference": "TestHerdt2010EmergencyStopTestFGPI.datref (the reference's golden file, printed to 1e-7), "
                         "every MPC tick on the GPU, B = 1"}


def config5_leg(dev, flags, ticks=40, warm=10):
    """BASELINE configs[4] at its stated size on this GPU: N = 32 with foot-placement variables, batch = 8192, in its own
    context (the N = 16 context of the main measurement stays configured).  flags = 0: Q_b from the reference-order host loop
    (bit-exact tick); flags = WG_FLAG_GRAMIAN_MFMA_F32: Q_b from the fp32 matrix-core Gramian (tolerance mode).  The QL
    solve is fp64 either way.  Same workload recipe as the main measurement (seeds 20100 + gait, redraw every 50 ticks)."""
    model = wg.model_defaults()
    model.N = 32
    model.flags = flags
    B = CONFIG5_BATCH
    with wg.Context(dev.index or 0) as ctx:
        ctx.mpc_configure(model)
        n_seg = (warm + ticks + REDRAW_TICKS - 1) // REDRAW_TICKS
        vtab = torch.from_numpy(velocity_table(0, B, n_seg)).to(dev)
        states = start_states(model, B).to(dev)
        diag = torch.zeros(warm + ticks, B, 6, dtype=torch.int32, device=dev)
        stream = torch.cuda.Stream(device=dev)
        sh, sp, dp, dstride = stream.cuda_stream, states.data_ptr(), diag.data_ptr(), B * 6 * 4

        def run(t0, t1, evs=None):
            for t, n in launch_plan(t0, t1):
                staged = n > 1 and t % REDRAW_TICKS == 0
                if t % REDRAW_TICKS == 0 and not staged:
                    ctx.mpc_set_velref_dev(B, sp, vtab[t // REDRAW_TICKS].data_ptr(), sh)
                adv = 1 if t == 0 else (19 if t == 1 else 20)
                if evs is not None:
                    evs.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), n))
                    evs[-1][0].record(stream)
                if n == 1:
                    ctx.mpc_tick_batch_dev(B, sp, None, dp + t * dstride, adv, stream=sh)
                elif staged:
                    ctx.mpc_run_sched_dev(B, sp, n, vtab[t // REDRAW_TICKS].data_ptr(), REDRAW_TICKS, adv, None, dp + t * dstride, stream=sh)
                else:
                    ctx.mpc_run_batch_dev(B, sp, n, adv, None, dp + t * dstride, stream=sh)
                if evs is not None:
                    evs[-1][1].record(stream)
        with torch.cuda.stream(stream):
            run(0, warm)
        torch.cuda.synchronize(dev)
        evs = []
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            run(warm, warm + ticks, evs)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        d = diag[warm:].cpu().numpy().reshape(-1, 6)
    kern_s = sum(a.elapsed_time(b) for a, b, _ in evs) * 1e-3
    alg = float(algorithmic_bytes(d[:, 3].astype(np.float64), d[:, 4].astype(np.float64)).sum())
    # HBM-side traffic of this kernel from the rocprofv3 --pmc passes filed in profiles/ (per gait-tick there, per step here)
    traffic = {"traffic": None}
    kern_name = "wg_mpc_run_xcd_kernel<32>"                   # N = 32 has its own instantiation (the any-horizon one is <-1>)
    try:
        e = json.load(open(PMC_SUMMARY))["elem_run_kernel"]
        kern_name = e.get("kernel", kern_name)
        traffic = {"traffic": e["hbm_bytes_per_gait_tick"] * B, "traffic_read": e["hbm_read_bytes_per_gait_tick"] * B,
                   "traffic_read_uncorrected": e["hbm_read_bytes_per_gait_tick_uncorrected"] * B,
                   "traffic_write": e["hbm_write_bytes_per_gait_tick"] * B,
                   "traffic_per_gait_tick": e["hbm_bytes_per_gait_tick"],
                   "traffic_source": "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s` (B = %d, launches of %s ticks), per "
                                     "gait-tick x this batch; FETCH_SIZE x 2 as calibrated on 8-byte lanes in this kernel's access "
                                     "shapes (%s)" % (os.path.relpath(PMC_SUMMARY, ROOT), e["command"], e["batch"], e["ticks_per_launch"],
                                                      e["calibration"])}
        # the rate at which this kernel moves those bytes: what actually bounds it
        traffic["traffic_gbs"] = traffic["traffic"] * ticks / kern_s / 1e9
        traffic["traffic_frac_of_peak"] = traffic["traffic_gbs"] / HBM_PEAK_GBS
    except Exception:                                                      # noqa: BLE001 -- a missing profile is not fatal
        pass
    return {"value": B * ticks / wall, "unit": "ticks/s", "batch": B, "horizon_N": 32, "steps": ticks, "warmup": warm,
            "ms_per_step": 1e3 * wall / ticks, "kernel_ms_per_step": 1e3 * kern_s / ticks,
            "hessian_source": "fp32 matrix-core Gramian (WG_FLAG_GRAMIAN_MFMA_F32)" if flags & 4 else "reference-order host loop (bit-exact)",
            "solve_dtype": "f64", "failed_qps": int((d[:, 0] != 0).sum()),
            "mean_iterations": float(d[:, 1].mean()), "max_iterations": int(d[:, 1].max()),
            "n_hist": {str(int(k)): int(v) for k, v in zip(*np.unique(d[:, 3], return_counts=True))},
            "roofline": dict({"bound": "hbm", "achieved": alg / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": alg / kern_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg / ticks,
                              "kernel": kern_name}, **traffic)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--per-tick-launch", action="store_true", help="one launch per tick (wg_mpc_tick_batch_dev) instead of "
                    "one launch per stretch of constant velocity references")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="gaits per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the golden-file replay (CoM RMSE) after the timed run")
    ap.add_argument("--no-per-tick-leg", action="store_true", help="skip the secondary one-launch-per-tick measurement")
    ap.add_argument("--no-config5", action="store_true", help="skip the N = 32, batch = 8192 leg (BASELINE configs[4])")
    ap.add_argument("--no-kernels", action="store_true", help="skip the `kernels` legs (PLDP, Dimitrov tick, dense ql0001_ boundary, "
                    "preview, zmpdisc, Gramian)")
    ap.add_argument("--preroll", type=int, default=PREROLL_TICKS, help="untimed ticks run before the warm-up steps (redraws every "
                    "50 as everywhere): a short timed window then sees gaits in every support phase, QPs of n = 32, 34 and 36")
    ap.add_argument("--outs-on", action="store_true", help="the timed launches store the tick's deliverable (wg_tick_out_t: the 20 CoM / "
                    "ZMP / feet samples per tick the reference pushes on its deques, ZMPVelocityReferencedQP.cpp:405-442) for every "
                    "gait-tick -- the primary figure then is the outs-on rate (profiling runs; the default line carries it as `outs_on`)")
    ap.add_argument("--no-outs-leg", action="store_true", help="skip the secondary outs-on measurement")
    ap.add_argument("--no-staged-refs", action="store_true", help="one launch per stretch of constant velocity references "
                    "(wg_mpc_set_velref_dev + wg_mpc_run_batch_dev, round 1's plan) instead of one launch with the references "
                    "of every stretch staged on the device (wg_mpc_run_sched_dev)")
    args = ap.parse_args()

    # CPU baseline first: it forks one worker per host core, which must happen before this process touches the GPU
    cpu_line = None
    oracle_rows = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        if not args.no_parity:
            try:
                oracle_rows = oracle_golden_rows()
            except Exception:                                             # noqa: BLE001 -- the checker's figure, never fatal
                oracle_rows = None
        ng, nt = 1024, 200                                   # ~13 s on one core
        one, allc = cpu_baseline(ng, nt)
        solver = ("QP solve by the reference's qld.cpp compiled -O3 -DNDEBUG (oracle/_ref)" if one["kind"] == "reference"
                  else "QP solve by the restated QL (oracle/ql_oracle.c)")
        cpu_line = {"value": one["value"], "unit": "ticks/s", "cores": 1, "kind": one["kind"],
                    "sample": f"first {ng} gaits x {nt} ticks of the same workload ({ng * nt} ticks, "
                              f"{one['seconds']:.1f} s); tick assembly by the oracle/ C restatement, {solver}; "
                              f"1 core of {allc.get('cores_used', allc['cores'])} usable ({os.cpu_count()} visible) host cores",
                    "value_is": "assemble + solve (the whole tick)",
                    "solve_only": one["solve_only"],          # ticks/s counting only the time inside ql0001_ (qp-problem.cpp:275-279)
                    "all_cores": allc}

    # rehearsal knobs (tools/rehearse_ranks.sh): several ranks on ONE card with the gloo backend exercise the multi-rank
    # bookkeeping where no multi-GPU node is at hand; the driver's runs use neither
    backend = os.environ.get("WG_BENCH_BACKEND", "nccl")
    one_card = os.environ.get("WG_BENCH_DEVICE")
    if one_card is not None:
        os.environ["LOCAL_RANK"] = one_card
    rank, local_rank, world = shard.init_process_group(backend)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    wg.init(local_rank)

    # ---- the one collective: constant model block from rank 0 (RCCL broadcast over xGMI) ----
    model = wg.model_defaults() if rank == 0 else wg.Model()
    shard.broadcast_struct(model, dev, src=0)
    wg.mpc_configure(model)

    B = args.batch
    lo, hi = rank * B, (rank + 1) * B                         # weak scaling: B gaits per GPU
    K, W0 = args.steps, args.warmup
    W = W0 + max(0, args.preroll)                             # pre-roll + warm-up: all untimed, same launch plan
    n_seg = (K + W + min(K, 100) + K + 2 * REDRAW_TICKS - 1) // REDRAW_TICKS    # + one period: the outs leg starts at the timed region's phase
    vtab = torch.from_numpy(velocity_table(lo, hi, n_seg)).to(dev)
    states = start_states(model, B).to(dev)
    diag = torch.zeros(K + W, B, 6, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    sp, dp = states.data_ptr(), diag.data_ptr()
    dstride = B * 6 * 4

    def launches(t0, t1):
        return launch_plan(t0, t1, args.per_tick_launch, staged=not args.no_staged_refs)

    def is_staged(t, n):
        return n > 1 and t % REDRAW_TICKS == 0 and not args.no_staged_refs

    def redraw(t, n=1):
        if t % REDRAW_TICKS == 0 and not is_staged(t, n):
            wg.mpc_set_velref_dev(B, sp, vtab[t // REDRAW_TICKS].data_ptr(), sh)

    entry_points = {}                                         # entry point -> launches in the timed region
    # the tick's deliverable (the 20 samples per tick the reference pushes on its four deques): K x B structs, tick-major, when a
    # leg asks for it.  Throughput figures without it advance the closed loop only (outs = NULL: SURVEY 8d's metric)
    out_bytes = C.sizeof(wg.TickOut)
    outs_buf = None
    if args.outs_on or not (args.no_outs_leg or args.per_tick_launch):
        outs_buf = torch.empty(K * B * out_bytes, dtype=torch.uint8, device=dev)

    def fire(t, n, timed=False, outs_base=None, diag_ptr=True):
        """outs_base = (first tick of the window that stores outs): tick t's structs go to slot t - outs_base of outs_buf"""
        adv = 1 if t == 0 else (19 if t == 1 else 20)
        op = None if outs_base is None else outs_buf.data_ptr() + (t - outs_base) * B * out_bytes
        dq = dp + t * dstride if diag_ptr else None
        if n == 1 and (t < 2 or args.per_tick_launch):
            name = "wg_mpc_tick_batch_dev"
            wg.mpc_tick_batch_dev(B, sp, op, dq, adv, stream=sh)
        elif is_staged(t, n):                                 # the references of every stretch of the launch wait on the device
            name = "wg_mpc_run_sched_dev"
            wg.mpc_run_sched_dev(B, sp, n, vtab[t // REDRAW_TICKS].data_ptr(), REDRAW_TICKS, adv, op, dq, stream=sh)
        else:
            name = "wg_mpc_run_batch_dev"
            wg.mpc_run_batch_dev(B, sp, n, adv, op, dq, stream=sh)
        if timed:
            entry_points[name] = entry_points.get(name, 0) + 1

    with torch.cuda.stream(stream):
        for t, n in launches(0, W):
            redraw(t, n)
            fire(t, n)
    torch.cuda.synchronize(dev)
    timed = launches(W, W + K)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in timed]

    shard.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for k, (t, n) in enumerate(timed):
            redraw(t, n)
            ev[k][0].record(stream)
            fire(t, n, timed=True, outs_base=W if args.outs_on else None)
            ev[k][1].record(stream)
    torch.cuda.synchronize(dev)
    shard.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dev)

    # secondary figure, same process: one launch per tick (the batch drains between ticks) over the next ticks
    alt = None
    if not args.per_tick_launch and not args.no_per_tick_leg:
        K2 = min(K, 100)
        need = W + K + K2
        if need <= vtab.shape[0] * REDRAW_TICKS:
            shard.barrier(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            with torch.cuda.stream(stream):
                for t in range(W + K, need):
                    redraw(t)
                    wg.mpc_tick_batch_dev(B, sp, None, None, 20, stream=sh)
            torch.cuda.synchronize(dev); shard.barrier()
            e2 = shard.max_over_ranks(time.perf_counter() - t1, dev)
            alt = {"value": shard.sum_over_ranks(B * K2, dev) / e2, "steps": K2, "ms_per_step": 1e3 * e2 / K2,
                   "note": "same workload continued with one launch per tick (wg_mpc_tick_batch_dev)"}

    # secondary figure: the same launches with the deliverable stored (outs non-NULL), on the following ticks of the same gaits
    outs_leg = None
    if outs_buf is not None and not args.outs_on:
        t_now = W + K + (min(K, 100) if alt is not None else 0)
        # same position relative to the reference redraws as the timed region (so that the leg is the same launch plan: a window
        # that straddles a redraw is two launches, and every launch pays its ramp and tail once): the ticks in between run untimed
        t_first = t_now + (W - t_now) % REDRAW_TICKS
        if (t_first + K + REDRAW_TICKS - 1) // REDRAW_TICKS > vtab.shape[0]:
            t_first = t_now
        with torch.cuda.stream(stream):
            for t, n in launches(t_now, t_first):
                redraw(t, n)
                fire(t, n, diag_ptr=False)
        plan = launches(t_first, t_first + K)
        shard.barrier(); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        with torch.cuda.stream(stream):
            for t, n in plan:
                redraw(t, n)
                fire(t, n, outs_base=t_first, diag_ptr=False)
        torch.cuda.synchronize(dev); shard.barrier()
        e3 = shard.max_over_ranks(time.perf_counter() - t1, dev)
        outs_leg = {"value": shard.sum_over_ranks(B * K, dev) / e3, "unit": "ticks/s", "steps": K, "ms_per_step": 1e3 * e3 / K,
                    "launches": len(plan), "out_bytes_per_gait_tick": out_bytes,
                    "stored_gb": B * K * out_bytes / 1e9, "store_rate_gbs": B * K * out_bytes / e3 / 1e9,
                    "first_tick": t_first,
                    "note": "the same launch plan on %d later ticks (same position relative to the reference redraws) with outs non-NULL: every gait-tick stores its wg_tick_out_t (20 "
                            "CoM / ZMP / feet / trunk samples, what ZMPVelocityReferencedQP::OnLine pushes on its deques, "
                            "ZMPVelocityReferencedQP.cpp:405-442), tick-major, K x B structs" % K}

    durs = [a.elapsed_time(b) for a, b in ev]
    kern_ms_total = float(np.sum(durs))
    kern_ms = float(np.mean(durs))
    ticks_per_launch = max(n for _, n in timed)
    d = diag[W:].cpu().numpy().reshape(-1, 6)
    n_fail = int((d[:, 0] != 0).sum())
    alg_bytes_total = float(algorithmic_bytes(d[:, 3].astype(np.float64), d[:, 4].astype(np.float64)).sum())
    alg_bytes_per_launch = alg_bytes_total / len(timed)
    ticks_total = shard.sum_over_ranks(B * K, dev)
    value = ticks_total / elapsed

    if rank == 0:
        achieved = alg_bytes_total / (kern_ms_total * 1e-3) / 1e9          # = bytes per launch / average launch duration
        # HBM traffic and issue-slot counters come from rocprofv3 --pmc passes of this same command (they cannot be read
        # from inside the process): measured per gait-tick there, scaled here to THIS run's launch so that `traffic` and
        # `algorithmic_bytes_per_launch` describe the same launch
        traffic = None
        pmc = None
        if os.path.exists(PMC_SUMMARY):
            try:
                pmc = json.load(open(PMC_SUMMARY))
                kname = "per_tick_kernel" if args.per_tick_launch else "run_kernel"
                traffic = pmc[kname]["hbm_bytes_per_gait_tick"] * (B * K / len(timed))
            except Exception:                                              # noqa: BLE001 -- a missing profile is not fatal
                traffic, pmc = None, None
        iters = d[:, 1].astype(np.float64)
        # flops per tick = (2/3) n^3 + 2 m n + 4 n^2 + it (2 m n + 10 n^2), here with each tick's own n, m, it
        nn, mm = d[:, 3].astype(np.float64), d[:, 4].astype(np.float64)
        flops_total = float(((2.0 / 3.0) * nn ** 3 + 2 * mm * nn + 4 * nn ** 2 + iters * (2 * mm * nn + 10 * nn ** 2)).sum())
        useful_tflops = flops_total / (kern_ms_total * 1e-3) / 1e12
        line = {
            "metric": "QP-MPC ticks/sec (batch=4096, N=16)",
            "value": value, "unit": "ticks/s", "n_gpus": world, "steps": K, "warmup": W0,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Herdt2010 N=16 fp64, batch=4096 independent gaits per GPU "
                                   "(fused tick: preview + QP assembly + QL solve + LIPM + feet)",
                       "batch_per_gpu": B, "horizon_N": int(model.N), "qp_T": model.T,
                       "velocity_refs": "U[-0.1,0.3] x U[-0.1,0.1] x U[-0.2,0.2], redrawn every 50 ticks, "
                                        "MT19937-64 seed 20100+gait",
                       "sharding": "gaits by contiguous index range, one RCCL broadcast of the model block",
                       "preroll_ticks": W - W0,
                       "launch": "%s; %d launch(es) in the timed region, up to %d ticks each (multi-tick launches run the device-side "
                                 "work queue; wg_mpc_run_sched_dev is the same kernel with the velocity references of later "
                                 "stretches staged on the device)"
                                 % (", ".join("%s x %d" % kv for kv in sorted(entry_points.items())), len(timed), ticks_per_launch),
                       "timed_ticks": [W, W + K]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "wg_mpc_tick_kernel" if args.per_tick_launch else "wg_mpc_run_xcd_kernel",
                         "kernel_ms": kern_ms, "launches": len(timed), "ticks_per_launch": ticks_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "traffic_is_of_this_kernel": None if pmc is None else pmc_file_is_current(pmc),
                         "traffic_source": None if pmc is None else
                         "%s: %.0f B per gait-tick measured by rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over launches of %s ticks, "
                         "scaled to this run's %.1f ticks per launch" % (os.path.relpath(PMC_SUMMARY, ROOT),
                                                                        pmc[kname]["hbm_bytes_per_gait_tick"],
                                                                        pmc[kname]["ticks_per_launch"], K / len(timed)),
                         # second axis: HBM is the formal roof but not what binds -- the vector ALUs are
                         "second_axis": {"useful_fp64_tflops": useful_tflops, "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                         "useful_flop_frac": useful_tflops / FP64_VECTOR_PEAK_TFLOPS,
                                         "flops_per_tick": flops_total / len(d),
                                         "valu_busy": None if pmc is None else pmc[kname].get("valu_busy"),
                                         "valu_insts_per_tick": None if pmc is None else pmc[kname].get("valu_insts_per_gait_tick"),
                                         "note": "useful flops = SURVEY 8(d)'s operation count of the reference algorithm for each "
                                                 "tick's own n, m and iteration count; valu_busy = 2 waves per SIMD x "
                                                 "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES from the same rocprofv3 passes (share of the "
                                                 "vector ALUs' issue time in use)"}},
            "solver": {"mean_iterations": float(d[:, 1].mean()), "max_iterations": int(d[:, 1].max()),
                       "mean_active": float(d[:, 2].mean()), "failed_qps": n_fail,
                       "n_hist": {str(int(k)): int(v) for k, v in zip(*np.unique(d[:, 3], return_counts=True))},
                       "iteration_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique(d[:, 1], return_counts=True))},
                       "active_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique(d[:, 2], return_counts=True))}},
        }
        if alt is not None:
            line["per_tick_launch"] = alt
        if args.outs_on:
            line["config"]["outs"] = "stored: every gait-tick writes its wg_tick_out_t (%d B)" % out_bytes
        else:
            line["config"]["outs"] = "NULL in the timed region (closed loop advanced only, SURVEY 8d); `outs_on` has the rate with them stored"
        if outs_leg is not None:
            outs_leg["delta_vs_value"] = outs_leg["value"] / value - 1.0
            if pmc is not None and "run_kernel_outs" in pmc:
                outs_leg["hbm_write_bytes_per_gait_tick_measured"] = pmc["run_kernel_outs"].get("hbm_write_bytes_per_gait_tick")
                outs_leg["hbm_bytes_per_gait_tick_measured"] = pmc["run_kernel_outs"].get("hbm_bytes_per_gait_tick")
                outs_leg["traffic_source"] = "%s: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of `bench.py --outs-on`" % os.path.relpath(PMC_SUMMARY, ROOT)
            line["outs_on"] = outs_leg
        if world == 1 and not args.no_kernels:
            line["kernels"] = bench_kernels.run_all(wg, dev, stream, B, states, model, algorithmic_bytes)
            wg.mpc_configure(model)
        if world == 1 and not args.no_config5:
            try:
                line["config5"] = {"default": config5_leg(dev, 0), "gramian_mfma_f32": config5_leg(dev, 4),
                                   "note": "BASELINE configs[4]: Herdt2010 N = 32 with foot-placement variables, batch = 8192; "
                                           "fp64 solve in both (DESIGN 7), the fp32 part is the matrix-core Gramian as Hessian source"}
            except Exception as e:                                         # noqa: BLE001 -- the main figure stands without it
                line["config5"] = {"error": str(e)}
            wg.mpc_configure(model)
        if world == 1 and not args.no_parity:
            line["parity"] = golden_parity(oracle_rows)
            if cpu_line is not None and "b1_tick_latency_us" in line["parity"]:
                line["parity"]["b1_tick_latency_us"]["cpu_reference_us_per_tick_one_core"] = 1e6 / cpu_line["value"]
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
