"""Whole-batch parity soak (evidence, not a test): EVERY gait of the benchmark workload, advanced on the GPU exactly as bench.py
does it (two single ticks, then multi-tick launches with the velocity references staged on the device), against the CPU checker
(oracle/ with the portable trigonometry, the bit-exact partner of the kernels) run on the host cores -- final gait states compared
byte for byte.  N = 16: 4096 gaits x 250 ticks; N = 32: 8192 gaits x 50 ticks.   python tools/soak_parity.py > profiles/<tag>_soak_parity.txt
The checker runs beside the product path here, as in tests/: nothing of it is measured or shipped."""
import ctypes as C
import importlib
import importlib.util
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wg = importlib.import_module("jrl-walkgen_amd")
spec = importlib.util.spec_from_file_location("wg_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); sys.modules["wg_bench"] = bench; spec.loader.exec_module(bench)
import oraclelib as ol  # noqa: E402

# SOAK_VSCALE=k: the benchmark's velocity references times k (both sides; the spawned checkers inherit the variable): at k = 3 .. 6
# the QPs of many ticks are infeasible or inconsistent -- the failure paths of the tick, not only its walking
VSCALE = float(os.environ.get("SOAK_VSCALE", "1"))
if VSCALE != 1.0:
    _vt = bench.velocity_table
    bench.velocity_table = lambda lo, hi, n_seg: _vt(lo, hi, n_seg) * VSCALE


def same_state(a, b):
    """0: bytes equal, or different only in words that are NaNs on both sides (a NaN's sign / payload bits differ between x86 and
    gfx950); 2: a real difference.  (Until the tick's views followed the reference through non-finite iterates -- DESIGN 3.3 --
    there was a class 1 here, "lost on both sides", for states beyond 1e100; every gait is compared to the end now.)"""
    if a == b:
        return 0
    wa, wb = np.frombuffer(a, dtype=np.uint64), np.frombuffer(b, dtype=np.uint64)
    expo = lambda w: (w >> np.uint64(52)) & np.uint64(0x7ff)                        # noqa: E731
    d = wa != wb
    if bool(((expo(wa[d]) == 0x7ff) & (expo(wb[d]) == 0x7ff)).all()):
        return 0
    return 2


def cpu_chunk(args):
    N, g0, ng, n_ticks = args
    ol.build_oracle()
    lib = C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))
    model = wg.Model(); lib.wgo_model_defaults(C.byref(model)); model.N = N
    tab = np.ascontiguousarray(bench.velocity_table(g0, g0 + ng, (n_ticks + bench.REDRAW_TICKS - 1) // bench.REDRAW_TICKS))
    states = (wg.GaitState * ng)()
    s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
    for g in range(ng):
        C.memmove(C.byref(states[g]), C.byref(s0), C.sizeof(wg.GaitState))
    rc = lib.wgo_mpc_run(C.byref(model), states, ng, n_ticks, tab.ctypes.data_as(C.c_void_p), bench.REDRAW_TICKS)
    assert rc == 0
    return g0, bytes(memoryview(states).cast("B"))


def gpu_run(N, B, n_ticks):
    dev = torch.device("cuda:0")
    model = wg.model_defaults(); model.N = N
    with wg.Context(0) as ctx:
        ctx.mpc_configure(model)
        vtab = torch.from_numpy(bench.velocity_table(0, B, (n_ticks + bench.REDRAW_TICKS - 1) // bench.REDRAW_TICKS)).to(dev)
        states = bench.start_states(model, B).to(dev)
        diag = torch.zeros(n_ticks, B, 6, dtype=torch.int32, device=dev)
        sp, dp, ds = states.data_ptr(), diag.data_ptr(), B * 6 * 4
        t0 = time.perf_counter()
        for t, n in bench.launch_plan(0, n_ticks):
            adv = 1 if t == 0 else (19 if t == 1 else 20)
            staged = n > 1 and t % bench.REDRAW_TICKS == 0
            if t % bench.REDRAW_TICKS == 0 and not staged:
                ctx.mpc_set_velref_dev(B, sp, vtab[t // bench.REDRAW_TICKS].data_ptr())
            if n == 1:
                ctx.mpc_tick_batch_dev(B, sp, None, dp + t * ds, adv)
            elif staged:
                ctx.mpc_run_sched_dev(B, sp, n, vtab[t // bench.REDRAW_TICKS].data_ptr(), bench.REDRAW_TICKS, adv, None, dp + t * ds)
            else:
                ctx.mpc_run_batch_dev(B, sp, n, adv, None, dp + t * ds)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        d = diag.cpu().numpy()
        return states.cpu().numpy().tobytes(), dt, d


def soak(N, B, n_ticks, workers):
    per = max(1, (B + 6 * workers - 1) // (6 * workers))            # six chunks per worker: the heartbeat below has something to count
    jobs = [(N, g0, min(per, B - g0), n_ticks) for g0 in range(0, B, per)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
        cpu = {}
        t_said = t0
        for k, v in pool.imap_unordered(cpu_chunk, jobs):
            cpu[k] = v
            if time.perf_counter() - t_said > 60:                    # lost gaits run maxit iterations per tick: say that it is alive
                t_said = time.perf_counter()
                print("   ... CPU checker: %d of %d chunks" % (len(cpu), len(jobs)), flush=True)
    t_cpu = time.perf_counter() - t0
    got, t_gpu, d = gpu_run(N, B, n_ticks)
    sz = C.sizeof(wg.GaitState)
    bad = lost = 0
    which = []
    for g0, blob in cpu.items():
        ng = len(blob) // sz
        for g in range(ng):
            r = same_state(blob[g * sz:(g + 1) * sz], got[(g0 + g) * sz:(g0 + g + 1) * sz])
            if r == 2:
                bad += 1
                which.append(g0 + g)
            w = np.frombuffer(blob[g * sz:(g + 1) * sz], dtype=np.uint64)
            lost += bool((((w >> np.uint64(52)) & np.uint64(0x7ff)) == np.uint64(0x7ff)).any())
    if which:
        print("   differing gaits: %s%s" % (sorted(which)[:24], " ..." if len(which) > 24 else ""), flush=True)
    print("N = %d%s: %d gaits x %d ticks = %d MPC ticks; gaits whose final state differs from the CPU checker's: %d%s; "
          "failed QPs %d; QL iterations mean %.1f max %d; n in %s; GPU %.2f s (launch plan %s), CPU checker %.1f s on %d processes"
          % (N, "" if VSCALE == 1.0 else " (references x %g)" % VSCALE, B, n_ticks, B * n_ticks, bad,
             "" if not lost else " (%d gaits end with NaNs in their state, on both sides)" % lost, int((d[..., 0] != 0).sum()), float(d[..., 1].mean()), int(d[..., 1].max()),
             sorted(set(int(v) for v in np.unique(d[..., 3]))), t_gpu, bench.launch_plan(0, n_ticks), t_cpu, min(workers, len(jobs))), flush=True)
    return bad


if __name__ == "__main__":
    wg.init(0)
    workers = min(os.cpu_count() or 8, 64)
    # SOAK_LONG=1: four times the ticks at N = 16 and other horizons through the element view as well
    long_run = os.environ.get("SOAK_LONG") == "1"
    if os.environ.get("SOAK_ONLY"):                              # "N:B:T[,N:B:T...]": these runs instead of the standard ones
        bad = sum(soak(*(int(v) for v in spec.split(":")), workers) for spec in os.environ["SOAK_ONLY"].split(","))
        print("soak parity: %s" % ("PASS (bit-identical)" if bad == 0 else "FAIL"))
        sys.exit(1 if bad else 0)
    bad = soak(16, 4096, 1000 if long_run else 250, workers)
    bad += soak(32, 8192, 50, workers)
    if long_run:
        for N, B, T in ((20, 2048, 100), (24, 2048, 80), (28, 2048, 60)):
            bad += soak(N, B, T, workers)
    print("soak parity: %s" % ("PASS (bit-identical)" if bad == 0 else "FAIL"))
    sys.exit(1 if bad else 0)
