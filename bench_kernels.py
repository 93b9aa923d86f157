"""bench.py's `kernels` object: every kernel BASELINE's north_star names besides the fused Herdt tick, each on its own unit,
driver-timed (HIP events on the stream the kernel is launched on), priced against the roof that bounds it.

    pldp            PLDPSolver::SolveProblem + OptCholesky, hot-started          /root/reference/src/Mathematics/PLDPSolver.cpp:654-1007
    dimitrov_tick   the Dimitrov-2008 receding-horizon tick around it             ZMPConstrainedQPFastFormulation.cpp:1180-1400
    ql0001_dense    the ql0001_ boundary on the REAL QPs of the Herdt workload    qld.cpp:378-612, called at qp-problem.cpp:275-279
    preview         PreviewControl::OneIterationOfPreview                        PreviewControl.cpp:324-374
    zmpdisc         ZMPDiscretization::GetZMPDiscretization                      ZMPDiscretization.cpp:143-173, 319-513
    gramian         build_invariant_part on the matrix cores                     generator-vel-ref.cpp:587-614

Inputs are synthetic and built by the product's own host entry points (wg_dimitrov_get_constants, wg_mpc_assemble_batch_dev,
wg_riccati_gains) plus pure-numpy footstep geometry (tests/footplans.py); nothing here touches oracle/.  Algorithmic bytes = what
the reference's routine reads and writes at its own argument list, per unit, stated next to each figure.  All of these kernels
are chain- or latency-bound: `frac` is small by construction and is reported so that the bench line alone reproduces
DESIGN 4's table."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}       # dense MFMA peaks, MI355X_MICROARCH.md


def _ev():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def _roof(bytes_per_unit, units, seconds):
    ach = bytes_per_unit * units / seconds / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "algorithmic_bytes_per_unit": bytes_per_unit}


def _poly_table(wg, n_plans, length, seed0=20100):
    """[n_plans, length] wg_zmp_polytope_t records: one footstep plan per row, one polytope per 0.1 s slot"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import footplans as fp
    PT = np.dtype([("nrows", "i4"), ("pad", "i4"), ("similar", "i4", 8), ("A", "f8", (8, 2)), ("B", "f8", 8), ("centre", "f8", 2)])
    assert PT.itemsize == C.sizeof(wg.ZmpPolytope)
    table = np.zeros((n_plans, length), PT)
    for p in range(n_plans):
        slots = fp.plan(np.random.default_rng(seed0 + p), n_steps=24)
        uniq = {}
        for k in range(length):
            poly = slots[min(k, len(slots) - 1)]
            rec = uniq.get(id(poly))
            if rec is None:
                A, Bv, c, sim = poly
                rec = np.zeros((), PT); r = len(Bv)
                rec["nrows"] = r; rec["similar"][:r] = sim; rec["A"][:r] = A; rec["B"][:r] = Bv; rec["centre"] = c
                uniq[id(poly)] = rec
            table[p, k] = rec
    return table


def dimitrov_and_pldp(wg, dev, stream, B=4096):
    """Both legs share the footstep plans: 256 plans x 23 phase offsets = one distinct receding-horizon gait per wave."""
    sh = stream.cuda_stream
    model = wg.dimitrov_defaults(); wg.dimitrov_configure(model); N = int(model.N); n = 2 * N
    k = wg.dimitrov_constants(N)
    wg.pldp_configure(N, k["iPu"], k["Px"], k["Pu"])
    MC = wg.PLDP_MMAX
    NPLAN = 256
    table = _poly_table(wg, NPLAN, 80)
    plan_id = np.arange(B) % NPLAN
    offs = (7 * np.arange(B)) % 23
    lib = wg.lib()
    res = {}

    # ---- the fused Dimitrov tick: polytope windows in, jerk + state out; with PLDP, then with ql0001_ (mode QLDANDLQ) ----
    ST = np.dtype([("xk", "f8", 6), ("pldp", "u1", C.sizeof(wg.PldpState)), ("n_removed", "i4"), ("starting", "i4")])
    assert ST.itemsize == C.sizeof(wg.DimitrovState)
    OUT = C.sizeof(wg.DimitrovOut)
    odt = np.dtype([("jerk", "f8", 2), ("ret", "i4"), ("n_iter", "i4"), ("n_active", "i4"), ("m", "i4"), ("rest", "u1", OUT - 32)])

    def tick_leg(solver, kernel, iter_key):
        mdl = wg.dimitrov_defaults(); mdl.solver = solver; wg.dimitrov_configure(mdl)
        offs_l = offs.copy()
        st = np.zeros(B, ST); st["starting"] = 1
        dst = torch.from_numpy(st.view(np.uint8)).to(dev)
        dout = torch.zeros(B * OUT, dtype=torch.uint8, device=dev)
        WARM, MEAS = 3, 20
        ms, iters, failed = [], [], 0
        for it in range(WARM + MEAS):
            win = table[plan_id[:, None], (it + offs_l)[:, None] + np.arange(N)[None, :]]
            dpoly = torch.from_numpy(np.ascontiguousarray(win).view(np.uint8)).to(dev)
            torch.cuda.synchronize(dev)
            e0, e1 = _ev()
            with torch.cuda.stream(stream):
                e0.record(stream)
                rc = lib.wg_dimitrov_tick_batch_dev(B, dpoly.data_ptr(), dst.data_ptr(), dout.data_ptr(), 0, sh)
                e1.record(stream)
            assert rc == 0, wg.lib().wg_last_error()
            torch.cuda.synchronize(dev)
            o = np.frombuffer(dout.cpu().numpy().tobytes(), dtype=odt)
            if it >= WARM:
                ms.append(e0.elapsed_time(e1)); iters.append(float(o["n_iter"].mean())); failed += int((o["ret"] != 0).sum())
            bad = o["ret"] != 0
            if bad.any():                              # the reference would have stopped there: those gaits restart from rest
                h = np.frombuffer(dst.cpu().numpy().tobytes(), dtype=ST).copy()
                h["xk"][bad] = 0.0; h["starting"][bad] = 1; h["n_removed"][bad] = 0; h["pldp"][bad] = 0
                offs_l = offs_l.copy(); offs_l[bad] = -it - 1 + (offs_l[bad] % 5)
                dst = torch.from_numpy(h.view(np.uint8)).to(dev)
        sec = float(np.sum(ms)) * 1e-3
        per_tick = N * C.sizeof(wg.ZmpPolytope) + 2 * C.sizeof(wg.DimitrovState) + OUT      # windows in, state in + out, outputs
        return {"value": B * MEAS / sec, "unit": "ticks/s", "batch": B, "launches": MEAS, "kernel": kernel,
                "kernel_ms": float(np.mean(ms)), iter_key: float(np.mean(iters)),
                "solves_ended_by_the_references_exit_condition": failed,
                "roofline": _roof(per_tick, B * MEAS, sec),
                "bytes_are": "N polytopes (248 B each) + state in and out + outputs per gait-tick"}

    res["dimitrov_tick"] = tick_leg(0, "wg_dimitrov_tick_kernel", "mean_pldp_iterations")
    try:
        res["dimitrov_tick_qldandlq"] = tick_leg(2, "wg_dimitrov_qld_tick_kernel<true>", "mean_ql_iterations")
        res["dimitrov_tick_qldandlq"]["solver"] = ("ql0001_ on the preconditioned problem (m_FastFormulationMode == QLDANDLQ): the in-wave "
                                                   "ql0002, bit-exact against the reference's compiled qld.cpp")
    finally:
        wg.dimitrov_configure(model)

    # ---- PLDP alone, hot-started, on the dense arguments PLDPSolver::SolveProblem takes ----
    Pu, Px, OptB, OptC, iLQ = k["Pu"], k["Px"], k["OptB"], k["OptC"], k["iLQ"]
    T = float(model.T)
    A3 = np.array([[1, T, T * T / 2], [0, 1, T], [0, 0, 1.0]]); B3 = np.array([T ** 3 / 6, T * T / 2, T])
    offs = (7 * np.arange(B)) % 23
    xk = np.zeros((B, 6))
    pst = torch.zeros(B * C.sizeof(wg.PldpState), dtype=torch.uint8, device=dev)
    X = torch.zeros(B, n, dtype=torch.float64, device=dev); ret = torch.zeros(B, dtype=torch.int32, device=dev)
    nit = torch.zeros(B, dtype=torch.int32, device=dev); nact = torch.zeros(B, dtype=torch.int32, device=dev)
    n_removed = np.zeros(B, np.int32); starting = np.ones(B, np.int32)
    karr = np.arange(N)
    WARM, MEAS = 6, 8                                        # hot-start state settles over the first ticks of a plan (tools/probe_pldp.py: 30)
    ms, iters, mrows, bytes_tot = [], [], [], 0.0
    for it in range(WARM + MEAS):
        win = table[plan_id[:, None], (it + offs)[:, None] + np.arange(N)[None, :]]              # B x N polytopes
        valid = np.arange(8)[None, None, :] < win["nrows"][:, :, None]                          # B x N x 8
        flat = valid.reshape(B, -1)
        idx = np.cumsum(flat, axis=1) - 1
        m = flat.sum(axis=1).astype(np.int32)
        gi, fi = np.nonzero(flat); ii = fi // 8; ei = fi % 8
        row = idx[gi, fi]
        Ax = win["A"][gi, ii, ei, 0]; Ay = win["A"][gi, ii, ei, 1]; Bv = win["B"][gi, ii, ei]
        zx = xk[:, :3] @ Px.T; zy = xk[:, 3:] @ Px.T                                            # B x N
        b = np.zeros((B, MC)); sim = np.zeros((B, MC), np.int32); A = np.zeros((B, (MC + 1) * n))
        b[gi, row] = zx[gi, ii] * Ax + zy[gi, ii] * Ay + Bv
        sim[gi, row] = win["similar"][gi, ii, ei]
        ld = (m[gi] + 1)[:, None]
        col = row[:, None] + karr[None, :] * ld                                                 # entries x N
        PuS = Pu[:, ii].T                                                                       # entries x N : Pu[k, instant]
        lin = gi[:, None] * ((MC + 1) * n) + col                                                # flat scatter: 4 x faster than 2-D fancy
        Af = A.reshape(-1)
        Af[lin.ravel()] = (Ax[:, None] * PuS).ravel()
        Af[(lin + N * ld).ravel()] = (Ay[:, None] * PuS).ravel()
        z = np.concatenate([win["centre"][:, :, 0], win["centre"][:, :, 1]], axis=1)           # B x 2N
        D = xk @ OptB.T - z @ OptC.T
        first = win["nrows"][:, 0].astype(np.int32)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)                         # noqa: E731
        dm_, dD, dA, db, dz, dx, dsim, dnr, dstart = t(m), t(D), t(A), t(b), t(z), t(xk), t(sim), t(n_removed), t(starting)
        torch.cuda.synchronize(dev)
        e0, e1 = _ev()
        with torch.cuda.stream(stream):
            e0.record(stream)
            rc = lib.wg_pldp_solve_batch_dev(B, MC, dm_.data_ptr(), dD.data_ptr(), dA.data_ptr(), db.data_ptr(), dz.data_ptr(),
                                             dx.data_ptr(), dsim.data_ptr(), dnr.data_ptr(), dstart.data_ptr(), 0, pst.data_ptr(),
                                             X.data_ptr(), ret.data_ptr(), nit.data_ptr(), None, nact.data_ptr(), sh)
            e1.record(stream)
        assert rc == 0, wg.lib().wg_last_error()
        torch.cuda.synchronize(dev)
        r = ret.cpu().numpy(); Xh = X.cpu().numpy()
        if it >= WARM:
            ms.append(e0.elapsed_time(e1)); iters.append(float(nit.double().mean().item())); mrows.append(float(m.mean()))
            # PLDPSolver::SolveProblem's arguments (PLDPSolver.cpp:654-662): D, A ((m+1) x 2N), b, ZMPRef, XkYk, SimilarConstraints in;
            # the solution out
            bytes_tot += float((8.0 * (n + (m + 1.0) * n + m + n + 6 + n) + 4.0 * m).sum())
        ok = r == 0
        u = Xh @ iLQ                                                   # un-preconditioning: NewX = iLQ' X  (row vectors)
        nx = xk[:, :3] @ A3.T + u[:, 0:1] * B3[None, :]; ny = xk[:, 3:] @ A3.T + u[:, N:N + 1] * B3[None, :]
        xk[ok] = np.concatenate([nx, ny], axis=1)[ok]
        dead = ~ok
        xk[dead] = 0.0; offs[dead] = -it - 1
        n_removed = first.copy(); starting[:] = 0; starting[dead] = 1
    sec = float(np.sum(ms)) * 1e-3
    res["pldp"] = {"value": B * MEAS / sec, "unit": "solves/s", "batch": B, "launches": MEAS, "kernel": "wg_pldp_kernel",
                   "kernel_ms": float(np.mean(ms)), "mean_iterations": float(np.mean(iters)), "mean_rows": float(np.mean(mrows)),
                   "hot_started": True,
                   "roofline": _roof(bytes_tot / (B * MEAS), B * MEAS, sec),
                   "bytes_are": "PLDPSolver::SolveProblem's arguments: D, A ((m+1) x 2N), b, ZMPRef, XkYk, SimilarConstraints in, X out"}
    return res


def ql_dense_on_real_qps(wg, dev, stream, B, states, N, algorithmic_bytes):
    """The ql0001_ boundary on the QPs the Herdt workload really poses, as an MPC loop poses them: the QPs of the benchmark's own
    gaits at their next tick are assembled on the device (wg_mpc_assemble_batch_dev) and solved by wg_qp_solve_batch_dev (timed),
    then the gaits advance one tick (on a COPY of the states: the caller's stay what they are) and the next tick's QPs follow --
    problem k of consecutive batches is the same robot a tick later, which is what the library's longest-solve-first start order
    predicts from (the previous batch's iteration counts; the first timed batch follows an untimed one of the tick before).
    `states`: the uint8 tensor of the gait states (or its device pointer: then the same QPs are solved again and again, the
    form of rounds 1 - 4, whose start order is predicted perfectly)."""
    sh = stream.cuda_stream
    nmax, mmax = 2 * N + 4, 1 + 4 * N + 10 + 1
    f = lambda *s: torch.zeros(*s, dtype=torch.float64, device=dev)                             # noqa: E731
    Cq, dq, Aq, bq, xl, xu = f(B, nmax * nmax), f(B, nmax), f(B, mmax * nmax), f(B, mmax), f(B, nmax), f(B, nmax)
    nn = torch.zeros(B, dtype=torch.int32, device=dev); mm = torch.zeros(B, dtype=torch.int32, device=dev)
    walk = torch.is_tensor(states)
    st = states.clone() if walk else None
    sp = st.data_ptr() if walk else states
    x = f(B, nmax); u = f(B, mmax + 2 * nmax)
    ifail = torch.zeros(B, dtype=torch.int32, device=dev); nit = torch.zeros(B, dtype=torch.int32, device=dev)

    def assemble():
        rc = wg.lib().wg_mpc_assemble_batch_dev(B, sp, 20, nmax, mmax, Cq.data_ptr(), dq.data_ptr(), Aq.data_ptr(), bq.data_ptr(),
                                                xl.data_ptr(), xu.data_ptr(), nn.data_ptr(), mm.data_ptr(), sh)
        assert rc == 0, wg.lib().wg_last_error()
    run = lambda: wg.qp_solve_batch_dev(B, nmax, mmax, nn, mm, None, Cq, dq, Aq, bq, xl, xu, 1e-8, x, u, ifail, nit, stream=sh)  # noqa: E731
    REP = 3
    evs, alg, iters, fails, nh_all = [], 0.0, [], 0, []
    with torch.cuda.stream(stream):
        for rep in range(REP + 1):                             # rep 0: untimed (warm, and the tick before the first timed batch)
            assemble()
            e0, e1 = _ev(); e0.record(stream); run(); e1.record(stream)
            if rep:
                evs.append((e0, e1))
                torch.cuda.synchronize(dev)
                nh, mh = nn.cpu().numpy().astype(np.float64), mm.cpu().numpy().astype(np.float64)
                alg += float(algorithmic_bytes(nh, mh - 1).sum())   # m_ counts the dummy row; the formula's m does not
                iters.append(float(nit.double().mean().item())); fails += int((ifail != 0).sum().item()); nh_all.append(nh)
            if walk:
                wg.mpc_tick_batch_dev(B, sp, None, None, 20, stream=sh)
    torch.cuda.synchronize(dev)
    ms = [a.elapsed_time(b) for a, b in evs]
    sec = float(np.sum(ms)) * 1e-3
    nh = np.concatenate(nh_all)
    return {"value": B * REP / sec, "unit": "QPs/s", "batch": B, "launches": REP, "kernel": "wg_ql_dense_kernel",
            "kernel_ms": float(np.mean(ms)), "mean_iterations": float(np.mean(iters)),
            "failed_qps": fails,
            "n_hist": {str(int(a)): int(c) for a, c in zip(*np.unique(nh, return_counts=True))},
            "qps_are": ("the benchmark's own gaits at %d consecutive ticks, each tick's QPs assembled by wg_mpc_assemble_batch_dev "
                        "(feasible, de-synchronised); start order of a batch from the previous tick's iteration counts" % REP) if walk else
                       "the benchmark's own gaits at their next tick, assembled by wg_mpc_assemble_batch_dev; the same batch solved %d times" % REP,
            "roofline": _roof(alg / (B * REP), B * REP, sec),
            "bytes_are": "ql0001_'s arguments for each QP's own n, m (SURVEY 8d)"}


def preview_and_zmpdisc(wg, dev, stream, B=4096, S=16):
    """Step sequences -> ZMP queues (wg_zmpdisc_batch_dev) -> stage-1 CoM (wg_preview_run_batch_dev), resident."""
    sh = stream.cuda_stream
    m = wg.zmpdisc_defaults()
    g, F = wg.preview_gains(0.005, 0.814, 1.6)
    wg.preview_configure(g, F)
    rng = np.random.default_rng(1)
    tr = np.zeros((B, S, 3))
    side = rng.choice([-1.0, 1.0], B)
    for i in range(S):
        tr[:, i, 0] = 0.0 if i == 0 else rng.uniform(0.1, 0.25, B)
        tr[:, i, 1] = side * (0.105 if i == 0 else 0.21)
        tr[:, i, 2] = 0.0 if i == 0 else rng.uniform(-5, 5, B)
        side = -side
    rec = np.zeros((B, S), dtype=[("sx", "f8"), ("sy", "f8"), ("theta", "f8"), ("ss", "f8"), ("ds", "f8"), ("ty", "i4"), ("pad", "i4")])
    rec["sx"], rec["sy"], rec["theta"] = tr[:, :, 0], tr[:, :, 1], tr[:, :, 2]
    rec["ss"], rec["ds"], rec["ty"] = m.t_single, m.t_double, 1
    steps = (wg.RelStep * S).from_buffer_copy(rec[0].tobytes())
    L = wg.zmpdisc_length(m, steps)
    d_steps = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev)
    d_ns = torch.full((B,), S, dtype=torch.int32, device=dev)
    d_init = torch.from_numpy(np.tile(np.array([0.0, 0.095, 0.0, 0.0, -0.095, 0.0]), (B, 1))).to(dev)
    zx = torch.zeros(L, B, dtype=torch.float64, device=dev); zy = torch.zeros_like(zx)
    d_len = torch.zeros(B, dtype=torch.int32, device=dev)
    Lrun = L - g.nl + 1
    st = torch.zeros(B, 8, dtype=torch.float64, device=dev)
    com = torch.zeros(Lrun, 6, B, dtype=torch.float64, device=dev)
    tz, tp = [], []
    for rep in range(3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        st.zero_()
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(stream):
            e0.record(stream)
            wg.zmpdisc_batch_dev(m, B, S, d_steps.data_ptr(), d_ns.data_ptr(), d_init.data_ptr(), L, zx.data_ptr(), zy.data_ptr(),
                                 d_len.data_ptr(), sh)
            e1.record(stream)
            wg.preview_run_batch_dev(B, Lrun, zx.data_ptr(), zy.data_ptr(), st.data_ptr(), com.data_ptr(), None, stream=sh)
            e2.record(stream)
        torch.cuda.synchronize(dev)
        if rep:
            tz.append(e0.elapsed_time(e1) * 1e-3); tp.append(e1.elapsed_time(e2) * 1e-3)
    assert int(d_len.min()) == L == int(d_len.max())
    sz, sp = float(np.sum(tz)), float(np.sum(tp))
    reps = len(tz)
    return {
        "zmpdisc": {"value": B * L * reps / sz, "unit": "gait-samples/s", "batch": B, "samples_per_gait": L, "steps_per_gait": S,
                    "launches": reps, "kernel": "wg_zmpdisc_kernel", "kernel_ms": 1e3 * sz / reps,
                    "roofline": _roof(16.0 + 48.0 * S / L, B * L * reps, sz),
                    "bytes_are": "16 B of ZMP queue written per gait-sample + 48 B read per step"},
        "preview": {"value": B * Lrun * reps / sp, "unit": "gait-steps/s", "batch": B, "control_steps": Lrun, "window": int(g.nl),
                    "launches": reps, "kernel": "wg_preview_split_kernel", "kernel_ms": 1e3 * sp / reps,
                    "roofline": _roof(80.0, B * Lrun * reps, sp),
                    "bytes_are": "2 new ZMP samples in + 8 doubles of CoM / ZMP out per gait-step"},
        "steps_to_com": {"value": B * reps / (sz + sp), "unit": "gaits/s", "note": "zmpdisc + preview back to back, nothing leaves the device"}}


def gramian(wg, dev, stream, B=65536):
    sh = stream.cuda_stream
    rng = np.random.default_rng(1)
    T = torch.from_numpy(rng.uniform(0.05, 0.2, B)).to(dev); h = torch.from_numpy(rng.uniform(0.6, 0.9, B)).to(dev)
    lib = wg.lib()
    v = lambda t: C.c_void_p(t.data_ptr())                                                     # noqa: E731
    out = {}
    N = 32
    Q = torch.zeros(B, N, N, dtype=torch.float64, device=dev)
    for prec, name in ((0, "f64"), (1, "f32")):
        with torch.cuda.stream(stream):
            assert lib.wg_gramian_batch_dev(B, N, v(T), v(h), 1.0, 1e-5, 1e-6, prec, v(Q), C.c_void_p(sh)) == 0
        torch.cuda.synchronize(dev)
        REP = 5
        e0, e1 = _ev()
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(REP):
                lib.wg_gramian_batch_dev(B, N, v(T), v(h), 1.0, 1e-5, 1e-6, prec, v(Q), C.c_void_p(sh))
            e1.record(stream)
        torch.cuda.synchronize(dev)
        sec = e0.elapsed_time(e1) * 1e-3 / REP
        flop = 2.0 * 2 * N ** 3 * B
        out[name] = {"value": B / sec, "unit": "models/s", "batch": B, "horizon_N": N, "kernel": "wg_gramian_kernel", "kernel_ms": 1e3 * sec,
                     "roofline": {"bound": "mfma", "achieved": flop / sec / 1e12, "peak": MFMA_PEAK_TFLOPS[name], "unit": "TFLOP/s",
                                  "frac": flop / sec / 1e12 / MFMA_PEAK_TFLOPS[name], "flops_per_unit": 2.0 * 2 * N ** 3,
                                  "output_gbs": B * N * N * 8 / sec / 1e9}}
    return out


def run_all(wg, dev, stream, B, states, model, algorithmic_bytes):
    """-> the `kernels` object.  A leg that fails reports its error and leaves the others standing."""
    t0 = time.perf_counter()
    out = {}

    def leg(name, fn):
        try:
            r = fn()
            if name in ("dimitrov_pldp", "preview_zmpdisc"):
                out.update(r)
            else:
                out[name] = r
        except Exception as e:                                              # noqa: BLE001 -- the main figure stands without it
            out[name] = {"error": repr(e)}

    leg("ql0001_dense", lambda: ql_dense_on_real_qps(wg, dev, stream, B, states, int(model.N), algorithmic_bytes))
    leg("dimitrov_pldp", lambda: dimitrov_and_pldp(wg, dev, stream))
    leg("preview_zmpdisc", lambda: preview_and_zmpdisc(wg, dev, stream))
    leg("gramian", lambda: gramian(wg, dev, stream))
    out["wall_seconds"] = time.perf_counter() - t0
    out["note"] = ("each entry: its own unit, HIP-event time on the launch stream, algorithmic bytes (or flops) per unit at the "
                   "reference routine's own argument list; all but the Gramian are dependent-chain bound (DESIGN 3), so `frac` of the "
                   "HBM roof is small by construction")
    return out
