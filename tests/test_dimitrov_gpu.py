"""The fused Dimitrov-2008 tick (ZMP polytopes in, LIPM state out; constraint matrices, cost vector, PLDP solve,
un-preconditioning and LIPM step inside one kernel) against oracle/pldp_oracle.c, through the C ABI:
constants against the numpy model of the reference's InitConstants (tests/dimitrov.py), ticks bit for bit."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dimitrov as dv  # noqa: E402
import oraclelib as ol  # noqa: E402

wg = importlib.import_module("jrl-walkgen_amd")
pytestmark = pytest.mark.gpu


def _setup():
    wg.init(0)
    model = wg.dimitrov_defaults()
    wg.dimitrov_configure(model)
    return model, wg.dimitrov_constants(model.N)


def test_constants_match_the_numpy_model_of_initconstants():
    model, K = _setup()
    dm = dv.Dimitrov(model.N, model.T, model.com_height, model.alpha, model.beta)
    for name, want in (("iLQ", dm.iLQ), ("OptB", dm.OptB), ("OptC", dm.OptC), ("Pu", dm.Pu), ("iPu", dm.iPu), ("Px", dm.Px)):
        np.testing.assert_allclose(K[name], want, rtol=1e-9, atol=1e-12 * np.abs(want).max(), err_msg=name)
    # the LQ factor really preconditions the Hessian the reference factors (lower triangle of OptA)
    L = np.linalg.inv(K["iLQ"][:model.N, :model.N])
    assert np.allclose(np.triu(L, 1), 0.0)


def _fill(poly, p):
    A, B, centre, sim = p
    poly.nrows = len(B)
    for j in range(len(B)):
        poly.A[j][0], poly.A[j][1] = A[j]
        poly.B[j] = B[j]
        poly.similar[j] = int(sim[j])
    poly.centre[0], poly.centre[1] = centre


def test_fused_tick_bit_exact_over_gaits():
    model, K = _setup()
    N = model.N
    M = ol.pldp_setup(N, K["iPu"], K["Px"], K["Pu"])
    lib = ol.oracle()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    B, T = 20, 40
    plans = [dv.plan(np.random.default_rng(500 + g), n_steps=3 + g % 6) for g in range(B)]
    offs = [(5 * g) % 11 for g in range(B)]
    sg = (wg.DimitrovState * B)(); so = (wg.DimitrovState * B)()
    for g in range(B):
        for s in (sg[g], so[g]):
            s.starting = 1
            s.xk[0] = 0.002 * g; s.xk[4] = 0.001 * (g % 5 - 2)
    alive = np.ones(B, bool); n_solves = 0; rets = set(); max_act = 0
    for it in range(T):
        polys = (wg.ZmpPolytope * (B * N))()
        for g in range(B):
            for i, p in enumerate(dv.polys_at(plans[g], it + offs[g], N)):
                _fill(polys[g * N + i], p)
        outs = wg.dimitrov_tick_batch(polys, sg)
        for g in range(B):
            if not alive[g]:
                C.memmove(C.byref(sg[g]), C.byref(so[g]), C.sizeof(wg.DimitrovState))   # frozen: keep both sides equal
                continue
            oo = wg.DimitrovOut()
            rc = lib.wgo_dimitrov_tick(C.byref(M), dp(K["OptB"]), dp(K["OptC"]), dp(K["iLQ"]), C.c_double(model.T),
                                       C.c_double(model.Tctrl), C.c_double(model.com_height),
                                       C.byref(polys, g * N * C.sizeof(wg.ZmpPolytope)), C.byref(so[g]), C.byref(oo),
                                       C.c_int(0))
            assert outs[g].ret == rc, (it, g, outs[g].ret, rc)
            assert bytes(sg[g]) == bytes(so[g]), (it, g)
            if rc == 0:
                assert bytes(outs[g]) == bytes(oo), (it, g)
            else:
                assert (outs[g].jerk_x, outs[g].n_iter, outs[g].n_active) == (oo.jerk_x, oo.n_iter, oo.n_active)
                alive[g] = False
            n_solves += 1; rets.add(rc); max_act = max(max_act, oo.n_active)
    assert n_solves > 400 and 0 in rets and max_act >= 10
    x_end = np.array([so[g].xk[0] for g in range(B)])
    assert x_end.max() > 0.2                                  # the gaits did walk


def test_bad_polytope_and_unconfigured_calls():
    model, K = _setup()
    N = model.N
    polys = (wg.ZmpPolytope * N)()
    for i in range(N):
        _fill(polys[i], dv.box(0.0, 0.0, 0.07, 0.12))
    polys[3].similar[1] = 5
    st = (wg.DimitrovState * 1)(); st[0].starting = 1
    outs = wg.dimitrov_tick_batch(polys, st)
    assert outs[0].ret == -4
    assert wg.lib().wg_dimitrov_tick_batch(1, None, None, None, 0) != 0
    bad = wg.dimitrov_defaults(); bad.N = 40
    assert wg.lib().wg_dimitrov_configure(C.byref(bad)) != 0


def _polytopes_for_tick(polys, ts, te, k, t0, N, T):
    """the queue walk of BuildConstraintMatrices (ZMPConstrainedQPFastFormulation.cpp:783-795, 836-840): the polytope whose
    interval holds t0, then the next one whenever a previewed instant passes the current one's EndingTime"""
    q = 0
    while q < k and not (ts[q] <= t0 <= te[q]):
        q += 1
    assert q < k
    sel = []
    for i in range(N):
        if t0 + i * T > te[q]:
            q += 1
        sel.append(q)
    return sel


def test_step_sequences_to_com_through_zmpdisc_footconstraints_and_the_tick():
    """the Dimitrov-2008 pipeline of BuildZMPTrajectoryFromFootTrajectory (:1096-1463) with real data either side:
    ":stepseq" -> ZMPDiscretization feet (GPU) -> FootConstraintsAsLinearSystem polytopes (library, host) -> one fused tick
    per 0.1 s (GPU), against the oracle tick on the same polytopes, bit for bit, while the CoM follows the footprints.
    PLDP leaves its solutions up to its tolerance (1e-8) outside a constraint (ComputeAlpha, PLDPSolver.cpp:617-618) and
    then refuses the hot start of a later tick for being more than 1e-8 outside (:611-616, its exit(0) path): with real
    footprints most gaits end that way after a few steps -- on the GPU at the same tick with the same state."""
    from test_zmpdisc_oracle import kajita_model
    from test_zmpdisc_gpu import gait_steps, random_fleet
    model, K = _setup()
    N = model.N
    M = ol.pldp_setup(N, K["iPu"], K["Px"], K["Pu"])
    lib = ol.oracle()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    zm = kajita_model()
    zm.t_single, zm.t_double = 0.7, 0.13                 # phase boundaries off the 0.1 s grid of the previewed instants
    B, smax = 6, 9
    steps, n_steps, init = random_fleet(np.random.default_rng(31), B, smax, zm, exotic=False)
    init[:] = [0.0, 0.095, 0.0, 0.0, -0.095, 0.0]       # axis-aligned soles: ComputeLinearSystem's slope form stays tame
    for b in range(B):
        for i in range(smax):
            s = steps[b * smax + i]
            s.theta = 0.0; s.ss_time, s.ds_time = zm.t_single, 0.0
    lens = [wg.zmpdisc_length(zm, gait_steps(steps, b, smax, int(n_steps[b]))) for b in range(B)]
    r = wg.zmpdisc_batch(zm, steps, n_steps, init, smax, max(lens))
    queues = []
    for b in range(B):
        L = lens[b]
        time = np.cumsum(np.full(L, zm.T)) - zm.T
        queues.append(wg.foot_constraints(time, r["left"][b, :L], r["left_type"][b, :L], r["right"][b, :L], 0.24, 0.138,
                                          0.02, 0.02))
        assert [queues[b][0][q].nrows for q in range(queues[b][3])][:4] == [4, 4, 6, 4]
    n_ticks = int(min((q[2][-1] - N * model.T) / model.T for q in queues)) - 1
    assert n_ticks > 40
    sg = (wg.DimitrovState * B)(); so = (wg.DimitrovState * B)()
    for b in range(B):
        sg[b].starting = so[b].starting = 1
    t0 = 0.0
    inside = 0.0
    alive = np.ones(B, bool)
    lived = np.zeros(B, int)
    for it in range(n_ticks):
        polys = (wg.ZmpPolytope * (B * N))()
        for b in range(B):
            pq, ts, te, k = queues[b]
            for i, q in enumerate(_polytopes_for_tick(pq, ts, te, k, t0, N, model.T)):
                C.memmove(C.byref(polys[b * N + i]), C.byref(pq[q]), C.sizeof(wg.ZmpPolytope))
        outs = wg.dimitrov_tick_batch(polys, sg)
        for b in range(B):
            if not alive[b]:
                C.memmove(C.byref(sg[b]), C.byref(so[b]), C.sizeof(wg.DimitrovState))
                continue
            oo = wg.DimitrovOut()
            rc = lib.wgo_dimitrov_tick(C.byref(M), dp(K["OptB"]), dp(K["OptC"]), dp(K["iLQ"]), C.c_double(model.T),
                                       C.c_double(model.Tctrl), C.c_double(model.com_height),
                                       C.byref(polys, b * N * C.sizeof(wg.ZmpPolytope)), C.byref(so[b]), C.byref(oo),
                                       C.c_int(0))
            assert outs[b].ret == rc, (it, b, rc, outs[b].ret)
            assert bytes(sg[b]) == bytes(so[b]), (it, b)
            if rc != 0:
                assert rc == -2 and (outs[b].n_iter, outs[b].n_active) == (oo.n_iter, oo.n_active)
                alive[b] = False
                continue
            assert bytes(outs[b]) == bytes(oo), (it, b)
            lived[b] = it + 1
            # the ZMP of the first previewed instant respects the polytope it was constrained to
            P = polys[b * N]
            zx = so[b].xk[0] - model.com_height / 9.81 * so[b].xk[2]; zy = so[b].xk[3] - model.com_height / 9.81 * so[b].xk[5]
            inside = min(inside, min(P.A[j][0] * zx + P.A[j][1] * zy + P.B[j] for j in range(P.nrows)))
        t0 += model.T
    assert inside > -1e-6
    assert lived.min() >= 30 and lived.max() >= 55       # everyone leaves the rest phase, some walk several steps
    for b in range(B):                                   # the CoM is where the feet are
        i_end = int(round(lived[b] * model.T / zm.T))
        feet_mid = 0.5 * (r["left"][b, i_end, :2] + r["right"][b, i_end, :2])
        assert np.hypot(so[b].xk[0] - feet_mid[0], so[b].xk[3] - feet_mid[1]) < 0.15, b


def _qld_setup(mode, N=None):
    wg.init(0)
    model = wg.dimitrov_defaults()
    if N:
        model.N = N                                            # n = 2N < 32: the register rows' unused tail
    model.solver = mode                                        # 1 = WG_DIMITROV_QLD, 2 = WG_DIMITROV_QLDANDLQ
    wg.dimitrov_configure(model)
    return model, wg.dimitrov_constants(model.N), wg.dimitrov_qld_constants(model.N)


@pytest.mark.parametrize("mode,ql", [(2, "reference"), (2, "restated"), (1, "reference"), (1, "restated"), (2, "restated-N12"), (2, "reference-N10")])
def test_fused_tick_with_the_ql_back_ends_bit_exact(mode, ql):
    """The reference's modes QLDANDLQ (2) and QLD (1) (ZMPConstrainedQPFastFormulation.cpp:1297-1320): ql0001_ as the tick's
    solver.  The GPU tick solves with the in-wave ql0002 through a structured view of DPu; the oracle tick builds the dense arrays
    as the driver does and calls ql0001_ -- the reference's own COMPILED qld.cpp called the driver's way, iwar[0] = 0 / 1
    ("reference": oracle/_ref; states, jerk sequences, return codes and samples bit for bit), or the restatement ("restated":
    iteration and active-set counts as well).  These are the back-ends of this tick whose CPU counterpart is pinned to the
    reference.  Mode QLDANDLQ walks; mode QLD is the reference's, literally: its OptA carries alpha VPu' instead of alpha VPu'VPu
    (:524-527), not symmetric, upper triangle not positive definite -- ql0001_ answers ifail = 2 ("accuracy insufficient") on
    the first tick there (the driver prints IFAIL and stops, :1348-1352) and here."""
    ql, _, small = ql.partition("-N")
    model, K, Kq = _qld_setup(mode, int(small) if small else None)
    N = model.N
    lib = ol.oracle()
    if ql == "reference":
        if not ol.have_ref():
            pytest.skip("oracle/_ref not built")
        lib.wgo_set_reference_ql(C.cast(getattr(ol.ref(), ol.REF_SYM), C.c_void_p))
    else:
        lib.wgo_set_reference_ql(None)
    try:
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        B, T = 20, 40
        plans = [dv.plan(np.random.default_rng(500 + g), n_steps=3 + g % 6) for g in range(B)]
        offs = [(5 * g) % 11 for g in range(B)]
        sg = (wg.DimitrovState * B)(); so = (wg.DimitrovState * B)()
        for g in range(B):
            for s in (sg[g], so[g]):
                s.starting = 1
                s.xk[0] = 0.002 * g; s.xk[4] = 0.001 * (g % 5 - 2)
        n_ok = 0; max_act = 0; iters = []; rets = set()
        OB, OC, PU = (K["OptB"], K["OptC"], K["Pu"]) if mode == 2 else (Kq["OptB"], Kq["OptC"], Kq["PuT"])
        for it in range(T):
            polys = (wg.ZmpPolytope * (B * N))()
            for g in range(B):
                for i, p in enumerate(dv.polys_at(plans[g], it + offs[g], N)):
                    _fill(polys[g * N + i], p)
            outs = wg.dimitrov_tick_batch(polys, sg)
            for g in range(B):
                oo = wg.DimitrovOut()
                rc = lib.wgo_dimitrov_qld_tick(C.c_int(mode), C.c_int(N), dp(Kq["Q"]), dp(OB), dp(OC), dp(PU), dp(K["Px"]), dp(K["iLQ"]),
                                               C.c_double(model.T), C.c_double(model.Tctrl), C.c_double(model.com_height),
                                               C.byref(polys, g * N * C.sizeof(wg.ZmpPolytope)), C.byref(so[g]), C.byref(oo))
                assert outs[g].ret == rc, (it, g, outs[g].ret, rc)
                rets.add(rc)
                assert bytes(sg[g]) == bytes(so[g]), (it, g)
                assert (outs[g].jerk_x, outs[g].jerk_y, outs[g].m) == (oo.jerk_x, oo.jerk_y, oo.m), (it, g)
                assert bytes(outs[g].X) == bytes(oo.X), (it, g)
                if rc == 0:
                    assert bytes(outs[g].com_x) == bytes(oo.com_x) and bytes(outs[g].zmp_y) == bytes(oo.zmp_y), (it, g)
                    n_ok += 1
                if ql == "restated":
                    assert (outs[g].n_iter, outs[g].n_active) == (oo.n_iter, oo.n_active), (it, g)
                max_act = max(max_act, outs[g].n_active); iters.append(outs[g].n_iter)
        if mode == 2:
            # a plan that ends inside the preview window poses inconsistent constraints: ql0001_ says so (ifail = 10 + row), the gait
            # stops there like the reference's driver does (:1348-1352) -- the same code from the GPU and from the reference
            assert all(r == 0 or r > 10 for r in rets) and n_ok > 0.6 * B * T and max_act >= 6 and np.mean(iters) > 3, (rets, n_ok)
            x_end = np.array([so[g].xk[0] for g in range(B)])
            assert x_end.max() > 0.2                              # the gaits did walk
        else:
            assert 2 in rets                                      # the reference's QLD mode: "IFAIL: 2" (see the docstring)
    finally:
        lib.wgo_set_reference_ql(None)
        wg.dimitrov_configure(wg.dimitrov_defaults())


@pytest.mark.parametrize("mode,B", [(2, 2200)])
def test_tick_longest_first_start_order_is_scheduling_only(mode, B, monkeypatch):
    """More gaits than resident waves through the QL back-end (eight per CU): from the second tick on the gaits start
    longest-solve-first by the previous tick's iteration counts.  Scheduling only: states and outputs of every tick equal the
    index-order run (WG_QL_LPT=0)."""
    model, K, Kq = _qld_setup(mode)
    N = model.N
    T = 4
    plans = [dv.plan(np.random.default_rng(700 + g % 37), n_steps=3 + g % 6) for g in range(B)]
    offs = [(5 * g) % 11 for g in range(B)]

    def run():
        st = (wg.DimitrovState * B)()
        for g in range(B):
            st[g].starting = 1
            st[g].xk[0] = 0.002 * (g % 50); st[g].xk[4] = 0.001 * (g % 5 - 2)
        res = []
        for it in range(T):
            polys = (wg.ZmpPolytope * (B * N))()
            for g in range(B):
                for i, p in enumerate(dv.polys_at(plans[g], it + offs[g], N)):
                    _fill(polys[g * N + i], p)
            outs = wg.dimitrov_tick_batch(polys, st)
            res.append((bytes(st), bytes(outs)))
        return res, [o.n_iter for o in outs]
    try:
        ordered, iters = run()
        monkeypatch.setenv("WG_QL_LPT", "0")
        plain, _ = run()
        assert ordered == plain
        assert max(iters) > min(iters) + (3 if mode else 0)       # there was something to order
    finally:
        wg.dimitrov_configure(wg.dimitrov_defaults())


def test_qld_tick_with_empty_polytopes():
    """Every polytope of the window without a row (m = 0): the QL back-end's register-row loader has no row to clamp its surplus
    lanes to (it indexed slot[-1]); the tick is the unconstrained minimiser, the same bytes as the oracle tick's."""
    model, K, Kq = _qld_setup(2)
    N = model.N
    lib = ol.oracle()
    lib.wgo_set_reference_ql(None)
    try:
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        B, T = 3, 4
        sg = (wg.DimitrovState * B)(); so = (wg.DimitrovState * B)()
        for g in range(B):
            for s in (sg[g], so[g]):
                s.starting = 1
                s.xk[0] = 0.01 * g; s.xk[3] = -0.02 * g
        for it in range(T):
            polys = (wg.ZmpPolytope * (B * N))()                  # zero-initialised: nrows = 0 everywhere
            outs = wg.dimitrov_tick_batch(polys, sg)
            for g in range(B):
                oo = wg.DimitrovOut()
                rc = lib.wgo_dimitrov_qld_tick(C.c_int(2), C.c_int(N), dp(Kq["Q"]), dp(K["OptB"]), dp(K["OptC"]), dp(K["Pu"]), dp(K["Px"]),
                                               dp(K["iLQ"]), C.c_double(model.T), C.c_double(model.Tctrl), C.c_double(model.com_height),
                                               C.byref(polys, g * N * C.sizeof(wg.ZmpPolytope)), C.byref(so[g]), C.byref(oo))
                assert outs[g].ret == rc == 0, (it, g, outs[g].ret, rc)
                assert outs[g].m == 0 and outs[g].n_active == 0
                assert bytes(sg[g]) == bytes(so[g]) and bytes(outs[g].X) == bytes(oo.X), (it, g)
    finally:
        wg.dimitrov_configure(wg.dimitrov_defaults())


def test_pldp_against_qldandlq_objective_gap_per_tick():
    """The same gaits through PLDP and through ql0001_ on the SAME preconditioned problem (mode QLDANDLQ), both from the same state
    at every tick.  In jerk coordinates u the common objective is 1/2 u' H u + D' u with H = LQ LQ' (the lower triangle of OptA,
    mirrored: what the LQ factor factors) and D = OptB xk - OptC ZMPRef as built; v = LQ' u turns it into 1/2 |v|^2 + (iLQ D)' v.
    ql0001_ returns the minimiser over the polytopes; PLDP stops on a face (it never drops a constraint inside a solve) -- feasible,
    objective at least QL's.  The gap per tick is reported (pytest -s) and bounded."""
    wg.init(0)
    mp = wg.dimitrov_defaults()
    mq = wg.dimitrov_defaults(); mq.solver = 2
    N = mp.N; n = 2 * N
    B, T = 16, 30
    plans = [dv.plan(np.random.default_rng(900 + g), n_steps=4 + g % 5) for g in range(B)]
    sp = (wg.DimitrovState * B)(); sq = (wg.DimitrovState * B)()
    for g in range(B):
        for s in (sp[g], sq[g]):
            s.starting = 1; s.xk[0] = 0.002 * g
    with wg.Context(0) as cp, wg.Context(0) as cq:
        assert cp.call("wg_dimitrov_configure", C.byref(mp)) == 0 and cq.call("wg_dimitrov_configure", C.byref(mq)) == 0
        Q = np.zeros((n, n)); OB = np.zeros((n, 6)); OC = np.zeros((n, n)); PuT = np.zeros((N, N)); Px = np.zeros((N, 3))
        hp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        assert cq.call("wg_dimitrov_get_qld_constants", hp(Q), hp(OB), hp(OC), hp(PuT)) == 0
        assert cq.call("wg_dimitrov_get_constants", None, None, None, None, None, hp(Px)) == 0
        OptA = Q.T                                               # Q is ql0001_'s column-major layout of OptA
        H = np.tril(OptA) + np.tril(OptA, -1).T                  # what OptCholesky factors (lower triangle)
        rel, worst = [], 0.0
        for it in range(T):
            polys = (wg.ZmpPolytope * (B * N))()
            for g in range(B):
                for i, p in enumerate(dv.polys_at(plans[g], it, N)):
                    _fill(polys[g * N + i], p)
            for g in range(B):                                   # the same state for both solvers; PLDP cold-started on it (its
                for k in range(6):                               # hot-start members describe ITS previous tick, not this state)
                    sp[g].xk[k] = sq[g].xk[k]
                sp[g].starting = 1
            xk = np.array([[sq[g].xk[k] for k in range(6)] for g in range(B)])
            op = (wg.DimitrovOut * B)(); oq = (wg.DimitrovOut * B)()
            assert cp.call("wg_dimitrov_tick_batch", B, C.addressof(polys), C.addressof(sp), C.addressof(op), 0) == 0
            assert cq.call("wg_dimitrov_tick_batch", B, C.addressof(polys), C.addressof(sq), C.addressof(oq), 0) == 0
            for g in range(B):
                if op[g].ret != 0 or oq[g].ret != 0:
                    continue
                zr = np.array([polys[g * N + i].centre[0] for i in range(N)] + [polys[g * N + i].centre[1] for i in range(N)])
                D = OB @ xk[g] - OC @ zr
                f = lambda X: 0.5 * X @ H @ X + D @ X             # noqa: E731
                Xp, Xq = np.array(op[g].X[:n]), np.array(oq[g].X[:n])
                fp, fq = f(Xp), f(Xq)
                rel.append((fp - fq) / max(abs(fq), 1e-12))
                for X in (Xp, Xq):                               # feasibility: A (Px xk + Pu X) + B >= 0
                    for i in range(N):
                        pu = np.zeros(N); pu[:i + 1] = PuT[:i + 1, i]
                        zx = Px[i] @ xk[g][:3] + pu @ X[:N]; zy = Px[i] @ xk[g][3:] + pu @ X[N:]
                        for j in range(polys[g * N + i].nrows):
                            worst = min(worst, polys[g * N + i].A[j][0] * zx + polys[g * N + i].A[j][1] * zy + polys[g * N + i].B[j])
        rel = np.array(rel)
        assert len(rel) > 100
        print("PLDP vs QLDANDLQ objective over %d ticks: relative gap mean %.3g, median %.3g, max %.3g, min %.3g; PLDP reaches the "
              "optimum (gap < 1e-9) on %.0f %% of the ticks; worst constraint value %.3g"
              % (len(rel), rel.mean(), np.median(rel), rel.max(), rel.min(), 100.0 * (np.abs(rel) < 1e-9).mean(), worst))
        assert rel.min() > -1e-7                                   # ql0001_'s solution is the minimiser
        assert worst > -1e-6                                       # both are feasible (PLDP's own 1e-8 slack, rows of norm ~1e-3)
