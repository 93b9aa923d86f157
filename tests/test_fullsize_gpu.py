"""Full-size and edge-size checks of the HIP path (through the C ABI).

At BASELINE.json's size (4096 gaits x 200 ticks) the oracle cannot follow every gait in test time, so the run is
checked through size-independent properties -- determinism, batch-composition invariance (a gait's trajectory does not
depend on its neighbours or its slot), shard consistency, physical sanity -- plus a seeded sample of gaits followed
bit-for-bit by the oracle over the whole 200 ticks.  Edge sizes: n = 1, config-5-sized dense QPs (n = 72, m = 149),
ragged batches, other horizons N through the element view (the default) and the dense tick policy (WG_TICK_DENSE=1), config 5's
N = 32 through the element view."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oraclelib as ol  # noqa: E402
import qpgen  # noqa: E402

wg = importlib.import_module("jrl-walkgen_amd")
pytestmark = pytest.mark.gpu
REDRAW = 50


def _vel(g, n_seg):
    r = np.random.Generator(np.random.MT19937(20100 + g))          # bench.py's table
    return np.stack([r.uniform(-0.1, 0.3, n_seg), r.uniform(-0.1, 0.1, n_seg), r.uniform(-0.2, 0.2, n_seg)], 1)


def _start(model, B):
    s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
    s0.nb_steps_left = 2
    st = (wg.GaitState * B)()
    for g in range(B):
        C.memmove(C.byref(st[g]), C.byref(s0), C.sizeof(wg.GaitState))
    return st


def _run(model, gaits, n_ticks):
    """advance the listed global gait indices n_ticks; returns the final state bytes per gait + per-tick diag"""
    B = len(gaits)
    st = _start(model, B)
    vt = [_vel(g, (n_ticks + REDRAW - 1) // REDRAW) for g in gaits]
    fails = 0; iters = []
    for t in range(n_ticks):
        if t % REDRAW == 0:
            for k in range(B):
                st[k].vref[0], st[k].vref[1], st[k].vref[2] = vt[k][t // REDRAW]
        adv = 1 if t == 0 else (19 if t == 1 else 20)
        _, diag, _, _ = wg.mpc_tick_batch(st, want_out=False, advance_calls=adv)
        fails += int((diag[:, 0] != 0).sum()); iters.append(diag[:, 1].copy())
    sz = C.sizeof(wg.GaitState)
    raw = bytes(memoryview(st).cast("B"))
    return [raw[k * sz:(k + 1) * sz] for k in range(B)], st, fails, np.array(iters)


@pytest.fixture(scope="module")
def full_run():
    wg.init(0)
    model = wg.model_defaults()
    wg.mpc_configure(model)
    B, T = 4096, 200
    fin, st, fails, iters = _run(model, list(range(B)), T)
    return model, B, T, fin, st, fails, iters


def test_full_size_run_is_sane(full_run):
    model, B, T, fin, st, fails, iters = full_run
    assert fails == 0                                              # every one of the 819 200 QPs solved (ifail == 0)
    assert iters.max() < 200 and iters.mean() > 5
    com = np.array([[s.com_x[0], s.com_y[0], s.com_x[1], s.com_y[1]] for s in st])
    feet = np.array([[s.lf[2].x, s.lf[2].y, s.rf[2].x, s.rf[2].y] for s in st])
    assert np.isfinite(com).all() and np.isfinite(feet).all()
    mid = 0.5 * (feet[:, :2] + feet[:, 2:])
    assert np.abs(com[:, :2] - mid).max() < 0.35                   # the CoM stays between the feet
    assert np.abs(com[:, 2:]).max() < 1.0                          # |v| bounded (references are <= 0.3 m/s)
    assert all(s.tick_count == T and s.running == 1 for s in st)
    assert len({f for f in fin}) > 4000                            # gaits really are different problems


def test_full_size_sample_followed_by_the_oracle(full_run):
    model, B, T, fin, st, fails, iters = full_run
    pt = C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))
    ol.build_oracle()
    rng = np.random.default_rng(4096)
    sample = sorted(rng.choice(B, 48, replace=False).tolist()) + [0, B - 1]
    for g in sample:
        s = _start(model, 1)[0]
        vt = _vel(g, (T + REDRAW - 1) // REDRAW)
        for t in range(T):
            if t % REDRAW == 0:
                s.vref[0], s.vref[1], s.vref[2] = vt[t // REDRAW]
            c = s.clock
            for _ in range(1 if t == 0 else (19 if t == 1 else 20)):
                c += model.Tctrl
            s.clock = c
            assert pt.wgo_mpc_tick(C.byref(model), C.byref(s), None, None) == 0
        assert bytes(memoryview(s).cast("B")) == fin[g], g


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("wg_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["wg_bench"] = mod
    spec.loader.exec_module(mod)
    return mod


def bench_plan_run(ctx, model, B, t_end, timed_from, bench):
    """bench.py's own launch sequence on device-resident states (its launch_plan, its velocity_table, its entry points:
    wg_mpc_tick_batch_dev for the control loop's first two ticks, wg_mpc_set_velref_dev + wg_mpc_run_batch_dev for a stretch
    that does not start on a redraw, wg_mpc_run_sched_dev with the references of every later stretch staged for one that
    does): ticks [0, timed_from) as its pre-roll + warm-up, [timed_from, t_end) as its timed region.  Returns the final states
    (host bytes per gait), the per-tick diagnostics and the entry point of every launch."""
    import torch
    n_seg = (t_end + bench.REDRAW_TICKS - 1) // bench.REDRAW_TICKS
    vtab = torch.from_numpy(bench.velocity_table(0, B, n_seg)).cuda()
    states = bench.start_states(model, B).cuda()
    diag = torch.zeros(t_end, B, 6, dtype=torch.int32, device="cuda")
    sp, dp, dstride = states.data_ptr(), diag.data_ptr(), B * 6 * 4
    stream = torch.cuda.Stream()
    sh = stream.cuda_stream
    names = []
    with torch.cuda.stream(stream):
        for t, n in bench.launch_plan(0, timed_from) + bench.launch_plan(timed_from, t_end):
            staged = n > 1 and t % bench.REDRAW_TICKS == 0
            if t % bench.REDRAW_TICKS == 0 and not staged:
                ctx.mpc_set_velref_dev(B, sp, vtab[t // bench.REDRAW_TICKS].data_ptr(), sh)
            adv = 1 if t == 0 else (19 if t == 1 else 20)
            if n == 1 and t < 2:
                names.append("wg_mpc_tick_batch_dev")
                ctx.mpc_tick_batch_dev(B, sp, None, dp + t * dstride, adv, stream=sh)
            elif staged:
                names.append("wg_mpc_run_sched_dev")
                ctx.mpc_run_sched_dev(B, sp, n, vtab[t // bench.REDRAW_TICKS].data_ptr(), bench.REDRAW_TICKS, adv, None,
                                      dp + t * dstride, stream=sh)
            else:
                names.append("wg_mpc_run_batch_dev")
                ctx.mpc_run_batch_dev(B, sp, n, adv, None, dp + t * dstride, stream=sh)
    torch.cuda.synchronize()
    raw = states.cpu().numpy().tobytes()
    sz = C.sizeof(wg.GaitState)
    return [raw[k * sz:(k + 1) * sz] for k in range(B)], diag.cpu().numpy(), names


def test_bench_plan_at_the_timed_size_ends_in_the_per_tick_runs_bytes(full_run):
    """The entry point bench.py times (wg_mpc_run_sched_dev, B = 4096, one launch from a redraw boundary with the later
    stretches' references staged) at the size it times it: bench.py's own pre-roll / warm-up / timed launch sequence over the
    first 200 ticks must leave every gait in the bytes the one-launch-per-tick host-pointer run left it in (which a sample of
    the oracle follows, test above), with the same ifail / iteration count / active-set size for every one of the 819 200
    QPs.  tools/soak_parity.py is the longer soak with the oracle on every gait."""
    model, B, T, fin, st, fails, iters = full_run
    bench = _bench_module()
    assert B == bench.BATCH_PER_GPU and bench.PREROLL_TICKS + 50 == 150       # the default W: timed region starts at tick 150
    fin2, diag, names = bench_plan_run(wg, model, B, T, 150, bench)
    assert names == ["wg_mpc_tick_batch_dev", "wg_mpc_tick_batch_dev", "wg_mpc_run_batch_dev", "wg_mpc_run_sched_dev",
                     "wg_mpc_run_sched_dev"]                                  # ticks 0 | 1 | 2-49 | 50-149 (staged) | 150-199 (timed)
    assert int((diag[:, :, 0] != 0).sum()) == 0
    assert np.array_equal(diag[:, :, 1], iters)
    assert fin2 == fin


def test_determinism_and_batch_composition_invariance(full_run):
    model, B, T, fin, st, fails, iters = full_run
    # same gaits in another order, other batch size, other neighbours -> identical bytes per gait
    perm = np.random.default_rng(1).permutation(B)[:600].tolist()
    fin2, _, _, _ = _run(model, perm, T)
    for k, g in enumerate(perm):
        assert fin2[k] == fin[g], g
    # shard consistency: two "ranks" with contiguous ranges reproduce the single-process run
    for lo, hi in ((0, 256), (3840, 4096)):
        fr, _, _, _ = _run(model, list(range(lo, hi)), T)
        assert fr == fin[lo:hi]


def test_dense_qp_edge_sizes_bit_exact():
    wg.init(0)
    rng = np.random.default_rng(72)
    qps = [qpgen.random_pd(rng, 1, 1), qpgen.random_pd(rng, 1, 3), qpgen.random_pd(rng, 2, 1),
           qpgen.random_pd(rng, 72, 149),                          # config-5-sized (N = 32: n <= 72, m <= 149), fp64
           qpgen.herdt_like(rng, 32, 4), qpgen.random_pd(rng, 64, 10), qpgen.boxed(rng, 50, 120),
           qpgen.random_pd(rng, 3, 100)]
    import test_ql_gpu as tq
    pk = wg.pack_qps(qps)                                          # one ragged batch: strides = the largest member
    res = wg.qp_solve_batch(pk, hist_cap=1024)
    bad = tq._compare(qps, res, "edge", pk)
    assert not bad, bad[:5]
    assert int(res["ifail"][3]) == 0


def _horizon_vs_oracle(N, T, step, B, ticks, redraw):
    pt = C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))
    model = wg.model_defaults()
    model.N = N; model.T = T; model.t_double = T; model.step_period = step
    model.Tctrl = T / 20.0                                         # the ABI fixes 20 control samples per tick
    wg.mpc_configure(model)
    try:
        st = _start(model, B)
        ref = _start(model, B)
        rng = np.random.default_rng(N)
        per_tick = int(round(T / model.Tctrl))
        sizes = set()
        for t in range(ticks):
            if t % redraw == 0:
                for g in range(B):
                    v = [rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2)]
                    for s in (st[g], ref[g]):
                        s.vref[0], s.vref[1], s.vref[2] = v
            adv = 1 if t == 0 else (per_tick - 1 if t == 1 else per_tick)
            _, diag, _, _ = wg.mpc_tick_batch(st, want_out=False, advance_calls=adv)
            assert (diag[:, 0] == 0).all(), diag[:, 0]
            sizes |= set(int(v) for v in diag[:, 3])
            for g in range(B):
                c = ref[g].clock
                for _ in range(adv):
                    c += model.Tctrl
                ref[g].clock = c
                assert pt.wgo_mpc_tick(C.byref(model), C.byref(ref[g]), None, None) == 0
            assert bytes(memoryview(st).cast("B")) == bytes(memoryview(ref).cast("B")), t
        return sizes
    finally:
        wg.mpc_configure(wg.model_defaults())


@pytest.mark.parametrize("N,T,step", [(8, 0.1, 0.8), (12, 0.1, 0.8), (16, 0.05, 0.8), (20, 0.1, 0.8), (24, 0.1, 0.8)])
def test_other_horizons_run_through_the_dense_policy_bit_exact(N, T, step, monkeypatch):
    """The register-resident problem view is instantiated for N = 16 with <= 2 previewed steps; the generic (dense, LDS) policy
    of the same solver takes any other model whose matrices fit a CU and must agree with the oracle just the same.  (Since round
    3 the element view is preferred wherever it applies -- WG_TICK_DENSE=1 keeps the dense policy under test.)"""
    monkeypatch.setenv("WG_TICK_DENSE", "1")
    wg.init(0)
    sizes = _horizon_vs_oracle(N, T, step, B=6, ticks=40, redraw=15)
    assert max(sizes) > 2 * N                                      # foot-placement variables did appear


@pytest.mark.parametrize("N,T,step", [(4, 0.1, 0.8), (8, 0.1, 0.8), (12, 0.1, 0.8), (16, 0.05, 0.8), (20, 0.1, 0.8), (24, 0.1, 0.8)])
def test_other_horizons_by_default_bit_exact(N, T, step):
    """the same models through the view wg_mpc_configure picks: the element view for every model but the benchmark's (at most
    12.6 KB of LDS per gait, twelve per CU, where the dense view's G and A take 100 KB at N = 20 -- one gait per CU, a quarter
    of the rate).  Horizons up to 13 have a smaller R than the pre-solve group that lies over it: theirs gets bytes of its own
    behind the tick's arrays (TickLds::elem_overlay_apart)."""
    wg.init(0)
    model = wg.model_defaults(); model.N = N; model.T = T; model.t_double = T; model.step_period = step; model.Tctrl = T / 20.0
    lds = wg.lib().wg_mpc_tick_lds_bytes_for(C.byref(model))
    assert lds <= 12800 or N == 16, lds                            # twelve gaits per CU (N = 16: the compact view, eight)
    sizes = _horizon_vs_oracle(N, T, step, B=6, ticks=40, redraw=15)
    assert max(sizes) > 2 * N


@pytest.mark.parametrize("generic", [False, True])
def test_config5_horizon_32_runs_through_the_element_view_bit_exact(generic, monkeypatch):
    """BASELINE config 5's problem size (N = 32, foot-placement variables kept: n up to 72, m up to 149), in fp64: its
    dense matrices do not fit a CU's LDS, so the tick regenerates G / A per element from the compact tables.  Z (42 KB) sits
    in a per-block slot of global memory.  N = 32 has a kernel instantiated for that horizon (every slot / LDS offset a constant);
    WG_TICK_ELEM_GENERIC=1 sends it through the any-horizon kernel: both against the oracle."""
    if generic:
        monkeypatch.setenv("WG_TICK_ELEM_GENERIC", "1")
    wg.init(0)
    sizes = _horizon_vs_oracle(32, 0.1, 0.8, B=5, ticks=36, redraw=12)
    assert max(sizes) == 72 and min(sizes) >= 64                   # all four previewed steps appeared


@pytest.mark.parametrize("N,abort_at", [(28, None), (30, "9"), (31, None)])
def test_horizons_next_to_32_take_the_element_view_with_their_own_column_cap(N, abort_at, monkeypatch):
    """N = 28 .. 31 do not fit the dense view either: element view, generic factorisation (the constant factor blocks are
    instantiated for N = 32 only), n <= 64 at N <= 30 (one row per lane), the host picks the column cap of R per model."""
    if abort_at:
        monkeypatch.setenv("WG_ELEM_ABORT_AT", abort_at)
    wg.init(0)
    model = wg.model_defaults(); model.N = N
    lds = wg.lib().wg_mpc_tick_lds_bytes_for(C.byref(model))
    assert lds <= 12800                                            # twelve gaits per CU
    sizes = _horizon_vs_oracle(N, 0.1, 0.8, B=4, ticks=28, redraw=9)
    assert max(sizes) > 2 * N


@pytest.mark.parametrize("abort_at", ["6", "31", "47", "31-generic"])
def test_config5_solves_that_outgrow_the_lds_part_of_R_are_repeated_in_global_memory(abort_at, monkeypatch):
    """At N = 32 the LDS holds the first 41 columns of R (twelve gaits per CU); a solve whose active set grows past them hands
    its loop state out (QlResume), R's finished columns move to the per-block slot of global memory and the same solve goes on
    there -- the same bytes as an uncapped solve.  WG_ELEM_ABORT_AT makes the hand-over happen at a smaller active set: at 6
    nearly every solve continues in global memory almost from the start, at 31 most of them, at 47 (beyond the layout's own
    41) the layout's cap decides."""
    if abort_at.endswith("-generic"):
        monkeypatch.setenv("WG_TICK_ELEM_GENERIC", "1")                # the any-horizon kernel at N = 32
        abort_at = abort_at.split("-")[0]
    monkeypatch.setenv("WG_ELEM_ABORT_AT", abort_at)
    wg.init(0)
    sizes = _horizon_vs_oracle(32, 0.1, 0.8, B=5, ticks=36, redraw=12)
    assert max(sizes) == 72 and min(sizes) >= 64


def test_benchmark_horizon_through_the_element_view(monkeypatch):
    """N = 16 takes the compact view (rows in registers, Z in LDS); WG_TICK_VIEW=element sends the same model through the element
    view -- half the rate (tools/n16_view_probe.sh), the same bytes"""
    wg.init(0)
    monkeypatch.setenv("WG_TICK_VIEW", "element")
    model = wg.model_defaults()
    assert wg.lib().wg_mpc_tick_lds_bytes_for(C.byref(model)) <= 12800
    sizes = _horizon_vs_oracle(16, 0.1, 0.8, B=4, ticks=30, redraw=10)
    assert max(sizes) == 36


def test_horizon_beyond_the_tables_is_refused():
    wg.init(0)
    model = wg.model_defaults()
    model.N = 48
    assert wg.lib().wg_mpc_configure(C.byref(model)) == -2        # WG_ERR_BAD_ARG: beyond the table size
    wg.mpc_configure(wg.model_defaults())


def _closed_loop(model, B, ticks, redraw, seed, oracle=False):
    """CoM trajectories [ticks, B, 2] (+ iterations) of B gaits under seeded velocity references"""
    pt = C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so")) if oracle else None
    st = _start(model, B)
    rng = np.random.default_rng(seed)
    per_tick = int(round(model.T / model.Tctrl))
    com = np.zeros((ticks, B, 2)); its = np.zeros((ticks, B), int)
    for t in range(ticks):
        if t % redraw == 0:
            for g in range(B):
                st[g].vref[0], st[g].vref[1], st[g].vref[2] = rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2)
        adv = 1 if t == 0 else (per_tick - 1 if t == 1 else per_tick)
        if oracle:
            for g in range(B):
                c = st[g].clock
                for _ in range(adv):
                    c += model.Tctrl
                st[g].clock = c
                assert pt.wgo_mpc_tick(C.byref(model), C.byref(st[g]), None, None) == 0
        else:
            _, diag, _, _ = wg.mpc_tick_batch(st, want_out=False, advance_calls=adv)
            assert (diag[:, 0] == 0).all(), diag[:, 0]
            its[t] = diag[:, 1]
        for g in range(B):
            com[t, g] = st[g].com_x[0], st[g].com_y[0]
    return com, its


@pytest.mark.parametrize("N", [16, 32])
def test_matrix_core_gramian_as_the_hessian_source(N):
    """BASELINE config 5, literally: N = 32 with foot-placement variables, Q_b from the fp32 MFMA Gramian
    (WG_FLAG_GRAMIAN_MFMA_F32).  Q_b's smallest eigenvalue is beta = 1e-5 and the fp32 operands cost 8e-8 of its largest
    entry (~1), so this is a tolerance mode: against the fp64 oracle (reference-order Q_b) the closed loop stays within
    1 mm over 6 s of walking and every QP still solves; the fp64 MFMA Gramian differs from the reference-order loop by
    rounding only, and the closed loop by less than a micrometre."""
    wg.init(0)
    B, ticks, redraw = 24, 60, 20
    base = wg.model_defaults()
    base.N = N
    try:
        ref, _ = _closed_loop(base, B, ticks, redraw, seed=5, oracle=True)
        out = {}
        for name, flag, tol in (("f64", 2, 1e-6), ("f32", 4, 1e-3)):
            m = wg.model_defaults(); m.N = N; m.flags = flag
            wg.mpc_configure(m)
            com, its = _closed_loop(m, B, ticks, redraw, seed=5)
            dev = np.abs(com - ref).max()
            out[name] = dev
            assert dev < tol, (name, dev)
            assert np.abs(com[-1] - com[0]).max() > 0.3                # they did walk
        assert out["f32"] > out["f64"]                                 # and the fp32 operands are visible
        print("N=%d: closed-loop CoM deviation from the fp64 oracle: f64 MFMA %.2e m, f32 MFMA %.2e m" % (N, out["f64"], out["f32"]))
    finally:
        wg.mpc_configure(wg.model_defaults())
