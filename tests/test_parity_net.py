"""Gates between the two builds of the CPU checker (no GPU needed).

The HIP kernels take their trigonometry from include/wg_trig.h; their bit-exact partner is the oracle built with
-DWGO_PORTABLE_TRIG.  The oracle that is pinned to the reference (golden file, compiled ql0001_) calls libm.  The two
differ by < 1 ulp per sin/cos; what must NOT differ is every discrete decision of the path: QP sizes, iteration counts,
ifail and the complete active-set add/drop history ("bit-exact active-set index sequences", BASELINE north_star).  These
tests make that a gate:
  * the reference's EmergencyStop scenario (tests/TestHerdt2010.cpp:260-265), legacy and today's semantics;
  * the reference's OnLine scenario (tests/TestHerdt2010.cpp:231-244; its golden file is absent from the reference tree,
    so this scenario is checked oracle-vs-oracle here and oracle-vs-GPU in test_online_gpu.py);
  * a 50-gait sample of the benchmark workload (config 3) over its 200 ticks."""
import ctypes as C
import os

import numpy as np

import herdt_replay as hr
import oraclelib as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "herdt_emergency_stop_datref.npz")


def _ptrig():
    ol.build_oracle()
    so = os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", ol.ORACLE_DIR, "libwg_oracle_ptrig.so"])
    return C.CDLL(so)


def _trace(lib, setup, max_calls, legacy_running=False):
    model, state, events = setup
    log = []

    def on_tick(it, clock, st, out, dump):
        log.append((it, dump.n, dump.m, dump.ifail, dump.n_iter, dump.nact, tuple(dump.hist[:min(dump.hist_len, hr.HIST_CAP)]),
                    st.com_x[0], st.com_y[0]))
    rows = hr.replay(model, state, events, max_calls, tick=hr.lib_tick(lib), on_tick=on_tick, legacy_running=legacy_running)
    return rows, log


def _same_decisions(la, lb):
    assert len(la) == len(lb)
    worst = 0.0
    for a, b in zip(la, lb):
        assert a[:7] == b[:7], ("discrete decisions differ at control step", a[0], a[1:6], b[1:6])
        worst = max(worst, abs(a[7] - b[7]), abs(a[8] - b[8]))
    return worst


def test_emergency_stop_same_active_set_history_libm_vs_portable_trig():
    datref = np.load(GOLD)["datref"]
    for legacy in (True, False):
        def setup():
            m, s, e = hr.emergency_stop_setup(datref)
            if not legacy:
                m.flags = 0
            return m, s, e
        ra, la = _trace(ol.oracle(), setup(), 6000, legacy_running=legacy)
        rb, lb = _trace(_ptrig(), setup(), 6000, legacy_running=legacy)
        assert len(la) >= 215 and ra.shape == rb.shape
        worst = _same_decisions(la, lb)
        assert worst < 1e-12, worst
        # CoM / ZMP columns to rounding; the feet acceleration columns amplify it (end-of-swing polynomials), still far
        # below the 1e-7 the golden file is printed to
        assert np.abs(ra - rb)[:, [1, 2, 5, 6, 8, 9]].max() < 1e-12 and np.abs(ra - rb).max() < 1e-7
        assert sum(len(a[6]) for a in la) > 1000               # the histories are not trivially empty


def test_online_walking_schedule_same_active_set_history_libm_vs_portable_trig():
    """TestHerdt2010's OnLine profile: 110 s, forward / sideways / turning on the spot at +-10 rad/s (saturated by the
    hip-yaw preview) / curves at +-6.08 rad/s, then :stoppg.  No golden exists for it in the reference tree."""
    datref = np.load(GOLD)["datref"]
    ra, la = _trace(ol.oracle(), hr.online_walking_setup(datref), 24000)
    rb, lb = _trace(_ptrig(), hr.online_walking_setup(datref), 24000)
    assert len(la) > 1000                                          # > 100 s of walking: one tick per 0.1 s
    assert ra.shape == rb.shape and ra.shape[0] > 20000
    worst = _same_decisions(la, lb)
    assert worst < 1e-10, worst
    sizes = {a[1] for a in la}
    assert sizes == {32, 34, 36}
    assert all(a[3] == 0 for a in la)                              # every QP of the schedule solves
    # the robot did what the schedule says: walked forward, sideways, and turned both ways
    com = ra[:, 1:3]; yaw = ra[:, 4]
    assert np.ptp(com[:, 0]) > 1.0 and np.ptp(com[:, 1]) > 0.5 and np.ptp(yaw) > 0.5
    # and came to rest between its feet once :stoppg had been sent
    lf, rf = ra[-1, 10:12], ra[-1, 22:24]
    assert np.abs(ra[-1, 1:3] - 0.5 * (lf + rf)).max() < 5e-3


def test_benchmark_sample_same_active_set_history_libm_vs_portable_trig():
    """50 gaits of config 3's workload (seeds 20100 + g, references redrawn every 50 ticks), 200 ticks each."""
    import importlib
    wg = importlib.import_module("jrl-walkgen_amd")
    libs = (ol.oracle(), _ptrig())
    model = hr.default_model()
    rng = np.random.default_rng(4096)
    sample = sorted(rng.choice(4096, 48, replace=False).tolist()) + [0, 4095]
    n_hist = 0
    for g in sample:
        r = np.random.Generator(np.random.MT19937(20100 + g))
        vt = np.stack([r.uniform(-0.1, 0.3, 4), r.uniform(-0.1, 0.1, 4), r.uniform(-0.2, 0.2, 4)], 1)
        st = []
        for _ in libs:
            s = hr.init_state(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
            s.nb_steps_left = 2
            st.append(s)
        for t in range(200):
            dumps = []
            for lib, s in zip(libs, st):
                if t % 50 == 0:
                    s.vref[0], s.vref[1], s.vref[2] = vt[t // 50]
                c = s.clock
                for _ in range(1 if t == 0 else (19 if t == 1 else 20)):
                    c += model.Tctrl
                s.clock = c
                d = hr.QpDump()
                assert lib.wgo_mpc_tick(C.byref(model), C.byref(s), None, C.byref(d)) == 0
                dumps.append((d.n, d.m, d.ifail, d.n_iter, d.nact, tuple(d.hist[:min(d.hist_len, hr.HIST_CAP)])))
            assert dumps[0] == dumps[1], (g, t, dumps[0][:5], dumps[1][:5])
            n_hist += len(dumps[0][5])
        assert abs(st[0].com_x[0] - st[1].com_x[0]) < 1e-10 and abs(st[0].com_y[0] - st[1].com_y[0]) < 1e-10   # after 20 s and metres of walking
    assert n_hist > 50 * 200 * 5
