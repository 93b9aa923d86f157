"""Pins the oracle's Herdt-2010 tick restatement (oracle/herdt_oracle.c) end to end against the
reference's own golden file tests/TestHerdt2010EmergencyStopTestFGPI.datref.cmake (committed as data in
tests/golden/herdt_emergency_stop_datref.npz): 4508 control steps x 38 columns, |delta| < 1e-6 per field
exactly like the reference's TestObject::compareDebugFiles (tests/TestObject.cpp:477-478)."""
import os

import numpy as np
import pytest

import herdt_replay as hr
import oraclelib as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "herdt_emergency_stop_datref.npz")


def _replay():
    datref = np.load(GOLD)["datref"]
    model, state, events = hr.emergency_stop_setup(datref)
    log = []
    rows = hr.replay(model, state, events, 6000, legacy_running=True,
                     on_tick=lambda it, clock, st, out, dump: log.append((out.ifail, out.n_iter, out.nact, out.n)))
    return datref, rows, np.array(log)


def test_emergency_stop_matches_reference_golden_file():
    datref, rows, log = _replay()
    assert rows.shape == datref.shape == (4508, 38)          # same number of control steps, too
    d = np.abs(rows - datref)
    assert d.max() < 1e-6, (d.max(), np.unravel_index(d.argmax(), d.shape))
    # CoM RMSE: limited by the 1e-7 truncation of the golden file (tests/TestObject.cpp:48-56)
    rmse = np.sqrt((d[:, 1:3] ** 2).mean())
    assert rmse < 1e-7
    assert len(log) == 225 and (log[:, 0] == 0).all()        # every QP solved, 225 MPC ticks
    assert set(np.unique(log[:, 3])) == {32, 34, 36}         # 0, 1 and 2 previewed steps all occur


def test_current_source_semantics_stop_centering():
    """Without the legacy flag the tick follows today's source: once NbStepsLeft reaches 0 the jerk is the
    closed form of ZMPVelocityReferencedQP.cpp:410-421 and Running() finally drops."""
    datref = np.load(GOLD)["datref"]
    model, state, events = hr.emergency_stop_setup(datref)
    model.flags = 0
    rows = hr.replay(model, state, events, 6000)
    assert 4000 < len(rows) < 4508
    assert np.abs(rows[:3700] - datref[:3700]).max() < 1e-6   # identical until the robot starts to stop
    lf, rf = rows[-1, 10:12], rows[-1, 22:24]
    assert np.abs(rows[-1, 1:3] - 0.5 * (lf + rf)).max() < 2e-3   # CoM ends between the feet


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref/libqld_ref.so not built")
def test_tick_with_the_reference_compiled_solver_is_bit_identical():
    """The whole EmergencyStop scenario with every QP solved by the REFERENCE's own compiled ql0001_ (oracle/_ref)
    instead of the restated solver: same 38-column trace, bit for bit."""
    import ctypes as C
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "herdt_emergency_stop_datref.npz"))["datref"]
    lib = ol.oracle()
    try:
        model, state, events = hr.emergency_stop_setup(gold)
        lib.wgo_set_reference_ql(None)
        rows_port = hr.replay(model, state, events, 6000, legacy_running=True)
        model, state, events = hr.emergency_stop_setup(gold)
        lib.wgo_set_reference_ql(C.cast(getattr(ol.ref(), ol.REF_SYM), C.c_void_p))
        rows_ref = hr.replay(model, state, events, 6000, legacy_running=True)
    finally:
        lib.wgo_set_reference_ql(None)
    assert rows_ref.shape == gold.shape
    assert np.array_equal(rows_ref, rows_port)
    assert np.abs(rows_ref - gold).max() < 1e-6


def test_c_run_loop_equals_the_python_loop():
    """wgo_mpc_run (bench.py's CPU leg) advances gaits exactly like the per-tick calls the tests make."""
    import ctypes as C
    import importlib
    wg = importlib.import_module("jrl-walkgen_amd")
    lib = ol.oracle()
    model = hr.default_model()
    ng, nt, redraw = 3, 30, 10
    rng = np.random.default_rng(5)
    vel = np.ascontiguousarray(rng.uniform(-0.1, 0.2, (3, ng, 3)))
    a = (wg.GaitState * ng)(); b = (wg.GaitState * ng)()
    for g in range(ng):
        s = hr.init_state(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s.nb_steps_left = 2
        C.memmove(C.byref(a[g]), C.byref(s), C.sizeof(wg.GaitState)); C.memmove(C.byref(b[g]), C.byref(s), C.sizeof(wg.GaitState))
    assert lib.wgo_mpc_run(C.byref(model), a, ng, nt, vel.ctypes.data_as(C.c_void_p), redraw) == 0
    for tick in range(nt):
        adv = 1 if tick == 0 else (19 if tick == 1 else 20)
        for g in range(ng):
            if tick % redraw == 0:
                b[g].vref[0], b[g].vref[1], b[g].vref[2] = vel[tick // redraw, g]
            c = b[g].clock
            for _ in range(adv):
                c += model.Tctrl
            b[g].clock = c
            assert lib.wgo_mpc_tick(C.byref(model), C.byref(b[g]), None, None) == 0
    assert bytes(memoryview(a).cast("B")) == bytes(memoryview(b).cast("B"))
