"""Pins the oracle's Herdt-2010 tick restatement (oracle/herdt_oracle.c) end to end against the
reference's own golden file tests/TestHerdt2010EmergencyStopTestFGPI.datref.cmake (committed as data in
tests/golden/herdt_emergency_stop_datref.npz): 4508 control steps x 38 columns, |delta| < 1e-6 per field
exactly like the reference's TestObject::compareDebugFiles (tests/TestObject.cpp:477-478)."""
import os

import numpy as np

import herdt_replay as hr

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
