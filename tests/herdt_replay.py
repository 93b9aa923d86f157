"""Replays the reference's 5 ms control loop around a tick function.

Mirrors PatternGeneratorInterfacePrivate::RunOneStepOfTheControlLoop
(src/PatternGeneratorInterfacePrivate.cpp:1246-1514, Herdt branch :1316-1327)
plus CoMAndFootOnlyStrategy::OneGlobalStepOfControl
(src/GlobalStrategyManagers/CoMAndFootOnlyStrategy.cpp:56-127) and the event
loop of tests/TestObject.cpp:515-605: the four deques, one pop per call, one
tick whenever  clock + 1e-5 > UpperTimeLimitToUpdate_.  The tick itself is a
callable so that the same replay drives the CPU oracle and the HIP path.

Row layout = the 38 columns of tests/TestObject.cpp:344-385.
"""
import ctypes as C

import numpy as np

import oraclelib as ol

SAMPLES = 20


import importlib as _il

_wg = _il.import_module("jrl-walkgen_amd")       # POD layouts only (no library load, no GPU needed)
FootSample, Model, GaitState, TickOut = _wg.FootSample, _wg.Model, _wg.GaitState, _wg.TickOut


HIST_CAP = 512


class QpDump(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("mmax", C.c_int), ("ifail", C.c_int), ("nact", C.c_int),
                ("n_iter", C.c_int), ("hist_len", C.c_int), ("iact", C.c_int * 128), ("hist", C.c_int * HIST_CAP),
                ("C", C.c_double * (80 * 80)), ("A", C.c_double * (200 * 80)), ("d", C.c_double * 80),
                ("b", C.c_double * 200), ("x", C.c_double * 80),
                ("lf_back_rewritten", FootSample), ("rf_back_rewritten", FootSample)]


def default_model():
    m = Model()
    ol.oracle().wgo_model_defaults(C.byref(m))
    return m


def init_state(model, com0, left_xyt, right_xyt):
    s = GaitState()
    a3 = lambda v: (C.c_double * 3)(*v)
    ol.oracle().wgo_gait_init(C.byref(model), C.byref(s), a3(com0), a3(left_xyt), a3(right_xyt))
    return s


def oracle_tick(model, state, want_dump=False):
    out = TickOut()
    dump = QpDump() if want_dump else None
    rc = ol.oracle().wgo_mpc_tick(C.byref(model), C.byref(state), C.byref(out), C.byref(dump) if dump else None)
    assert rc == 0, rc
    return out, dump


def _foot_row(f):
    return [f.x, f.y, f.z, f.dx, f.dy, f.dz, f.ddx, f.ddy, f.ddz, f.theta, f.omega, f.omega2]


def _copy_foot(f):
    g = FootSample()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(FootSample))
    return g


def replay(model, state, events, max_calls, tick=oracle_tick, zmp0=(0.0, 0.0), on_tick=None,
           legacy_running=False):
    """events: {iteration: callable(state)} applied after that call like TestObject::generateEvent.
    legacy_running: rows are produced until the queues run dry (behaviour of the revision that recorded
    the golden file; today RunOneStepOfTheControlLoop returns ZMPVelocityReferencedQP::Running()).
    Returns an (n_calls x 38) array."""
    dt = model.Tctrl
    # InitOnLine: 8 start samples in every queue (ZMPVelocityReferencedQP.cpp:240-275)
    nbuf = int(0.04 / dt)
    com_q = [dict(x=[state.com_x[0], 0.0, 0.0], y=[state.com_y[0], 0.0, 0.0], z=state.com_z, yaw=0.0) for _ in range(nbuf)]
    zmp_q = [tuple(zmp0) for _ in range(nbuf)]
    lf_q = [_copy_foot(state.lf[2]) for _ in range(nbuf)]
    rf_q = [_copy_foot(state.rf[2]) for _ in range(nbuf)]
    clock = 0.0
    rows = []
    for it in range(1, max_calls + 1):
        clock += dt
        was_online = bool(state.online)                                            # :332-335
        if was_online and state.ending_phase and clock >= state.time_to_stop:      # :338-340
            state.online = 0       # (the call that switches on-line mode off still runs its tick)
        if was_online and clock + 0.00001 > state.upper_time_limit:                # :346
            state.clock = clock
            res = tick(model, state, True)
            out, dump = res
            if on_tick:
                on_tick(it, clock, state, out, dump)
            lf_q[-1] = _copy_foot(out.lf_back)
            rf_q[-1] = _copy_foot(out.rf_back)
            for k in range(SAMPLES):
                com_q.append(dict(x=list(out.com_x[k]), y=list(out.com_y[k]), z=state.com_z, yaw=out.com_yaw[k][0]))
                zmp_q.append((out.zmp_x[k], out.zmp_y[k]))
                lf_q.append(_copy_foot(out.lf[k]))
                rf_q.append(_copy_foot(out.rf[k]))
        running = bool(state.running) or legacy_running
        if not com_q:
            break
        c = com_q.pop(0); z = zmp_q.pop(0); lf = lf_q.pop(0); rf = rf_q.pop(0)
        if not running:
            break
        row = [it * 0.005, c["x"][0], c["y"][0], c["z"], c["yaw"], c["x"][1], c["y"][1], 0.0, z[0], z[1]]
        row += _foot_row(lf) + _foot_row(rf) + [z[0], z[1], 0.0, 0.0]
        rows.append(row)
        if it in events:
            events[it](state)
    return np.array(rows)


def emergency_stop_events():
    """tests/TestHerdt2010.cpp:260-265 (+ handlers :128-200)."""
    def vel(x, y, w):
        def f(s):
            s.vref[0], s.vref[1], s.vref[2] = x, y, w
        return f

    def stop(s):
        s.vref[0], s.vref[1], s.vref[2] = 0.0, 0.0, 0.0
        s.ending_phase = 1
    return {int(5 * 200): vel(0.0, 0.0, 0.4), int(10 * 200): vel(0.2, 0.0, -0.2), int(15.2 * 200): vel(0.0, 0.0, 0.0),
            int(20.8 * 200): stop}


def online_walking_events():
    """The 12-event schedule of TestHerdt2010's OnLine profile, tests/TestHerdt2010.cpp:231-244 (+ handlers :121-200).
    Its golden file is not in the reference tree (.MISSING_LARGE_BLOBS): the scenario pins nothing against the reference,
    it is run oracle against oracle and oracle against GPU."""
    def vel(x, y, w):
        def f(s):
            s.vref[0], s.vref[1], s.vref[2] = x, y, w
        return f

    def stop(s):
        s.vref[0], s.vref[1], s.vref[2] = 0.0, 0.0, 0.0
        s.ending_phase = 1
    fwd, side = vel(0.2, 0.0, 0.0), vel(0.0, 0.2, 0.0)
    return {5 * 200: fwd, 10 * 200: side, 25 * 200: vel(0.0, 0.0, -10.0), 35 * 200: fwd, 45 * 200: vel(0.0, 0.0, 10.0),
            55 * 200: fwd, 65 * 200: vel(0.0, 0.0, -10.0), 75 * 200: fwd, 85 * 200: vel(0.2, 0.0, 6.0832),
            95 * 200: vel(0.2, 0.0, -6.0832), 105 * 200: vel(0.0, 0.0, 0.0), 110 * 200: stop}


def online_walking_setup(datref):
    """TestHerdt2010's OnLine profile (startOnLineWalking, tests/TestHerdt2010.cpp:64-91) on the same robot and start
    state as the EmergencyStop profile; today's source semantics (no legacy switches)."""
    m, s, _ = emergency_stop_setup(datref)
    m.flags = 0
    s.sup_y = s.lf[2].y                # today's InitOnLine (ZMPVelocityReferencedQP.cpp:283)
    return m, s, online_walking_events()


def lib_tick(lib):
    """tick callable for replay() on any build of the oracle (libm or include/wg_trig.h trigonometry)"""
    def tick(model, state, want_dump=False):
        out = TickOut()
        dump = QpDump() if want_dump else None
        rc = lib.wgo_mpc_tick(C.byref(model), C.byref(state), C.byref(out), C.byref(dump) if dump else None)
        assert rc == 0, rc
        return out, dump
    return tick


def load_datref(path):
    return np.loadtxt(path)


def emergency_stop_setup(datref):
    """Model / start state / events of TestHerdt2010's EmergencyStop profile on jrl-dynamics' sample robot.

    Robot constants that the reference reads from the (absent) sample model were identified from the
    golden file itself: sole 0.25 x 0.14 m, no hip-yaw limits (fallback -30/+45 deg on both
    legs, zero velocity bound), start CoM (0.0316055, 0, 0.7116911), feet at (0, +-0.09).
    Two things in the golden file predate the current reference source (ChangeLog [3.1.8]):
      * the initial support state had Y hard-coded to 0.1 (half the default feet distance) instead of the
        left foot's y,
      * the "move the CoM to the feet centre when stopping" branch did not exist, and rows were written
        until the queues ran dry.
    """
    m = default_model()
    m.flags = 1                      # WG_FLAG_NO_STOP_CENTERING
    s = init_state(m, [datref[0, 1], datref[0, 2], datref[0, 3]], [datref[0, 10], datref[0, 11], 0.0],
                   [datref[0, 22], datref[0, 23], 0.0])
    s.nb_steps_left = 2              # ":numberstepsbeforestop 2"  (ZMPVelocityReferencedQP.cpp:197-202)
    s.nb_steps_ssds = 2
    s.sup_y = 0.1
    return m, s, emergency_stop_events()
