"""The drop-in boundary the way the reference's own harness uses it (tests/TestObject.cpp:138-260, 515-605): a
CjrlHumanoidDynamicRobot goes into patternGeneratorInterfaceFactory, joint values into SetCurrentJointValues, and the
generator evaluates its starting state from the robot's forward kinematics (PatternGeneratorInterfacePrivate.cpp:573-617).
jrl-walkgen_amd/host/test_herdt2010_robot.cpp does that with a small kinematic robot written against
include/wg_abstract_robot.hh; here its output is checked against
  * an independent evaluation of the same kinematics (numpy): CoM, feet, ZMP start, waist height;
  * the numbers the generator must read from the robot (mass, sole size of the LEFT foot, hip-yaw limits and velocity);
  * the Python replay of the control loop through the C ABI started from that state with that model: the whole
    EmergencyStop trace, also with ":setfeetconstraint XY mx my";
  * the rest of the interface (odometry, step stack, methods of other generators)."""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import herdt_replay as hr  # noqa: E402

wg = importlib.import_module("jrl-walkgen_amd")
EXE = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_herdt2010_robot")

# the reference's pure virtuals, include/jrl/walkgen/patterngeneratorinterface.hh:72-301, in declaration order (names only)
REFERENCE_VIRTUALS = ["AddStepInStack", "CommonInitializationOfWalking", "RunOneStepOfTheControlLoop", "RunOneStepOfTheControlLoop",
                      "RunOneStepOfTheControlLoop", "RunOneStepOfTheControlLoop", "SetCurrentJointValues", "GetWalkMode",
                      "GetLegJointVelocity", "ReadSequenceOfSteps", "StartOnLineStepSequencing", "StopOnLineStepSequencing",
                      "AddOnLineStep", "ChangeOnLineStep", "ChangeOnLineStep", "UpdateAbsolutePosition",
                      "getWaistPositionAndOrientation", "setWaistPositionAndOrientation", "getWaistVelocity",
                      "getWaistPositionMatrix", "setZMPInitialPoint", "getZMPInitialPoint", "ParseCmd", "EvaluateStartingState",
                      "setVelocityReference", "setCoMPerturbationForce"]


def test_interface_declares_every_reference_virtual_in_order():
    """no GPU needed: same methods, same order (=> same vtable slots) as the reference's abstract class"""
    txt = open(os.path.join(ROOT, "include", "wg_walkgen.hh")).read()
    body = txt[txt.index("class PatternGeneratorInterface {"):]
    body = body[:body.index("};")]
    names = re.findall(r"virtual\s+[\w:<>\s\*&]+?\b(\w+)\s*\(", body)
    names = [n for n in names if n != "PatternGeneratorInterface"]
    assert names == REFERENCE_VIRTUALS, names
    assert len(re.findall(r"=\s*0\s*;", body)) == len(REFERENCE_VIRTUALS)
    robot = open(os.path.join(ROOT, "include", "wg_abstract_robot.hh")).read()
    for m in ("mass", "leftFoot", "rightFoot", "waist", "jointsBetween", "numberDof", "getActuatedJoints", "setProperty",
              "currentConfiguration", "computeForwardKinematics", "positionCenterOfMass", "associatedAnkle", "getSoleSize",
              "getAnklePositionInLocalFrame", "lowerBound", "upperBound", "upperVelocityBound", "currentTransformation",
              "initialPosition"):
        assert re.search(r"virtual [^;]*\b%s\(" % m, robot), m


def _mock_kinematics(a):
    """the robot of test_herdt2010_robot.cpp in its start posture (hip pitch -a, knee 2a, ankle -a, waist at the origin)"""
    thigh = shank = 0.3; hip_y = 0.09; hip_z = -0.05; com_x = 0.04; ankle_z = 0.105
    m_waist, m_thigh, m_shank, m_foot = 40.0, 4.0, 3.0, 1.0

    def roty(t):
        return np.array([[np.cos(t), 0, np.sin(t), 0], [0, 1, 0, 0], [-np.sin(t), 0, np.cos(t), 0], [0, 0, 0, 1.0]])

    def tr(x, y, z):
        M = np.eye(4); M[:3, 3] = [x, y, z]
        return M
    acc = m_waist * np.array([com_x, 0.0, 0.0]); m = m_waist
    feet = []
    for y in (hip_y, -hip_y):
        hip = tr(0, y, hip_z) @ roty(-a)
        knee = hip @ tr(0, 0, -thigh) @ roty(2 * a)
        ankle = knee @ tr(0, 0, -shank) @ roty(-a)
        acc = acc + m_thigh * (hip @ tr(0, 0, -thigh / 2))[:3, 3] + m_shank * (knee @ tr(0, 0, -shank / 2))[:3, 3] + m_foot * ankle[:3, 3]
        m += m_thigh + m_shank + m_foot
        feet.append((ankle @ tr(0, 0, -ankle_z))[:3, 3])
    com = acc / m
    lf, rf = feet
    return dict(com=np.array([com[0], com[1], com[2] - rf[2]]), lf=lf, rf=rf, waist_z=-rf[2], zmp=0.5 * (lf + rf), mass=m)


def _run(tmp_path, *args):
    assert os.path.exists(EXE), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    out = tmp_path / "robot.dat"
    r = subprocess.run([EXE, str(out), *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(out).read().splitlines()
    assert lines[0].startswith("start ") and lines[-1].startswith("end ")
    start = np.array(lines[0].split()[1:], dtype=float)
    end = np.array(lines[-1].split()[1:], dtype=float)
    rows = np.loadtxt(lines[1:-1])
    return start, rows, end


def _gpu_tick(model, state, want_dump):
    arr = (wg.GaitState * 1)()
    C.memmove(C.byref(arr[0]), C.byref(state), C.sizeof(wg.GaitState))
    outs, diag, _, _ = wg.mpc_tick_batch(arr, want_out=True)
    C.memmove(C.byref(state), C.byref(arr[0]), C.sizeof(wg.GaitState))
    return outs[0], None


def _replay_from(start, margins=None):
    model = wg.model_defaults()
    model.sole_w, model.sole_h = start[14], start[15]
    model.hip_l_lo, model.hip_l_hi, model.hip_r_lo, model.hip_r_hi = start[16], start[17], start[16], start[17]
    model.hip_vmax = start[18]
    if margins:
        model.margin_x, model.margin_y = margins
    state = wg.gait_init(model, [start[0], start[1], start[2]], [start[6], start[7], start[8]], [start[9], start[10], start[11]])
    state.nb_steps_left = 2; state.nb_steps_ssds = 2
    wg.mpc_configure(model)
    try:
        return hr.replay(model, state, hr.emergency_stop_events(), 6000, tick=_gpu_tick, zmp0=(start[3], start[4]))
    finally:
        wg.mpc_configure(wg.model_defaults())


@pytest.mark.gpu
@pytest.mark.parametrize("knee_deg", [25.0, 32.0])
def test_testherdt2010_on_an_abstract_robot(tmp_path, knee_deg):
    wg.init(0)
    start, rows, end = _run(tmp_path, str(knee_deg))
    k = _mock_kinematics(knee_deg / 180.0 * np.pi)
    # EvaluateStartingState from the robot's forward kinematics
    assert np.abs(start[0:3] - k["com"]).max() < 1e-12
    assert np.abs(start[3:6] - k["zmp"]).max() < 1e-12                   # COGInitialAnkles (z before the feet are grounded)
    assert np.abs(start[[6, 7]] - k["lf"][:2]).max() < 1e-12 and np.abs(start[[9, 10]] - k["rf"][:2]).max() < 1e-12
    assert abs(start[8]) < 1e-12 and abs(start[11]) < 1e-12             # feet yaw (degrees)
    assert abs(start[12] - k["waist_z"]) < 1e-12                         # waist height above the ground
    # what the generator read from the robot
    assert abs(start[13] - k["mass"]) < 1e-12 and (start[14], start[15]) == (0.25, 0.14)
    assert abs(start[16] + 35.0 / 180.0 * np.pi) < 1e-15 and abs(start[17] - 40.0 / 180.0 * np.pi) < 1e-15 and start[18] == 3.5
    assert start[19] == 0                                                # GetWalkMode
    # the whole scenario: same as the control loop replayed through the C ABI from that state with that model
    want = _replay_from(start)
    assert rows.shape == want.shape and rows.shape[0] > 4000
    assert np.abs(rows - want).max() < 1e-9                              # 13-digit text round trip of the trace
    # hip-yaw limits and velocity bound of THIS robot were in force: another robot, another walk
    other = start.copy(); other[16:19] = [-30.0 / 180.0 * np.pi, 45.0 / 180.0 * np.pi, 0.0]
    assert np.abs(_replay_from(other)[:3000] - want[:3000]).max() > 1e-4
    # odometry at the end of the motion: the waist pose the caller reported last, in the (still initial) motion frame
    # (the pose handed in with the last successful call: the one the caller built from the call before)
    x, y, yaw = rows[-2, 1], rows[-2, 2], rows[-2, 4]
    assert abs(end[0] - x) < 1e-9 and abs(end[1] - y) < 1e-9 and abs(end[6] - x) < 1e-9 and abs(end[7] - y) < 1e-9
    assert abs(end[3] - np.sin(yaw / 2)) < 1e-9 and abs(end[4] - np.cos(yaw / 2)) < 1e-9 and abs(end[5] - np.fmod(yaw, 2 * np.pi)) < 1e-9
    assert end[11] == 6 and abs(end[12] - 0.03) < 1e-15                  # GetLegJointVelocity sizes, set/getZMPInitialPoint
    assert end[13] == -1 and end[14] == 1                                # ChangeOnLineStep (Morisawa only), StartOnLineStepSequencing


@pytest.mark.gpu
def test_setfeetconstraint_command_reaches_the_device_model(tmp_path):
    """":setfeetconstraint XY mx my" (relative-feet-inequalities.cpp:322-342) shrinks the ZMP polygon the QP enforces"""
    wg.init(0)
    start, rows, _ = _run(tmp_path, "25", "0.02", "0.055")
    want = _replay_from(start, margins=(0.02, 0.055))
    assert rows.shape == want.shape and np.abs(rows - want).max() < 1e-9
    base = _replay_from(start)
    n = min(len(base), len(rows))
    assert np.abs(rows[:n] - base[:n]).max() > 1e-4                      # and it is not the default polygon's walk
