"""The host C++ facade (include/wg_walkgen.hh: PatternGeneratorInterface / SimplePlugin API of the reference) driven by
the reference's TestHerdt2010 EmergencyStop scenario (jrl-walkgen_amd/host/test_herdt2010.cpp), on the GPU:
  * --legacy run  vs the reference's own golden file (38 columns, 1e-6 like the reference's test harness);
  * default run   vs the Python replay of the same control loop through the C ABI (tests/herdt_replay.py)."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import herdt_replay as hr  # noqa: E402

wg = importlib.import_module("jrl-walkgen_amd")
pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_herdt2010")
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "herdt_emergency_stop_datref.npz"))["datref"]


def _run(tmp_path, *flags):
    assert os.path.exists(BIN), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    out = tmp_path / "trace.dat"
    r = subprocess.run([BIN, *flags, str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.loadtxt(out)


def test_facade_reproduces_reference_golden_file(tmp_path):
    rows = _run(tmp_path, "--legacy")
    assert rows.shape == GOLD.shape == (4508, 38)
    # the reference's comparison tolerance is 1e-6
    assert np.abs(rows - GOLD).max() < 1e-6
    com = rows[:, [1, 2]] - GOLD[:, [1, 2]]
    assert np.sqrt((com ** 2).mean()) < 1e-7


def _gpu_tick(model, state, want_dump):
    arr = (wg.GaitState * 1)()
    C.memmove(C.byref(arr[0]), C.byref(state), C.sizeof(wg.GaitState))
    outs, diag, _, _ = wg.mpc_tick_batch(arr, want_out=True)
    C.memmove(C.byref(state), C.byref(arr[0]), C.sizeof(wg.GaitState))
    return outs[0], None


def test_facade_equals_python_replay_of_the_control_loop(tmp_path):
    rows = _run(tmp_path)
    wg.init(0)
    model, state, events = hr.emergency_stop_setup(GOLD)
    model.flags = 0
    state.sup_y = state.lf[2].y            # today's InitOnLine (ZMPVelocityReferencedQP.cpp:283)
    wg.mpc_configure(model)
    want = hr.replay(model, state, events, 6000, tick=_gpu_tick)
    assert rows.shape == want.shape and rows.shape[0] > 4000
    # same arithmetic on both sides; the only difference is the 13-digit text round trip of the trace
    assert np.abs(rows - want).max() < 1e-9


def test_facade_todays_semantics_equals_the_oracle_over_the_full_run(tmp_path):
    """No legacy switches: the facade's whole EmergencyStop run (stop-centring branch of ZMPVelocityReferencedQP.cpp:410-421
    included, until Running() drops) against the CPU oracle driving the same control loop -- the libm build, i.e. the one
    pinned to the reference's golden file and to its compiled ql0001_.  Every row, not only those before the robot starts
    to stop (the golden file itself predates that branch: DESIGN 5.2)."""
    rows = _run(tmp_path)
    model, state, events = hr.emergency_stop_setup(GOLD)
    model.flags = 0
    state.sup_y = state.lf[2].y
    want = hr.replay(model, state, events, 6000)                    # tick = the oracle
    assert rows.shape == want.shape and 4000 < rows.shape[0] < 4508
    d = np.abs(rows - want)
    assert d[:, [1, 2, 5, 6, 8, 9]].max() < 1e-11                   # CoM, CoM velocity, ZMP: libm vs wg_trig.h, < 1 ulp per call
    assert d.max() < 1e-7                                           # feet accelerations amplify it; the file precision is 1e-7


def test_kajita_stage1_driver_matches_oracle(tmp_path):
    """BASELINE config[0] plumbing (jrl-walkgen_amd/host/test_kajita_preview.cpp): TestKajita2003's StraightWalking step
    sequence through the facade's StepStackHandler, ZMPDiscretization and PreviewControl on the GPU -- the reference's one-call-per-step pattern and the
    batched run agree inside the driver; here its trace is compared with the oracle, bit for bit (%.17g round-trips)."""
    from test_preview_oracle import oracle_run
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_kajita_preview")
    assert os.path.exists(exe), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    out = tmp_path / "kajita.dat"
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = np.loadtxt(out)
    g, F = wg.preview_gains(0.005, 0.814, 1.6, wg.RICCATI_WITHOUT_INITIALPOS)
    L = rows.shape[0]
    assert L > 2500 and g.nl == 320
    # the queue the driver built: its first L samples are in the trace, the tail rests at the last footprint
    zx = np.concatenate([rows[:, 5], np.full(g.nl - 1, rows[-1, 5])]); zy = np.concatenate([rows[:, 6], np.full(g.nl - 1, rows[-1, 6])])
    st = np.zeros((1, 8))
    com, z2 = oracle_run(g, F, zx[None, :], zy[None, :], st, L)
    assert np.array_equal(rows[:, 1], com[0, :, 0]) and np.array_equal(rows[:, 2], com[0, :, 3])
    assert np.array_equal(rows[:, 3], z2[0, :, 0]) and np.array_equal(rows[:, 4], z2[0, :, 1])
    # walked 14 x 0.2 m, ends at rest between the feet
    assert abs(rows[-1, 1] - 2.8) < 1e-3 and abs(rows[-1, 2] - rows[-1, 6]) < 1e-3
    # the queue and the feet are ZMPDiscretization's (facade class over wg_zmpdisc_batch): the reference's golden file
    # TestKajita2003StraightWalkingTestFGPI.datref holds them (columns 35-36, 11-13, 23-25), to its print precision
    gold = np.load(os.path.join(ROOT, "tests", "golden", "kajita_zmpdisc_datref.npz"))["StraightWalking_rows"]
    n = gold.shape[0]
    assert L == n + 2 * g.nl - g.nl + 1            # queue of n + 2 nl samples, L = queue - nl + 1 control steps
    assert np.abs(rows[:n, 5:7] - gold[:, 13:15]).max() < 2e-7
    assert np.abs(rows[:n, 7:10] - gold[:, 1:4]).max() < 2e-7 and np.abs(rows[:n, 10:13] - gold[:, 7:10]).max() < 2e-7


def test_kajita_circle_through_the_facade_matches_golden(tmp_path):
    """TestKajita2003's TurningOnTheCircle through the facade: StepStackHandler's ":supportfoot" / ":arc" / ":lastsupport"
    generators -> ZMPDiscretization (GPU) -> PreviewControl (GPU); queue and feet against the reference's golden file"""
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_kajita_preview")
    out = tmp_path / "circle.dat"
    r = subprocess.run([exe, str(out), "Circle"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = np.loadtxt(out)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "kajita_zmpdisc_datref.npz"))["Circle_rows"]
    n = gold.shape[0]
    assert rows.shape[0] == n + 320 + 1
    assert np.abs(rows[:n, 5:7] - gold[:, 13:15]).max() < 2e-7
    assert np.abs(rows[:n, 7:10] - gold[:, 1:4]).max() < 2e-7 and np.abs(rows[:n, 10:13] - gold[:, 7:10]).max() < 2e-7
    assert np.abs(rows[:n, 13] - gold[:, 4]).max() < 2e-7 and np.abs(rows[:n, 14] - gold[:, 10]).max() < 2e-7   # feet yaw
    assert abs(rows[-1, 13] - 30.0) < 1e-6


@pytest.mark.parametrize("profile", ["StraightWalking", "PbFlorentSeq1", "Circle"])
def test_testkajita2003_rehosted_on_the_interface(profile, tmp_path):
    """The reference's TestKajita2003 re-hosted on PatternGeneratorInterface (jrl-walkgen_amd/host/test_kajita2003.cpp):
    CommonInitialization + the profile's commands through ParseCmd, then RunOneStepOfTheControlLoop until it stops.  The
    loop ends after exactly as many calls as the reference's golden file has rows, and the feet and ZMP-reference columns
    are the golden's to its print precision; the CoM columns are stage 1's (the reference prints stage 2's) and must
    follow the ZMP reference."""
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_kajita2003")
    assert os.path.exists(exe), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    out = tmp_path / "k.dat"
    r = subprocess.run([exe, profile, str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = np.loadtxt(out)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "kajita_zmpdisc_datref.npz"))[profile + "_rows"]
    assert rows.shape == (gold.shape[0], 38)
    tol = 2e-7
    assert np.abs(rows[:, 0] - gold[:, 0]).max() < 1e-9                                      # time
    assert np.abs(rows[:, [10, 11, 12, 19, 20, 21]] - gold[:, 1:7]).max() < tol               # left foot x y z theta omega omega2
    assert np.abs(rows[:, [22, 23, 24, 31, 32, 33]] - gold[:, 7:13]).max() < tol              # right foot
    assert np.abs(rows[:, 34:36] - gold[:, 13:15]).max() < tol                                # world-frame ZMP reference
    # stage-1 CoM: at rest at the start, within a step length of the ZMP reference throughout, height = :comheight
    assert np.abs(rows[:200, 1:3]).max() < 1e-3 and np.abs(rows[:, 3] - 0.8078).max() < 1e-12
    assert np.abs(rows[:, 1:3] - rows[:, 34:36]).max() < 0.25


def test_cpp_fleet_bench_runs_through_the_c_abi():
    """jrl-walkgen_amd/host/fleet_bench.cpp: the fleet path from plain C++ (hipMalloc'd states, wg_mpc_run_batch_dev and
    wg_mpc_tick_batch_dev) -- both launch modes advance every gait by the same number of ticks"""
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "fleet_bench")
    assert os.path.exists(exe)
    for flags in ([], ["--per-tick"]):
        r = subprocess.run([exe, "--batch", "300", "--ticks", "60", *flags], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "%d ticks done in all" % (300 * (50 + 60)) in r.stdout, r.stdout


def test_cpp_fleet_bench_rank_launcher_with_rccl():
    """`fleet_bench --ranks 1`: the program starts one copy of itself per GPU (here one) before touching the GPU; the rank
    creates an RCCL communicator (ncclCommInitRank), receives wg_model_t through ncclBroadcast, configures its own context
    from it, takes its wg_shard_range and prints the job's JSON line.  World size 1 is the only size a one-GPU box can
    run; the N > 1 path is the same code and waits for the 8-GPU node (unmeasured on hardware until then)."""
    import json
    import signal
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "fleet_bench")

    def run(args, limit):
        # the launcher and its ranks in their own process group, so that a hang ends with every one of them killed
        p = subprocess.Popen([exe, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            out, err = p.communicate()
            raise AssertionError("fleet_bench %s hung:\n%s\n%s" % (args, out, err))
        return subprocess.CompletedProcess(args, p.returncode, out, err)

    r = run(["--ranks", "1", "--batch", "300", "--ticks", "60"], 150)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    d = json.loads(line[0])
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["scaling"] == "weak" and d["value"] > 0
    assert "ncclBroadcast" in d["config"]["collective"]
    assert d["ticks_done_in_all"] == 300 * (50 + 60)
    # more ranks than GPUs: every rank fails loudly, and so does the launcher
    r = run(["--ranks", "2", "--batch", "8", "--ticks", "4"], 150)
    assert r.returncode != 0 and "FAILED" in r.stderr


def test_two_facade_objects_of_a_kind_do_not_share_device_state():
    """jrl-walkgen_amd/host/test_two_objects.cpp: two PreviewControl objects (different windows / CoM heights) and two
    PatternGeneratorInterface objects (different robots), interleaved, each bit-identical to the same object run alone --
    the reference keeps this state per object; here every such object owns a context of the C ABI"""
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_two_objects")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "two objects ok" in r.stdout, r.stdout + r.stderr


def test_cpp_kajita_fleet_runs_through_the_c_abi():
    """jrl-walkgen_amd/host/kajita_fleet.cpp: step sequences -> wg_zmpdisc_batch_dev -> wg_preview_run_batch_dev from plain C++
    (hipMalloc'd buffers, one stream); the program itself compares gait 0 of the device chain with the host-pointer entry
    points bit for bit"""
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "kajita_fleet")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "--batch", "300", "--steps", "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "device chain == host entry points" in r.stdout and "4002 samples" in r.stdout, r.stdout
    assert "gait 0 ends at x = 2.8" in r.stdout, r.stdout           # fourteen 0.2 m steps


def test_every_reference_command_string_through_parsecmd(tmp_path):
    """host/test_commands.cpp: the fifteen commands PatternGeneratorInterfacePrivate registers (PatternGeneratorInterfacePrivate.cpp:
    186-201) and the step-stack generators, each sent through ParseCmd and checked for its effect (":ZMPShiftParameters" reaching
    ZMPDiscretization::SetZMPShift, ":SetAutoFirstStep" = AutomaticallyAddFirstStep, ":arccentered" geometry, the failed-QP dump of
    ZMPVelocityReferencedQP.cpp:399-402) or for the documented refusal (on-line step sequencing: NotOnThisPath)."""
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_commands")
    assert os.path.exists(exe), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("ok  ", "FAIL"))]
    assert r.returncode == 0 and len(lines) >= 10 and all(ln.startswith("ok  ") for ln in lines), r.stdout + r.stderr
    assert "all checks passed" in r.stdout
