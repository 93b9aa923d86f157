"""The reference's TestHerdt2010 *OnLine* profile (tests/TestHerdt2010.cpp:64-91, 231-244: walk forward, sideways, turn on
the spot at +-10 rad/s, curves at +-6.08 rad/s, stop, :stoppg -- 110 s, > 1100 MPC ticks) on the GPU.

The profile's golden file (TestHerdt2010OnLineTestFGPI.datref) is NOT in the reference tree (.MISSING_LARGE_BLOBS), so
nothing here is pinned to the reference's output; what is checked:
  * every MPC tick of the scenario on the GPU (through the C ABI) against the CPU oracle: state, tick outputs, QP sizes,
    iteration counts and the complete active-set add/drop history, bit for bit;
  * the C++ facade (PatternGeneratorInterface::ParseCmd / RunOneStepOfTheControlLoop, test_herdt2010 --online) against
    the Python replay of the same control loop through the C ABI.
(tests/test_parity_net.py holds the same scenario oracle-vs-oracle: libm against include/wg_trig.h.)"""
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
import oraclelib as ol  # noqa: E402

wg = importlib.import_module("jrl-walkgen_amd")
pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "herdt_emergency_stop_datref.npz"))["datref"]


def _bytes(x):
    return bytes(memoryview(x).cast("B"))


def test_online_schedule_every_tick_bit_exact_vs_oracle():
    wg.init(0)
    ol.build_oracle()
    pt = C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))
    model, s_gpu, events = hr.online_walking_setup(GOLD)
    _, s_cpu, _ = hr.online_walking_setup(GOLD)
    wg.mpc_configure(model)
    clock = 0.0
    n_ticks = 0; n_events = 0; sizes = set()
    for it in range(1, 24000):
        clock += model.Tctrl
        was_online = bool(s_cpu.online)
        if was_online and s_cpu.ending_phase and clock >= s_cpu.time_to_stop:
            s_cpu.online = 0; s_gpu.online = 0
        if was_online and clock + 0.00001 > s_cpu.upper_time_limit:
            s_gpu.clock = clock; s_cpu.clock = clock
            arr = (wg.GaitState * 1)()
            C.memmove(C.byref(arr[0]), C.byref(s_gpu), C.sizeof(wg.GaitState))
            outs, diag, hist, hlen = wg.mpc_tick_batch(arr, want_out=True, hist_cap=512)
            C.memmove(C.byref(s_gpu), C.byref(arr[0]), C.sizeof(wg.GaitState))
            out_c = wg.TickOut(); dump = hr.QpDump()
            assert pt.wgo_mpc_tick(C.byref(model), C.byref(s_cpu), C.byref(out_c), C.byref(dump)) == 0
            assert _bytes(arr[0]) == _bytes(s_cpu), ("state differs", it)
            assert _bytes(outs[0]) == _bytes(out_c), ("tick output differs", it)
            assert list(diag[0]) == [dump.ifail, dump.n_iter, dump.nact, dump.n, dump.m, out_c.nb_prw_steps]
            assert int(hlen[0]) == dump.hist_len and list(hist[0, :dump.hist_len]) == list(dump.hist[:dump.hist_len]), it
            n_ticks += 1; n_events += dump.hist_len; sizes.add(dump.n)
        if not s_cpu.online:
            break
        if it in events:
            events[it](s_gpu); events[it](s_cpu)
    assert n_ticks > 1100 and n_events > 10000 and sizes == {32, 34, 36}
    assert not s_cpu.online and it > 110 * 200                      # :stoppg ended the on-line mode


def _gpu_tick(model, state, want_dump):
    arr = (wg.GaitState * 1)()
    C.memmove(C.byref(arr[0]), C.byref(state), C.sizeof(wg.GaitState))
    outs, diag, _, _ = wg.mpc_tick_batch(arr, want_out=True)
    C.memmove(C.byref(state), C.byref(arr[0]), C.sizeof(wg.GaitState))
    return outs[0], None


def test_online_schedule_through_the_facade(tmp_path):
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_herdt2010")
    assert os.path.exists(exe), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    out = tmp_path / "online.dat"
    r = subprocess.run([exe, "--online", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = np.loadtxt(out)
    wg.init(0)
    model, state, events = hr.online_walking_setup(GOLD)
    wg.mpc_configure(model)
    want = hr.replay(model, state, events, 24000, tick=_gpu_tick)
    assert rows.shape == want.shape and rows.shape[0] > 22000
    assert np.abs(rows - want).max() < 1e-9                          # 13-digit text round trip of the trace
    # the robot walked, side-stepped and turned as scheduled, and ended at rest between its feet
    assert np.ptp(rows[:, 1]) > 1.0 and np.ptp(rows[:, 2]) > 0.5 and np.ptp(rows[:, 4]) > 0.5
    lf, rf = rows[-1, 10:12], rows[-1, 22:24]
    assert np.abs(rows[-1, 1:3] - 0.5 * (lf + rf)).max() < 5e-3
