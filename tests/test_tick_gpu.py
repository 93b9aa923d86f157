"""GPU parity of the fused Herdt-2010 tick (wg_mpc_tick_batch, through the C ABI) against the CPU oracle.

Bit-exact partner: oracle/libwg_oracle_ptrig.so (same restatement, trigonometry from include/wg_trig.h,
which the HIP kernels also use).  The libm oracle (pinned to the reference's golden file) is compared
with a tolerance, and the golden file itself is replayed THROUGH THE GPU at the reference's 1e-6."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import herdt_replay as hr
import oraclelib as ol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "herdt_emergency_stop_datref.npz")


def _wg():
    wg = importlib.import_module("jrl-walkgen_amd")
    wg.init(0)
    return wg


def _ptrig():
    ol.build_oracle()
    return C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))


def _bytes(x):
    return bytes(memoryview(x).cast("B"))


def _gpu_tick_factory(wg):
    def tick(model, state, want_dump):
        arr = (wg.GaitState * 1)()
        C.memmove(C.byref(arr[0]), C.byref(state), C.sizeof(wg.GaitState))
        outs, diag, hist, hlen = wg.mpc_tick_batch(arr, want_out=True, hist_cap=256)
        C.memmove(C.byref(state), C.byref(arr[0]), C.sizeof(wg.GaitState))
        out = wg.TickOut()
        C.memmove(C.byref(out), C.byref(outs[0]), C.sizeof(wg.TickOut))
        return out, None
    return tick


def test_golden_file_replayed_through_the_gpu():
    """The reference's EmergencyStop golden file (4508 x 38) with every MPC tick done by the HIP kernel."""
    wg = _wg()
    datref = np.load(GOLD)["datref"]
    model, state, events = hr.emergency_stop_setup(datref)
    wg.mpc_configure(model)
    rows = hr.replay(model, state, events, 6000, tick=_gpu_tick_factory(wg), legacy_running=True)
    assert rows.shape == datref.shape
    d = np.abs(rows - datref)
    assert d.max() < 1e-6, (d.max(), np.unravel_index(d.argmax(), d.shape))
    assert np.sqrt((d[:, 1:3] ** 2).mean()) < 1e-7


def test_tick_bit_exact_vs_portable_trig_oracle_on_scenario():
    """Same scenario, tick by tick: state, outputs, QP sizes, iteration count and the whole active-set add/drop
    history must be IDENTICAL to the portable-trig oracle; CoM must agree with the libm oracle to 1e-12."""
    wg = _wg()
    pt = _ptrig()
    datref = np.load(GOLD)["datref"]
    model, s_gpu, events = hr.emergency_stop_setup(datref)
    model.flags = 0                        # current-source semantics (stop centring on)
    wg.mpc_configure(model)
    _, s_cpu, _ = hr.emergency_stop_setup(datref)
    _, s_libm, _ = hr.emergency_stop_setup(datref)
    clock = 0.0
    n_ticks = 0
    worst = 0.0
    for it in range(1, 4400):
        clock += model.Tctrl
        if s_cpu.online and clock + 0.00001 > s_cpu.upper_time_limit:
            for st in (s_gpu, s_cpu, s_libm):
                st.clock = clock
            arr = (wg.GaitState * 1)()
            C.memmove(C.byref(arr[0]), C.byref(s_gpu), C.sizeof(wg.GaitState))
            outs, diag, hist, hlen = wg.mpc_tick_batch(arr, want_out=True, hist_cap=256)
            C.memmove(C.byref(s_gpu), C.byref(arr[0]), C.sizeof(wg.GaitState))
            out_c = wg.TickOut(); dump = hr.QpDump()
            assert pt.wgo_mpc_tick(C.byref(model), C.byref(s_cpu), C.byref(out_c), C.byref(dump)) == 0
            out_l, _ = hr.oracle_tick(model, s_libm)
            assert _bytes(arr[0]) == _bytes(s_cpu), ("state differs", it)
            assert _bytes(outs[0]) == _bytes(out_c), ("tick output differs", it)
            assert list(diag[0]) == [dump.ifail, dump.n_iter, dump.nact, dump.n, dump.m, out_c.nb_prw_steps]
            assert int(hlen[0]) == dump.hist_len and list(hist[0, :dump.hist_len]) == list(dump.hist[:dump.hist_len])
            worst = max(worst, abs(s_gpu.com_x[0] - s_libm.com_x[0]), abs(s_gpu.com_y[0] - s_libm.com_y[0]))
            n_ticks += 1
        if it in events:
            for st in (s_gpu, s_cpu, s_libm):
                events[it](st)
    assert n_ticks >= 215
    assert worst < 1e-12, worst            # libm vs include/wg_trig.h: < 1 ulp per call


def _random_gaits(wg, model, B, seed):
    rng = np.random.default_rng(seed)
    states = (wg.GaitState * B)()
    for g in range(B):
        s = wg.gait_init(model, [0.0316055 + rng.normal(0, 0.003), rng.normal(0, 0.003), 0.7116911],
                         [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
        s.nb_steps_left = 2
        C.memmove(C.byref(states[g]), C.byref(s), C.sizeof(wg.GaitState))
    return states, rng


def test_batch_of_desynchronised_gaits_bit_exact():
    """B independent gaits with their own velocity references (redrawn every 2.5 s), 60 ticks: every gait's
    state must stay bit-identical to the oracle run on the host."""
    wg = _wg()
    pt = _ptrig()
    model = wg.model_defaults()
    wg.mpc_configure(model)
    B = 96
    states, rng = _random_gaits(wg, model, B, 20100)
    ref_states = (wg.GaitState * B)()
    C.memmove(ref_states, states, C.sizeof(states))
    sizes = set()
    for tick in range(60):
        if tick % 25 == 0:
            for g in range(B):
                v = [rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2)]
                for st in (states[g], ref_states[g]):
                    st.vref[0], st.vref[1], st.vref[2] = v
        adv = 1 if tick == 0 else (19 if tick == 1 else 20)
        outs, diag, _, _ = wg.mpc_tick_batch(states, want_out=False, advance_calls=adv)
        for g in range(B):
            c = ref_states[g].clock
            for _ in range(adv):
                c += model.Tctrl
            ref_states[g].clock = c
            assert pt.wgo_mpc_tick(C.byref(model), C.byref(ref_states[g]), None, None) == 0
        assert _bytes(states) == _bytes(ref_states), tick
        sizes |= set(int(v) for v in diag[:, 3])
        assert (diag[:, 0] == 0).all()
    assert {34, 36} <= sizes <= {32, 34, 36}


def test_pushed_gaits_including_infeasible_qps_bit_exact():
    """Gaits started with large CoM velocities (a shove): many active rows while n = 32/34/36, QPs that QL declares
    inconsistent (ifail > 10) and their aftermath -- every state byte and every ifail code must still match the oracle."""
    wg = _wg()
    pt = _ptrig()
    model = wg.model_defaults()
    wg.mpc_configure(model)
    B = 160
    rng = np.random.default_rng(77)
    states = (wg.GaitState * B)()
    for g in range(B):
        s = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
        s.nb_steps_left = 2
        amp = 0.05 + 0.45 * g / B
        s.com_x[1] = rng.uniform(-amp, amp); s.com_y[1] = rng.uniform(-0.6 * amp, 0.6 * amp)
        s.com_x[2] = rng.uniform(-amp, amp)
        C.memmove(C.byref(states[g]), C.byref(s), C.sizeof(wg.GaitState))
    ref_states = (wg.GaitState * B)()
    C.memmove(ref_states, states, C.sizeof(states))
    seen_fail = set(); sizes = set(); max_act = 0
    for tick in range(45):
        if tick % 12 == 0:
            for g in range(B):
                v = [rng.uniform(-0.2, 0.4), rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.3)] if g % 3 else [0.0, 0.0, 0.0]
                for st in (states[g], ref_states[g]):
                    st.vref[0], st.vref[1], st.vref[2] = v
        adv = 1 if tick == 0 else (19 if tick == 1 else 20)
        outs, diag, _, _ = wg.mpc_tick_batch(states, want_out=True, advance_calls=adv)
        for g in range(B):
            c = ref_states[g].clock
            for _ in range(adv):
                c += model.Tctrl
            ref_states[g].clock = c
            o = hr.TickOut()
            assert pt.wgo_mpc_tick(C.byref(model), C.byref(ref_states[g]), C.byref(o), None) == 0
            assert o.ifail == diag[g, 0] and o.n_iter == diag[g, 1] and o.nact == diag[g, 2], (tick, g, o.ifail, diag[g])
            assert bytes(o) == bytes(outs[g]), (tick, g)
        assert _bytes(states) == _bytes(ref_states), tick
        seen_fail |= set(int(v) for v in diag[:, 0]); sizes |= set(int(v) for v in diag[:, 3])
        max_act = max(max_act, int(diag[:, 2].max()))
    assert sizes == {32, 34, 36} and max_act >= 28
    assert any(f > 10 for f in seen_fail) and 0 in seen_fail


@pytest.mark.parametrize("N,B,T,scale", [(16, 32, 150, 3.0), (32, 6, 160, 8.0)])
def test_overdriven_gaits_through_failed_and_nan_solves_end_in_the_oracles_states(N, B, T, scale):
    """The benchmark's recipe with three times its velocity references (0.9 m/s asked of a 0.7 m robot; eight times at N = 32,
    whose longer preview copes with three): QPs that QL declares inconsistent, states that grow past 1e140, then solves whose
    iterate becomes NaN -- the reference runs those to maxit = 40 (m + n) and returns ifail = 1 and an all-NaN x, which the tick
    integrates: the gait is lost.  The tick's views follow the reference through all of it decision by decision.
    Every tick: ifail and the iteration count equal the oracle's, the state and the tick's outputs (wg_tick_out_t: every sample
    the reference pushes on its deques) equal the oracle's (a NaN matches a NaN whatever its sign bit).  (Round 5: this test found a Givens rotation with a DENORMAL second operand -- gb = q / norm underflows to 0,
    ga = -1 -- that sweep_flat took for a skipped one.)"""
    wg = _wg()
    pt = _ptrig()
    model = wg.model_defaults()
    model.N = N
    wg.mpc_configure(model)                                      # (N = 16: the first maxit exit of this seed comes at tick 87)
    rng = np.random.default_rng(333)
    states = (wg.GaitState * B)()
    for g in range(B):
        s = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
        s.nb_steps_left = 2
        C.memmove(C.byref(states[g]), C.byref(s), C.sizeof(wg.GaitState))
    ref_states = (wg.GaitState * B)()
    C.memmove(ref_states, states, C.sizeof(states))
    sz = C.sizeof(wg.GaitState)

    def same(a, b):
        if a == b:
            return True
        wa, wb = np.frombuffer(a, dtype=np.uint64), np.frombuffer(b, dtype=np.uint64)
        d = wa != wb
        nan = lambda w: ((w >> np.uint64(52)) & np.uint64(0x7ff)) == np.uint64(0x7ff)   # noqa: E731
        return bool((nan(wa[d]) & nan(wb[d])).all())
    seen = set(); n_maxit = 0
    for tick in range(T):
        if tick % 50 == 0:
            for g in range(B):
                v = [scale * rng.uniform(-0.1, 0.3), scale * rng.uniform(-0.1, 0.1), scale * rng.uniform(-0.2, 0.2)]
                for st in (states[g], ref_states[g]):
                    st.vref[0], st.vref[1], st.vref[2] = v
        adv = 1 if tick == 0 else (19 if tick == 1 else 20)
        outs, diag, _, _ = wg.mpc_tick_batch(states, want_out=True, advance_calls=adv)
        got = _bytes(states)
        got_outs = _bytes(outs); osz = C.sizeof(wg.TickOut)
        for g in range(B):
            c = ref_states[g].clock
            for _ in range(adv):
                c += model.Tctrl
            ref_states[g].clock = c
            o = hr.TickOut()
            assert pt.wgo_mpc_tick(C.byref(model), C.byref(ref_states[g]), C.byref(o), None) == 0
            assert (o.ifail, o.n_iter) == (int(diag[g, 0]), int(diag[g, 1])), (tick, g, o.ifail, o.n_iter, diag[g])
            assert same(bytes(ref_states[g]), got[g * sz:(g + 1) * sz]), (tick, g)
            assert same(bytes(o), got_outs[g * osz:(g + 1) * osz]), (tick, g, "outputs")
            seen.add(int(o.ifail)); n_maxit += o.ifail == 1
    wg.mpc_configure(wg.model_defaults())
    assert 0 in seen and 1 in seen and any(f > 10 for f in seen), seen        # walking, the NaN regime and inconsistent QPs all occurred
    assert n_maxit >= 20
