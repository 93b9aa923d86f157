"""Multi-tick launches (wg_mpc_run_batch_dev: a device-side work queue of (gait, next tick), no batch-wide drain between
ticks) against one launch per tick and against the CPU oracle: gait states, per-tick diagnostics and per-tick outputs
must be the same bytes, whatever the order in which the queue happened to serve the gaits.  Shapes cover a batch smaller
than the resident wave slots (waves poll the queue), a batch of several rounds, a single tick, and the dense view."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest
import torch

import oraclelib as ol

pytestmark = pytest.mark.gpu
wg = importlib.import_module("jrl-walkgen_amd")


def _ptrig():
    ol.build_oracle()
    return C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))


def _start(model, B, rng):
    s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
    s0.nb_steps_left = 2
    arr = (wg.GaitState * B)()
    for g in range(B):
        C.memmove(C.byref(arr[g]), C.byref(s0), C.sizeof(wg.GaitState))
        arr[g].vref[0], arr[g].vref[1], arr[g].vref[2] = rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2)
    return arr


def _dev(arr):
    return torch.frombuffer(bytearray(bytes(memoryview(arr).cast("B"))), dtype=torch.uint8).cuda()


@pytest.mark.parametrize("B,T,want_out", [(40, 30, True), (1, 7, False), (3000, 12, False), (64, 1, True), (2600, 10, True)])
def test_run_batch_equals_single_tick_launches_and_oracle(B, T, want_out):
    wg.init(0)
    model = wg.model_defaults()
    wg.mpc_configure(model)
    rng = np.random.default_rng(B + T)
    host = _start(model, B, rng)
    per_tick = (model.T / model.Tctrl)
    adv = int(round(per_tick))
    a = _dev(host); b = _dev(host)
    # the two special clock advances of the control loop's first ticks, the same way on both copies
    for st in (a, b):
        wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1)
        wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, adv - 1)
    osz = C.sizeof(wg.TickOut)
    da = torch.zeros(T, B, 6, dtype=torch.int32, device="cuda"); db = torch.zeros_like(da)
    oa = torch.zeros(T, B, osz, dtype=torch.uint8, device="cuda") if want_out else None
    ob = torch.zeros_like(oa) if want_out else None
    for t in range(T):
        wg.mpc_tick_batch_dev(B, a.data_ptr(), oa[t].data_ptr() if want_out else None, da[t].data_ptr(), adv)
    wg.mpc_run_batch_dev(B, b.data_ptr(), T, adv, ob.data_ptr() if want_out else None, db.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(da, db)
    if want_out:
        assert torch.equal(oa, ob)
    assert int((db[:, :, 0] != 0).sum()) == 0                     # every QP solved
    # and the oracle, on a sample of the gaits
    pt = _ptrig()
    final = b.cpu().numpy().reshape(B, -1)
    for g in sorted(set([0, B // 2, B - 1])):
        ref = wg.GaitState()
        C.memmove(C.byref(ref), C.byref(host[g]), C.sizeof(wg.GaitState))
        for k in [1, adv - 1] + [adv] * T:
            c = ref.clock
            for _ in range(k):
                c += model.Tctrl
            ref.clock = c
            assert pt.wgo_mpc_tick(C.byref(model), C.byref(ref), None, None) == 0
        assert bytes(memoryview(ref).cast("B")) == final[g].tobytes(), g


def test_run_batch_dense_view_and_arguments():
    wg.init(0)
    model = wg.model_defaults()
    model.N = 12                                                  # dense view
    wg.mpc_configure(model)
    try:
        rng = np.random.default_rng(12)
        B, T, adv = 20, 10, 20
        host = _start(model, B, rng)
        a = _dev(host); b = _dev(host)
        for t in range(T):
            wg.mpc_tick_batch_dev(B, a.data_ptr(), None, None, adv)
        wg.mpc_run_batch_dev(B, b.data_ptr(), T, adv)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        lib = wg.lib()
        assert lib.wg_mpc_run_batch_dev(-1, b.data_ptr(), 1, adv, None, None, None) == -2
        assert lib.wg_mpc_run_batch_dev(B, None, 1, adv, None, None, None) == -2
        assert lib.wg_mpc_run_batch_dev(B, b.data_ptr(), 0, adv, None, None, None) == 0     # nothing to do
    finally:
        wg.mpc_configure(wg.model_defaults())


@pytest.mark.parametrize("N,B,T,keep", [(16, 5000, 15, "off"), (16, 5000, 15, "0"), (16, 5000, 15, "3"), (16, 300, 9, "0"),
                                        (32, 3500, 6, "0"), (32, 3500, 6, "off"), (20, 700, 8, "1"), (-32, 3500, 6, "0")])
def test_hand_over_policy_does_not_change_a_byte(N, B, T, keep, monkeypatch):
    """The multi-tick kernel lets a wave keep a gait that is behind its XCD's mean progress (WG_RUN_KEEP = margin in ticks; "off":
    the plain ring; by default 0, and off when every gait has a wave of its own).  Which wave runs which tick of which gait in
    which order is scheduling only: states and per-tick diagnostics are the bytes of one launch per tick."""
    wg.init(0)
    monkeypatch.setenv("WG_RUN_KEEP", keep)
    if N < 0:                                                  # N = 32 through the any-horizon element kernel
        monkeypatch.setenv("WG_TICK_ELEM_GENERIC", "1"); N = -N
    model = wg.model_defaults(); model.N = N
    wg.mpc_configure(model)
    try:
        rng = np.random.default_rng(N + B + T)
        host = _start(model, B, rng)
        adv = int(round(model.T / model.Tctrl))
        a = _dev(host); b = _dev(host)
        for st in (a, b):
            wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1)
            wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, adv - 1)
        da = torch.zeros(T, B, 6, dtype=torch.int32, device="cuda"); db = torch.zeros_like(da)
        for t in range(T):
            wg.mpc_tick_batch_dev(B, a.data_ptr(), None, da[t].data_ptr(), adv)
        wg.mpc_run_batch_dev(B, b.data_ptr(), T, adv, None, db.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(da, db)
        assert int(da[:, :, 0].abs().sum()) == 0
    finally:
        wg.mpc_configure(wg.model_defaults())


@pytest.mark.parametrize("N,B", [(16, 2600), (32, 3300)])
def test_start_order_of_a_one_launch_tick_does_not_change_a_byte(N, B, monkeypatch):
    """With more gaits than resident waves wg_mpc_tick_batch_dev starts the gaits longest-solve-first, by the iteration counts of
    the previous call on the same state array (WG_TICK_LPT=0: index order).  States, diagnostics and outputs are the same bytes."""
    wg.init(0)
    model = wg.model_defaults(); model.N = N
    wg.mpc_configure(model)
    try:
        rng = np.random.default_rng(N + B)
        host = _start(model, B, rng)
        adv = int(round(model.T / model.Tctrl))
        osz = C.sizeof(wg.TickOut)
        res = []
        for lpt in ("0", "1"):
            monkeypatch.setenv("WG_TICK_LPT", lpt)
            st = _dev(host)
            T = 5
            dg = torch.zeros(T + 2, B, 6, dtype=torch.int32, device="cuda")
            out = torch.zeros(T + 2, B, osz, dtype=torch.uint8, device="cuda")
            for t, a in enumerate([1, adv - 1] + [adv] * T):
                wg.mpc_tick_batch_dev(B, st.data_ptr(), out[t].data_ptr(), dg[t].data_ptr(), a)
            torch.cuda.synchronize()
            res.append((st, dg, out))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
        assert int(res[0][1][:, :, 0].abs().sum()) == 0
        its = res[1][1][2:, :, 1].cpu().numpy()
        assert its.max() > its.min() + 5                               # there was something to order
    finally:
        wg.mpc_configure(wg.model_defaults())


@pytest.mark.parametrize("N,B,T,period,queue", [(16, 700, 37, 10, "xcd"), (16, 2500, 23, 23, "xcd"), (16, 64, 9, 4, "global"),
                                              (32, 96, 11, 5, "xcd"), (32, 200, 9, 4, "xcd-abort")])
def test_staged_references_equal_a_loop_of_set_velref_and_run(N, B, T, period, queue, monkeypatch):
    """wg_mpc_run_sched_dev: the velocity references of every stretch staged on the device, ONE launch for all the ticks --
    the same bytes (states, diagnostics, outputs) as wg_mpc_set_velref_dev + wg_mpc_run_batch_dev per stretch.  Also through
    the device-wide queue (which takes the per-stretch path inside the library) and the element view (N = 32)."""
    wg.init(0)
    if queue == "global":
        monkeypatch.setenv("WG_RUN_QUEUE", "global")
    if queue == "xcd-abort":                                   # N = 32: most solves repeated with R in global memory
        monkeypatch.setenv("WG_ELEM_ABORT_AT", "20")
    model = wg.model_defaults(); model.N = N
    wg.mpc_configure(model)
    try:
        rng = np.random.default_rng(N * 1000 + B + T)
        host = _start(model, B, rng)
        adv = int(round(model.T / model.Tctrl))
        a = _dev(host); b = _dev(host)
        for st in (a, b):
            wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1)
            wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, adv - 1)
        nblk = (T + period - 1) // period
        sched = torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, (nblk, B)), rng.uniform(-0.1, 0.1, (nblk, B)),
                                           rng.uniform(-0.2, 0.2, (nblk, B))], axis=2)).cuda().contiguous()
        osz = C.sizeof(wg.TickOut)
        da = torch.zeros(T, B, 6, dtype=torch.int32, device="cuda"); db = torch.zeros_like(da)
        oa = torch.zeros(T, B, osz, dtype=torch.uint8, device="cuda"); ob = torch.zeros_like(oa)
        for k in range(nblk):
            t0 = k * period; n = min(period, T - t0)
            wg.mpc_set_velref_dev(B, a.data_ptr(), sched[k].data_ptr())
            wg.mpc_run_batch_dev(B, a.data_ptr(), n, adv, oa[t0].data_ptr(), da[t0].data_ptr())
        wg.mpc_run_sched_dev(B, b.data_ptr(), T, sched.data_ptr(), period, adv, ob.data_ptr(), db.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(da, db) and torch.equal(oa, ob) and torch.equal(a, b)
        assert int(da[:, :, 0].abs().sum()) == 0                               # every QP solved
        # the references did change: the last block's values are in the states
        st = np.frombuffer(b.cpu().numpy().tobytes(), dtype=np.uint8).reshape(B, -1)
        off = wg.GaitState.vref.offset
        got = np.frombuffer(st[:, off:off + 24].tobytes(), dtype=np.float64).reshape(B, 3)
        assert np.array_equal(got, sched[nblk - 1].cpu().numpy())
        lib = wg.lib()
        assert lib.wg_mpc_run_sched_dev(B, b.data_ptr(), 1, adv, sched.data_ptr(), 0, None, None, None) == -2   # period < 1
    finally:
        wg.mpc_configure(wg.model_defaults())
