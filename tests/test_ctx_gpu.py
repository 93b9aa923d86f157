"""Contexts of the C ABI (include/wg_mpc.h, "Contexts"): two configured models side by side, launches of both in flight
at the same time on two streams, each bit-exact against the CPU oracle -- and the default context untouched by either.
Before contexts the library held ONE model per process and its tick launches shared one workspace."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oraclelib as ol  # noqa: E402

wg = importlib.import_module("jrl-walkgen_amd")
pytestmark = pytest.mark.gpu
SZ = C.sizeof(wg.GaitState)


def _ptrig():
    ol.build_oracle()
    return C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))


def _fleet(model, B, seed):
    rng = np.random.default_rng(seed)
    arr = (wg.GaitState * B)()
    for g in range(B):
        s = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
        s.nb_steps_left = 2
        s.vref[0], s.vref[1], s.vref[2] = rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2)
        C.memmove(C.byref(arr[g]), C.byref(s), SZ)
    return arr


def _dev(arr):
    return torch.frombuffer(bytearray(bytes(memoryview(arr).cast("B"))), dtype=torch.uint8).cuda()


def _oracle(pt, model, start, advs):
    s = wg.GaitState()
    C.memmove(C.byref(s), C.byref(start), SZ)
    for k in advs:
        c = s.clock
        for _ in range(k):
            c += model.Tctrl
        s.clock = c
        assert pt.wgo_mpc_tick(C.byref(model), C.byref(s), None, None) == 0
    return bytes(memoryview(s).cast("B"))


def test_two_models_two_streams_interleaved_bit_exact():
    wg.init(0)
    pt = _ptrig()
    m16 = wg.model_defaults()
    m32 = wg.model_defaults(); m32.N = 32
    wg.mpc_configure(m16)                                           # the default context: must stay what it is
    with wg.Context(0) as ca, wg.Context(0) as cb:
        ca.mpc_configure(m16)
        cb.mpc_configure(m32)
        assert ca.mpc_tick_lds_bytes() == wg.mpc_tick_lds_bytes_for(m16)
        assert cb.mpc_tick_lds_bytes() == wg.mpc_tick_lds_bytes_for(m32) != ca.mpc_tick_lds_bytes()
        Ba, Bb = 3000, 900                                          # both beyond what the device keeps resident at once
        ha, hb = _fleet(m16, Ba, 16), _fleet(m32, Bb, 32)
        da, db = _dev(ha), _dev(hb)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        advs = [1, 19] + [20] * 14
        # launches of the two contexts alternate from the host and overlap on the device (two streams, no events)
        ca.mpc_tick_batch_dev(Ba, da.data_ptr(), None, None, 1, stream=sa.cuda_stream)
        cb.mpc_tick_batch_dev(Bb, db.data_ptr(), None, None, 1, stream=sb.cuda_stream)
        ca.mpc_tick_batch_dev(Ba, da.data_ptr(), None, None, 19, stream=sa.cuda_stream)
        cb.mpc_tick_batch_dev(Bb, db.data_ptr(), None, None, 19, stream=sb.cuda_stream)
        for _ in range(2):
            ca.mpc_run_batch_dev(Ba, da.data_ptr(), 5, 20, stream=sa.cuda_stream)          # multi-tick launch: its own queue
            cb.mpc_run_batch_dev(Bb, db.data_ptr(), 3, 20, stream=sb.cuda_stream)
            cb.mpc_tick_batch_dev(Bb, db.data_ptr(), None, None, 20, stream=sb.cuda_stream)
            ca.mpc_tick_batch_dev(Ba, da.data_ptr(), None, None, 20, stream=sa.cuda_stream)
            cb.mpc_run_batch_dev(Bb, db.data_ptr(), 3, 20, stream=sb.cuda_stream)
            ca.mpc_tick_batch_dev(Ba, da.data_ptr(), None, None, 20, stream=sa.cuda_stream)
        torch.cuda.synchronize()
        fa = da.cpu().numpy().reshape(Ba, SZ); fb = db.cpu().numpy().reshape(Bb, SZ)
        for g in sorted(set(np.random.default_rng(1).choice(Ba, 20, replace=False).tolist()) | {0, Ba - 1}):
            assert _oracle(pt, m16, ha[g], advs) == fa[g].tobytes(), ("N=16 context", g)
        for g in sorted(set(np.random.default_rng(2).choice(Bb, 10, replace=False).tolist()) | {0, Bb - 1}):
            assert _oracle(pt, m32, hb[g], advs) == fb[g].tobytes(), ("N=32 context", g)
        # the whole batches again, one context at a time on the default stream: same bytes as the overlapped run
        for ctx, h, B, fin in ((ca, ha, Ba, fa), (cb, hb, Bb, fb)):
            d = _dev(h)
            for k in advs:
                ctx.mpc_tick_batch_dev(B, d.data_ptr(), None, None, k)
            torch.cuda.synchronize()
            assert np.array_equal(d.cpu().numpy().reshape(B, SZ), fin)
    # the default context still holds ITS model (N = 16) and works after the other two are gone
    assert wg.mpc_tick_lds_bytes() == wg.mpc_tick_lds_bytes_for(m16)
    h = _fleet(m16, 5, 3)
    ref = [_oracle(pt, m16, h[g], [1]) for g in range(5)]
    wg.mpc_tick_batch(h, want_out=False, advance_calls=1)
    assert [bytes(memoryview(h[g]).cast("B")) for g in range(5)] == ref


def test_two_robots_of_the_same_horizon_do_not_share_tables():
    """two contexts, both N = 16, different robots (sole size, QP weights): each follows ITS model's oracle"""
    wg.init(0)
    pt = _ptrig()
    ma = wg.model_defaults()
    mb = wg.model_defaults(); mb.sole_w = 0.20; mb.sole_h = 0.12; mb.beta = 2e-5; mb.com_height_qp = 0.75
    with wg.Context(0) as ca, wg.Context(0) as cb:
        ca.mpc_configure(ma)
        cb.mpc_configure(mb)                                        # configured later: must not replace ca's tables
        B = 64
        advs = [1, 19] + [20] * 18
        ha, hb = _fleet(ma, B, 7), _fleet(mb, B, 7)
        for k in advs:
            _, diag_a, _, _ = ca.mpc_tick_batch(ha, want_out=False, advance_calls=k)
            _, diag_b, _, _ = cb.mpc_tick_batch(hb, want_out=False, advance_calls=k)
            assert (diag_a[:, 0] == 0).all() and (diag_b[:, 0] == 0).all()
        st = _fleet(ma, B, 7)
        differ = 0
        for g in range(B):
            a, b = bytes(memoryview(ha[g]).cast("B")), bytes(memoryview(hb[g]).cast("B"))
            assert a == _oracle(pt, ma, st[g], advs), g
            assert b == _oracle(pt, mb, st[g], advs), g
            differ += a != b
        assert differ == B                                          # and the two robots really walk differently


def test_context_argument_checks():
    wg.init(0)
    lib = wg.lib()
    h = C.c_void_p()
    assert lib.wg_ctx_create(99, C.byref(h)) == -2 and not h.value  # no such device
    assert lib.wg_mpc_tick_batch_ctx(None, 1, None, None, None, 0, None, 0, None) == -2
    with wg.Context(0) as c:
        assert c.device() == 0
        st = _fleet(wg.model_defaults(), 1, 0)
        assert lib.wg_mpc_tick_batch_ctx(c.handle, 1, C.addressof(st), None, None, 0, None, 0, None) == -2   # not configured
        assert b"not been called on this context" in lib.wg_last_error()
        assert c.mpc_tick_lds_bytes() == 0


def test_host_calls_of_one_context_do_not_wait_for_another_contexts_launch():
    """The host-pointer entry points stage, launch and copy back on their context's own non-blocking stream and wait for that
    stream alone (they used to end in hipDeviceSynchronize: one facade object's tick waited for a fleet's whole launch, two
    PatternGeneratorInterface objects serialised on each other).  A long multi-tick launch of context A is in flight on a stream
    -- half of the device's wave slots, so that there is room beside it -- while context B ticks ONE robot through host pointers:
    every one of B's calls returns while A's launch is still running, at a latency within 2 x what it has on an idle device (+ a
    fixed 150 us for sharing the chip), and with the bytes of the same ticks made alone.  So does a re-configuration of B."""
    import time
    wg.init(0)
    model = wg.model_defaults()
    with wg.Context(0) as ca, wg.Context(0) as cb:
        ca.mpc_configure(model); cb.mpc_configure(model)
        fleet = _dev(_fleet(model, 1024, 11))
        ca.mpc_tick_batch_dev(1024, fleet.data_ptr(), None, None, 1)
        ca.mpc_tick_batch_dev(1024, fleet.data_ptr(), None, None, 19)
        torch.cuda.synchronize()

        def robot_ticks(n):
            st = _fleet(model, 1, 12)
            cb.mpc_tick_batch(st, want_out=True, advance_calls=1)
            cb.mpc_tick_batch(st, want_out=True, advance_calls=19)
            lat = []
            for _ in range(n):
                t0 = time.perf_counter()
                cb.mpc_tick_batch(st, want_out=True, advance_calls=20)
                lat.append(time.perf_counter() - t0)
            return lat, bytes(memoryview(st).cast("B"))

        robot_ticks(5)                                             # warm: buffers sized, kernels loaded
        alone, ref_bytes = robot_ticks(40)
        s1 = torch.cuda.Stream()
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        ca.mpc_run_batch_dev(1024, fleet.data_ptr(), 600, 20, None, None, s1.cuda_stream)       # ~ 0.2 s of device time
        done.record(s1)
        beside, got_bytes = robot_ticks(40)
        cb.mpc_configure(model)                                    # waits for B's own launches only
        t_host = time.perf_counter() - t0
        still_running = not done.query()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        assert still_running, "context A's launch ended before B's 42 host ticks did (%.1f ms): nothing was measured" % (1e3 * t_host)
        assert t_all > 2.0 * t_host                                 # B's calls took a fraction of A's launch, not all of it
        med_alone, med_beside = float(np.median(alone)), float(np.median(beside))
        assert med_beside < 2.0 * med_alone + 150e-6, (med_alone, med_beside)
        assert got_bytes == ref_bytes
