"""wg_mpc_assemble_batch (the QP at the ql0001_ boundary, QPProblem::dump_problem), wg_mpc_tick_pinned (the one-robot path in
host-mapped memory) and the ordering of one context's launches -- through the C ABI on the GPU, against the CPU oracle."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import herdt_replay as hr  # noqa: E402
import oraclelib as ol  # noqa: E402

pytestmark = pytest.mark.gpu


def _wg():
    wg = importlib.import_module("jrl-walkgen_amd")
    wg.init(0)
    return wg


def _ptrig():
    ol.build_oracle()
    return C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))


def _bytes(x):
    return bytes(memoryview(x).cast("B"))


def _gaits(wg, model, B, seed):
    rng = np.random.default_rng(seed)
    states = (wg.GaitState * B)()
    for g in range(B):
        s = wg.gait_init(model, [0.0316055 + rng.normal(0, 0.003), rng.normal(0, 0.003), 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
        s.nb_steps_left = 2
        s.vref[0], s.vref[1], s.vref[2] = rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2)
        C.memmove(C.byref(states[g]), C.byref(s), C.sizeof(wg.GaitState))
    return states


@pytest.mark.parametrize("N", [16, 12])
def test_assembled_qp_is_the_oracles_and_its_dense_solve_is_the_fused_ticks(N):
    """Every tick of a de-synchronised batch: (1) the assembled Q, D, DU, DS equal what the oracle's tick hands to QL, bit for
    bit; (2) the states are left alone; (3) wg_qp_solve_batch on the assembled problems returns the solution, the iteration
    count and the add/drop history of the fused tick."""
    wg = _wg()
    pt = _ptrig()
    model = wg.model_defaults()
    model.N = N
    wg.mpc_configure(model)
    B = 12
    gpu = _gaits(wg, model, B, 31 + N)
    cpu = (wg.GaitState * B)()
    C.memmove(cpu, gpu, C.sizeof(gpu))
    sizes = set()
    for tick in range(30):
        adv = 1 if tick == 0 else (19 if tick == 1 else 20)
        before = _bytes(gpu)
        pk = wg.mpc_assemble_batch(gpu, advance_calls=adv, model=model)
        assert _bytes(gpu) == before, "assembling must not touch the states"
        dumps = []
        for g in range(B):
            c = cpu[g].clock
            for _ in range(adv):
                c += model.Tctrl
            cpu[g].clock = c
            d = hr.QpDump()
            assert pt.wgo_mpc_tick(C.byref(model), C.byref(cpu[g]), None, C.byref(d)) == 0
            dumps.append(d)
        nmax, mmax = pk["nmax"], pk["mmax"]
        for g, d in enumerate(dumps):
            n, m = d.n, d.m
            sizes.add(n)
            assert (int(pk["n"][g]), int(pk["m"][g])) == (n, m)
            Cg = pk["C"][g].reshape(nmax, nmax, order="F"); Ag = pk["A"][g].reshape(mmax, nmax, order="F")
            Co = np.frombuffer(d.C, dtype=np.float64, count=n * n).reshape(n, n, order="F")
            Ao = np.frombuffer(d.A, dtype=np.float64, count=d.mmax * n).reshape(d.mmax, n, order="F")
            assert ol.same_bits(Cg[:n, :n], Co) and not Cg[n:, :].any() and not Cg[:, n:].any(), (tick, g, "Q")
            assert ol.same_bits(Ag[:m, :n], Ao[:m, :]) and not Ag[m:, :].any() and not Ag[:, n:].any(), (tick, g, "DU")
            assert ol.same_bits(pk["d"][g, :n], np.frombuffer(d.d, dtype=np.float64, count=n)), (tick, g, "D")
            assert ol.same_bits(pk["b"][g, :m], np.frombuffer(d.b, dtype=np.float64, count=m)), (tick, g, "DS")
            assert (pk["xl"][g, :n] == -1e8).all() and (pk["xu"][g, :n] == 1e8).all()      # qp-problem.cpp:118-121
        res = wg.qp_solve_batch(pk, hist_cap=512)
        outs, diag, hist, hlen = wg.mpc_tick_batch(gpu, want_out=True, advance_calls=adv, hist_cap=512)
        assert _bytes(gpu) == _bytes(cpu), ("state differs", tick)
        for g, d in enumerate(dumps):
            n = d.n
            assert ol.same_bits(res["x"][g, :n], np.frombuffer(d.x, dtype=np.float64, count=n)), (tick, g, "x")
            assert (int(res["ifail"][g]), int(res["n_iter"][g])) == (int(diag[g, 0]), int(diag[g, 1])) == (d.ifail, d.n_iter)
            hl = int(hlen[g])
            assert int(res["hist_len"][g]) == hl == d.hist_len and list(res["hist"][g, :hl]) == list(hist[g, :hl])
    assert len(sizes) >= 2, sizes            # problems with and without previewed steps were seen


def test_one_robot_in_host_mapped_memory_equals_the_host_pointer_call():
    wg = _wg()
    model = wg.model_defaults()
    wg.mpc_configure(model)
    ref = _gaits(wg, model, 1, 5)
    hm = wg.HostMapped()
    try:
        C.memmove(C.addressof(hm.state), C.byref(ref[0]), C.sizeof(wg.GaitState))
        for tick in range(40):
            adv = 1 if tick == 0 else (19 if tick == 1 else 20)
            outs, diag, _, _ = wg.mpc_tick_batch(ref, want_out=True, advance_calls=adv)
            hm.tick(adv)
            assert _bytes(hm.state) == _bytes(ref[0]), tick
            assert _bytes(hm.out) == _bytes(outs[0]), tick
            assert list(hm.diag) == list(diag[0]), tick
    finally:
        hm.close()
    # memory that is not host-mapped is refused, nothing is launched
    plain = (wg.GaitState * 1)()
    rc = wg.lib().wg_mpc_tick_pinned(C.addressof(plain), None, None, 0)
    assert rc == -2 and b"wg_host_alloc" in wg.lib().wg_last_error()


def test_overlapping_launches_of_one_context_are_ordered(monkeypatch):
    """The tick / run kernels keep their queue and solver slots in the context.  A launch on a second stream while the first is
    in flight is enqueued BEHIND it (hipStreamWaitEvent on the event the first left) -- an event-ordered double-buffered pipeline
    is accepted as it is, an unordered one is serialised, nothing is corrupted; another context overlaps freely.  With
    WG_OVERLAP_STRICT=1 such a launch comes back WG_ERR_BUSY instead and launches nothing."""
    import torch
    wg = _wg()
    model = wg.model_defaults()
    B = 4096
    one = _bytes(_gaits(wg, model, 1, 9)[0])
    mk = lambda: torch.frombuffer(bytearray(one * B), dtype=torch.uint8).cuda()   # noqa: E731
    with wg.Context(0) as ctx, wg.Context(0) as other:
        ctx.mpc_configure(model); other.mpc_configure(model)
        a, b, c, ref = mk(), mk(), mk(), mk()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        # ---- default: ordered.  Two launches of ONE context on two streams, the second while the first is in flight
        ctx.mpc_run_batch_dev(B, a.data_ptr(), 60, 20, None, None, s1.cuda_stream)          # tens of milliseconds
        ctx.mpc_tick_batch_dev(B, b.data_ptr(), None, None, 20, stream=s2.cuda_stream)      # accepted: runs behind it
        ctx.mpc_run_batch_dev(B, b.data_ptr(), 1, 20, None, None, s2.cuda_stream)
        ctx.mpc_tick_batch_dev(B, a.data_ptr(), None, None, 20, stream=s1.cuda_stream)      # back on the first stream
        other.mpc_run_batch_dev(B, c.data_ptr(), 2, 20, None, None, s2.cuda_stream)         # another context: no ordering needed
        torch.cuda.synchronize()
        # nothing was corrupted: a = 61 ticks, b = c = 2 ticks of the same gait, against a plain sequence on one stream
        other.mpc_run_batch_dev(B, ref.data_ptr(), 2, 20, None, None, None)
        torch.cuda.synchronize()
        assert torch.equal(b, ref) and torch.equal(c, ref)
        other.mpc_run_batch_dev(B, ref.data_ptr(), 59, 20, None, None, None)
        torch.cuda.synchronize()
        assert torch.equal(a, ref)
        assert ctx.call("wg_overlap_serialised") == 2 and other.call("wg_overlap_serialised") == 0   # the library says what it ordered
        # ---- strict mode: refused (WG_OVERLAP_STRICT is read when a context is created; wg_set_overlap_strict toggles a live one)
        assert ctx.call("wg_set_overlap_strict", 1) == 0
        a2, b2 = mk(), mk()
        torch.cuda.synchronize()
        ctx.mpc_run_batch_dev(B, a2.data_ptr(), 60, 20, None, None, s1.cuda_stream)
        rc = ctx.call("wg_mpc_tick_batch_dev", B, C.c_void_p(b2.data_ptr()), None, None, 20, None, 0, None, C.c_void_p(s2.cuda_stream))
        assert rc == -5, rc                                                             # WG_ERR_BUSY
        assert b"in flight" in wg.lib().wg_last_error()
        rc = ctx.call("wg_mpc_run_batch_dev", B, C.c_void_p(b2.data_ptr()), 2, 20, None, None, C.c_void_p(s2.cuda_stream))
        assert rc == -5, rc
        assert ctx.call("wg_overlap_serialised") == 2                                   # refused launches are not counted as ordered
        ctx.mpc_tick_batch_dev(B, a2.data_ptr(), None, None, 20, stream=s1.cuda_stream)     # the same stream queues behind it
        torch.cuda.synchronize()
        ctx.mpc_tick_batch_dev(B, b2.data_ptr(), None, None, 20, stream=s2.cuda_stream)     # first launch done: accepted
        torch.cuda.synchronize()
        assert torch.equal(a2, a)
        one_tick = mk()
        other.mpc_tick_batch_dev(B, one_tick.data_ptr(), None, None, 20)
        torch.cuda.synchronize()
        assert torch.equal(b2, one_tick)


def test_assemble_launches_of_one_context_on_two_streams():
    """wg_mpc_assemble_batch_dev runs the tick on scratch copies of the states that belong to the context: two assemble launches
    of one context on two streams are ordered by the library and write the QPs of their own gaits."""
    import torch
    wg = _wg()
    model = wg.model_defaults()
    wg.mpc_configure(model)
    B = 1024
    nmax, mmax = 36, 76
    ga, gb = _gaits(wg, model, 1, 9), _gaits(wg, model, 1, 10)
    for g in (ga, gb):
        wg.mpc_tick_batch(g, advance_calls=1)
        for _ in range(30):
            wg.mpc_tick_batch(g, advance_calls=20)
    ref = {}
    for key, g in (("a", ga), ("b", gb)):
        ref[key] = wg.mpc_assemble_batch(g, 20, nmax, mmax)
    mk = lambda g: torch.frombuffer(bytearray(_bytes(g[0]) * B), dtype=torch.uint8).cuda()   # noqa: E731
    sa, sb = mk(ga), mk(gb)
    def bufs():
        return dict(C=torch.zeros(B, nmax * nmax, dtype=torch.float64, device="cuda"), d=torch.zeros(B, nmax, dtype=torch.float64, device="cuda"),
                    A=torch.zeros(B, mmax * nmax, dtype=torch.float64, device="cuda"), b=torch.zeros(B, mmax, dtype=torch.float64, device="cuda"),
                    xl=torch.zeros(B, nmax, dtype=torch.float64, device="cuda"), xu=torch.zeros(B, nmax, dtype=torch.float64, device="cuda"),
                    n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros(B, dtype=torch.int32, device="cuda"))
    with wg.Context(0) as ctx:
        ctx.mpc_configure(model)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        oa, ob = bufs(), bufs()
        torch.cuda.synchronize()
        for st, states, o in ((s1, sa, oa), (s2, sb, ob), (s1, sa, oa), (s2, sb, ob)):
            rc = ctx.call("wg_mpc_assemble_batch_dev", B, C.c_void_p(states.data_ptr()), 20, nmax, mmax, *[C.c_void_p(o[k].data_ptr()) for k in ("C", "d", "A", "b", "xl", "xu", "n", "m")],
                          C.c_void_p(st.cuda_stream))
            assert rc == 0, wg.lib().wg_last_error()
        torch.cuda.synchronize()
        for key, o in (("a", oa), ("b", ob)):
            r = ref[key]
            for k in ("C", "d", "A", "b"):
                got = o[k].cpu().numpy()
                assert (got == got[0]).all(), (key, k)                       # every copy of the gait gives the same QP
                assert got[0].tobytes() == np.ascontiguousarray(r[k][0]).tobytes(), (key, k)
