"""GPU parity: the HIP dual active-set solver (wg_qp_solve_batch, through the C ABI)
against the CPU oracle (oracle/ql_oracle.c, itself pinned bit-for-bit to the
compiled reference qld.cpp).  Bar: bit-exact x, u, ifail, final active set AND
the full add/drop history."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import oraclelib as ol
import qpgen

pytestmark = pytest.mark.gpu


def _wg():
    wg = importlib.import_module("jrl-walkgen_amd")
    wg.init(0)
    return wg


def _padded(pk, qps, k):
    """QP k exactly as the kernel sees it: batch strides nmax/mmax (matters for
    the c(nmax,nmax) patch of qld.cpp:442-444)."""
    nmax, mmax = pk["nmax"], pk["mmax"]
    q = qps[k]
    return dict(n=q["n"], m=q["m"], me=q["me"], nmax=nmax, mmax=mmax,
                C=np.asfortranarray(pk["C"][k].reshape((nmax, nmax), order="F")),
                A=np.asfortranarray(pk["A"][k].reshape((mmax, nmax), order="F")),
                d=pk["d"][k].copy(), b=pk["b"][k].copy(), xl=pk["xl"][k].copy(), xu=pk["xu"][k].copy())


def _compare(qps, res, label, pk):
    bad = []
    for k, q in enumerate(qps):
        o = ol.oracle_ql(_padded(pk, qps, k))
        n, m = q["n"], q["m"]
        ok = int(res["ifail"][k]) == o["ifail"]
        ok &= ol.same_bits(res["x"][k, :n], o["x"])
        ok &= int(res["n_iter"][k]) == o["n_iter"]
        ok &= int(res["nact"][k]) == o["nact"]
        ok &= np.array_equal(res["iact"][k, :o["nact"]], o["iact"])
        ok &= int(res["hist_len"][k]) == o["hist_len"]
        hl = min(o["hist_len"], res["hist"].shape[1])
        ok &= np.array_equal(res["hist"][k, :hl], o["hist"][:hl])
        if o["ifail"] == 0:
            ok &= ol.same_bits(res["u"][k, :m + 2 * n], o["u"])
        if not ok:
            bad.append((label, k, int(res["ifail"][k]), o["ifail"], int(res["n_iter"][k]), o["n_iter"],
                        float(np.abs(res["x"][k, :n] - o["x"]).max())))
    return bad


@pytest.mark.parametrize("family", sorted(qpgen.FAMILIES))
def test_family_bit_exact(family):
    wg = _wg()
    gen = qpgen.FAMILIES[family]
    qps = [gen(np.random.default_rng(7000 + 131 * s)) for s in range(96)]
    pk = wg.pack_qps(qps)
    res = wg.qp_solve_batch(pk, hist_cap=512)
    bad = _compare(qps, res, family, pk)
    assert not bad, bad[:5]


def test_golden_vectors_and_reference_history_through_the_gpu():
    """The committed fixture of the COMPILED REFERENCE (tests/golden/ql_golden.npz: x, u, ifail, final set and the reference's
    own add/drop history) against the HIP solver directly, no oracle in between: "bit-exact active-set index sequences"."""
    import test_ql_oracle as tqo
    wg = _wg()
    gold = list(tqo._golden())
    groups = {}
    for tag, q, ref in gold:      # one launch per problem shape family keeps the padding (nmax, mmax) as the fixture's
        groups.setdefault((q["n"], q["mmax"]), []).append((tag, q, ref))
    n_checked = 0
    for (n, mmax), items in sorted(groups.items()):
        qps = [q for _, q, _ in items]
        pk = wg.pack_qps(qps)
        assert pk["nmax"] == n and pk["mmax"] == mmax
        res = wg.qp_solve_batch(pk, hist_cap=max(512, max(len(r["hist"]) for _, _, r in items)))
        for k, (tag, q, ref) in enumerate(items):
            m = q["m"]
            assert int(res["ifail"][k]) == ref["ifail"], tag
            assert ol.same_bits_nan_aware(res["x"][k, :n], ref["x"]), tag     # NaN where the reference has NaN (8 such QPs)
            assert int(res["hist_len"][k]) == len(ref["hist"]), tag
            assert np.array_equal(res["hist"][k, :len(ref["hist"])], ref["hist"]), tag
            if ref["ifail"] == 0:
                assert ol.same_bits(res["u"][k, :m + 2 * n], ref["u"]), tag
                assert np.array_equal(res["iact"][k, :len(ref["iact"])], ref["iact"]), tag
            n_checked += 1
    assert n_checked >= 450


def test_herdt_shape_uniform_batch():
    """Uniform n/m batch through the NULL-size-array convention (m = mmax-1)."""
    wg = _wg()
    qps = [qpgen.herdt_like(np.random.default_rng(99 + s), 16, 2) for s in range(256)]
    pk = wg.pack_qps(qps)
    full = dict(pk)
    pk["n"] = None; pk["m"] = None; pk["me"] = None
    res = wg.qp_solve_batch(pk, hist_cap=512)
    bad = _compare(qps, res, "herdt_uniform", full)
    assert not bad, bad[:5]


@pytest.mark.parametrize("a_in_lds,g_in_lds,w_in_lds", [("1", "1", "1"), ("0", "1", "1"), ("0", "0", "1"), ("0", "0", "0"), ("0", "0", "0-generic")])
def test_matrices_in_lds_or_read_in_place(a_in_lds, g_in_lds, w_in_lds, monkeypatch):
    """A is staged in LDS only while that does not cost a resident QP (Herdt-sized QPs read it in place, from L2), and G -- cold
    once Z = R^-1 exists, its diagonal kept in LDS for ql0002's shift -- follows it out when that buys two more resident QPs,
    the constraint weights wa | b (a slot of global memory per QP) when that buys the eighth;
    every placement is the same arithmetic and is held to the oracle, on Herdt-shaped QPs and on small dense ones, on QPs whose
    Hessian needs the diagonal shift (singular) and on the c(nmax, nmax) == 0 patch.  With everything out of the LDS a batch of
    exactly the Herdt size (nmax = 36, mmax = 76) takes the kernel instantiated for that size (strides, LDS layout and the solver's
    loop bounds compile-time constants); "0-generic" (WG_QL_FIXED=0) keeps the any-size kernel on the same batch."""
    if w_in_lds.endswith("-generic"):
        monkeypatch.setenv("WG_QL_FIXED", "0"); w_in_lds = "0"
    monkeypatch.setenv("WG_QL_A_IN_LDS", a_in_lds)
    monkeypatch.setenv("WG_QL_G_IN_LDS", g_in_lds)
    monkeypatch.setenv("WG_QL_W_IN_LDS", w_in_lds)
    wg = _wg()
    qps = [qpgen.herdt_like(np.random.default_rng(4100 + s), 16, 2) for s in range(64)] + \
          [qpgen.random_pd(np.random.default_rng(4200 + s), 12, 9) for s in range(32)]
    pk = wg.pack_qps(qps)
    res = wg.qp_solve_batch(pk, hist_cap=512)
    bad = _compare(qps, res, "a_in_lds=%s g_in_lds=%s w_in_lds=%s" % (a_in_lds, g_in_lds, w_in_lds), pk)
    assert not bad, bad[:5]
    for family in sorted(qpgen.FAMILIES):
        gen = qpgen.FAMILIES[family]
        qps = [gen(np.random.default_rng(9000 + 17 * s)) for s in range(24)]
        pk = wg.pack_qps(qps)
        res = wg.qp_solve_batch(pk, hist_cap=512)
        bad = _compare(qps, res, family, pk)
        assert not bad, (family, bad[:5])


def test_bounds_only_qp_in_the_fixed_size_batch():
    """m = 0 (bounds only) inside a batch of exactly the Herdt size: the fixed 36 x 76 kernel keeps A's rows in registers and
    clamps the surplus lanes' row index -- with m = 0 that clamp must stay inside the caller's buffers (it read row -1)."""
    wg = _wg()
    rng = np.random.default_rng(360)
    qps = [qpgen.herdt_like(np.random.default_rng(8100 + s), 16, 2) for s in range(6)]
    for k in (0, 3, 5):                                       # first, middle, last QP of the batch: no general constraint at all
        q = qps[k]
        q["m"] = 0
        q["xl"] = np.full(q["n"], -0.05) if k else q["xl"]    # one of them with bounds that bind, the others effectively free
        q["xu"] = np.full(q["n"], 0.05) if k else q["xu"]
        q["A"] = np.asfortranarray(rng.standard_normal(q["A"].shape))   # whatever lies in A must not matter
        q["b"] = rng.standard_normal(q["b"].shape)
    pk = wg.pack_qps(qps)
    assert pk["nmax"] == 36 and pk["mmax"] == 76
    res = wg.qp_solve_batch(pk, hist_cap=512)
    bad = _compare(qps, res, "m=0", pk)
    assert not bad, bad[:5]
    assert all(int(res["ifail"][k]) == 0 for k in (0, 3, 5))
    assert int(res["nact"][3]) > 0 and (res["iact"][3, :int(res["nact"][3])] > 0).all()      # the bounds did bind: codes m + i ...


def test_non_finite_iterates_end_the_way_the_reference_ends_them():
    """QPs on which ql0002's iterate becomes NaN (7 of 6000 random Herdt-shaped problems, one of the `infeasible` family: found in
    round 5).  The reference does not notice: its running comparisons never skip a NaN, it adds and drops until maxit = 40 (m + n)
    and returns ifail = 1 with a NaN solution; the oracle follows it bit for bit.  A wave arg-max does not pick what those serial
    comparisons pick once NaNs compete -- the solver used to "converge" there with ifail = 0 -- so it takes NaN-exact forms of
    the two selections when the iterate is not a number (wg_ql_device.hpp: scan_nan_exact, pick_drop_serial_reference).  Held here: ifail, the NaN solution's bits, the iteration count and the whole add / drop history
    (8 877 events) against the oracle, alone and inside a batch of ordinary QPs, through the fixed-size and the generic kernel."""
    wg = _wg()
    bad = [qpgen.herdt_like(np.random.default_rng(61000 + s), 16, 2) for s in (72, 1732, 3422, 3928, 4265, 5716, 5797)]
    good = [qpgen.herdt_like(np.random.default_rng(61000 + s), 16, 2) for s in (5, 6, 7)]
    qps = [good[0]] + bad[:4] + [good[1]] + bad[4:] + [good[2]]
    for fixed in ("1", "0"):
        os.environ["WG_QL_FIXED"] = fixed
        try:
            pk = wg.pack_qps(qps)
            res = wg.qp_solve_batch(pk, hist_cap=16384)
        finally:
            os.environ.pop("WG_QL_FIXED", None)
        n_maxit = 0
        for k, q in enumerate(qps):
            o = ol.oracle_ql(_padded(pk, qps, k), hist_cap=16384)
            assert int(res["ifail"][k]) == o["ifail"], (fixed, k, int(res["ifail"][k]), o["ifail"])
            assert int(res["n_iter"][k]) == o["n_iter"] and int(res["hist_len"][k]) == o["hist_len"], (fixed, k)
            assert np.array_equal(res["hist"][k, :o["hist_len"]], o["hist"]), (fixed, k)
            assert ol.same_bits_nan_aware(res["x"][k, :q["n"]], o["x"]), (fixed, k)     # NaN where the reference has NaN
            if o["ifail"] == 1:
                n_maxit += 1
                assert np.isnan(o["x"]).all() and o["n_iter"] == 40 * (q["m"] + q["n"]) + 1
        assert n_maxit == 7
    q = qpgen.FAMILIES["infeasible"](np.random.default_rng(5282))
    pk = wg.pack_qps([q])
    res = wg.qp_solve_batch(pk, hist_cap=16384)
    o = ol.oracle_ql(_padded(pk, [q], 0), hist_cap=16384)
    assert int(res["ifail"][0]) == o["ifail"] == 1 and int(res["n_iter"][0]) == o["n_iter"]
    assert int(res["hist_len"][0]) == o["hist_len"] and np.array_equal(res["hist"][0, :o["hist_len"]], o["hist"])
    assert ol.same_bits_nan_aware(res["x"][0, :q["n"]], o["x"])


def test_fuzz_families_scaled_and_config5_sized():
    """A slice of tools/fuzz_ql.py inside the suite: every family, the magnitude-scaled variants (Hessian and constraints scaled by
    2^k, |k| up to 480: the edges of the double range) and config-5-sized problems on seeds no other test uses -- ifail, iterations, history, x (NaN-aware), u.
    The full run (322 000 QPs of seventeen families, 0 mismatches) is filed as profiles/round5_fuzz_ql.txt."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_ql", os.path.join(root, "tools", "fuzz_ql.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    wg = _wg()
    n_checked = 0
    for name, gen in fz.variants().items():
        seeds = [770000 + 104729 * k for k in range(16 if name == "config5_sized" else 120)]
        qps = [gen(np.random.default_rng(s)) for s in seeds]
        pk = wg.pack_qps(qps)
        res = wg.qp_solve_batch(pk, hist_cap=fz.HIST)
        for k, q in enumerate(qps):
            o = ol.oracle_ql(_padded(pk, qps, k), hist_cap=fz.HIST)
            n, m = q["n"], q["m"]
            assert int(res["ifail"][k]) == o["ifail"] and int(res["n_iter"][k]) == o["n_iter"], (name, seeds[k])
            assert int(res["hist_len"][k]) == o["hist_len"], (name, seeds[k])
            assert np.array_equal(res["hist"][k, :len(o["hist"])], o["hist"]), (name, seeds[k])
            assert ol.same_bits_nan_aware(res["x"][k, :n], o["x"]), (name, seeds[k])
            if o["ifail"] == 0:
                assert np.array_equal(res["iact"][k, :o["nact"]], o["iact"]) and ol.same_bits(res["u"][k, :m + 2 * n], o["u"]), (name, seeds[k])
            n_checked += 1
    assert n_checked >= 1500


def test_longest_first_start_order_is_scheduling_only(monkeypatch):
    """More QPs than resident waves: a batch that follows another one of the same size on the same arrays starts its QPs
    longest-solve-first by the previous batch's iteration counts (wg_lpt_order_kernel, as the tick does).  Scheduling only: the
    second and third call return the first call's bytes, the oracle's on a sample, and WG_QL_LPT=0 (index order) the same."""
    import torch
    wg = _wg()
    B = 2304                                                   # > 2048 resident waves (eight per CU)
    qps = [qpgen.herdt_like(np.random.default_rng(61000 + s), 16, 2) for s in range(B)]
    pk = wg.pack_qps(qps)
    assert pk["nmax"] == 36 and pk["mmax"] == 76
    dev = {k: torch.from_numpy(np.ascontiguousarray(pk[k])).cuda() for k in ("C", "d", "A", "b", "xl", "xu")}
    ints = {k: torch.from_numpy(np.ascontiguousarray(pk[k]).astype(np.int32)).cuda() for k in ("n", "m", "me")}

    def solve():
        x = torch.zeros(B, 36, dtype=torch.float64, device="cuda"); u = torch.zeros(B, 76 + 72, dtype=torch.float64, device="cuda")
        ifail = torch.full((B,), -9, dtype=torch.int32, device="cuda"); nit = torch.zeros(B, dtype=torch.int32, device="cuda")
        iact = torch.zeros(B, 36, dtype=torch.int32, device="cuda"); nact = torch.zeros(B, dtype=torch.int32, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())                 # noqa: E731
        rc = wg.lib().wg_qp_solve_batch_dev(B, 36, 76, p(ints["n"]), p(ints["m"]), p(ints["me"]), p(dev["C"]), p(dev["d"]), p(dev["A"]),
                                            p(dev["b"]), p(dev["xl"]), p(dev["xu"]), C.c_double(1e-8), p(x), p(u), p(ifail), p(nit), p(iact),
                                            p(nact), None, 0, None, None)
        assert rc == 0, wg.lib().wg_last_error()
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (x, u, ifail, nit, iact, nact)]

    first = solve()                                            # no prediction yet: index order
    second = solve()                                           # ordered by the first call's iteration counts
    third = solve()
    monkeypatch.setenv("WG_QL_LPT", "0")
    plain = solve()
    for other in (second, third, plain):
        assert all(a.tobytes() == b.tobytes() for a, b in zip(first, other))     # bytes: some of these QPs end in NaN solutions
    assert first[3].max() > first[3].min() + 5                 # there was something to order
    for k in (0, 1, B // 2, B - 1):
        o = ol.oracle_ql(_padded(pk, qps, k))
        assert int(first[2][k]) == o["ifail"] and ol.same_bits(first[0][k, :36], o["x"]) and int(first[3][k]) == o["n_iter"]


def test_two_streams_share_one_context(monkeypatch):
    """Herdt-sized QPs keep wa | b in a slot of global memory per QP (a buffer of the context): a launch that arrives on another
    stream while the slots are in use takes the placement with wa | b in LDS instead -- both batches solve to the oracle's
    bytes (here: the same problems on two streams at once give the same answers as one after the other)."""
    import torch
    wg = _wg()
    qps = [qpgen.herdt_like(np.random.default_rng(31000 + s), 16, 2) for s in range(3000)]
    pk = wg.pack_qps(qps)
    ref = wg.qp_solve_batch(pk, hist_cap=64)
    B, nmax, mmax = pk["B"], pk["nmax"], pk["mmax"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dev_in = {k: t(pk[k]) for k in ("n", "m", "me", "C", "d", "A", "b", "xl", "xu")}
    outs = []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    # the outputs exist BEFORE the launches: torch fills them on its default stream, which nothing orders against the two
    # non-blocking streams -- with the device full of the first launch's blocks a fill enqueued next to the second launch could
    # land after that launch's first blocks had written their results (seen: a few early QPs of launch 1 read back as zeros)
    for st in streams:
        x = torch.zeros(B, nmax, dtype=torch.float64, device="cuda"); u = torch.zeros(B, mmax + 2 * nmax, dtype=torch.float64, device="cuda")
        ifail = torch.full((B,), -99, dtype=torch.int32, device="cuda"); nit = torch.zeros(B, dtype=torch.int32, device="cuda")
        outs.append((x, u, ifail, nit))
    torch.cuda.synchronize()
    for st, (x, u, ifail, nit) in zip(streams, outs):
        wg.qp_solve_batch_dev(B, nmax, mmax, dev_in["n"], dev_in["m"], dev_in["me"], dev_in["C"], dev_in["d"], dev_in["A"], dev_in["b"],
                              dev_in["xl"], dev_in["xu"], 1e-8, x, u, ifail, nit, stream=st.cuda_stream)
    torch.cuda.synchronize()
    for k, (x, u, ifail, nit) in enumerate(outs):
        bad = np.nonzero((x.cpu().numpy() != ref["x"]).any(axis=1))[0]
        assert len(bad) == 0, "launch %d: %d of %d QPs differ, first %s" % (k, len(bad), B, bad[:8])
        assert np.array_equal(u.cpu().numpy(), ref["u"])
        assert np.array_equal(ifail.cpu().numpy(), ref["ifail"]) and np.array_equal(nit.cpu().numpy(), ref["n_iter"])


def test_two_host_threads_share_one_context():
    """The test "are the wa | b slots free", the launch and the event behind it are one critical section of the context: two host
    threads that submit Herdt-sized batches on two streams of ONE context at the same moment, over and over, both get the
    oracle's bytes (before, both could find the slots free and run on the same slots)."""
    import threading
    import torch
    wg = _wg()
    qps = [qpgen.herdt_like(np.random.default_rng(32000 + s), 16, 2) for s in range(2048)]
    pk = wg.pack_qps(qps)
    ref = wg.qp_solve_batch(pk, hist_cap=64)
    B, nmax, mmax = pk["B"], pk["nmax"], pk["mmax"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    dev_in = {k: t(pk[k]) for k in ("n", "m", "me", "C", "d", "A", "b", "xl", "xu")}
    rounds = 6
    outs = [[None] * rounds for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for w in range(2):
        for r in range(rounds):
            outs[w][r] = (torch.zeros(B, nmax, dtype=torch.float64, device="cuda"), torch.zeros(B, mmax + 2 * nmax, dtype=torch.float64, device="cuda"),
                          torch.full((B,), -99, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    gate = threading.Barrier(2)
    errs = []

    def worker(w):
        try:
            for r in range(rounds):
                x, u, ifail, nit = outs[w][r]
                gate.wait()
                wg.qp_solve_batch_dev(B, nmax, mmax, dev_in["n"], dev_in["m"], dev_in["me"], dev_in["C"], dev_in["d"], dev_in["A"], dev_in["b"],
                                      dev_in["xl"], dev_in["xu"], 1e-8, x, u, ifail, nit, stream=streams[w].cuda_stream)
        except Exception as e:            # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for w in range(2):
        for r in range(rounds):
            x, u, ifail, nit = outs[w][r]
            assert np.array_equal(x.cpu().numpy(), ref["x"]) and np.array_equal(u.cpu().numpy(), ref["u"]), (w, r)
            assert np.array_equal(ifail.cpu().numpy(), ref["ifail"]) and np.array_equal(nit.cpu().numpy(), ref["n_iter"]), (w, r)


@pytest.mark.parametrize("log2_scale", [0, 380, 450, -380, -450])
def test_sweep_norm_range_guard(log2_scale):
    """The sweep's norm chain takes a shorter instruction sequence when every operand is zero or within [2^-400, 2^400]
    (wg_ql_device.hpp, givens_norm_fast / sweep_range_ok) and the reference-shaped one otherwise.  Constraint rows scaled by
    2^k put the operands (Z^T a) on either side of the guard, close to it and far from it; n > 64 takes the generic sweep."""
    wg = _wg()
    qps = []
    for s_ in range(24):
        q = qpgen.random_pd(np.random.default_rng(8800 + s_), 70, 48)
        f = float(2.0 ** log2_scale)                       # exact scaling: the same feasible set
        q["A"] = np.asfortranarray(q["A"] * f); q["b"] = q["b"] * f
        qps.append(q)
    pk = wg.pack_qps(qps)
    res = wg.qp_solve_batch(pk, hist_cap=512)
    bad = _compare(qps, res, "scale=2^%d" % log2_scale, pk)
    assert not bad, bad[:5]
    assert int(np.max(res["n_iter"])) > 3                  # the solves did add constraints (sweeps ran)


def test_empty_batch_and_errors():
    wg = _wg()
    q = qpgen.random_pd(np.random.default_rng(1), 4, 3)
    pk = wg.pack_qps([q])
    pk["B"] = 0
    res = wg.qp_solve_batch(pk)
    assert res["x"].shape[0] == 0
