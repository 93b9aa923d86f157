"""Dimitrov back-end oracle (oracle/pldp_oracle.c): OptCholesky against the reference's own self-checking test
(tests/TestOptCholesky.cpp replayed with the same libc rand() stream), PLDP against the optimality conditions of the
QP it solves and against the QL oracle on the same QP.  Parity with a compiled reference is UNPINNED for these two
files (they need jrl-mal headers the image lacks), see the oracle's header."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dimitrov as dv  # noqa: E402
import oraclelib as ol  # noqa: E402


def _libc_rand_matrix(rows, cols, seed=0):
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(seed)
    RAND_MAX = 2147483647
    return np.array([[libc.rand() / RAND_MAX for _ in range(cols)] for _ in range(rows)])


def test_optcholesky_reference_self_check():
    # tests/TestOptCholesky.cpp:88-170: srand(0); A 12 x 15 uniform; rows added one by one; ||A A' - L L'||_F <= 1e-6
    A = _libc_rand_matrix(12, 15, 0)
    L = ol.optchol_rows_normal(A, list(range(12)))
    AAT = A @ A.T
    assert np.sqrt(((AAT - L @ L.T) ** 2).sum()) <= 1e-6
    assert np.allclose(L, np.linalg.cholesky(AAT), rtol=1e-10, atol=1e-12)
    # second half of the test: plain Cholesky of A A' and its inverse
    L2 = ol.chol_normal(AAT)
    assert np.sqrt(((AAT - L2 @ L2.T) ** 2).sum()) <= 1e-6
    iL = ol.chol_inverse(L2)
    assert np.allclose(iL @ L2, np.eye(12), atol=1e-9)


def test_optcholesky_fortran_layout_is_the_same_arithmetic():
    rng = np.random.default_rng(3)
    m, cu = 20, 32
    A = rng.standard_normal((m, cu))
    order = [7, 3, 19, 0, 11, 12, 5]
    Ln = ol.optchol_rows_normal(A, order)
    Af = np.zeros((m + 1) * cu)
    for r in range(m):
        for c in range(cu):
            Af[r + c * (m + 1)] = A[r, c]
    Lf = ol.optchol_rows_fortran(Af, m, cu, order)
    assert np.array_equal(Ln, Lf)


def _solve(M, st, pr, nr, starting):
    return ol.pldp_solve(M, st, pr["D"], pr["m"], pr["A"], pr["b"], pr["zmpref"], pr["xk"], pr["similar"], nr, starting)


@pytest.fixture(scope="module")
def setup():
    dm = dv.Dimitrov()
    return dm, ol.pldp_setup(dm.N, dm.iPu, dm.Px, dm.Pu)


def _rows(pr, N):
    m = pr["m"]
    return pr["A"].reshape((2 * N, m + 1)).T[:m]


def test_pldp_gaits_feasible_and_optimal_where_kkt_signs_hold(setup):
    dm, M = setup
    n_opt = 0; n_solves = 0
    for seed in range(6):
        recs = dv.run_gait(dm, M, dv.plan(np.random.default_rng(seed)), 60, _solve)
        assert len(recs) >= 5
        for k, r in enumerate(recs):
            pr, res = r["prob"], r["res"]
            if res["ret"] != 0:
                # the reference's "initial solution is incorrect" exit (PLDPSolver.cpp:833-838): only ever the last record
                assert res["ret"] == -2 and k == len(recs) - 1
                continue
            n_solves += 1
            A = _rows(pr, dm.N)
            v = res["X"]
            # primal feasibility up to the solver's own tolerance games (m_tol = 1e-8, :49, :613-621)
            assert (A @ v + pr["b"]).min() > -5e-8
            act = res["active"]
            assert len(set(act.tolist())) == len(act)
            # KKT: if the projected gradient vanished with multipliers of the right sign the point is THE optimum of
            #   min 1/2|v|^2 + D'v  s.t.  A v + b >= 0 ; cross-check with the (reference-pinned) QL oracle
            if len(act):
                E = A[act]
                lam, *_ = np.linalg.lstsq(E.T, v + pr["D"], rcond=None)
                resid = np.abs(E.T @ lam - (v + pr["D"])).max()
            else:
                lam = np.zeros(0); resid = np.abs(v + pr["D"]).max()
            if resid < 1e-9 and (lam > -1e-12).all():
                m, n = pr["m"], 2 * dm.N
                q = dict(n=n, m=m, me=0, mmax=m + 1, nmax=n, C=np.asfortranarray(np.eye(n)), d=pr["D"].copy(),
                         A=np.asfortranarray(np.vstack([A, np.zeros((1, n))])), b=np.concatenate([pr["b"], [0.0]]),
                         xl=np.full(n, -1e8), xu=np.full(n, 1e8))
                o = ol.oracle_ql(q)
                assert o["ifail"] == 0
                f = lambda z: 0.5 * z @ z + pr["D"] @ z  # noqa: E731
                # PLDP may sit up to ~m_tol outside a face (see above), which buys it sum(lam)*slack of objective
                assert abs(f(v) - f(o["x"])) <= 5e-8 * lam.sum() + 1e-9 * max(1.0, abs(f(v)))
                assert np.abs(v - o["x"]).max() < 2e-5      # 1e-8 of slack over rows of norm ~1e-3
                n_opt += 1
    assert n_solves > 100 and n_opt > 20


def test_pldp_hot_start_bookkeeping(setup):
    dm, M = setup
    recs = dv.run_gait(dm, M, dv.plan(np.random.default_rng(2)), 30, _solve)
    st = ol.PldpState()
    for k, r in enumerate(recs[:-1]):
        pr = r["prob"]
        res = ol.pldp_solve(M, st, pr["D"], pr["m"], pr["A"], pr["b"], pr["zmpref"], pr["xk"], pr["similar"],
                            r["n_removed"], r["starting"])
        assert np.array_equal(res["X"], r["res"]["X"]) and np.array_equal(res["active"], r["res"]["active"])
        kept = list(st.prev_active[:st.n_prev])
        assert set(kept) <= set(res["active"].tolist())
        # the next solve starts from the kept rows shifted by the rows of the slot that left the horizon
        nxt = recs[k + 1]
        want = [a - nxt["n_removed"] for a in kept if a - nxt["n_removed"] >= 0]
        assert nxt["res"]["active"][:len(want)].tolist() == want


def test_pldp_iteration_cap_and_bad_input(setup):
    dm, M = setup
    segs = dv.plan(np.random.default_rng(4))
    pr = dm.problem(np.zeros(6), dv.polys_at(segs, 12, dm.N))
    st = ol.PldpState()
    full = ol.pldp_solve(M, st, pr["D"], pr["m"], pr["A"], pr["b"], pr["zmpref"], pr["xk"], pr["similar"], 0, True)
    assert full["ret"] == 0 and full["n_iter"] >= 2
    st = ol.PldpState()
    capped = ol.pldp_solve(M, st, pr["D"], pr["m"], pr["A"], pr["b"], pr["zmpref"], pr["xk"], pr["similar"], 0, True,
                           max_iter=1)
    assert capped["n_iter"] == 1 and len(capped["active"]) <= 1
    bad = pr["similar"].copy(); bad[0] = 2
    st = ol.PldpState()
    assert ol.pldp_solve(M, st, pr["D"], pr["m"], pr["A"], pr["b"], pr["zmpref"], pr["xk"], bad, 0, True)["ret"] == -100
