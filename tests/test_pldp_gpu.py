"""GPU parity of the PLDP/OptCholesky back-end (through the C ABI) against oracle/pldp_oracle.c: bit-identical
solutions, iteration counts, active-set index sequences (activation order) and hot-start states, over whole
receding-horizon gaits with de-synchronised footstep plans, including the solves that end in the reference's
"initial solution is incorrect" exit."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dimitrov as dv  # noqa: E402
import oraclelib as ol  # noqa: E402

wg = importlib.import_module("jrl-walkgen_amd")
pytestmark = pytest.mark.gpu


def _pack(dm, probs, mcap):
    B = len(probs); n = 2 * dm.N
    m = np.array([p["m"] for p in probs], dtype=np.int32)
    D = np.stack([p["D"] for p in probs])
    A = np.zeros((B, (mcap + 1) * n)); b = np.zeros((B, mcap)); sim = np.zeros((B, mcap), dtype=np.int32)
    for i, p in enumerate(probs):
        A[i, :p["A"].size] = p["A"]; b[i, :p["m"]] = p["b"]; sim[i, :p["m"]] = p["similar"]
    z = np.stack([p["zmpref"] for p in probs]); xk = np.stack([p["xk"] for p in probs])
    return m, D, A, b, z, xk, sim


def _state_tuple(s):
    return (s.n_prev, list(s.prev_active[:s.n_prev]), list(s.prev_zmp), s.internal_time)


def _ql_gate(dm, p, X, active, gate):
    """One solved problem against the reference-pinned QL oracle (oracle/ql_oracle.c == the reference's compiled qld.cpp):
    min 1/2 |v|^2 + D'v  s.t.  A v + b >= 0  has ONE optimum v*.  PLDP is a primal active-set method that never drops a
    constraint inside a solve (PLDPSolver.cpp:654-1007), so it ends either AT v* (projected gradient gone, multipliers of the
    right sign) or on a vertex it activated on the way and could not leave -- feasible, objective above the optimum.  Checked on
    EVERY solve: feasibility, f(X) >= f(v*) (up to what PLDP's own 1e-8 slack outside a face can buy, ComputeAlpha :613-621);
    where the KKT signs hold: X == v* to 2e-5 (1e-8 of slack over constraint rows of norm ~1e-3 -- 1e-9 on x is below what
    the method's own tolerance allows) and the objectives to 1e-9 relative."""
    m, n = p["m"], 2 * dm.N
    A = p["A"].reshape((n, m + 1)).T[:m]
    q = dict(n=n, m=m, me=0, mmax=m + 1, nmax=n, C=np.asfortranarray(np.eye(n)), d=p["D"].copy(),
             A=np.asfortranarray(np.vstack([A, np.zeros((1, n))])), b=np.concatenate([p["b"], [0.0]]),
             xl=np.full(n, -1e8), xu=np.full(n, 1e8))
    o = ol.oracle_ql(q)
    assert o["ifail"] == 0
    f = lambda z: 0.5 * z @ z + p["D"] @ z  # noqa: E731
    lam_sum = float(np.abs(o["u"][:m]).sum())
    assert (A @ X + p["b"]).min() > -5e-8
    gap = f(X) - f(o["x"])
    assert gap >= -5e-8 * lam_sum - 1e-9 * max(1.0, abs(f(X))), gap
    act = np.asarray(active, dtype=int)                        # PLDP's own active set, in activation order
    if len(act):
        lam, *_ = np.linalg.lstsq(A[act].T, X + p["D"], rcond=None)
        kkt = np.abs(A[act].T @ lam - (X + p["D"])).max() < 1e-9 and (lam > -1e-12).all()
    else:
        kkt = np.abs(X + p["D"]).max() < 1e-9
    gate["solves"] += 1
    if kkt:
        assert np.abs(X - o["x"]).max() < 2e-5 and abs(gap) <= 5e-8 * lam_sum + 1e-9 * max(1.0, abs(f(X)))
        gate["optimal"] += 1
    else:
        gate["stuck"] += 1
        gate["worst_gap"] = max(gate["worst_gap"], gap / max(1e-12, abs(f(o["x"]))))
        gate["worst_dx"] = max(gate["worst_dx"], float(np.abs(X - o["x"]).max()))


def _run_lockstep(B, n_ticks, seed0, max_iter=0, mcap=wg.PLDP_MMAX, gate=None):
    dm = dv.Dimitrov()
    M = ol.pldp_setup(dm.N, dm.iPu, dm.Px, dm.Pu)
    wg.init(0)
    wg.pldp_configure(dm.N, dm.iPu, dm.Px, dm.Pu)
    plans = [dv.plan(np.random.default_rng(seed0 + g), n_steps=4 + g % 5) for g in range(B)]
    offs = [(3 * g) % 9 for g in range(B)]                  # de-synchronise the gaits in time
    xk = [np.zeros(6) for _ in range(B)]
    st_o = [ol.PldpState() for _ in range(B)]
    st_g = (wg.PldpState * B)()
    alive = np.ones(B, dtype=bool)
    n_removed = np.zeros(B, dtype=np.int32); starting = np.ones(B, dtype=np.int32)
    stats = dict(solves=0, neg_alpha=0, iters=[], nact=[])
    for it in range(n_ticks):
        probs = [dm.problem(xk[g], dv.polys_at(plans[g], it + offs[g], dm.N)) for g in range(B)]
        m, D, A, b, z, xkk, sim = _pack(dm, probs, mcap)
        out = wg.pldp_solve_batch(dm.N, mcap, m, D, A, b, z, xkk, sim, n_removed, starting, st_g, max_iter=max_iter)
        for g in range(B):
            if not alive[g]:
                continue
            p = probs[g]
            o = ol.pldp_solve(M, st_o[g], p["D"], p["m"], p["A"], p["b"], p["zmpref"], p["xk"], p["similar"],
                              int(n_removed[g]), bool(starting[g]), max_iter=max_iter)
            assert out["ret"][g] == o["ret"], (it, g)
            assert out["n_iter"][g] == o["n_iter"], (it, g)
            assert np.array_equal(out["active"][g], o["active"]), (it, g, out["active"][g], o["active"])
            assert np.array_equal(out["X"][g], o["X"]), (it, g, np.abs(out["X"][g] - o["X"]).max())
            assert _state_tuple(st_g[g]) == _state_tuple(st_o[g]), (it, g)
            stats["solves"] += 1; stats["iters"].append(o["n_iter"]); stats["nact"].append(len(o["active"]))
            if o["ret"] != 0:
                stats["neg_alpha"] += (o["ret"] == -2)
                alive[g] = False                            # the reference process would have exited here
                continue
            if gate is not None:
                _ql_gate(dm, p, out["X"][g], out["active"][g], gate)
            xk[g] = dm.step(xk[g], out["X"][g])
        n_removed = np.array([p["first_rows"] for p in probs], dtype=np.int32)
        starting[:] = 0
    return stats


@pytest.mark.parametrize("a_in_lds", ["0", "1"])
def test_pldp_gaits_bit_exact(a_in_lds, monkeypatch):
    """both placements of the constraint matrix (staged in LDS / read in place from L2, the default at m = 128)"""
    monkeypatch.setenv("WG_PLDP_A_IN_LDS", a_in_lds)
    st = _run_lockstep(B=24, n_ticks=45, seed0=100)
    assert st["solves"] > 600 and max(st["nact"]) >= 10 and max(st["iters"]) >= 6


def test_pldp_solutions_against_the_pinned_ql_oracle_on_every_solve():
    """the property gate of _ql_gate on every successful solve of 24 de-synchronised gaits x 45 ticks (GPU solutions)"""
    gate = dict(solves=0, optimal=0, stuck=0, worst_gap=0.0, worst_dx=0.0)
    _run_lockstep(B=24, n_ticks=45, seed0=100, gate=gate)
    print("PLDP vs QL:", gate)
    # measured (same numbers on the CPU restatement, whose bits the GPU reproduces): ~7 % of the solves end at the optimum,
    # the others on a vertex PLDP could not leave (objective up to 2 x the optimum's) -- the method trades optimality for
    # speed by design (Dimitrov 2008); what holds on every solve is feasibility and f(X) >= f(v*)
    assert gate["solves"] > 600 and gate["optimal"] >= 30 and gate["optimal"] + gate["stuck"] == gate["solves"], gate


def test_pldp_iteration_cap_bit_exact():
    st = _run_lockstep(B=8, n_ticks=25, seed0=300, max_iter=3)
    assert st["solves"] > 100 and max(st["iters"]) == 3


def test_pldp_small_slots_and_rejections():
    # mcap below the largest m is refused per problem, positive SimilarConstraint offsets too
    dm = dv.Dimitrov()
    wg.init(0)
    wg.pldp_configure(dm.N, dm.iPu, dm.Px, dm.Pu)
    segs = dv.plan(np.random.default_rng(5))
    p = dm.problem(np.zeros(6), dv.polys_at(segs, 0, dm.N))
    assert p["m"] == 64
    m, D, A, b, z, xk, sim = _pack(dm, [p, p], 64)
    st = (wg.PldpState * 2)()
    sim[1, 0] = 3
    out = wg.pldp_solve_batch(dm.N, 64, m, D, A, b, z, xk, sim, np.zeros(2, np.int32), np.ones(2, np.int32), st)
    assert out["ret"][0] == 0 and out["ret"][1] == -4
    M = ol.pldp_setup(dm.N, dm.iPu, dm.Px, dm.Pu)
    o = ol.pldp_solve(M, ol.PldpState(), p["D"], p["m"], p["A"], p["b"], p["zmpref"], p["xk"], p["similar"], 0, True)
    assert np.array_equal(out["X"][0], o["X"])
    m2 = m.copy(); m2[0] = 65
    out = wg.pldp_solve_batch(dm.N, 64, m2, D, A, b, z, xk, sim, np.zeros(2, np.int32), np.ones(2, np.int32), st)
    assert out["ret"][0] == -4
    lib = wg.lib()
    assert lib.wg_pldp_solve_batch(1, 500, None, None, None, None, None, None, None, None, None, 0, None, None, None, None,
                                   None, None) != 0
    assert wg.pldp_lds_bytes() <= 64 * 1024
