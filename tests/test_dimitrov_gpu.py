"""The fused Dimitrov-2008 tick (ZMP polytopes in, LIPM state out; constraint matrices, cost vector, PLDP solve,
un-preconditioning and LIPM step inside one kernel) against oracle/pldp_oracle.c, through the C ABI:
constants against the numpy model of the reference's InitConstants (tests/dimitrov.py), ticks bit for bit."""
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


def _setup():
    wg.init(0)
    model = wg.dimitrov_defaults()
    wg.dimitrov_configure(model)
    return model, wg.dimitrov_constants(model.N)


def test_constants_match_the_numpy_model_of_initconstants():
    model, K = _setup()
    dm = dv.Dimitrov(model.N, model.T, model.com_height, model.alpha, model.beta)
    for name, want in (("iLQ", dm.iLQ), ("OptB", dm.OptB), ("OptC", dm.OptC), ("Pu", dm.Pu), ("iPu", dm.iPu), ("Px", dm.Px)):
        np.testing.assert_allclose(K[name], want, rtol=1e-9, atol=1e-12 * np.abs(want).max(), err_msg=name)
    # the LQ factor really preconditions the Hessian the reference factors (lower triangle of OptA)
    L = np.linalg.inv(K["iLQ"][:model.N, :model.N])
    assert np.allclose(np.triu(L, 1), 0.0)


def _fill(poly, p):
    A, B, centre, sim = p
    poly.nrows = len(B)
    for j in range(len(B)):
        poly.A[j][0], poly.A[j][1] = A[j]
        poly.B[j] = B[j]
        poly.similar[j] = int(sim[j])
    poly.centre[0], poly.centre[1] = centre


def test_fused_tick_bit_exact_over_gaits():
    model, K = _setup()
    N = model.N
    M = ol.pldp_setup(N, K["iPu"], K["Px"], K["Pu"])
    lib = ol.oracle()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    B, T = 20, 40
    plans = [dv.plan(np.random.default_rng(500 + g), n_steps=3 + g % 6) for g in range(B)]
    offs = [(5 * g) % 11 for g in range(B)]
    sg = (wg.DimitrovState * B)(); so = (wg.DimitrovState * B)()
    for g in range(B):
        for s in (sg[g], so[g]):
            s.starting = 1
            s.xk[0] = 0.002 * g; s.xk[4] = 0.001 * (g % 5 - 2)
    alive = np.ones(B, bool); n_solves = 0; rets = set(); max_act = 0
    for it in range(T):
        polys = (wg.ZmpPolytope * (B * N))()
        for g in range(B):
            for i, p in enumerate(dv.polys_at(plans[g], it + offs[g], N)):
                _fill(polys[g * N + i], p)
        outs = wg.dimitrov_tick_batch(polys, sg)
        for g in range(B):
            if not alive[g]:
                C.memmove(C.byref(sg[g]), C.byref(so[g]), C.sizeof(wg.DimitrovState))   # frozen: keep both sides equal
                continue
            oo = wg.DimitrovOut()
            rc = lib.wgo_dimitrov_tick(C.byref(M), dp(K["OptB"]), dp(K["OptC"]), dp(K["iLQ"]), C.c_double(model.T),
                                       C.c_double(model.Tctrl), C.c_double(model.com_height),
                                       C.byref(polys, g * N * C.sizeof(wg.ZmpPolytope)), C.byref(so[g]), C.byref(oo),
                                       C.c_int(0))
            assert outs[g].ret == rc, (it, g, outs[g].ret, rc)
            assert bytes(sg[g]) == bytes(so[g]), (it, g)
            if rc == 0:
                assert bytes(outs[g]) == bytes(oo), (it, g)
            else:
                assert (outs[g].jerk_x, outs[g].n_iter, outs[g].n_active) == (oo.jerk_x, oo.n_iter, oo.n_active)
                alive[g] = False
            n_solves += 1; rets.add(rc); max_act = max(max_act, oo.n_active)
    assert n_solves > 400 and 0 in rets and max_act >= 10
    x_end = np.array([so[g].xk[0] for g in range(B)])
    assert x_end.max() > 0.2                                  # the gaits did walk


def test_bad_polytope_and_unconfigured_calls():
    model, K = _setup()
    N = model.N
    polys = (wg.ZmpPolytope * N)()
    for i in range(N):
        _fill(polys[i], dv.box(0.0, 0.0, 0.07, 0.12))
    polys[3].similar[1] = 5
    st = (wg.DimitrovState * 1)(); st[0].starting = 1
    outs = wg.dimitrov_tick_batch(polys, st)
    assert outs[0].ret == -4
    assert wg.lib().wg_dimitrov_tick_batch(1, None, None, None, 0) != 0
    bad = wg.dimitrov_defaults(); bad.N = 40
    assert wg.lib().wg_dimitrov_configure(C.byref(bad)) != 0
