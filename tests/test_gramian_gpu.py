"""The horizon-condensing Gramian on the matrix cores (wg_gramian_batch, SURVEY 8(a) a2 / BASELINE config 5's MFMA
path) against the oracle's loop in the reference's summation order.  MFMA fuses and reorders the sums, so this is a
floating-point check with a stated tolerance: 1e-14 of the block's largest entry for v_mfma_f64, 1e-6 for the
f32-operand form (operands are rounded to float)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol  # noqa: E402

wg = importlib.import_module("jrl-walkgen_amd")
pytestmark = pytest.mark.gpu
TOL = {wg.GRAMIAN_F64: 1e-14, wg.GRAMIAN_F32: 1e-6}


def _oracle_qb(N, T, h, alpha, beta, gamma):
    lib = ol.oracle()
    m = wg.Model()
    lib.wgo_model_defaults(C.byref(m))
    m.N, m.T, m.com_height_qp, m.alpha, m.beta, m.gamma = N, T, h, alpha, beta, gamma
    Qb = np.zeros((N, N))
    assert lib.wgo_invariant_hessian(C.byref(m), Qb.ctypes.data_as(C.c_void_p)) == 0
    return Qb


@pytest.mark.parametrize("N", [16, 32, 8, 20, 1])
@pytest.mark.parametrize("prec", [wg.GRAMIAN_F64, wg.GRAMIAN_F32])
def test_gramian_matches_reference_order_loop(N, prec):
    wg.init(0)
    rng = np.random.default_rng(N)
    B = 37
    T = rng.uniform(0.02, 0.2, B); h = rng.uniform(0.5, 1.0, B)
    T[0], h[0] = 0.1, 0.814                                      # the reference's model
    alpha, beta, gamma = 1.0, 1e-5, 1e-6                         # ZMPVelocityReferencedQP.cpp:94-96 weights
    Qb = wg.gramian_batch(N, T, h, alpha, beta, gamma, prec)
    for b in range(B):
        want = _oracle_qb(N, T[b], h[b], alpha, beta, gamma)
        assert np.abs(Qb[b] - want).max() <= TOL[prec] * np.abs(want).max(), (b, N)
        assert np.array_equal(Qb[b], Qb[b].T) or np.abs(Qb[b] - Qb[b].T).max() <= TOL[prec] * np.abs(want).max()


def test_gramian_weights_enter_like_the_reference():
    """beta only on the diagonal, alpha / gamma scale the two products (generator-vel-ref.cpp:592-613)"""
    wg.init(0)
    T = np.array([0.1]); h = np.array([0.814])
    q_v = wg.gramian_batch(16, T, h, 1.0, 0.0, 0.0)[0]
    q_z = wg.gramian_batch(16, T, h, 0.0, 0.0, 1.0)[0]
    q_j = wg.gramian_batch(16, T, h, 0.0, 1.0, 0.0)[0]
    assert np.array_equal(q_j, np.eye(16))
    q = wg.gramian_batch(16, T, h, 2.0, 3.0, 5.0)[0]
    assert np.abs(q - (3.0 * q_j + 2.0 * q_v + 5.0 * q_z)).max() <= 1e-14 * np.abs(q).max()
    assert wg.lib().wg_gramian_batch(1, 33, T.ctypes.data, h.ctypes.data, 1.0, 1.0, 1.0, 0, q.ctypes.data) == -2
