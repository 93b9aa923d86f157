"""ctypes binding of include/wg_mpc.h (no compute here)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwg_mpc.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lib = None


class WgError(RuntimeError):
    pass


def lib():
    """Load libwg_mpc.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WgError(f"{LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(LIB_PATH)
        _lib.wg_last_error.restype = C.c_char_p
        _lib.wg_qp_lds_bytes.restype = C.c_size_t
        qp_args = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_double] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p]
        _lib.wg_qp_solve_batch.argtypes = qp_args
        _lib.wg_qp_solve_batch_dev.argtypes = qp_args + [C.c_void_p]
    return _lib


def _check(rc):
    if rc != 0:
        raise WgError(f"wg error {rc}: {lib().wg_last_error().decode()}")


def init(device=0):
    _check(lib().wg_init(int(device)))


def shutdown():
    lib().wg_shutdown()


def qp_lds_bytes(n, m):
    return int(lib().wg_qp_lds_bytes(int(n), int(m)))


def _hp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_qps(qps):
    """Pad a list of qpgen-style dicts into the strided batch layout of wg_qp_solve_batch."""
    B = len(qps)
    nmax = max(q["n"] for q in qps)
    mmax = max(q["mmax"] for q in qps)
    Cb = np.zeros((B, nmax * nmax))
    Ab = np.zeros((B, mmax * nmax))
    d = np.zeros((B, nmax)); xl = np.zeros((B, nmax)); xu = np.zeros((B, nmax))
    b = np.zeros((B, mmax))
    n = np.zeros(B, dtype=np.int32); m = np.zeros(B, dtype=np.int32); me = np.zeros(B, dtype=np.int32)
    for k, q in enumerate(qps):
        nn, mm = q["n"], q["m"]
        Cf = np.zeros((nmax, nmax), order="F"); Cf[:nn, :nn] = q["C"][:nn, :nn]
        Af = np.zeros((mmax, nmax), order="F"); Af[:mm, :nn] = q["A"][:mm, :nn]
        Cb[k] = Cf.ravel(order="F"); Ab[k] = Af.ravel(order="F")
        d[k, :nn] = q["d"]; xl[k, :nn] = q["xl"]; xu[k, :nn] = q["xu"]; b[k, :mm] = q["b"][:mm]
        n[k], m[k], me[k] = nn, mm, q["me"]
    return dict(B=B, nmax=nmax, mmax=mmax, n=n, m=m, me=me, C=Cb, d=d, A=Ab, b=b, xl=xl, xu=xu)


def qp_solve_batch(pk, eps=1e-8, hist_cap=256):
    """Host-pointer entry point (wg_qp_solve_batch): numpy in, numpy out."""
    B, nmax, mmax = pk["B"], pk["nmax"], pk["mmax"]
    x = np.zeros((B, nmax)); u = np.zeros((B, mmax + 2 * nmax))
    ifail = np.full(B, -99, dtype=np.int32); n_iter = np.zeros(B, dtype=np.int32)
    iact = np.zeros((B, nmax), dtype=np.int32); nact = np.zeros(B, dtype=np.int32)
    hist = np.zeros((B, hist_cap), dtype=np.int32); hist_len = np.zeros(B, dtype=np.int32)
    rc = lib().wg_qp_solve_batch(B, nmax, mmax, _hp(pk["n"]), _hp(pk["m"]), _hp(pk["me"]), _hp(pk["C"]), _hp(pk["d"]),
                                 _hp(pk["A"]), _hp(pk["b"]), _hp(pk["xl"]), _hp(pk["xu"]), eps, _hp(x), _hp(u),
                                 _hp(ifail), _hp(n_iter), _hp(iact), _hp(nact), _hp(hist), hist_cap, _hp(hist_len))
    _check(rc)
    return dict(x=x, u=u, ifail=ifail, n_iter=n_iter, iact=iact, nact=nact, hist=hist, hist_len=hist_len)


def qp_solve_batch_dev(B, nmax, mmax, n, m, me, Cd, d, A, b, xl, xu, eps, x, u, ifail, n_iter=None, iact=None,
                       nact=None, hist=None, hist_cap=0, hist_len=None, stream=None):
    """Device-pointer entry point (wg_qp_solve_batch_dev).  Arguments are torch CUDA tensors or None."""
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rc = lib().wg_qp_solve_batch_dev(B, nmax, mmax, p(n), p(m), p(me), p(Cd), p(d), p(A), p(b), p(xl), p(xu), eps,
                                     p(x), p(u), p(ifail), p(n_iter), p(iact), p(nact), p(hist), hist_cap,
                                     p(hist_len), C.c_void_p(stream) if stream else None)
    _check(rc)
