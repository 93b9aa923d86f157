"""ctypes access to the CPU checkers (test infrastructure):

  * `oracle/libwg_oracle.so`     -- this repo's C restatement (oracle/*.c)
  * `oracle/_ref/libqld_ref.so`  -- the reference's own qld.cpp, compiled by
                                    oracle/Makefile (present where the
                                    reference tree was available at build time)
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libwg_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libqld_ref.so")
REF_SYM = "_Z7ql0001_PiS_S_S_S_S_PdS0_S0_S0_S0_S0_S0_S0_S_S_S_S0_S_S_S_S0_"

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def build_oracle():
    """(Re)build the oracle .so if a source is newer than it."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libwg_oracle.so"])
    return ORACLE_SO


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = C.CDLL(build_oracle())
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
    return _ref


def oracle_ql(q, eps=1e-8, hist_cap=4096):
    """Run the oracle's QL restatement on a qpgen dict."""
    n, m = q["n"], q["m"]
    x = np.zeros(n)
    u = np.zeros(m + 2 * n)
    ifail = C.c_int(-99)
    iact = np.zeros(n, dtype=np.int32)
    nact = C.c_int(0)
    nit = C.c_int(0)
    hist = np.zeros(hist_cap, dtype=np.int32)
    hlen = C.c_int(0)
    oracle().wgo_ql_solve(C.c_int(m), C.c_int(q["me"]), C.c_int(q["mmax"]), C.c_int(n), C.c_int(q["nmax"]),
                          _d(q["C"]), _d(q["d"]), _d(q["A"]), _d(q["b"]), _d(q["xl"]), _d(q["xu"]),
                          C.c_double(eps), _d(x), _d(u), C.byref(ifail), _i(iact), C.byref(nact),
                          C.byref(nit), _i(hist), C.c_int(hist_cap), C.byref(hlen))
    return dict(x=x, u=u, ifail=ifail.value, iact=iact[:nact.value].copy(), nact=nact.value,
                n_iter=nit.value, hist=hist[:min(hlen.value, hist_cap)].copy(), hist_len=hlen.value)


def ref_ql(q, eps=1e-8, lib=None):
    """Run the compiled reference ql0001_ (argument order of qld.hh:27-31;
    workspace sizes as qp-problem.cpp:256-263).  Not re-entrant (static locals)."""
    n, m = q["n"], q["m"]
    Cc = q["C"].copy(order="F")
    x = np.zeros(n)
    mnn = m + 2 * n
    u = np.zeros(mnn)
    lwar = 2 * (3 * n * n // 2 + 10 * n + 2 * (m + 1) + 20000)
    war = np.zeros(lwar)
    liwar = 2 * n + 1000
    iwar = np.zeros(liwar, dtype=np.int32)
    iwar[0] = 1
    ci = lambda v: C.byref(C.c_int(v))
    ifail = C.c_int(-99)
    fn = getattr(lib if lib is not None else ref(), REF_SYM)
    d = q["d"].copy()
    A = q["A"].copy(order="F")
    b = q["b"].copy()
    xl = q["xl"].copy()
    xu = q["xu"].copy()
    fn(ci(m), ci(q["me"]), ci(q["mmax"]), ci(n), ci(q["nmax"]), ci(mnn),
       _d(Cc), _d(d), _d(A), _d(b), _d(xl), _d(xu), _d(x), _d(u), ci(0), C.byref(ifail), ci(0),
       _d(war), ci(lwar), _i(iwar), ci(liwar), C.byref(C.c_double(eps)))
    # the final ordered active set is left in iwar[0..nact); nact itself is a
    # static local, so recover it from the non-zero multipliers' count bound
    return dict(x=x, u=u, ifail=ifail.value, iwar=iwar[:n].copy(), C_after=Cc)


# ---- add/drop history of the REFERENCE itself -------------------------------------------------------------------
# ql0002 keeps nact / iact in static locals and returns only the final ordered set, so the history needs a
# throw-away build of the reference's qld.cpp with one log call at the add site (qld.cpp:1766, `iact[*nact] = knext;`)
# and one at the drop site (qld.cpp:1903, label L800).  The patched text exists only in a temp dir outside the repo and
# is deleted with it; nothing of it is committed or travels.  Event codes as oracle/ql_oracle.c:log_event:
# +k = constraint k (1-based QL code) added, -k = dropped.
REF_QLD_SRC = "/root/reference/src/Mathematics/qld.cpp"
HIST_CAP = 1 << 16
_LOGGER = """
extern "C" {
__attribute__((visibility("default"))) int wg_hist_len = 0;
__attribute__((visibility("default"))) int wg_hist_buf[%d];
}
static inline void wg_hist_log(int code) { if (wg_hist_len < %d) wg_hist_buf[wg_hist_len] = code; ++wg_hist_len; }
""" % (HIST_CAP, HIST_CAP)
_ref_hist = None
_ref_hist_dir = None


def have_ref_source():
    return os.path.exists(REF_QLD_SRC)


def ref_hist():
    """The reference's qld.cpp with the two log calls, built -O3 -DNDEBUG (oracle/Makefile's flags) in a temp dir."""
    global _ref_hist, _ref_hist_dir
    if _ref_hist is None:
        import atexit
        import re
        import shutil
        import tempfile
        lines = open(REF_QLD_SRC).read().split("\n")
        add = [i for i, l in enumerate(lines) if re.match(r"^\s*iact\[\*nact\] = knext;\s*$", l)]
        drop = [i for i, l in enumerate(lines) if re.match(r"^L800:\s*$", l)]
        assert len(add) == 1 and len(drop) == 1, (add, drop)
        lines[add[0]] += " wg_hist_log(knext);"
        lines[drop[0]] += " wg_hist_log(-iact[kdrop]);"
        _ref_hist_dir = tempfile.mkdtemp(prefix="wg_qld_hist_")
        atexit.register(shutil.rmtree, _ref_hist_dir, True)
        open(os.path.join(_ref_hist_dir, "logger.h"), "w").write(_LOGGER)
        open(os.path.join(_ref_hist_dir, "qld_hist.cpp"), "w").write("\n".join(lines))
        so = os.path.join(_ref_hist_dir, "libqld_hist.so")
        subprocess.check_call(["g++", "-O3", "-DNDEBUG", "-fPIC", "-shared", "-include",
                               os.path.join(_ref_hist_dir, "logger.h"), os.path.join(_ref_hist_dir, "qld_hist.cpp"),
                               "-o", so])
        _ref_hist = C.CDLL(so)
    return _ref_hist


def ref_ql_hist(q, eps=1e-8):
    """ref_ql on the instrumented build: the same outputs plus the reference's own add/drop history."""
    lib = ref_hist()
    hlen = C.c_int.in_dll(lib, "wg_hist_len")
    hlen.value = 0
    r = ref_ql(q, eps, lib=lib)
    assert hlen.value <= HIST_CAP
    buf = (C.c_int * HIST_CAP).in_dll(lib, "wg_hist_buf")
    r["hist"] = np.array(buf[:hlen.value], dtype=np.int32)
    return r


def same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def same_bits_nan_aware(a, b):
    """same_bits, except that a NaN matches a NaN whatever its sign / payload bits: an invalid operation gives the negative default
    NaN on x86-64 SSE (0xFFF8...) and the positive one on gfx950 (0x7FF8...); which of the two is not part of any contract."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a.view(np.uint64)[~na], b.view(np.uint64)[~nb]))


# ---- Dimitrov back-end (oracle/pldp_oracle.c) ------------------------------------------------------------------

PLDP_N = 16
PLDP_MMAX = 8 * PLDP_N


class PldpState(C.Structure):            # wg_pldp_state_t (include/wg_mpc.h)
    _fields_ = [("n_prev", C.c_int), ("prev_active", C.c_int * PLDP_MMAX), ("pad_", C.c_int),
                ("prev_zmp", C.c_double * (2 * PLDP_N)), ("internal_time", C.c_double)]


class PldpModel(C.Structure):            # wgo_pldp_model_t (oracle/wg_oracle.h)
    _fields_ = [("N", C.c_int), ("pad_", C.c_int), ("iPu", C.c_double * (PLDP_N * PLDP_N)),
                ("Px", C.c_double * (PLDP_N * 3)), ("Pu", C.c_double * (PLDP_N * PLDP_N)),
                ("iPuPx", C.c_double * (2 * PLDP_N * 6))]


def chol_normal(A):
    """OptCholesky::ComputeNormalCholeskyOnANormal -> L (lower, row-major)"""
    A = np.ascontiguousarray(A, dtype=np.float64); n = A.shape[0]
    L = np.zeros((n, n))
    oracle().wgo_chol_normal(_d(A), C.c_int(n), _d(L))
    return L


def chol_inverse(L):
    L = np.ascontiguousarray(L, dtype=np.float64); n = L.shape[0]
    iL = np.zeros((n, n))
    oracle().wgo_chol_inverse(_d(L), C.c_int(n), C.c_int(n), _d(iL))
    return iL


def optchol_rows_normal(A, order):
    """OptCholesky in MODE_NORMAL: AddActiveConstraint(order[0]), ... -> L (len(order) x len(order))"""
    A = np.ascontiguousarray(A, dtype=np.float64)
    k = len(order)
    L = np.zeros((k, k))
    st = np.asarray(order, dtype=np.int32)
    for i in range(1, k + 1):
        oracle().wgo_optchol_update_normal(_d(A), C.c_int(A.shape[1]), _i(st), C.c_int(i), _d(L), C.c_int(k))
    return L


def optchol_rows_fortran(A_colmajor_ld, m, card_u, order):
    """OptCholesky in MODE_FORTRAN on a column-major array with leading dimension m+1"""
    k = len(order)
    L = np.zeros((k, k))
    st = np.asarray(order, dtype=np.int32)
    for i in range(1, k + 1):
        oracle().wgo_optchol_update_fortran(_d(A_colmajor_ld), C.c_int(m), C.c_int(card_u), _i(st), C.c_int(i), _d(L),
                                            C.c_int(k))
    return L


def pldp_setup(N, iPu, Px, Pu):
    M = PldpModel()
    rc = oracle().wgo_pldp_setup(C.byref(M), C.c_int(N), _d(np.ascontiguousarray(iPu, dtype=np.float64)),
                                 _d(np.ascontiguousarray(Px, dtype=np.float64)),
                                 _d(np.ascontiguousarray(Pu, dtype=np.float64)))
    assert rc == 0
    return M


def pldp_solve(M, st, D, m, A, b, zmpref, xkyk, similar, n_removed, starting, max_iter=0):
    """one PLDPSolver::SolveProblem on the oracle; A is the flat column-major (m+1) x 2N array"""
    n = 2 * M.N
    X = np.zeros(n)
    act = np.zeros(PLDP_MMAX, dtype=np.int32)
    nit = C.c_int(0); nact = C.c_int(0)
    sim = np.ascontiguousarray(similar, dtype=np.int32)
    rc = oracle().wgo_pldp_solve(C.byref(M), C.byref(st), _d(D), C.c_int(m), _d(A), _d(b), _d(zmpref), _d(xkyk), _i(sim),
                                 C.c_int(n_removed), C.c_int(1 if starting else 0), C.c_int(max_iter), _d(X),
                                 C.byref(nit), _i(act), C.byref(nact))
    return dict(ret=rc, X=X, n_iter=nit.value, active=act[:nact.value].copy())


# ---- ZMPDiscretization / FootConstraintsAsLinearSystem restatement (oracle/zmpdisc_oracle.c) ----
def zmpdisc(model, steps, init_feet, n_steps=None, lib=None):
    """wgo_zmpdisc for one gait: dict(zmp [L,2], zmp_theta, zmp_type, left [L,6], left_type, right, right_type, time)."""
    lib = lib or oracle()
    S = len(steps) if n_steps is None else int(n_steps)
    lib.wgo_zmpdisc_length.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L = lib.wgo_zmpdisc_length(C.byref(model), C.addressof(steps), S)
    if L < 0:
        return dict(length=L)
    r = dict(zmp=np.zeros((L, 2)), zmp_theta=np.zeros(L), zmp_type=np.zeros(L, np.int32), left=np.zeros((L, 6)),
             left_type=np.zeros(L, np.int32), right=np.zeros((L, 6)), right_type=np.zeros(L, np.int32), time=np.zeros(L))
    f = np.ascontiguousarray(init_feet, dtype=np.float64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    lib.wgo_zmpdisc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 8
    r["length"] = lib.wgo_zmpdisc(C.byref(model), C.addressof(steps), S, vp(f), L, vp(r["zmp"]), vp(r["zmp_theta"]),
                                  vp(r["zmp_type"]), vp(r["left"]), vp(r["left_type"]), vp(r["right"]), vp(r["right_type"]),
                                  vp(r["time"]))
    return r


def foot_constraints(poly_type, time, left, left_type, right, sole_w, sole_h, cx, cy, cap=256):
    """wgo_foot_constraints: (polys (poly_type * cap), t_start, t_end, count)."""
    lib = oracle()
    time = np.ascontiguousarray(time, dtype=np.float64); left = np.ascontiguousarray(left, dtype=np.float64)
    right = np.ascontiguousarray(right, dtype=np.float64); left_type = np.ascontiguousarray(left_type, dtype=np.int32)
    polys = (poly_type * cap)(); ts = np.zeros(cap); te = np.zeros(cap)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    lib.wgo_foot_constraints.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_double] * 4 + [C.c_int] + [C.c_void_p] * 3
    k = lib.wgo_foot_constraints(time.shape[0], vp(time), vp(left), vp(left_type), vp(right), sole_w, sole_h, cx, cy, cap,
                                 C.addressof(polys), vp(ts), vp(te))
    return polys, ts[:max(k, 0)], te[:max(k, 0)], k


def circle_steps(step_type, ss, ds, lib=None):
    """TurningOnTheCircle's step stack (":supportfoot 1", ":arc 0.0 0.75 30.0 -1", ":lastsupport") from the oracle's
    restatement of StepStackHandler's generators: (steps array, count)"""
    lib = lib or oracle()
    steps = (step_type * 64)(); n = C.c_int(0); keep = C.c_int(0)
    lib.wgo_steps_arc.argtypes = [C.c_double] * 3 + [C.c_int] + [C.c_double] * 2 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.wgo_steps_support_foot.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    lib.wgo_steps_last_support.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    assert lib.wgo_steps_support_foot(1, ss, ds, C.addressof(steps), C.byref(n), 64) == 0
    assert lib.wgo_steps_arc(0.0, 0.75, 30.0, -1, ss, ds, C.addressof(steps), C.byref(n), 64, C.byref(keep)) == 0
    assert lib.wgo_steps_last_support(keep.value, ss, ds, C.addressof(steps), C.byref(n), 64) == 0
    return steps, n.value
