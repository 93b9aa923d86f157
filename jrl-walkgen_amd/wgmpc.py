"""ctypes binding of include/wg_mpc.h (no compute here)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WG_LIB_PATH", os.path.join(_HERE, "lib", "libwg_mpc.so"))

SAMPLES = 20   # WG_SAMPLES_PER_TICK


class FootSample(C.Structure):          # wg_foot_sample_t
    _fields_ = [(k, C.c_double) for k in
                ("x", "y", "z", "theta", "omega", "omega2", "dx", "dy", "dz", "dtheta", "domega", "domega2",
                 "ddx", "ddy", "ddz", "ddtheta", "ddomega", "ddomega2")]


class Model(C.Structure):               # wg_model_t
    _fields_ = [("N", C.c_int), ("flags", C.c_int)] + [(k, C.c_double) for k in
                ("T", "Tctrl", "com_height_qp", "alpha", "beta", "gamma", "sole_w", "sole_h", "margin_x", "margin_y",
                 "ds_feet_distance", "hip_l_lo", "hip_l_hi", "hip_r_lo", "hip_r_hi", "hip_vmax", "hip_amax",
                 "feet_cross_max", "step_period", "ds_period", "dsss_period", "t_single", "t_double", "step_height",
                 "feet_distance")]


class GaitState(C.Structure):           # wg_gait_state_t
    _fields_ = [("clock", C.c_double), ("upper_time_limit", C.c_double), ("time_to_stop", C.c_double),
                ("tick_count", C.c_int), ("running", C.c_int), ("ending_phase", C.c_int), ("online", C.c_int),
                ("vref", C.c_double * 3),
                ("com_x", C.c_double * 3), ("com_y", C.c_double * 3), ("com_z", C.c_double),
                ("phase", C.c_int), ("foot", C.c_int), ("nb_steps_left", C.c_int), ("step_number", C.c_int),
                ("state_changed", C.c_int), ("pad0_", C.c_int),
                ("time_limit", C.c_double), ("start_time", C.c_double), ("sup_x", C.c_double), ("sup_y", C.c_double),
                ("sup_yaw", C.c_double),
                ("in_translation", C.c_int), ("in_rotation", C.c_int), ("nb_steps_after_rotation", C.c_int),
                ("rot_support_foot", C.c_int), ("post_rotation_phase", C.c_int), ("nb_steps_ssds", C.c_int),
                ("trunk_yaw", C.c_double * 3), ("trunkT_yaw", C.c_double * 3),
                ("lf", FootSample * 3), ("rf", FootSample * 3),
                ("front_com_x", C.c_double * 3), ("front_com_y", C.c_double * 3),
                ("poly_z", C.c_double * 5)]


class TickOut(C.Structure):             # wg_tick_out_t
    _fields_ = [("jerk_x", C.c_double), ("jerk_y", C.c_double),
                ("ifail", C.c_int), ("n_iter", C.c_int), ("nact", C.c_int), ("n", C.c_int), ("m", C.c_int),
                ("nb_prw_steps", C.c_int),
                ("com_x", (C.c_double * 3) * SAMPLES), ("com_y", (C.c_double * 3) * SAMPLES),
                ("com_yaw", (C.c_double * 2) * SAMPLES),
                ("zmp_x", C.c_double * SAMPLES), ("zmp_y", C.c_double * SAMPLES),
                ("lf", FootSample * SAMPLES), ("rf", FootSample * SAMPLES),
                ("lf_back", FootSample), ("rf_back", FootSample),
                ("pad_", C.c_double * 15)]          # sizeof == 7808 = 61 x 128 B (ABI 5)


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
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  If the
        # product library were loaded first and torch later, two runtimes would coexist and the second one finds no GPU;
        # with torch first the loader binds libwg_mpc.so to the runtime already in the process, and torch tensors'
        # device pointers / streams can be handed to the _dev entry points.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.wg_last_error.restype = C.c_char_p
        _lib.wg_qp_lds_bytes.restype = C.c_size_t
        qp_args = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_double] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p]
        _lib.wg_qp_solve_batch.argtypes = qp_args
        _lib.wg_qp_solve_batch_dev.argtypes = qp_args + [C.c_void_p]
        _lib.wg_mpc_tick_lds_bytes.restype = C.c_size_t
        _lib.wg_mpc_tick_lds_bytes_for.restype = C.c_size_t
        _lib.wg_mpc_tick_lds_bytes_for.argtypes = [C.c_void_p]
        _lib.wg_mpc_tick_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p]
        _lib.wg_mpc_tick_batch_dev.argtypes = _lib.wg_mpc_tick_batch.argtypes + [C.c_void_p]
        _lib.wg_mpc_set_velref_dev.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.wg_mpc_run_batch_dev.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.wg_mpc_run_sched_dev.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
        _lib.wg_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
        _lib.wg_host_free.argtypes = [C.c_void_p]
        _lib.wg_host_free.restype = None
        _lib.wg_mpc_tick_pinned.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.wg_mpc_assemble_batch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8
        _lib.wg_mpc_assemble_batch_dev.argtypes = _lib.wg_mpc_assemble_batch.argtypes + [C.c_void_p]
        _lib.wg_pldp_lds_bytes.restype = C.c_size_t
        _lib.wg_pldp_configure.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.wg_pldp_solve_batch.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_int] + [C.c_void_p] * 6
        _lib.wg_pldp_solve_batch_dev.argtypes = _lib.wg_pldp_solve_batch.argtypes + [C.c_void_p]
        _lib.wg_dimitrov_tick_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.wg_dimitrov_tick_batch_dev.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.wg_dimitrov_get_constants.argtypes = [C.c_void_p] * 6
        if hasattr(_lib, "wg_dimitrov_get_qld_constants"):
            _lib.wg_dimitrov_get_qld_constants.argtypes = [C.c_void_p] * 4
        _lib.wg_riccati_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p]
        _lib.wg_riccati_gains.argtypes = [C.c_double] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.wg_preview_configure.argtypes = [C.c_void_p, C.c_void_p]
        _lib.wg_gramian_batch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p] + [C.c_double] * 3 + [C.c_int, C.c_void_p]
        _lib.wg_gramian_batch_dev.argtypes = _lib.wg_gramian_batch.argtypes + [C.c_void_p]
        _lib.wg_preview_run_batch.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
        _lib.wg_preview_run_batch_dev.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        _lib.wg_zmpdisc_defaults.argtypes = [C.c_void_p]
        _lib.wg_zmpdisc_length.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.wg_zmpdisc_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + \
            [C.c_void_p] * 8
        _lib.wg_zmpdisc_batch_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + \
            [C.c_void_p] * 4
        _lib.wg_zmpdisc_full_batch_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + \
            [C.c_void_p] * 10
        _lib.wg_foot_constraints.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_double] * 4 + [C.c_int] + [C.c_void_p] * 3
        # contexts: NAME_ctx(ctx, args...) for every entry point NAME(args...) that keeps or stages device-side state
        _lib.wg_ctx_create.argtypes = [C.c_int, C.c_void_p]
        _lib.wg_ctx_destroy.argtypes = [C.c_void_p]
        _lib.wg_ctx_destroy.restype = None
        _lib.wg_ctx_device.argtypes = [C.c_void_p]
        _lib.wg_mpc_configure.argtypes = [C.c_void_p]
        if hasattr(_lib, "wg_set_overlap_strict"):          # absent from older experiment builds
            _lib.wg_set_overlap_strict.argtypes = [C.c_int]
            _lib.wg_overlap_serialised.argtypes = []
            _lib.wg_overlap_serialised.restype = C.c_longlong
        if hasattr(_lib, "wg_mpc_reserve"):                 # absent from older experiment builds (WG_LIB_PATH, A/B runs)
            _lib.wg_mpc_reserve.argtypes = [C.c_int]
        for name in CTX_ENTRY_POINTS:
            if not hasattr(_lib, name):
                continue
            base, fn = getattr(_lib, name), getattr(_lib, name + "_ctx")
            fn.argtypes = [C.c_void_p] + list(base.argtypes or [])
            fn.restype = base.restype
    return _lib


CTX_ENTRY_POINTS = ("wg_set_overlap_strict", "wg_overlap_serialised", "wg_qp_solve_batch", "wg_qp_solve_batch_dev", "wg_mpc_configure", "wg_mpc_reserve", "wg_mpc_tick_lds_bytes", "wg_mpc_tick_batch",
                    "wg_mpc_tick_batch_dev", "wg_mpc_run_batch_dev", "wg_mpc_run_sched_dev", "wg_mpc_set_velref_dev", "wg_mpc_tick_pinned",
                    "wg_mpc_assemble_batch", "wg_mpc_assemble_batch_dev", "wg_pldp_configure",
                    "wg_pldp_solve_batch", "wg_pldp_solve_batch_dev", "wg_dimitrov_configure", "wg_dimitrov_get_constants", "wg_dimitrov_get_qld_constants",
                    "wg_dimitrov_tick_batch", "wg_dimitrov_tick_batch_dev", "wg_preview_configure", "wg_preview_window",
                    "wg_preview_run_batch", "wg_preview_run_batch_dev", "wg_gramian_batch", "wg_gramian_batch_dev",
                    "wg_zmpdisc_batch", "wg_zmpdisc_batch_dev", "wg_zmpdisc_full_batch_dev")


class Context:
    """wg_ctx_t: one configured model with its device tables and workspaces (include/wg_mpc.h, "Contexts").  Two contexts
    may hold different models and run overlapping launches on different streams.  Only the Herdt-tick family is wrapped
    here (what the tests and bench.py need); every other NAME_ctx is reachable through `call`."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib().wg_ctx_create(int(device), C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            lib().wg_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:            # noqa: BLE001 -- interpreter shutdown
            pass

    def call(self, name, *args):
        """NAME_ctx(ctx, *args), raw"""
        return getattr(lib(), name + "_ctx")(self.handle, *args)

    def device(self):
        return int(lib().wg_ctx_device(self.handle))

    def mpc_configure(self, model):
        _check(self.call("wg_mpc_configure", C.byref(model)))

    def mpc_tick_lds_bytes(self):
        return int(self.call("wg_mpc_tick_lds_bytes"))

    def mpc_tick_batch(self, states, want_out=True, advance_calls=0, hist_cap=0):
        B = len(states)
        outs = (TickOut * B)() if want_out else None
        diag = np.zeros((B, 6), dtype=np.int32)
        hist = np.zeros((B, hist_cap), dtype=np.int32) if hist_cap else None
        hlen = np.zeros(B, dtype=np.int32) if hist_cap else None
        _check(self.call("wg_mpc_tick_batch", B, C.addressof(states), C.addressof(outs) if outs is not None else None,
                         _hp(diag), advance_calls, _hp(hist), hist_cap, _hp(hlen)))
        return outs, diag, hist, hlen

    def mpc_tick_batch_dev(self, B, states_ptr, outs_ptr=None, diag_ptr=None, advance_calls=0, hist_ptr=None, hist_cap=0,
                           hist_len_ptr=None, stream=None):
        v = lambda p: C.c_void_p(p) if p else None
        _check(self.call("wg_mpc_tick_batch_dev", B, v(states_ptr), v(outs_ptr), v(diag_ptr), advance_calls, v(hist_ptr),
                         hist_cap, v(hist_len_ptr), v(stream)))

    def mpc_run_batch_dev(self, B, states_ptr, n_ticks, advance_calls=20, outs_ptr=None, diag_ptr=None, stream=None):
        _check(self.call("wg_mpc_run_batch_dev", B, states_ptr, int(n_ticks), int(advance_calls), outs_ptr, diag_ptr, stream))

    def mpc_run_sched_dev(self, B, states_ptr, n_ticks, vref_sched_ptr, period, advance_calls=20, outs_ptr=None, diag_ptr=None,
                          stream=None):
        _check(self.call("wg_mpc_run_sched_dev", B, states_ptr, int(n_ticks), int(advance_calls), vref_sched_ptr, int(period),
                         outs_ptr, diag_ptr, stream))

    def mpc_set_velref_dev(self, B, states_ptr, vref_ptr, stream=None):
        v = lambda p: C.c_void_p(p) if p else None
        _check(self.call("wg_mpc_set_velref_dev", B, v(states_ptr), v(vref_ptr), v(stream)))


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


# ---------------------------------------------------------------------------
# Herdt-2010 tick
# ---------------------------------------------------------------------------
def model_defaults():
    m = Model()
    lib().wg_model_defaults(C.byref(m))
    return m


def gait_init(model, com0, left_xyt, right_xyt):
    s = GaitState()
    a3 = lambda v: (C.c_double * 3)(*v)
    lib().wg_gait_init(C.byref(model), C.byref(s), a3(com0), a3(left_xyt), a3(right_xyt))
    return s


def mpc_configure(model):
    _check(lib().wg_mpc_configure(C.byref(model)))


def mpc_tick_lds_bytes_for(model):
    """LDS bytes per gait the tick kernel would use for `model` (host arithmetic, no GPU needed)."""
    return int(lib().wg_mpc_tick_lds_bytes_for(C.byref(model)))


def mpc_tick_lds_bytes():
    return int(lib().wg_mpc_tick_lds_bytes())


def mpc_tick_batch(states, want_out=True, advance_calls=0, hist_cap=0):
    """Host-pointer entry point.  `states` is a ctypes array (GaitState * B), updated in place."""
    B = len(states)
    outs = (TickOut * B)() if want_out else None
    diag = np.zeros((B, 6), dtype=np.int32)
    hist = np.zeros((B, hist_cap), dtype=np.int32) if hist_cap else None
    hlen = np.zeros(B, dtype=np.int32) if hist_cap else None
    rc = lib().wg_mpc_tick_batch(B, C.addressof(states), C.addressof(outs) if outs is not None else None, _hp(diag),
                                 advance_calls, _hp(hist), hist_cap, _hp(hlen))
    _check(rc)
    return outs, diag, hist, hlen


def mpc_assemble_batch(states, advance_calls=0, nmax=None, mmax=None, model=None):
    """The QP of every gait's next tick as QPProblem::solve hands it to ql0001_ (wg_mpc_assemble_batch): a pack_qps-style
    dict (C, d, A, b, xl, xu column-major with strides nmax / mmax, n, m, me) ready for qp_solve_batch.  `states` is a ctypes
    array (GaitState * B), not modified."""
    B = len(states)
    N = model.N if model is not None else 16
    nmax = nmax or 2 * N + 8
    mmax = mmax or 1 + 4 * N + 20 + 1
    Cb = np.zeros((B, nmax * nmax)); Ab = np.zeros((B, mmax * nmax)); d = np.zeros((B, nmax)); b = np.zeros((B, mmax))
    xl = np.zeros((B, nmax)); xu = np.zeros((B, nmax)); n = np.zeros(B, dtype=np.int32); m = np.zeros(B, dtype=np.int32)
    _check(lib().wg_mpc_assemble_batch(B, C.addressof(states), int(advance_calls), nmax, mmax, _hp(Cb), _hp(d), _hp(Ab), _hp(b),
                                       _hp(xl), _hp(xu), _hp(n), _hp(m)))
    return dict(B=B, nmax=nmax, mmax=mmax, n=n, m=m, me=np.zeros(B, dtype=np.int32), C=Cb, d=d, A=Ab, b=b, xl=xl, xu=xu)


class HostMapped:
    """A block of host-mapped memory from wg_host_alloc holding one GaitState, one TickOut and the 6 diagnostic ints: what
    wg_mpc_tick_pinned works on (the one-robot path: no staging copies, no device synchronisation)."""

    def __init__(self):
        p = C.c_void_p()
        size = C.sizeof(GaitState) + C.sizeof(TickOut) + 64
        _check(lib().wg_host_alloc(C.byref(p), size))
        self.ptr = p
        self.state = GaitState.from_address(p.value)
        self.out = TickOut.from_address(p.value + C.sizeof(GaitState))
        self.diag = (C.c_int * 6).from_address(p.value + C.sizeof(GaitState) + C.sizeof(TickOut))

    def tick(self, advance_calls=0, ctx=None):
        args = (C.addressof(self.state), C.addressof(self.out), C.addressof(self.diag), int(advance_calls))
        _check(ctx.call("wg_mpc_tick_pinned", *args) if ctx is not None else lib().wg_mpc_tick_pinned(*args))

    def close(self):
        if self.ptr:
            lib().wg_host_free(self.ptr)
            self.ptr = None


def mpc_tick_batch_dev(B, states_ptr, outs_ptr=None, diag_ptr=None, advance_calls=0, hist_ptr=None, hist_cap=0,
                       hist_len_ptr=None, stream=None):
    """Device-pointer entry point; pointers are plain ints (e.g. torch tensor.data_ptr())."""
    v = lambda p: C.c_void_p(p) if p else None
    _check(lib().wg_mpc_tick_batch_dev(B, v(states_ptr), v(outs_ptr), v(diag_ptr), advance_calls, v(hist_ptr), hist_cap,
                                       v(hist_len_ptr), v(stream)))


def mpc_run_batch_dev(B, states_ptr, n_ticks, advance_calls=20, outs_ptr=None, diag_ptr=None, stream=None):
    """n_ticks ticks of every gait in one launch (device-side work queue); device pointers as integers."""
    _check(lib().wg_mpc_run_batch_dev(B, states_ptr, int(n_ticks), int(advance_calls), outs_ptr, diag_ptr, stream))


def mpc_run_sched_dev(B, states_ptr, n_ticks, vref_sched_ptr, period, advance_calls=20, outs_ptr=None, diag_ptr=None, stream=None):
    """mpc_run_batch_dev with the velocity references staged ahead: block t // period of vref_sched (ceil(n_ticks / period)
    x B x 3 doubles on the device) takes effect before tick t = 0, period, 2 period, ... of the launch."""
    _check(lib().wg_mpc_run_sched_dev(B, states_ptr, int(n_ticks), int(advance_calls), vref_sched_ptr, int(period), outs_ptr,
                                      diag_ptr, stream))


def mpc_set_velref_dev(B, states_ptr, vref_ptr, stream=None):
    v = lambda p: C.c_void_p(p) if p else None
    _check(lib().wg_mpc_set_velref_dev(B, v(states_ptr), v(vref_ptr), v(stream)))


RICCATI_WITH_INITIALPOS = 0
RICCATI_WITHOUT_INITIALPOS = 1


def riccati_solve(A, b, c, Q, R, Nl, mode):
    """OptimalControllerSolver(A,b,c,Q,R,Nl).ComputeWeights(mode) -> (K[n], F[Nl]); host-side entry point."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(n)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(n)
    K = np.zeros(n)
    F = np.zeros(max(int(Nl), 1))
    rc = lib().wg_riccati_solve(n, _hp(A), _hp(b), _hp(c), float(Q), float(R), int(Nl), int(mode), _hp(K), _hp(F))
    if rc != 0:
        raise WgError(f"wg_riccati_solve failed ({rc})")
    return K, F[:int(Nl)]


def riccati_gains(T, zc, Q, R, Nl, mode):
    """PreviewControl::ComputeOptimalWeights -> (K[4], F[Nl]); see include/wg_mpc.h for the layout of K."""
    K = np.zeros(4)
    F = np.zeros(max(int(Nl), 1))
    rc = lib().wg_riccati_gains(float(T), float(zc), float(Q), float(R), int(Nl), int(mode), _hp(K), _hp(F))
    if rc != 0:
        raise WgError(f"wg_riccati_gains failed ({rc})")
    return K, F[:int(Nl)]


PLDP_N = 16
PLDP_MMAX = 8 * PLDP_N


class PldpState(C.Structure):           # wg_pldp_state_t
    _fields_ = [("n_prev", C.c_int), ("prev_active", C.c_int * PLDP_MMAX), ("pad_", C.c_int),
                ("prev_zmp", C.c_double * (2 * PLDP_N)), ("internal_time", C.c_double)]


def pldp_configure(N, iPu, Px, Pu):
    iPu = np.ascontiguousarray(iPu, dtype=np.float64); Px = np.ascontiguousarray(Px, dtype=np.float64)
    Pu = np.ascontiguousarray(Pu, dtype=np.float64)
    _check(lib().wg_pldp_configure(int(N), _hp(iPu), _hp(Px), _hp(Pu)))


def pldp_lds_bytes():
    return int(lib().wg_pldp_lds_bytes())


def pldp_solve_batch(N, mcap, m, D, A, b, zmpref, xkyk, similar, n_removed, starting, states, max_iter=0):
    """Host-pointer batch solve.  Arrays follow include/wg_mpc.h (A: B x (mcap+1)*2N flat slots, leading dimension
    m[b]+1 inside a slot); `states` is a ctypes array of PldpState, updated in place."""
    B = len(m)
    n = 2 * N
    m = np.ascontiguousarray(m, dtype=np.int32); D = np.ascontiguousarray(D, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    zmpref = np.ascontiguousarray(zmpref, dtype=np.float64); xkyk = np.ascontiguousarray(xkyk, dtype=np.float64)
    similar = np.ascontiguousarray(similar, dtype=np.int32)
    n_removed = np.ascontiguousarray(n_removed, dtype=np.int32); starting = np.ascontiguousarray(starting, dtype=np.int32)
    assert A.size == B * (mcap + 1) * n and b.size == B * mcap and similar.size == B * mcap
    X = np.zeros((B, n)); ret = np.zeros(B, dtype=np.int32); nit = np.zeros(B, dtype=np.int32)
    act = np.zeros((B, mcap), dtype=np.int32); nact = np.zeros(B, dtype=np.int32)
    _check(lib().wg_pldp_solve_batch(B, int(mcap), _hp(m), _hp(D), _hp(A), _hp(b), _hp(zmpref), _hp(xkyk), _hp(similar),
                                     _hp(n_removed), _hp(starting), int(max_iter), C.addressof(states), _hp(X), _hp(ret),
                                     _hp(nit), _hp(act), _hp(nact)))
    return dict(X=X, ret=ret, n_iter=nit, active=[act[i, :nact[i]].copy() for i in range(B)])


POLY_MAX_ROWS = 8


class DimitrovModel(C.Structure):       # wg_dimitrov_model_t
    _fields_ = [("N", C.c_int), ("solver", C.c_int), ("T", C.c_double), ("Tctrl", C.c_double), ("com_height", C.c_double),
                ("alpha", C.c_double), ("beta", C.c_double)]       # solver: 0 = PLDP (the reference's default), 1 = QLD


class ZmpPolytope(C.Structure):         # wg_zmp_polytope_t
    _fields_ = [("nrows", C.c_int), ("pad_", C.c_int), ("similar", C.c_int * POLY_MAX_ROWS),
                ("A", (C.c_double * 2) * POLY_MAX_ROWS), ("B", C.c_double * POLY_MAX_ROWS), ("centre", C.c_double * 2)]


class DimitrovState(C.Structure):       # wg_dimitrov_state_t
    _fields_ = [("xk", C.c_double * 6), ("pldp", PldpState), ("n_removed", C.c_int), ("starting", C.c_int)]


class DimitrovOut(C.Structure):         # wg_dimitrov_out_t
    _fields_ = [("jerk_x", C.c_double), ("jerk_y", C.c_double), ("ret", C.c_int), ("n_iter", C.c_int),
                ("n_active", C.c_int), ("m", C.c_int),
                ("com_x", (C.c_double * 3) * 21), ("com_y", (C.c_double * 3) * 21),
                ("zmp_x", C.c_double * 21), ("zmp_y", C.c_double * 21), ("X", C.c_double * 32)]


def dimitrov_defaults():
    m = DimitrovModel()
    lib().wg_dimitrov_defaults(C.byref(m))
    return m


def dimitrov_configure(model):
    _check(lib().wg_dimitrov_configure(C.byref(model)))


def dimitrov_constants(N):
    n = 2 * N
    iLQ = np.zeros((n, n)); OptB = np.zeros((n, 6)); OptC = np.zeros((n, n))
    Pu = np.zeros((N, N)); iPu = np.zeros((N, N)); Px = np.zeros((N, 3))
    _check(lib().wg_dimitrov_get_constants(_hp(iLQ), _hp(OptB), _hp(OptC), _hp(Pu), _hp(iPu), _hp(Px)))
    return dict(iLQ=iLQ, OptB=OptB, OptC=OptC, Pu=Pu, iPu=iPu, Px=Px)


def dimitrov_qld_constants(N):
    """what mode QLD hands to ql0001_ and builds its cost vector from: Q (column-major 2N x 2N), OptB / OptC as built, Pu'"""
    n = 2 * N
    Q = np.zeros((n, n)); OptB = np.zeros((n, 6)); OptC = np.zeros((n, n)); PuT = np.zeros((N, N))
    _check(lib().wg_dimitrov_get_qld_constants(_hp(Q), _hp(OptB), _hp(OptC), _hp(PuT)))
    return dict(Q=Q, OptB=OptB, OptC=OptC, PuT=PuT)


def dimitrov_tick_batch(polys, states, want_out=True, max_iter=0):
    """polys: ctypes array (ZmpPolytope * (B*N)), states: (DimitrovState * B), updated in place."""
    B = len(states)
    outs = (DimitrovOut * B)() if want_out else None
    _check(lib().wg_dimitrov_tick_batch(B, C.addressof(polys), C.addressof(states),
                                        C.addressof(outs) if outs is not None else None, int(max_iter)))
    return outs


# ---- Kajita stage-1 preview control (PreviewControl::OneIterationOfPreview, batched) ----
class PreviewGains(C.Structure):        # wg_preview_gains_t
    _fields_ = [("T", C.c_double), ("zc", C.c_double), ("Ks", C.c_double), ("Kx", C.c_double * 3), ("nl", C.c_int),
                ("pad_", C.c_int)]


def preview_gains(T, zc, preview_time, mode=RICCATI_WITHOUT_INITIALPOS):
    """PreviewControl::ComputeOptimalWeights (PreviewControl.cpp:198-322): (PreviewGains, F[nl])."""
    nl = int(preview_time / T)
    R = 1e-6 if mode == RICCATI_WITHOUT_INITIALPOS else 1e-5
    K, F = riccati_gains(T, zc, 1.0, R, nl, mode)
    g = PreviewGains()
    g.T, g.zc, g.nl = T, zc, nl
    if mode == RICCATI_WITHOUT_INITIALPOS:
        g.Ks = K[0]; g.Kx[0], g.Kx[1], g.Kx[2] = K[1], K[2], K[3]
    else:
        g.Ks = K[0]; g.Kx[0], g.Kx[1], g.Kx[2] = K[0], K[1], K[2]
    return g, np.ascontiguousarray(F)


def preview_configure(gains, F):
    F = np.ascontiguousarray(F, dtype=np.float64)
    assert F.shape == (gains.nl,)
    _check(lib().wg_preview_configure(C.byref(gains), _hp(F)))


def preview_run_batch(zmp_x, zmp_y, state, L, simulation=True, want_com=True, want_zmp=True):
    """Host-pointer entry point: zmp_x/zmp_y [B, L+nl-1], state [B, 8] (advanced in place).  Returns (com, zmp2)."""
    zmp_x = np.ascontiguousarray(zmp_x, dtype=np.float64); zmp_y = np.ascontiguousarray(zmp_y, dtype=np.float64)
    B = zmp_x.shape[0]
    nl = int(lib().wg_preview_window())
    assert zmp_x.shape == zmp_y.shape == (B, L + nl - 1) and state.shape == (B, 8) and state.flags.c_contiguous
    com = np.zeros((B, L, 6)) if want_com else None
    z2 = np.zeros((B, L, 2)) if want_zmp else None
    _check(lib().wg_preview_run_batch(B, L, _hp(zmp_x), _hp(zmp_y), _hp(state), _hp(com), _hp(z2), int(bool(simulation))))
    return com, z2


def preview_run_batch_dev(B, L, zx_tm_ptr, zy_tm_ptr, state_ptr, com_tm_ptr=None, zmp2_tm_ptr=None, simulation=True,
                          stream=None):
    _check(lib().wg_preview_run_batch_dev(B, L, zx_tm_ptr, zy_tm_ptr, state_ptr, com_tm_ptr, zmp2_tm_ptr,
                                          int(bool(simulation)), stream))


# ---- Kajita stage-1 inputs: ZMPDiscretization, batched; FootConstraintsAsLinearSystem (host) ----
ZMPDISC_MAX_STEPS = 64


class ZmpDiscModel(C.Structure):        # wg_zmpdisc_model_t
    _fields_ = [(k, C.c_double) for k in ("T", "preview_time", "t_single", "t_double", "step_height", "omega",
                                          "modulation")] + \
               [("zmp_neutral", C.c_double * 2), ("zmp_shift", C.c_double * 4)] + \
               [(k, C.c_double) for k in ("foot_b", "foot_h", "foot_f")]


class RelStep(C.Structure):             # wg_rel_step_t
    _fields_ = [(k, C.c_double) for k in ("sx", "sy", "theta", "ss_time", "ds_time")] + \
               [("step_type", C.c_int), ("pad_", C.c_int)]


def zmpdisc_defaults():
    m = ZmpDiscModel()
    lib().wg_zmpdisc_defaults(C.byref(m))
    return m


def rel_steps(triples, ss_time, ds_time, step_type=1):
    """(RelStep * S) from [S, 3] (sx, sy, theta in degrees), as StepStackHandler leaves a ":stepseq" (walk mode 0)."""
    t = np.asarray(triples, dtype=np.float64).reshape(-1, 3)
    steps = (RelStep * len(t))()
    for i, (sx, sy, th) in enumerate(t):
        steps[i] = RelStep(sx, sy, th, ss_time, ds_time, step_type, 0)
    return steps


def zmpdisc_length(model, steps, n_steps=None):
    return int(lib().wg_zmpdisc_length(C.byref(model), C.addressof(steps), len(steps) if n_steps is None else int(n_steps)))


def zmpdisc_batch(model, steps, n_steps, init_feet, smax, lcap, want_feet=True):
    """steps: (RelStep * (B*smax)); n_steps [B]; init_feet [B, 6].  Returns a dict of arrays and `length` [B]."""
    n_steps = np.ascontiguousarray(n_steps, dtype=np.int32); B = n_steps.shape[0]
    init_feet = np.ascontiguousarray(init_feet, dtype=np.float64)
    assert init_feet.shape == (B, 6) and len(steps) == B * smax
    r = dict(zmp=np.zeros((B, lcap, 2)), zmp_theta=np.zeros((B, lcap)), zmp_type=np.zeros((B, lcap), np.int32),
             length=np.zeros(B, np.int32))
    if want_feet:
        r.update(left=np.zeros((B, lcap, 6)), right=np.zeros((B, lcap, 6)), left_type=np.zeros((B, lcap), np.int32),
                 right_type=np.zeros((B, lcap), np.int32))
    g = lambda k: _hp(r[k]) if k in r else None  # noqa: E731
    _check(lib().wg_zmpdisc_batch(C.byref(model), B, int(smax), C.addressof(steps), _hp(n_steps), _hp(init_feet), int(lcap),
                                  g("zmp"), g("zmp_theta"), g("zmp_type"), g("left"), g("left_type"), g("right"),
                                  g("right_type"), _hp(r["length"])))
    return r


def zmpdisc_batch_dev(model, B, smax, steps_ptr, n_steps_ptr, init_feet_ptr, lcap, zx_tm_ptr, zy_tm_ptr, length_ptr,
                      stream=None):
    _check(lib().wg_zmpdisc_batch_dev(C.byref(model), int(B), int(smax), steps_ptr, n_steps_ptr, init_feet_ptr, int(lcap),
                                      zx_tm_ptr, zy_tm_ptr, length_ptr, stream))


def foot_constraints(time, left, left_type, right, sole_w, sole_h, constraint_x, constraint_y, cap=256):
    """wg_foot_constraints: (polys (ZmpPolytope * n), t_start[n], t_end[n])."""
    time = np.ascontiguousarray(time, dtype=np.float64); left = np.ascontiguousarray(left, dtype=np.float64)
    right = np.ascontiguousarray(right, dtype=np.float64); left_type = np.ascontiguousarray(left_type, dtype=np.int32)
    n = time.shape[0]
    assert left.shape == right.shape == (n, 6) and left_type.shape == (n,)
    polys = (ZmpPolytope * cap)(); ts = np.zeros(cap); te = np.zeros(cap)
    k = lib().wg_foot_constraints(n, _hp(time), _hp(left), _hp(left_type), _hp(right), float(sole_w), float(sole_h),
                                  float(constraint_x), float(constraint_y), cap, C.addressof(polys), _hp(ts), _hp(te))
    _check(min(k, 0))
    if k > cap:
        raise WgError("foot_constraints: %d polytopes, capacity %d" % (k, cap))
    return polys, ts[:k], te[:k], k


# ---- invariant Hessian block on the matrix cores ----
GRAMIAN_F64 = 0
GRAMIAN_F32 = 1


def gramian_batch(N, T, h, alpha, beta, gamma, precision=GRAMIAN_F64):
    """Q_b[B, N, N] for models with sampling periods T[B] and CoM heights h[B] (wg_gramian_batch)."""
    T = np.ascontiguousarray(T, dtype=np.float64); h = np.ascontiguousarray(h, dtype=np.float64)
    B = T.shape[0]
    Qb = np.zeros((B, N, N))
    _check(lib().wg_gramian_batch(B, int(N), _hp(T), _hp(h), alpha, beta, gamma, int(precision), _hp(Qb)))
    return Qb
