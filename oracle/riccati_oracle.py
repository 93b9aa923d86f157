"""CPU oracle (test infrastructure only -- never imported by the product) for the Kajita preview-control gains.

Restates, with numpy/scipy,
  * OptimalControllerSolver::ComputeWeights   /root/reference/src/PreviewControl/OptimalControllerSolver.cpp:200-352
    (symplectic pencil H, E :210-263; ordered generalised Schur with the "inside the unit circle" selector sb02ox
    :54-61, 133-198; P = Z21 Z11^-1 :283-299; K :301-318; F recursion :320-350)
  * PreviewControl::ComputeOptimalWeights     /root/reference/src/PreviewControl/PreviewControl.cpp:198-322
  * PreviewControl::OneIterationOfPreview     /root/reference/src/PreviewControl/PreviewControl.cpp:324-374

Third-party arithmetic: the reference calls LAPACK dgges_ (system LAPACK, version unpinned, CMakeLists.txt:21) and
jrl-mal's MAL_INVERSE (jrl-mal >= 1.9.0, CMakeLists.txt:45); here scipy.linalg.ordqz (bundled OpenBLAS LAPACK dgges +
dtgsen) plays that part.  Pin: src/data/PreviewControlParameters.ini (5 significant digits, committed as
tests/golden/preview_control_parameters.npz).  The reference's own TestRiccatiEquation asserts nothing.
"""
import numpy as np
import scipy.linalg as sl

MODE_WITH_INITIALPOS = 0
MODE_WITHOUT_INITIALPOS = 1


def polish(A, b, c, Q, R, P, iters=400000):
    """Extended-precision fixed-point iteration of the Riccati map from P (x87 long double).  The QZ route loses digits
    when T is small (pencil eigenvalues crowd 1); this gives the fixed point itself, to ~1e-17 relative."""
    L = np.longdouble
    n = len(P)
    A = np.asarray(A, L); b = np.asarray(b, L).reshape(n, 1); c = np.asarray(c, L).reshape(1, n)
    H = (c.T * L(Q)) @ c; P = np.asarray(P, L); R = L(R)
    for _ in range(iters):
        PA = P @ A; btPA = b.T @ PA
        Pn = A.T @ PA + H - (btPA.T @ btPA) / (R + (b.T @ P @ b)[0, 0])
        Pn = (Pn + Pn.T) / 2
        d = np.max(np.abs(Pn - P)) / np.max(np.abs(Pn)); P = Pn
        if d < 1e-19:
            break
    return P


def compute_weights(A, b, c, Q, R, Nl, mode, refine=False):
    A = np.asarray(A, float); n = A.shape[0]
    b = np.asarray(b, float).reshape(n, 1); c = np.asarray(c, float).reshape(1, n)
    H = np.eye(2 * n); H[:n, :n] = A; H[n:, :n] = -(c.T * Q) @ c
    E = np.eye(2 * n); E[:n, n:] = (b * (1 / R)) @ b.T; E[n:, n:] = A.T
    # eigenvalues alpha/beta strictly inside the unit circle first (sb02ox: |alpha| < |beta|)
    _, _, _, _, _, Z = sl.ordqz(H, E, sort="iuc", output="real")
    Z11 = Z[:n, :n]; Z21 = Z[n:, :n]
    P = Z21 @ np.linalg.inv(Z11)
    if refine:
        P = polish(A, b, c, Q, R, P).astype(float)
    la = 1 / (R + (b.T @ P @ b)[0, 0])
    K = (b.T @ (P @ A)) * la
    pre = la * b.T
    base = (A - b @ K).T
    post = c.T * Q
    if mode == MODE_WITHOUT_INITIALPOS:
        post = P @ post
    F = np.empty(Nl); rec = post
    for k in range(Nl):
        F[k] = (pre @ rec)[0, 0]
        rec = base @ rec
    return K[0].copy(), F, P


def cart_table(T, zc):
    A = np.array([[1.0, T, T * T / 2.0], [0.0, 1.0, T], [0.0, 0.0, 1.0]])
    B = np.array([T * T * T / 6.0, T * T / 2.0, T])
    C = np.array([1.0, 0.0, -zc / 9.81])
    return A, B, C


def preview_gains(T, zc, preview_time, mode, refine=False):
    """-> (Ks, Kx[3], F[Nl]) like PreviewControl::ComputeOptimalWeights leaves them in m_Ks, m_Kx, m_F."""
    A, B, C = cart_table(T, zc)
    Nl = int(preview_time / T)
    if mode == MODE_WITHOUT_INITIALPOS:
        Ax = np.zeros((4, 4)); Ax[0, 0] = 1.0; Ax[0, 1:] = C @ A; Ax[1:, 1:] = A
        bx = np.concatenate(([C @ B], B)); cx = np.array([1.0, 0.0, 0.0, 0.0])
        K, F, _ = compute_weights(Ax, bx, cx, 1.0, 1e-6, Nl, mode, refine)
        return K[0], K[1:4].copy(), F
    K, F, _ = compute_weights(A, B, C, 1.0, 1e-5, Nl, mode, refine)
    return K[0], K[0:3].copy(), F


def one_iteration_of_preview(A, B, C, Kx, Ks, F, x, y, sx, sy, px, py, lindex, simulation=True):
    """PreviewControl.cpp:324-374; x, y are 3-vectors, px/py the ZMP reference queues.  Summation order kept."""
    Nl = len(F)
    ux = -(Kx @ x) + Ks * sx
    for i in range(Nl):
        ux += F[i] * px[lindex + i]
    uy = -(Kx @ y) + Ks * sy
    for i in range(Nl):
        uy += F[i] * py[lindex + i]
    x = A @ x + ux * B
    y = A @ y + uy * B
    zx = 0.0
    for i in range(3):
        zx += C[i] * x[i]
    zy = 0.0
    for i in range(3):
        zy += C[i] * y[i]
    if simulation:
        sx += px[lindex] - zx
        sy += py[lindex] - zy
    return x, y, sx, sy, zx, zy
