"""Diagnostic for BASELINE config 5 ("N = 32 ... fp32 (MFMA Gramian condensation path)"): what the fp32 matrix-core Gramian
does to the QP.  Q_b from wg_gramian_batch in both precisions against the reference-order fp64 loop: entry error, smallest
eigenvalue, Cholesky, and the effect on an unconstrained solve Q x = -d."""
import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
for N in (16, 32):
    T, h, alpha, beta, gamma = 0.1, 0.814, 1.0, 1e-5, 1e-6
    i = np.arange(N)[:, None]; j = np.arange(N)[None, :]; low = j <= i
    Uv = np.where(low, (2 * (i - j) + 1) * T * T / 2, 0.0)
    Uz = np.where(low, (1 + 3 * (i - j) + 3 * (i - j) ** 2) * T ** 3 / 6 - T * h / 9.81, 0.0)
    Q = beta * np.eye(N) + alpha * Uv.T @ Uv + gamma * Uz.T @ Uz
    w = np.linalg.eigvalsh(Q)
    rng = np.random.default_rng(N)
    d = Uv.T @ rng.uniform(-0.3, 0.3, N)                     # a gradient like alpha Uv'(Sv c - v_ref)
    x = np.linalg.solve(Q, -d)
    print(f"N={N}: cond {w[-1]/w[0]:.3g}, eigenvalues {w[0]:.3e} .. {w[-1]:.3e}, |Q|max {np.abs(Q).max():.3g}")
    for name, prec in (("f64 MFMA", wg.GRAMIAN_F64), ("f32 MFMA", wg.GRAMIAN_F32)):
        Qg = wg.gramian_batch(N, np.array([T]), np.array([h]), alpha, beta, gamma, prec)[0]
        Qs = 0.5 * (Qg + Qg.T)
        wg_ = np.linalg.eigvalsh(Qs)
        try:
            np.linalg.cholesky(Qs); chol = "ok"
        except np.linalg.LinAlgError:
            chol = "FAILS"
        xg = np.linalg.solve(Qs, -d) if wg_[0] > 0 else np.full(N, np.nan)
        print(f"   {name}: max entry error {np.abs(Qg - Q).max():.2e} ({np.abs(Qg - Q).max() / np.abs(Q).max():.1e} of |Q|max), "
              f"asymmetry {np.abs(Qg - Qg.T).max():.1e}, smallest eigenvalue {wg_[0]:.3e} (true {w[0]:.3e}), Cholesky {chol}, "
              f"solve: relative change of the jerk vector {np.linalg.norm(xg - x) / np.linalg.norm(x):.2e}")
