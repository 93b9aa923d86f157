"""Test-side driver around the PLDP back-end: builds the LQ-preconditioned problems the reference's Dimitrov-2008
generator hands to PLDPSolver::SolveProblem, for synthetic footstep plans.

Follows (as a data generator, numpy):
  ZMPConstrainedQPFastFormulation::InitializeMatrixPbConstants        ZMPConstrainedQPFastFormulation.cpp:158-246
  ::BuildingConstantPartOfTheObjectiveFunction[QLDANDLQ]              :384-560   (OptA = I + beta PPu'PPu + alpha VPu'
                                                                      -- the reference scales VPu', :524-527 -- and only
                                                                      the lower triangle reaches the Cholesky)
  ::BuildingConstantPartOfConstraintMatrices                          :597-690
  ::BuildConstraintMatrices                                           :759-1022  (DPu column-major, ld = m+1)
  ::BuildZMPTrajectoryFromFootTrajectory (solve loop, un-preconditioning, LIPM step)   :1096-1463
Defaults N=16, T=0.1, CoM height 0.80, alpha=200, beta=1000 (:81-97).
"""
import numpy as np

import oraclelib as ol


class Dimitrov:
    def __init__(self, N=16, T=0.1, h=0.80, alpha=200.0, beta=1000.0):
        self.N, self.T, self.h = N, T, h
        i = np.arange(N)[:, None]; j = np.arange(N)[None, :]
        low = j <= i
        PPu1 = np.where(low, (1 + 3 * (i - j) + 3 * (i - j) ** 2) * T * T * T / 6.0, 0.0)
        VPu1 = np.where(low, (2 * (i - j) + 1) * T * T * 0.5, 0.0)
        Z = np.zeros((N, N))
        PPu = np.block([[PPu1, Z], [Z, PPu1]]); VPu = np.block([[VPu1, Z], [Z, VPu1]])
        k = np.arange(1, N + 1) * T
        PPx1 = np.stack([np.ones(N), k, k * k * 0.5], axis=1); VPx1 = np.stack([np.zeros(N), np.ones(N), k], axis=1)
        Z3 = np.zeros((N, 3))
        PPx = np.block([[PPx1, Z3], [Z3, PPx1]]); VPx = np.block([[VPx1, Z3], [Z3, VPx1]])
        OptA = np.eye(2 * N) + beta * (PPu.T @ PPu) + alpha * VPu.T
        Q = np.tril(OptA[:N, :N]); Q = Q + np.tril(Q, -1).T           # what the Cholesky effectively factors
        LQ = ol.chol_normal(Q)
        iLQ1 = ol.chol_inverse(LQ)
        self.iLQ = np.block([[iLQ1, Z], [Z, iLQ1]])
        self.OptB = self.iLQ @ (alpha * (VPu.T @ VPx) + beta * (PPu.T @ PPx))
        self.OptC = self.iLQ @ (beta * PPu.T)
        PuT = np.where(j >= i, (1 + 3 * (j - i) + 3 * (j - i) ** 2) * T * T * T / 6.0 - T * h / 9.81, 0.0)   # Pu'
        self.Pu = iLQ1 @ PuT                                           # m_Pu = iLQ * Pu'
        self.iPu = np.linalg.inv(self.Pu)
        self.Px = np.stack([np.ones(N), k, k * k * 0.5 - h / 9.81], axis=1)
        self.A3 = np.array([[1, T, T * T / 2], [0, 1, T], [0, 0, 1.0]]); self.B3 = np.array([T ** 3 / 6, T * T / 2, T])

    def problem(self, xk, polys):
        """polys: N entries (A rows x 2, B rows, centre 2, similar rows) for the previewed instants.
        -> dict(m, D, A (flat, ld m+1), b, zmpref, similar, first_rows)"""
        N = self.N
        m = sum(len(p[1]) for p in polys)
        A = np.zeros((m + 1) * 2 * N); b = np.zeros(m); sim = np.zeros(m, dtype=np.int32); zr = np.zeros(2 * N)
        idx = 0
        for i, (Ai, Bi, Ci, Si) in enumerate(polys):
            zr[i], zr[i + N] = Ci
            zx = xk[0] * self.Px[i, 0] + xk[1] * self.Px[i, 1] + xk[2] * self.Px[i, 2]
            zy = xk[3] * self.Px[i, 0] + xk[4] * self.Px[i, 1] + xk[5] * self.Px[i, 2]
            for jrow in range(len(Bi)):
                b[idx] = zx * Ai[jrow, 0] + zy * Ai[jrow, 1] + Bi[jrow]
                sim[idx] = Si[jrow]
                for k in range(N):
                    A[idx + k * (m + 1)] = Ai[jrow, 0] * self.Pu[k, i]
                    A[idx + (k + N) * (m + 1)] = Ai[jrow, 1] * self.Pu[k, i]
                idx += 1
        D = self.OptB @ xk - self.OptC @ zr
        return dict(m=m, D=D, A=A, b=b, zmpref=zr, similar=sim, first_rows=len(polys[0][1]), xk=np.array(xk, float))

    def jerk(self, X):
        """un-preconditioning, :1355-1381: NewX = iLQ' X"""
        return self.iLQ.T @ X

    def step(self, xk, X):
        u = self.jerk(X)
        x = self.A3 @ xk[:3] + self.B3 * u[0]; y = self.A3 @ xk[3:] + self.B3 * u[self.N]
        return np.concatenate([x, y])


from footplans import box, hexagon, plan, polys_at  # noqa: E402,F401  (the plans are pure geometry: tests/footplans.py)


def run_gait(dm, model, segs, n_ticks, solve, xk0=None, max_iter=0):
    """Replays the reference's receding-horizon loop with `solve(model, st, prob, n_removed, starting)`.
    Returns the list of per-tick records."""
    st = ol.PldpState()
    xk = np.zeros(6) if xk0 is None else np.array(xk0, float)
    recs = []
    n_removed = 0
    starting = True
    prev_first = None
    for it in range(n_ticks):
        polys = polys_at(segs, it, dm.N)
        pr = dm.problem(xk, polys)
        r = solve(model, st, pr, n_removed, starting)
        recs.append(dict(prob=pr, res=r, xk=xk.copy(), n_removed=n_removed, starting=starting))
        starting = False
        n_removed = pr["first_rows"]
        if r["ret"] != 0:
            break
        xk = dm.step(xk, r["X"])
    return recs
