"""Seeded QP generators shared by the CPU and GPU parity tests.

All generators return a dict with Fortran-ordered float64 arrays laid out the
way ql0001_ takes them (qld.hh:27-31): C (nmax x nmax), d (n), A (mmax x n,
constraint rows  A x + b >= 0, first `me` rows equalities), b (mmax), xl, xu.
"""
import numpy as np


def _pack(C, d, A, b, xl, xu, me=0, mmax=None):
    n = C.shape[0]
    m = A.shape[0]
    mmax = mmax or (m + 1)
    Af = np.zeros((mmax, n), order="F")
    Af[:m, :] = A
    bf = np.zeros(mmax)
    bf[:m] = b
    return dict(n=n, m=m, me=me, mmax=mmax, nmax=n,
                C=np.asfortranarray(C, dtype=np.float64), d=np.ascontiguousarray(d, dtype=np.float64),
                A=Af, b=bf, xl=np.ascontiguousarray(xl, dtype=np.float64),
                xu=np.ascontiguousarray(xu, dtype=np.float64))


def random_pd(rng, n, m, me=0, bound=1e8, scale=1.0, feasible=True):
    """Strictly convex QP with m general rows; x0 strictly feasible if asked."""
    M = rng.standard_normal((n, n))
    C = M @ M.T + 0.1 * np.eye(n)
    C = 0.5 * (C + C.T)
    d = scale * rng.standard_normal(n)
    A = rng.standard_normal((m, n))
    x0 = rng.standard_normal(n)
    if feasible:
        b = -A @ x0 + rng.uniform(0.0, 1.0, m)
        b[:me] = -A[:me] @ x0
    else:
        b = rng.standard_normal(m)
    xl = np.full(n, -bound)
    xu = np.full(n, bound)
    return _pack(C, d, A, b, xl, xu, me=me)


def boxed(rng, n, m):
    """Tight simple bounds so that lower/upper-bound codes enter the active set."""
    q = random_pd(rng, n, m)
    q["xl"] = rng.uniform(-0.3, -0.05, n)
    q["xu"] = rng.uniform(0.05, 0.3, n)
    return q


def dependent_rows(rng, n, m):
    """Duplicate / opposite rows: exercises the linear-dependence branches."""
    q = random_pd(rng, n, m)
    A = q["A"][:m].copy()
    b = q["b"][:m].copy()
    k = m // 3
    for i in range(k):
        src = rng.integers(0, m)
        dst = rng.integers(0, m)
        if src == dst:
            continue
        sgn = rng.choice([1.0, -1.0, 2.0])
        A[dst] = sgn * A[src]
        b[dst] = sgn * b[src] + (rng.uniform(0.0, 0.5) if sgn > 0 else rng.uniform(0.5, 2.0))
    return _pack(q["C"], q["d"], A, b, q["xl"], q["xu"])


def infeasible(rng, n, m):
    """Contradictory pair  a x >= 1 and -a x >= 1."""
    q = random_pd(rng, n, m)
    A = q["A"][:m].copy()
    b = q["b"][:m].copy()
    A[1] = -A[0]
    b[0] = -1.0
    b[1] = -1.0
    return _pack(q["C"], q["d"], A, b, q["xl"], q["xu"])


def semidefinite(rng, n, m, rank=None):
    """Singular / indefinite-looking Hessian: exercises the diagonal bump loop."""
    rank = rank or max(1, n // 2)
    M = rng.standard_normal((n, rank))
    C = M @ M.T
    C = 0.5 * (C + C.T)
    if rng.random() < 0.5:
        C[-1, -1] = 0.0
    d = rng.standard_normal(n)
    A = rng.standard_normal((m, n))
    x0 = rng.standard_normal(n)
    b = -A @ x0 + rng.uniform(0.0, 1.0, m)
    return _pack(C, d, A, b, np.full(n, -10.0), np.full(n, 10.0))


def zero_rows(rng, n, m):
    """Some all-zero normals (the reference's dummy row 0 is one)."""
    q = random_pd(rng, n, m)
    A = q["A"][:m].copy()
    b = q["b"][:m].copy()
    A[0] = 0.0
    b[0] = 0.0
    if m > 3:
        A[3] = 0.0
        b[3] = 0.7          # zero normal, satisfied -> ignored
    return _pack(q["C"], q["d"], A, b, q["xl"], q["xu"])


def herdt_like(rng, N=16, s=2, T=0.1, h=0.814):
    """A QP with the Herdt-2010 shape (n = 2N+2s, m = 1+4N+5s) built directly
    from the cart-table matrices; used before/independently of the tick oracle."""
    n = 2 * N + 2 * s
    i = np.arange(N)[:, None]
    j = np.arange(N)[None, :]
    low = (j <= i)
    Uv = np.where(low, (2 * (i - j) + 1) * T * T * 0.5, 0.0)
    Uz = np.where(low, (1 + 3 * (i - j) + 3 * (i - j) ** 2) * T ** 3 / 6.0 - T * h / 9.81, 0.0)
    Sv = np.stack([np.zeros(N), np.ones(N), (np.arange(N) + 1) * T], 1)
    Sz = np.stack([np.ones(N), (np.arange(N) + 1) * T, ((np.arange(N) + 1) * T) ** 2 / 2 - h / 9.81], 1)
    alpha, beta, gamma = 1.0, 1e-5, 1e-6
    Qb = beta * np.eye(N) + alpha * Uv.T @ Uv + gamma * Uz.T @ Uz
    # step selection: first `k0` samples on the current foot, then 8 per step
    k0 = int(rng.integers(1, 9))
    V = np.zeros((N, s))
    for r in range(N):
        st = 0 if r < k0 else 1 + (r - k0) // 8
        if st >= 1 and st <= s:
            V[r, st - 1] = 1.0
    Vc = np.where(V.sum(1) == 0, 1.0, 0.0)
    C = np.zeros((n, n))
    C[:N, :N] = Qb
    C[N:2 * N, N:2 * N] = Qb
    if s:
        C[:N, 2 * N:2 * N + s] = -gamma * Uz.T @ V
        C[2 * N:2 * N + s, :N] = -gamma * V.T @ Uz
        C[N:2 * N, 2 * N + s:] = -gamma * Uz.T @ V
        C[2 * N + s:, N:2 * N] = -gamma * V.T @ Uz
        C[2 * N:2 * N + s, 2 * N:2 * N + s] = gamma * V.T @ V
        C[2 * N + s:, 2 * N + s:] = gamma * V.T @ V
    cx = np.array([rng.normal(0, 0.02), rng.normal(0.1, 0.1), rng.normal(0, 0.3)])
    cy = np.array([rng.normal(0, 0.03), rng.normal(0, 0.15), rng.normal(0, 0.5)])
    vref = np.array([rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1)])
    fx, fy = cx[0] + rng.normal(0, 0.01), cy[0] + rng.choice([-0.09, 0.09])
    d = np.zeros(n)
    d[:N] = alpha * Uv.T @ (Sv @ cx - vref[0])
    d[N:2 * N] = alpha * Uv.T @ (Sv @ cy - vref[1])
    if s:
        d[2 * N:2 * N + s] = -gamma * V.T @ (Sz @ cx) + gamma * V.T @ (Vc * fx)
        d[2 * N + s:] = -gamma * V.T @ (Sz @ cy) + gamma * V.T @ (Vc * fy)
    hx, hy = 0.0686, 0.029
    m = 1 + 4 * N + 5 * s
    A = np.zeros((m, n))
    b = np.zeros(m)
    edges = [(1.0, 0.0, hx), (-1.0, 0.0, hx), (0.0, 1.0, hy), (0.0, -1.0, hy)]
    for r in range(N):
        for e, (ax, ay, dd) in enumerate(edges):
            row = 1 + 4 * r + e
            # inside:  dd - ax*(z_x - p_x) - ay*(z_y - p_y) >= 0
            A[row, :N] = -ax * Uz[r]
            A[row, N:2 * N] = -ay * Uz[r]
            if s:
                A[row, 2 * N:2 * N + s] = ax * V[r]
                A[row, 2 * N + s:] = ay * V[r]
            b[row] = dd - ax * (Sz[r] @ cx) - ay * (Sz[r] @ cy) + ax * Vc[r] * fx + ay * Vc[r] * fy
    px = [-0.28, -0.2, 0.0, 0.2, 0.28]
    py = [-0.2, -0.3, -0.4, -0.3, -0.2]
    for k in range(s):
        sign = 1.0 if (k % 2 == 0) else -1.0
        for e in range(5):
            x1, y1 = px[e], sign * py[e]
            x2, y2 = px[(e + 1) % 5], sign * py[(e + 1) % 5]
            ax, ay = sign * (y1 - y2), sign * (x2 - x1)
            dd = ax * x1 + ay * y1
            row = 1 + 4 * N + 5 * k + e
            A[row, 2 * N + k] = -ax
            A[row, 2 * N + s + k] = -ay
            if k > 0:
                A[row, 2 * N + k - 1] = ax
                A[row, 2 * N + s + k - 1] = ay
                b[row] = dd
            else:
                b[row] = dd + ax * fx + ay * fy
    return _pack(C, d, A, b, np.full(n, -1e8), np.full(n, 1e8), mmax=m + 1)


FAMILIES = {
    "random_pd": lambda rng: random_pd(rng, int(rng.integers(2, 40)), int(rng.integers(1, 80))),
    "equalities": lambda rng: random_pd(rng, int(rng.integers(4, 30)), int(rng.integers(4, 40)), me=int(rng.integers(1, 4))),
    "boxed": lambda rng: boxed(rng, int(rng.integers(2, 30)), int(rng.integers(1, 40))),
    "dependent": lambda rng: dependent_rows(rng, int(rng.integers(3, 24)), int(rng.integers(6, 60))),
    "infeasible": lambda rng: infeasible(rng, int(rng.integers(3, 20)), int(rng.integers(4, 30))),
    "semidefinite": lambda rng: semidefinite(rng, int(rng.integers(3, 24)), int(rng.integers(2, 40))),
    "zero_rows": lambda rng: zero_rows(rng, int(rng.integers(3, 24)), int(rng.integers(5, 40))),
    "rand_infeas": lambda rng: random_pd(rng, int(rng.integers(2, 12)), int(rng.integers(10, 60)), feasible=False),
    "herdt_like": lambda rng: herdt_like(rng, 16, int(rng.integers(0, 3))),
}
