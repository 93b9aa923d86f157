"""Synthetic footstep plans as ZMP polytopes, one per 0.1 s slot (pure numpy geometry: no solver, no checker).  Inputs of the
Dimitrov-2008 / PLDP tests (tests/dimitrov.py) and of bench.py's `kernels` legs: what FootConstraintsAsLinearSystem would hand
the receding-horizon loop for a straight walk with random step lengths (a box under the stance foot in single support, the hull of
both feet in double support, 4 or 6 edges, antiparallel pairs marked the way FindSimilarConstraints marks them,
FootConstraintsAsLinearSystem.cpp:55-93)."""
import numpy as np


def box(cx, cy, hx, hy):
    A = np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [0.0, -1.0]])
    B = np.array([-(cx - hx), -(cy - hy), cx + hx, cy + hy])
    return A, B, (cx, cy), np.array([0, 0, -2, -2])


def hexagon(p0, p1, hx, hy):
    """hull of two equal boxes centred at p0, p1 (different x and y): 6 edges, edge i+3 antiparallel to edge i"""
    pts = []
    for (cx, cy) in (p0, p1):
        pts += [(cx - hx, cy - hy), (cx + hx, cy - hy), (cx + hx, cy + hy), (cx - hx, cy + hy)]
    pts = np.array(pts)
    c = pts.mean(axis=0)
    # gift wrap (tiny input)
    hull = []
    start = int(np.lexsort((pts[:, 1], pts[:, 0]))[0]); cur = start
    while True:
        hull.append(cur)
        nxt = (cur + 1) % len(pts)
        for k in range(len(pts)):
            u_, w_ = pts[nxt] - pts[cur], pts[k] - pts[cur]
            cr = u_[0] * w_[1] - u_[1] * w_[0]
            if cr < -1e-14 or (abs(cr) <= 1e-14 and np.linalg.norm(pts[k] - pts[cur]) > np.linalg.norm(pts[nxt] - pts[cur])):
                nxt = k
        cur = nxt
        if cur == start:
            break
    H = pts[hull]
    if len(H) != 6:
        return None
    A = np.zeros((6, 2)); B = np.zeros(6)
    for e in range(6):
        p, q = H[e], H[(e + 1) % 6]
        nrm = np.array([-(q[1] - p[1]), q[0] - p[0]])
        if nrm @ (c - p) < 0:
            nrm = -nrm
        A[e] = nrm; B[e] = -(nrm @ p)
    A[3:] = -A[:3]                                                     # exact antiparallel pairs, like the reference's test
    return A, B, (c[0], c[1]), np.array([0, 0, 0, -3, -3, -3])


def plan(rng, n_steps=8, hx=0.07, hy=0.03, ss=7, ds=1):
    """-> one polygon per 0.1 s slot: a start DS (10 slots), n_steps alternating SS (ss slots) separated by short DS
    phases (ds slots), a final DS.  Integer slots keep the slot -> polygon map identical from tick to tick (the
    reference compares floating times against the intervals' EndingTime, :846-869)."""
    slots = []
    lx, ly, rx, ry = 0.0, 0.095, 0.0, -0.095
    slots += [box(0.0, 0.0, hx, hy + 0.095)] * 10
    left_support = bool(rng.integers(2))
    for s in range(n_steps):
        cx, cy = (lx, ly) if left_support else (rx, ry)
        slots += [box(cx, cy, hx, hy)] * ss
        dx = rng.uniform(0.05, 0.25); dy = rng.uniform(-0.02, 0.02)
        if left_support:
            rx, ry = lx + dx, -0.095 + dy
        else:
            lx, ly = rx + dx, 0.095 + dy
        hp = hexagon((lx, ly), (rx, ry), hx, hy)
        slots += [hp if hp is not None else box(0.5 * (lx + rx), 0.0, hx + 0.5 * abs(lx - rx), hy + 0.095)] * ds
        left_support = not left_support
    slots += [box(0.5 * (lx + rx), 0.5 * (ly + ry), hx + 0.5 * abs(lx - rx), hy + 0.095)] * 40
    return slots


def polys_at(slots, it, N):
    return [slots[min(it + i, len(slots) - 1)] for i in range(N)]
