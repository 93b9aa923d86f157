"""Build-supplied ZMP reference queues for the Kajita stage-1 tests: the pattern the reference's ZMPDiscretization
produces for a step sequence (ZMP under the stance foot during single support, linear hand-over during double
support, at rest between the feet before and after) -- shape only, it is test input, not a restatement."""
import numpy as np


def step_sequence_zmp(n_steps, step_len, half_width, t_ss, t_ds, t_rest, T, tail, lateral_first=-1.0):
    """[x, y] samples at period T: rest, then n_steps alternating stance feet, rest, plus `tail` trailing samples."""
    pts = [(0.0, 0.0, t_rest)]
    x, side = 0.0, lateral_first
    for k in range(n_steps):
        pts.append((x, side * half_width, t_ss))
        x += step_len if k < n_steps - 1 else 0.0
        side = -side
    pts.append((x, 0.0, t_rest))
    zx, zy = [], []
    px, py = pts[0][0], pts[0][1]
    for (cx, cy, dur) in pts:
        nds = int(round(t_ds / T))
        for i in range(nds):                                   # hand-over from the previous point
            a = (i + 1) / nds
            zx.append(px + a * (cx - px)); zy.append(py + a * (cy - py))
        for _ in range(int(round(dur / T))):
            zx.append(cx); zy.append(cy)
        px, py = cx, cy
    zx += [px] * tail; zy += [py] * tail
    return np.array(zx), np.array(zy)


def random_batch(rng, B, L, nl, T=0.005):
    """B gaits with their own step length / width / timing, each L + nl - 1 samples long."""
    Lz = L + nl - 1
    ZX = np.zeros((B, Lz)); ZY = np.zeros((B, Lz))
    for b in range(B):
        zx, zy = step_sequence_zmp(n_steps=int(rng.integers(2, 9)), step_len=rng.uniform(0.05, 0.3),
                                   half_width=rng.uniform(0.08, 0.12), t_ss=rng.uniform(0.5, 0.9), t_ds=rng.uniform(0.02, 0.2),
                                   t_rest=rng.uniform(0.2, 1.0), T=T, tail=Lz, lateral_first=rng.choice([-1.0, 1.0]))
        ZX[b] = zx[:Lz]; ZY[b] = zy[:Lz]
    return ZX, ZY
