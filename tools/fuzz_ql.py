"""Parity fuzz of the QL solver (evidence, not a test): many more seeds than tests/test_ql_gpu.py runs, every family of
tests/qpgen.py plus magnitude-scaled variants of them (the whole problem scaled by 2^k, the constraints alone, the Hessian
alone; to the edges of the double range in the edge_* families) and config-5-sized problems, through wg_qp_solve_batch (the C ABI) against the CPU oracle -- ifail, iteration count,
final active set, the complete add / drop history, x bit for bit (NaN where the oracle has NaN), u where the solve succeeded.
The oracle runs on the host cores beside the product path here, as in tests/: nothing of it is measured or shipped.

    python tools/fuzz_ql.py [seeds per family, default 2000] > profiles/<tag>_fuzz_ql.txt
Round 5 wrote this after a longest-first scheduling test stumbled over QPs whose iterate becomes NaN: the families'
96 seeds per test had never produced one."""
import importlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol  # noqa: E402
import qpgen  # noqa: E402

HIST = 12000                      # the NaN regime's runs to maxit log 8 877 events at the Herdt size
BATCH = 4000


def scaled(q, kq=0, ka=0):
    """the Hessian and gradient scaled by 2^kq, the constraints (A, b) by 2^ka: exact scalings, other magnitudes everywhere"""
    q = dict(q)
    q["C"] = np.asfortranarray(q["C"] * 2.0 ** kq); q["d"] = q["d"] * 2.0 ** kq
    q["A"] = np.asfortranarray(q["A"] * 2.0 ** ka); q["b"] = q["b"] * 2.0 ** ka
    return q


def variants():
    out = {}
    for name, gen in qpgen.FAMILIES.items():
        out[name] = gen
    out["herdt_like_big_steps"] = lambda rng: qpgen.herdt_like(rng, 16, 2)
    out["scaled_up"] = lambda rng: scaled(qpgen.FAMILIES["random_pd"](rng), int(rng.integers(20, 200)), int(rng.integers(-100, 100)))
    out["scaled_down"] = lambda rng: scaled(qpgen.FAMILIES["dependent"](rng), -int(rng.integers(20, 200)), int(rng.integers(-100, 100)))
    out["herdt_scaled"] = lambda rng: scaled(qpgen.herdt_like(rng, 16, int(rng.integers(0, 3))), int(rng.integers(-60, 60)), int(rng.integers(-60, 60)))
    out["config5_sized"] = lambda rng: qpgen.herdt_like(rng, 32, int(rng.integers(0, 5)))
    # the edges of the double range: overflow, underflow and denormals inside the solver
    out["edge_up"] = lambda rng: scaled(qpgen.herdt_like(rng, 16, int(rng.integers(0, 3))), int(rng.integers(200, 480)), int(rng.integers(-480, 480)))
    out["edge_down"] = lambda rng: scaled(qpgen.herdt_like(rng, 16, int(rng.integers(0, 3))), -int(rng.integers(200, 480)), int(rng.integers(-480, 480)))
    out["edge_random"] = lambda rng: scaled(qpgen.FAMILIES["dependent"](rng), int(rng.integers(-480, 480)), int(rng.integers(-480, 480)))
    return out


def oracle_chunk(args):
    name, seeds, nmax, mmax = args
    gen = variants()[name]
    res = []
    for s in seeds:
        q = gen(np.random.default_rng(s))
        n, m = q["n"], q["m"]
        Cp = np.zeros((nmax, nmax), order="F"); Cp[:n, :n] = q["C"][:n, :n]
        Ap = np.zeros((mmax, nmax), order="F"); Ap[:m, :n] = q["A"][:m, :n]           # exactly wgmpc.pack_qps's padding
        bp = np.zeros(mmax); bp[:m] = q["b"][:m]
        qq = dict(n=n, m=m, me=q["me"], nmax=nmax, mmax=mmax, C=Cp, A=Ap, d=np.pad(q["d"], (0, nmax - n)), b=bp,
                  xl=np.pad(q["xl"], (0, nmax - n)), xu=np.pad(q["xu"], (0, nmax - n)))
        o = ol.oracle_ql(qq, hist_cap=HIST)
        res.append((s, o["ifail"], o["n_iter"], o["nact"], o["iact"].copy(), o["hist_len"], o["hist"][:min(o["hist_len"], HIST)].copy(), o["x"].copy(),
                    o["u"].copy()))
    return name, res

SEED0 = int(os.environ.get("FUZZ_SEED0", "910000"))          # another base = another set of problems


def main():
    per_family = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    wg = importlib.import_module("jrl-walkgen_amd")
    wg.init(0)
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    V = variants()
    print("# QL parity fuzz: %d seeds per family from %d, %d families, oracle on %d host cores" % (per_family, SEED0, len(V), cores))
    total = bad_total = 0
    t_all = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        for name, gen in V.items():
            t0 = time.time()
            n_seeds = per_family if name != "config5_sized" else max(50, per_family // 10)
            all_seeds = [SEED0 + 7919 * k for k in range(n_seeds)]
            bad = []
            fails = {}
            nan_x = longest = 0
            for b0 in range(0, n_seeds, BATCH):
                seeds = all_seeds[b0:b0 + BATCH]
                qps = [gen(np.random.default_rng(s)) for s in seeds]
                pk = wg.pack_qps(qps)
                res = wg.qp_solve_batch(pk, hist_cap=HIST)
                chunks = [seeds[i::cores * 4] for i in range(cores * 4)]
                ref = {}
                for _, rr in pool.imap_unordered(oracle_chunk, [(name, c, pk["nmax"], pk["mmax"]) for c in chunks if c]):
                    for r in rr:
                        ref[r[0]] = r
                for k, s in enumerate(seeds):
                    _, ifail, nit, nact, iact, hl, hist, x, u = ref[s]
                    n, m = qps[k]["n"], qps[k]["m"]
                    ok = int(res["ifail"][k]) == ifail and int(res["n_iter"][k]) == nit and int(res["hist_len"][k]) == hl
                    ok = ok and np.array_equal(res["hist"][k, :len(hist)], hist) and ol.same_bits_nan_aware(res["x"][k, :n], x)
                    if ifail == 0:
                        ok = ok and int(res["nact"][k]) == nact and np.array_equal(res["iact"][k, :nact], iact) and ol.same_bits(res["u"][k, :m + 2 * n], u)
                    if not ok:
                        bad.append((s, int(res["ifail"][k]), ifail, int(res["n_iter"][k]), nit))
                    key = ifail if ifail < 3 else 11
                    fails[key] = fails.get(key, 0) + 1
                    nan_x += int(np.isnan(x).any()); longest = max(longest, hl)
            seeds = all_seeds
            total += len(seeds); bad_total += len(bad)
            print("%-22s %6d QPs  mismatches %d  | ifail 0 / 1 / 2 / >10: %d / %d / %d / %d, NaN solutions %d, longest history %d  (%.0f s)%s" %
                  (name, len(seeds), len(bad), fails.get(0, 0), fails.get(1, 0), fails.get(2, 0), fails.get(11, 0), nan_x,
                   longest, time.time() - t0, ("  FIRST: %s" % (bad[:3],)) if bad else ""), flush=True)
    print("# total %d QPs, %d mismatches, %.0f s" % (total, bad_total, time.time() - t_all))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
