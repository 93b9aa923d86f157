"""The CPU restatement (oracle/ql_oracle.c) against the COMPILED REFERENCE qld.cpp on many more QPs than the golden fixture holds
(evidence, not a test; runs only where /root/reference is: the development container).  Every family of tools/fuzz_ql.py -- the
families of tests/qpgen.py, magnitude-scaled variants, config-5-sized problems, problems scaled to the edges of the double
range (2^+-480: overflow, underflow and denormals inside the solver).  Compared: ifail, x bit for bit (a NaN matches a NaN), u and
the final active set where the solve succeeded, and the complete add / drop history (instrumented build of the reference,
tests/oraclelib.py:ref_hist).

    python tools/fuzz_oracle_vs_reference.py [seeds per family, default 2000] > profiles/<tag>_fuzz_oracle_vs_reference.txt
Round 5 wrote this after an overdriven gait showed two places where the restatement's comparisons treated a NaN differently
from the reference's: the fixture's 450 QPs had never reached them."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oraclelib as ol  # noqa: E402
import qpgen  # noqa: E402
import fuzz_ql  # noqa: E402


variants = fuzz_ql.variants


def chunk(args):
    name, seeds = args
    gen = variants()[name]
    out = []
    for s in seeds:
        q = gen(np.random.default_rng(s))
        o = ol.oracle_ql(q, hist_cap=20000)
        r = ol.ref_ql_hist(q)
        ok = o["ifail"] == r["ifail"] and ol.same_bits_nan_aware(o["x"], r["x"])
        ok = ok and o["hist_len"] == len(r["hist"]) and np.array_equal(o["hist"][:o["hist_len"]], r["hist"])
        if r["ifail"] == 0:
            ok = ok and ol.same_bits(o["u"], r["u"]) and np.array_equal(o["iact"], r["iwar"][:o["nact"]])
        out.append((s, ok, r["ifail"], int(np.isnan(r["x"]).any()), len(r["hist"])))
    return name, out

SEED0 = int(os.environ.get("FUZZ_SEED0", "730000"))          # another base = another set of problems


def main():
    per_family = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cores = max(1, len(os.sched_getaffinity(0)))
    V = variants()
    print("# oracle/ql_oracle.c against the compiled reference qld.cpp: %d seeds per family from %d, %d families, %d processes" % (per_family, SEED0, len(V), cores))
    total = bad_total = 0
    t_all = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        for name in V:
            t0 = time.time()
            n_seeds = per_family if name != "config5_sized" else max(50, per_family // 10)
            seeds = [SEED0 + 6007 * k for k in range(n_seeds)]
            res = []
            for _, rr in pool.imap_unordered(chunk, [(name, seeds[i::cores * 4]) for i in range(cores * 4) if seeds[i::cores * 4]]):
                res += rr
            bad = sorted(s for s, ok, *_ in res if not ok)
            fails = {}
            for _, _, f, _, _ in res:
                fails[f if f < 3 else 11] = fails.get(f if f < 3 else 11, 0) + 1
            total += len(res); bad_total += len(bad)
            print("%-22s %6d QPs  mismatches %d  | ifail 0 / 1 / 2 / >10: %d / %d / %d / %d, NaN solutions %d, longest history %d  (%.0f s)%s" %
                  (name, len(res), len(bad), fails.get(0, 0), fails.get(1, 0), fails.get(2, 0), fails.get(11, 0), sum(r[3] for r in res),
                   max(r[4] for r in res), time.time() - t0, ("  FIRST: %s" % bad[:5]) if bad else ""), flush=True)
    print("# total %d QPs, %d mismatches, %.0f s" % (total, bad_total, time.time() - t_all))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
