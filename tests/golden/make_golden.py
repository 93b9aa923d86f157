"""Generates the committed golden fixtures.  Runs ONLY in the development container
(needs /root/reference and oracle/_ref/libqld_ref.so); the fixtures travel, this script's inputs do not.

  herdt_emergency_stop_datref.npz : the reference's own golden file
        /root/reference/tests/TestHerdt2010EmergencyStopTestFGPI.datref.cmake  (4508 x 38), as data.
  ql_golden.npz : QPs (inputs) + the COMPILED REFERENCE qld.cpp's outputs (x, u, ifail, final active set)
        - 24 QPs per family of tests/qpgen.py (branch coverage of ql0002), and
        - the 225 QPs the Herdt oracle assembles while replaying the EmergencyStop scenario.
  preview_control_parameters.npz : the reference's precomputed Kajita gains
        /root/reference/src/data/PreviewControlParameters.ini  (Zc, T, preview time, Kx[3], Ks, F[320]), as data.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import herdt_replay as hr  # noqa: E402
import oraclelib as ol  # noqa: E402
import qpgen  # noqa: E402

REF = "/root/reference/tests/TestHerdt2010EmergencyStopTestFGPI.datref.cmake"


def preview_ini():
    v = np.array(open("/root/reference/src/data/PreviewControlParameters.ini").read().split(), dtype=float)
    nl = int(v[2] / v[1])
    assert v.size == 7 + nl
    np.savez_compressed(os.path.join(HERE, "preview_control_parameters.npz"), zc=v[0], T=v[1], preview_time=v[2],
                        Kx=v[3:6], Ks=v[6], F=v[7:7 + nl])


def main():
    preview_ini()
    datref = np.loadtxt(REF)
    np.savez_compressed(os.path.join(HERE, "herdt_emergency_stop_datref.npz"), datref=datref)

    recs = []

    def add(tag, q):
        r = ol.ref_ql(q)
        o = ol.oracle_ql(q)   # only to learn nact (a static local inside the reference)
        recs.append(dict(tag=tag, n=q["n"], m=q["m"], me=q["me"], mmax=q["mmax"], C=q["C"], d=q["d"], A=q["A"], b=q["b"],
                         xl=q["xl"], xu=q["xu"], x=r["x"], u=r["u"], ifail=r["ifail"], iact=r["iwar"][:o["nact"]]))

    for fam in sorted(qpgen.FAMILIES):
        for s in range(24):
            add(fam, qpgen.FAMILIES[fam](np.random.default_rng(424200 + 977 * s)))

    model, state, events = hr.emergency_stop_setup(datref)

    def on_tick(it, clock, st, out, dump):
        n, m, mmax = dump.n, dump.m, dump.mmax
        q = dict(n=n, m=m, me=0, mmax=mmax, nmax=n,
                 C=np.array(dump.C[:n * n]).reshape((n, n), order="F"), d=np.array(dump.d[:n]),
                 A=np.array(dump.A[:mmax * n]).reshape((mmax, n), order="F"), b=np.array(dump.b[:mmax]),
                 xl=np.full(n, -1e8), xu=np.full(n, 1e8))
        add("herdt_tick_%d" % it, q)

    hr.replay(model, state, events, 6000, on_tick=on_tick, legacy_running=True)
    out = {}
    for k, r in enumerate(recs):
        for key, v in r.items():
            out["%04d_%s" % (k, key)] = np.asarray(v)
    out["count"] = np.array(len(recs))
    np.savez_compressed(os.path.join(HERE, "ql_golden.npz"), **out)
    print("wrote", len(recs), "QPs")


if __name__ == "__main__":
    main()
