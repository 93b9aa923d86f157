"""Generates the committed golden fixtures.  Runs ONLY in the development container
(needs /root/reference and oracle/_ref/libqld_ref.so); the fixtures travel, this script's inputs do not.

  herdt_emergency_stop_datref.npz : the reference's own golden file
        /root/reference/tests/TestHerdt2010EmergencyStopTestFGPI.datref.cmake  (4508 x 38), as data.
  ql_golden.npz : QPs (inputs) + the COMPILED REFERENCE qld.cpp's outputs (x, u, ifail, final active set, and the
        add/drop HISTORY: +k = code k added, -k = dropped, in order -- from a throw-away build of the reference's qld.cpp
        with one log call at its add site (qld.cpp:1766) and one at its drop site (:1903), tests/oraclelib.py:ref_hist;
        x, u, ifail and the final set are taken from the UNPATCHED oracle/_ref build and must agree with the patched one)
        - 24 QPs per family of tests/qpgen.py (branch coverage of ql0002),
        - eight QPs on which the reference's iterate becomes NaN (found in round 5: 7 of 6000 random Herdt-shaped problems, one of
          the `infeasible` family): it runs to maxit, ifail = 1, a NaN solution, histories of 8 877 / 2 237 events,
        - one QP of an overdriven gait (overdriven_tick_qp below: magnitudes to 1e240 and a rotation with a denormal operand), and
        - the 225 QPs the Herdt oracle assembles while replaying the EmergencyStop scenario.
  preview_control_parameters.npz : the reference's precomputed Kajita gains
        /root/reference/src/data/PreviewControlParameters.ini  (Zc, T, preview time, Kx[3], Ks, F[320]), as data.
  kajita_zmpdisc_datref.npz : the columns of the reference's golden files
        /root/reference/tests/TestKajita2003{StraightWalking,PbFlorentSeq1,Circle}TestFGPI.datref.cmake
        that hold the outputs of ZMPDiscretization (tests/TestObject.cpp:344-385): time (1), left foot x y z (11-13),
        theta omega omega2 (20-22), right foot (23-25, 32-34), world-frame ZMP reference (35-36), as data, plus the
        step sequences of tests/TestKajita2003.cpp:95-152 (inputs).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root: herdt_replay takes the POD layouts from the package
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


STRAIGHT = [0.0, -0.105, 0.0] + [0.2, 0.21, 0.0, 0.2, -0.21, 0.0] * 7 + [0.0, 0.21, 0.0]
PBFLORENT1 = [0, 0.1, 0, -0.0398822, -0.232351, 4.6646, -0.0261703, 0.199677, 4.6646, -0.0471999, -0.256672, 4.6646,
              -0.0305785, 0.200634, 4.6646, -0.0507024, -0.245393, 4.6646, -0.0339626, 0.197227, 4.6646,
              -0.0527259, -0.228579, 4.6646, -0.0362332, 0.199282, 4.6646, -0.0540087, -0.21638, 4.6646,
              -0.0373302, 0.196611, 4.6646, -0.0536928, -0.199019, 4.6646, -0.0372245, 0.204021, 4.6646,
              -0.0529848, -0.196642, 4.6646, -0.0355124, 0.2163, 4.6646, -0.000858977, -0.204807, 0.0767924, 0, 0.2, 0]
KAJITA_COLS = [0, 10, 11, 12, 19, 20, 21, 22, 23, 24, 31, 32, 33, 34, 35]


def kajita_datrefs():
    out = {}
    for name, seq in (("StraightWalking", STRAIGHT), ("PbFlorentSeq1", PBFLORENT1), ("Circle", None)):
        d = np.loadtxt("/root/reference/tests/TestKajita2003%sTestFGPI.datref.cmake" % name)
        out[name + "_rows"] = d[:, KAJITA_COLS]
        if seq is not None:
            out[name + "_steps"] = np.array(seq, dtype=float).reshape(-1, 3)
    # TurningOnTheCircle's commands (tests/TestKajita2003.cpp:68-93): the steps come from StepStackHandler's generators
    out["Circle_commands"] = np.array([":supportfoot 1", ":arc 0.0 0.75 30.0 -1", ":lastsupport", ":finish"])
    out["columns"] = np.array(KAJITA_COLS) + 1
    np.savez_compressed(os.path.join(HERE, "kajita_zmpdisc_datref.npz"), **out)


def overdriven_tick_qp(gait=18, tick_wanted=87, B=32, scale=3.0):
    """The QP of tick 87 of gait 18 of tests/test_tick_gpu.py's overdriven scenario (the benchmark's recipe with three times its
    velocity references; portable-trig oracle): a state near 1e143, b up to 1e160, an iterate that passes 1e230 -- and, after
    ~600 iterations of the same two bounds going in and out, an entry of Z that has become DENORMAL: the Givens rotation that
    meets it has gb = q / norm = 0 by underflow and ga = -1, which the reference carries out (two columns change sign)."""
    import ctypes as C
    ol.build_oracle()
    pt = C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))
    model = hr.default_model()
    rng = np.random.default_rng(333)
    s = hr.init_state(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
    s.nb_steps_left = 2
    for tick in range(tick_wanted + 1):
        if tick % 50 == 0:
            for g in range(B):
                v = [scale * rng.uniform(-0.1, 0.3), scale * rng.uniform(-0.1, 0.1), scale * rng.uniform(-0.2, 0.2)]
                if g == gait:
                    s.vref[0], s.vref[1], s.vref[2] = v
        c = s.clock
        for _ in range(1 if tick == 0 else (19 if tick == 1 else 20)):
            c += model.Tctrl
        s.clock = c
        out, dump = hr.TickOut(), hr.QpDump()
        assert pt.wgo_mpc_tick(C.byref(model), C.byref(s), C.byref(out), C.byref(dump)) == 0
    n, m, mmax = dump.n, dump.m, dump.mmax
    assert (out.ifail, out.n_iter) == (1, 40 * (m + n) + 1)
    return dict(n=n, m=m, me=0, mmax=mmax, nmax=n, C=np.array(dump.C[:n * n]).reshape((n, n), order="F"), d=np.array(dump.d[:n]),
                A=np.array(dump.A[:mmax * n]).reshape((mmax, n), order="F"), b=np.array(dump.b[:mmax]),
                xl=np.full(n, -1e8), xu=np.full(n, 1e8))


def main():
    preview_ini()
    kajita_datrefs()
    datref = np.loadtxt(REF)
    np.savez_compressed(os.path.join(HERE, "herdt_emergency_stop_datref.npz"), datref=datref)

    recs = []

    def add(tag, q):
        r = ol.ref_ql(q)
        h = ol.ref_ql_hist(q)   # the instrumented build: same arithmetic, plus the add/drop log
        assert h["ifail"] == r["ifail"] and ol.same_bits(h["x"], r["x"]) and ol.same_bits(h["u"], r["u"])
        assert np.array_equal(h["iwar"], r["iwar"])
        # nact is a static local inside the reference: it is the history's balance (adds minus drops)
        nact = int((h["hist"] > 0).sum() - (h["hist"] < 0).sum())
        o = ol.oracle_ql(q)
        assert r["ifail"] != 0 or nact == o["nact"]
        recs.append(dict(tag=tag, n=q["n"], m=q["m"], me=q["me"], mmax=q["mmax"], C=q["C"], d=q["d"], A=q["A"], b=q["b"],
                         xl=q["xl"], xu=q["xu"], x=r["x"], u=r["u"], ifail=r["ifail"], iact=r["iwar"][:o["nact"]],
                         hist=h["hist"]))

    for fam in sorted(qpgen.FAMILIES):
        for s in range(24):
            add(fam, qpgen.FAMILIES[fam](np.random.default_rng(424200 + 977 * s)))

    for sd in (72, 1732, 3422, 3928, 4265, 5716, 5797):
        add("nonfinite_herdt_like", qpgen.herdt_like(np.random.default_rng(61000 + sd), 16, 2))
    add("nonfinite_infeasible", qpgen.FAMILIES["infeasible"](np.random.default_rng(5282)))
    add("overdriven_tick_denormal_rotation", overdriven_tick_qp())

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
