"""Pins the oracle's QL restatement (oracle/ql_oracle.c):
   1. against committed golden vectors produced by the COMPILED REFERENCE qld.cpp
      (tests/golden/ql_golden.npz, made by tests/golden/make_golden.py), bit for bit;
   2. where oracle/_ref/libqld_ref.so is present (it is built from the reference source
      by oracle/Makefile and travels with the repo), live against that library on fresh seeds.
"""
import os

import numpy as np
import pytest

import oraclelib as ol
import qpgen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ql_golden.npz")


def _golden():
    z = np.load(GOLD)
    for k in range(int(z["count"])):
        g = lambda key: z["%04d_%s" % (k, key)]
        q = dict(n=int(g("n")), m=int(g("m")), me=int(g("me")), mmax=int(g("mmax")), nmax=int(g("n")),
                 C=np.asfortranarray(g("C")), d=g("d").copy(), A=np.asfortranarray(g("A")), b=g("b").copy(),
                 xl=g("xl").copy(), xu=g("xu").copy())
        yield str(g("tag")), q, dict(x=g("x"), u=g("u"), ifail=int(g("ifail")), iact=g("iact"), hist=g("hist"))


def test_oracle_matches_reference_golden_vectors_bit_for_bit():
    n_checked = 0
    fails = set()
    for tag, q, ref in _golden():
        o = ol.oracle_ql(q)
        assert o["ifail"] == ref["ifail"], tag
        assert ol.same_bits_nan_aware(o["x"], ref["x"]), tag          # a NaN is a NaN: sign and payload are no contract
        if ref["ifail"] == 0:
            assert ol.same_bits(o["u"], ref["u"]), tag
            assert np.array_equal(o["iact"], ref["iact"]), tag
        fails.add(ref["ifail"] if ref["ifail"] < 3 else 11)
        n_checked += 1
    assert n_checked >= 450
    assert {0, 1, 2, 11} <= fails  # success, "maxit" (the iterate went NaN), "accuracy insufficient" and "inconsistent" exits all covered


def test_oracle_history_matches_reference_golden():
    """north_star: "bit-exact active-set index sequences".  The golden history is the REFERENCE's own (its qld.cpp with a
    log call at the add site :1766 and the drop site :1903, tests/golden/make_golden.py); every add and every drop of the
    restatement, in order, for successful and failed solves alike."""
    n_events = n_drops = longest = 0
    for tag, q, ref in _golden():
        o = ol.oracle_ql(q, hist_cap=16384)
        assert o["hist_len"] == len(ref["hist"]), tag
        assert np.array_equal(o["hist"], ref["hist"]), tag
        n_events += len(ref["hist"])
        n_drops += int((ref["hist"] < 0).sum())
        longest = max(longest, len(ref["hist"]))
    assert n_events >= 70000 and n_drops >= 30000 and longest >= 8877   # drops, long solves and the NaN regime's 8 877-event runs


@pytest.mark.skipif(not ol.have_ref_source(), reason="reference source not present (development container only)")
@pytest.mark.parametrize("family", sorted(qpgen.FAMILIES))
def test_oracle_history_matches_instrumented_reference_live(family, capfd):
    gen = qpgen.FAMILIES[family]
    for s in range(60):
        q = gen(np.random.default_rng(77000 + 13 * s))
        r = ol.ref_ql_hist(q)
        o = ol.oracle_ql(q)
        assert r["ifail"] == o["ifail"] and ol.same_bits(r["x"], o["x"]), (family, s)
        assert o["hist_len"] == len(r["hist"]) and np.array_equal(o["hist"], r["hist"]), (family, s)
    capfd.readouterr()


@pytest.mark.skipif(not ol.have_ref(), reason="compiled reference qld not present")
@pytest.mark.parametrize("family", sorted(qpgen.FAMILIES))
def test_oracle_matches_compiled_reference_live(family, capfd):
    gen = qpgen.FAMILIES[family]
    for s in range(60):
        q = gen(np.random.default_rng(99000 + 31 * s))
        r = ol.ref_ql(q)
        o = ol.oracle_ql(q)
        assert r["ifail"] == o["ifail"], (family, s)
        assert ol.same_bits(r["x"], o["x"]), (family, s)
        if r["ifail"] == 0:
            assert ol.same_bits(r["u"], o["u"]), (family, s)
            assert np.array_equal(r["iwar"][:o["nact"]], o["iact"]), (family, s)
        # the reference restores the caller's Hessian diagonal on exit (qld.cpp:1804-1809)
        assert ol.same_bits(r["C_after"], q["C"]) or q["C"][-1, -1] == 0.0
    capfd.readouterr()   # swallow the reference's printf diagnostics


def test_history_is_consistent_with_final_active_set():
    for tag, q, ref in _golden():
        if ref["ifail"] != 0:
            continue
        o = ol.oracle_ql(q)
        active = []
        for ev in o["hist"][:o["hist_len"]]:
            if ev > 0:
                active.append(int(ev))
            else:
                active.remove(int(-ev))
        assert sorted(active) == sorted(int(v) for v in o["iact"]), tag


def test_oracle_against_the_compiled_reference_on_the_fuzz_families():
    """A slice of tools/fuzz_oracle_vs_reference.py inside the suite: every family of tools/fuzz_ql.py -- the magnitude-scaled ones
    reach overflow, underflow and denormals inside the solver -- on seeds no other test uses, restatement against the COMPILED
    reference (oracle/_ref): ifail, x (a NaN matches a NaN), u and the final active set; the add / drop history as well where the
    reference's source is present (instrumented throw-away build).  The full run (322 000 QPs, 0 mismatches) is filed as
    profiles/round5_fuzz_oracle_vs_reference.txt."""
    import importlib.util
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_ql", os.path.join(root, "tools", "fuzz_ql.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    with_hist = ol.have_ref_source()
    n_checked = 0
    classes = set()
    for name, gen in fz.variants().items():
        for k in range(8 if name == "config5_sized" else 60):
            seed = 550000 + 15485863 * k % 1000003
            q = gen(np.random.default_rng(seed))
            o = ol.oracle_ql(q, hist_cap=20000)
            r = ol.ref_ql_hist(q) if with_hist else ol.ref_ql(q)
            assert o["ifail"] == r["ifail"], (name, seed)
            assert ol.same_bits_nan_aware(o["x"], r["x"]), (name, seed)
            if r["ifail"] == 0:
                assert ol.same_bits(o["u"], r["u"]) and np.array_equal(o["iact"], r["iwar"][:o["nact"]]), (name, seed)
            if with_hist:
                assert o["hist_len"] == len(r["hist"]) and np.array_equal(o["hist"][:o["hist_len"]], r["hist"]), (name, seed)
            classes.add(r["ifail"] if r["ifail"] < 3 else 11)
            n_checked += 1
    assert n_checked >= 900 and {0, 2, 11} <= classes
