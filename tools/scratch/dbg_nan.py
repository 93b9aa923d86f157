import ctypes as C, importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpgen, oraclelib as ol
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
qps = [qpgen.herdt_like(np.random.default_rng(61000 + s), 16, 2) for s in (72, 1732, 5)]
pk = wg.pack_qps(qps)
for env in ({}, {"WG_QL_FIXED": "0"}):
    os.environ.update(env)
    res = wg.qp_solve_batch(pk, hist_cap=8192)
    for k in range(3):
        o = ol.oracle_ql(dict(qps[k], nmax=36, mmax=76), hist_cap=8192)
        hl = int(res["hist_len"][k])
        h = res["hist"][k, :min(hl, 60)]
        oh = o["hist"][:60]
        first_diff = next((i for i in range(min(len(h), len(oh))) if h[i] != oh[i]), None)
        print(env, k, "gpu ifail", int(res["ifail"][k]), "nit", int(res["n_iter"][k]), "hist_len", hl, "| oracle ifail", o["ifail"], "nit", o["n_iter"], "hist_len", o["hist_len"], "first diff at", first_diff)
        if first_diff is not None: print("   gpu", h[max(0,first_diff-3):first_diff+8], "\n   ora", oh[max(0,first_diff-3):first_diff+8])
        print("   x bits equal", ol.same_bits(res["x"][k,:36], o["x"]), "gpu x[:4]", res["x"][k,:4], "ora", o["x"][:4])
