import ctypes as C, importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpgen
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = 2304
qps = [qpgen.herdt_like(np.random.default_rng(61000 + s), 16, 2) for s in range(B)]
pk = wg.pack_qps(qps)
dev = {k: torch.from_numpy(np.ascontiguousarray(pk[k])).cuda() for k in ("C", "d", "A", "b", "xl", "xu")}
ints = {k: torch.from_numpy(np.ascontiguousarray(pk[k]).astype(np.int32)).cuda() for k in ("n", "m", "me")}
def solve():
    x = torch.zeros(B, 36, dtype=torch.float64, device="cuda"); u = torch.zeros(B, 76 + 72, dtype=torch.float64, device="cuda")
    ifail = torch.full((B,), -9, dtype=torch.int32, device="cuda"); nit = torch.zeros(B, dtype=torch.int32, device="cuda")
    iact = torch.zeros(B, 36, dtype=torch.int32, device="cuda"); nact = torch.zeros(B, dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = wg.lib().wg_qp_solve_batch_dev(B, 36, 76, p(ints["n"]), p(ints["m"]), p(ints["me"]), p(dev["C"]), p(dev["d"]), p(dev["A"]),
                                        p(dev["b"]), p(dev["xl"]), p(dev["xu"]), C.c_double(1e-8), p(x), p(u), p(ifail), p(nit), p(iact), p(nact), None, 0, None, None)
    assert rc == 0
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (x, u, ifail, nit, iact, nact)]
names = ["x", "u", "ifail", "nit", "iact", "nact"]
runs = [solve() for _ in range(3)]
os.environ["WG_QL_LPT"] = "0"
runs.append(solve()); runs.append(solve())
for r, lab in zip(runs[1:], ["second", "third", "plain", "plain2"]):
    for nm, a, b in zip(names, runs[0], r):
        if not np.array_equal(a, b):
            bad = np.unique(np.argwhere(a != b)[:, 0])
            print(lab, nm, "differs in", len(bad), "QPs, first", bad[:8], "ifail there", runs[0][2][bad[:8]], r[2][bad[:8]], "nit", runs[0][3][bad[:8]], r[3][bad[:8]])
print("ifail hist", np.unique(runs[0][2], return_counts=True), "nit range", runs[0][3].min(), runs[0][3].max())
import oraclelib as ol
sys.path.insert(0, os.path.join(ROOT, "tests"))
for k in (72, 1732):
    q = qps[k]
    o = ol.oracle_ql(dict(n=q["n"], m=q["m"], me=q["me"], nmax=36, mmax=76, C=np.asfortranarray(pk["C"][k].reshape((36, 36), order="F")),
                          A=np.asfortranarray(pk["A"][k].reshape((76, 36), order="F")), d=pk["d"][k].copy(), b=pk["b"][k].copy(), xl=pk["xl"][k].copy(), xu=pk["xu"][k].copy()))
    for lab, r in (("first", runs[0]), ("plain", runs[3])):
        print(k, lab, "x == oracle:", ol.same_bits(r[0][k, :36], o["x"]), "max|dx|", np.abs(r[0][k, :36] - o["x"]).max(), "nit", r[3][k], o["n_iter"], "nact", r[5][k], o["nact"],
              "iact eq", np.array_equal(r[4][k, :o["nact"]], o["iact"]))
