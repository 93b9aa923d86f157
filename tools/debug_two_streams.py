"""which of two concurrent dense-QP launches of one context goes wrong (tests/test_ql_gpu.py::test_two_streams_share_one_context)"""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpgen
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
qps = [qpgen.herdt_like(np.random.default_rng(31000 + s), 16, 2) for s in range(3000)]
pk = wg.pack_qps(qps)
ref = wg.qp_solve_batch(pk, hist_cap=64)
B, nmax, mmax = pk["B"], pk["nmax"], pk["mmax"]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dev_in = {k: t(pk[k]) for k in ("n", "m", "me", "C", "d", "A", "b", "xl", "xu")}
for rep in range(int(os.environ.get("REPS", "4"))):
    outs = []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in streams:
        x = torch.zeros(B, nmax, dtype=torch.float64, device="cuda"); u = torch.zeros(B, mmax + 2 * nmax, dtype=torch.float64, device="cuda")
        ifail = torch.full((B,), -99, dtype=torch.int32, device="cuda"); nit = torch.zeros(B, dtype=torch.int32, device="cuda")
        outs.append((x, u, ifail, nit))
    torch.cuda.synchronize()
    for st, (x, u, ifail, nit) in zip(streams, outs):
        wg.qp_solve_batch_dev(B, nmax, mmax, dev_in["n"], dev_in["m"], dev_in["me"], dev_in["C"], dev_in["d"], dev_in["A"], dev_in["b"],
                              dev_in["xl"], dev_in["xu"], 1e-8, x, u, ifail, nit, stream=st.cuda_stream)
    torch.cuda.synchronize()
    for k, (x, u, ifail, nit) in enumerate(outs):
        bad = np.nonzero((x.cpu().numpy() != ref["x"]).any(axis=1))[0]
        print("rep %d launch %d: %d of %d QPs differ%s" % (rep, k, len(bad), B, (" first " + str(bad[:8])) if len(bad) else ""), flush=True)
