"""Diagnostic: per-phase shader cycles of the QL kernel (libwg_mpc_prof.so)."""
import ctypes as C, importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import qpgen
wgmod = importlib.import_module("jrl-walkgen_amd")
import importlib as il
wm = il.import_module("jrl-walkgen_amd.wgmpc")
wm.LIB_PATH = os.path.join(ROOT, "jrl-walkgen_amd", "lib", "libwg_mpc_prof.so")
wm._lib = None
wm.init(0)
B = int(os.environ.get("PB", "256"))
fam = os.environ.get("PFAM", "herdt_like")
if fam == "herdt_like":
    qps = [qpgen.herdt_like(np.random.default_rng(5 + s), 16, 2) for s in range(B)]
else:
    qps = [qpgen.FAMILIES[fam](np.random.default_rng(5 + s)) for s in range(B)]
pk = wm.pack_qps(qps)
buf = (C.c_ulonglong * 48)()
wm.lib().wg_prof_read(buf)
res = wm.qp_solve_batch(pk)
wm.lib().wg_prof_read(buf)
v = np.array(list(buf), dtype=np.float64)
names = ["norms", "diagchk", "chol", "inverse", "resid/reset+shift", "ZT*ww(resid)", "x-shift", "backsub+lam(resid)", "xmag(resid)",
         "scan", "fdiff/wx", "newnormal ZTa", "sweep", "route sums", "step-pre", "backsub(step)", "pickdrop", "step/upd/drop", "add", "xmag(add)", "tail"]
its = res["n_iter"].sum()
print(f"QPs={B} total iters={its} mean iters={its/B:.1f} cycles/QP={v.sum()/B:.0f}")
for k, nme in enumerate(names):
    print(f"{k:2d} {nme:22s} {v[k]/B:12.0f} cyc/QP  {100*v[k]/v.sum():5.1f}%   {v[k]/its:9.0f} cyc/iter")
