import sys, time, importlib, numpy as np, torch
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,ROOT)
import qpgen
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B=4096
uniq=[qpgen.herdt_like(np.random.default_rng(5+s),16,2) for s in range(256)]
qps=[uniq[i%256] for i in range(B)]
pk=wg.pack_qps(qps)
dev='cuda'
t=lambda a: torch.from_numpy(a).to(dev)
C,d,A,b,xl,xu=[t(pk[k]) for k in ("C","d","A","b","xl","xu")]
x=torch.zeros(B,pk["nmax"],dtype=torch.float64,device=dev); u=torch.zeros(B,pk["mmax"]+2*pk["nmax"],dtype=torch.float64,device=dev)
ifail=torch.zeros(B,dtype=torch.int32,device=dev); nit=torch.zeros(B,dtype=torch.int32,device=dev)
def run():
    wg.qp_solve_batch_dev(B,pk["nmax"],pk["mmax"],None,None,None,C,d,A,b,xl,xu,1e-8,x,u,ifail,nit)
run(); torch.cuda.synchronize()
for rep in range(3):
    t0=time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/10
    print(f"B={B} ms={dt*1e3:.3f} qp/s={B/dt:.0f}")
print("ifail==0:", int((ifail==0).sum()), "mean iters", nit.double().mean().item(), "max", nit.max().item())
print("lds bytes", wg.qp_lds_bytes(36,75))
