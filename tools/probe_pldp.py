"""Throughput probe of the PLDP/OptCholesky kernel: B receding-horizon gaits in lock-step, device-resident buffers,
hot-started solves (what ZMPConstrainedQPFastFormulation's loop does per 0.1 s), timed with events on the launch stream."""
import ctypes as C, importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
import dimitrov as dv
B = int(os.environ.get("PB", "4096")); TICKS = int(os.environ.get("PT", "30")); NPLAN = 64
dm = dv.Dimitrov(); N = dm.N; n = 2 * N; MC = wg.PLDP_MMAX
wg.pldp_configure(N, dm.iPu, dm.Px, dm.Pu)
plans = [dv.plan(np.random.default_rng(20100 + g), n_steps=12) for g in range(NPLAN)]
offs = np.array([(7 * g) % 23 for g in range(B)])
xk = np.zeros((B, 6))
alive = np.ones(B, bool)


def build(it):
    m = np.zeros(B, np.int32); D = np.zeros((B, n)); A = np.zeros((B, (MC + 1) * n)); b = np.zeros((B, MC))
    z = np.zeros((B, n)); sim = np.zeros((B, MC), np.int32); first = np.zeros(B, np.int32)
    for g in range(B):
        polys = dv.polys_at(plans[g % NPLAN], it + offs[g], N)
        Ax = np.concatenate([p[0][:, 0] for p in polys]); Ay = np.concatenate([p[0][:, 1] for p in polys])
        Bv = np.concatenate([p[1] for p in polys]); slot = np.concatenate([np.full(len(p[1]), i) for i, p in enumerate(polys)])
        mm = len(Bv); m[g] = mm; first[g] = len(polys[0][1])
        sim[g, :mm] = np.concatenate([p[3] for p in polys])
        zx = dm.Px @ xk[g, :3]; zy = dm.Px @ xk[g, 3:]
        b[g, :mm] = zx[slot] * Ax + zy[slot] * Ay + Bv
        PuS = dm.Pu[:, slot]                                  # N x mm : Pu[k, slot[row]]
        Am = np.zeros((n, mm + 1)); Am[:N, :mm] = PuS * Ax; Am[N:, :mm] = PuS * Ay
        A[g, :(mm + 1) * n] = Am.reshape(-1)
        for i, p in enumerate(polys): z[g, i], z[g, i + N] = p[2]
        D[g] = dm.OptB @ xk[g] - dm.OptC @ z[g]
    return m, D, A, b, z, sim, first


st = torch.zeros(B * C.sizeof(wg.PldpState), dtype=torch.uint8, device="cuda")
X = torch.zeros(B, n, dtype=torch.float64, device="cuda"); ret = torch.zeros(B, dtype=torch.int32, device="cuda")
nit = torch.zeros(B, dtype=torch.int32, device="cuda"); nact = torch.zeros(B, dtype=torch.int32, device="cuda")
act = torch.zeros(B, MC, dtype=torch.int32, device="cuda")
n_removed = np.zeros(B, np.int32); starting = np.ones(B, np.int32)
stream = torch.cuda.Stream()
print("lds bytes/problem", wg.pldp_lds_bytes())
tot_ms = 0.0; tot_solves = 0
for it in range(TICKS):
    m, D, A, b, z, sim, first = build(it)
    t = lambda a: torch.from_numpy(a).cuda()
    dm_, dD, dA, db, dz, dx, dsim, dnr, dst = t(m), t(D), t(A), t(b), t(z), t(xk.copy()), t(sim), t(n_removed), t(starting)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        rc = wg.lib().wg_pldp_solve_batch_dev(B, MC, dm_.data_ptr(), dD.data_ptr(), dA.data_ptr(), db.data_ptr(), dz.data_ptr(),
                                              dx.data_ptr(), dsim.data_ptr(), dnr.data_ptr(), dst.data_ptr(), 0, st.data_ptr(),
                                              X.data_ptr(), ret.data_ptr(), nit.data_ptr(), act.data_ptr(), nact.data_ptr(),
                                              stream.cuda_stream)
        e1.record(stream)
    assert rc == 0
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    r = ret.cpu().numpy(); Xh = X.cpu().numpy(); ni = nit.cpu().numpy(); na = nact.cpu().numpy()
    ok = alive & (r == 0)
    if it >= 3: tot_ms += ms; tot_solves += B
    print("tick %2d  %.3f ms  iters mean %.2f max %d  nact mean %.1f max %d  alive %d  ret!=0 %d" %
          (it, ms, ni[alive].mean(), ni[alive].max(), na[alive].mean(), na[alive].max(), alive.sum(), (r[alive] != 0).sum()))
    for g in np.nonzero(ok)[0]: xk[g] = dm.step(xk[g], Xh[g])
    dead = alive & (r != 0)
    # a gait whose solve hit the reference's exit(0) condition is restarted from rest (fresh starting sequence)
    xk[dead] = 0.0; offs[dead] = (offs[dead] * 0) - it - 1
    n_removed = first.copy(); starting[:] = 0; starting[dead] = 1
print("PLDP solves/s %.0f (%.3f ms per batch of %d)" % (1e3 * tot_solves / tot_ms, tot_ms / max(1, TICKS - 3), B))
