"""Multi-tick launches (wg_mpc_run_batch_dev, device-side work queue) against one launch per tick: same bytes, and the rate."""
import ctypes as C, importlib, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "4096")); T = int(os.environ.get("PT", "50")); REPS = 4
model = wg.model_defaults()
if os.environ.get("PN"): model.N = int(os.environ["PN"])
wg.mpc_configure(model)
rng = np.random.default_rng(20100)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
one = bytes(memoryview(s0).cast("B"))
def fresh():
    return torch.frombuffer(bytearray(one * B), dtype=torch.uint8).cuda()
def vref():
    return torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)).cuda()
a = fresh(); b = fresh()
da = torch.zeros(REPS * T, B, 6, dtype=torch.int32, device="cuda"); db = torch.zeros_like(da)
# warm-up ticks with the special clock advances (both copies the same way)
for st in (a, b):
    wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1); wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 19)
vs = [vref() for _ in range(REPS)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for r in range(REPS):
    wg.mpc_set_velref_dev(B, a.data_ptr(), vs[r].data_ptr())
    for t in range(T):
        wg.mpc_tick_batch_dev(B, a.data_ptr(), None, da[r * T + t].data_ptr(), 20)
torch.cuda.synchronize(); t1 = time.perf_counter()
for r in range(REPS):
    wg.mpc_set_velref_dev(B, b.data_ptr(), vs[r].data_ptr())
    wg.mpc_run_batch_dev(B, b.data_ptr(), T, 20, None, db[r * T].data_ptr())
torch.cuda.synchronize(); t2 = time.perf_counter()
same_s = bool(torch.equal(a, b)); same_d = bool(torch.equal(da, db))
same = same_s and same_d
if not same:
    A = a.cpu().numpy().reshape(B, -1); Bm = b.cpu().numpy().reshape(B, -1)
    bad = np.nonzero((A != Bm).any(1))[0]
    print("states equal", same_s, "diag equal", same_d, "gaits differing", len(bad), bad[:10])
    D = (da != db).any(-1).cpu().numpy()
    print("first differing tick", np.nonzero(D.any(1))[0][:5], "n diag cells", int(D.sum()))
    g = int(bad[0]); xa = da[:, g, 1].cpu().numpy(); xb = db[:, g, 1].cpu().numpy()
    print("gait", g, "n_iter per tick, per-tick launches:", xa[40:56].tolist())
    print("gait", g, "n_iter per tick, work queue       :", xb[40:56].tolist())
    off = 1208 - 64
    ta = np.frombuffer(A[g].tobytes(), dtype=np.int32); tb_ = np.frombuffer(Bm[g].tobytes(), dtype=np.int32)
    print("int fields differing:", np.nonzero(ta != tb_)[0][:20].tolist())
print(f"B={B} T={T}: per-tick launches {B*REPS*T/(t1-t0):.0f} ticks/s, work-queue launches {B*REPS*T/(t2-t1):.0f} ticks/s, identical={same}")
if not same:
    A = a.cpu().numpy().reshape(B, -1); Bm = b.cpu().numpy().reshape(B, -1)
    bad = np.nonzero((A != Bm).any(1))[0]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "run_bad.npz"), bad=bad, per_tick=A[bad], queue=Bm[bad],
             vs=np.stack([v.cpu().numpy() for v in vs])[:, bad])
assert same
