"""Throughput probe of the fused tick kernel (device-resident states)."""
import ctypes as C, importlib, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "4096")); TICKS = int(os.environ.get("PT", "200"))
model = wg.model_defaults()
if os.environ.get("PN"): model.N = int(os.environ["PN"])       # other horizons: dense / element view
if os.environ.get("PFLAGS"): model.flags = int(os.environ["PFLAGS"])   # 4: Q_b from the fp32 MFMA Gramian (config 5)
wg.mpc_configure(model)
rng = np.random.default_rng(20100)
states = (wg.GaitState * B)()
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
for g in range(B): C.memmove(C.byref(states[g]), C.byref(s0), C.sizeof(wg.GaitState))
nb = C.sizeof(states)
host = torch.frombuffer(bytearray(bytes(memoryview(states).cast("B"))), dtype=torch.uint8)
dev = host.cuda()
diag = torch.zeros(B, 6, dtype=torch.int32, device="cuda")
def vrefs():
    v = np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)
    return torch.from_numpy(v).cuda()
print("lds bytes/gait", wg.mpc_tick_lds_bytes(), "state bytes", C.sizeof(wg.GaitState))
iters = []; t_total = 0.0
for tick in range(TICKS):
    if tick % 50 == 0:
        v = vrefs(); wg.mpc_set_velref_dev(B, dev.data_ptr(), v.data_ptr())
    adv = 1 if tick == 0 else (19 if tick == 1 else 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wg.mpc_tick_batch_dev(B, dev.data_ptr(), None, diag.data_ptr(), adv)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if tick >= 5: t_total += dt
    d = diag.cpu().numpy(); iters.append((d[:, 1].mean(), d[:, 1].max(), d[:, 2].mean(), (d[:, 0] != 0).sum(), dt * 1e3))
it = np.array(iters)
print("ticks/s %.0f  (ms per batch tick: mean %.3f)" % (B * (TICKS - 5) / t_total, 1e3 * t_total / (TICKS - 5)))
print("QL iterations mean %.2f max %d ; nact mean %.2f ; failed QPs total %d" % (it[:, 0].mean(), it[:, 1].max(), it[:, 2].mean(), it[:, 3].sum()))
for k in range(0, TICKS, max(1, TICKS // 10)): print(k, "iters mean %.1f max %d nact %.1f fails %d  %.3f ms" % tuple(it[k]))
