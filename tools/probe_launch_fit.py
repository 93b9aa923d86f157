"""Multi-tick launches of the bench workload at several lengths: duration(T) = a + b * T (least squares).  b is the steady rate,
a what every launch pays once (queue set-up, ramp, the tail in which the last gait-ticks finish on a chip that is running empty).
PB = batch (4096), PN = horizon (16), PTS = comma-separated launch lengths."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "4096"))
TS = [int(x) for x in os.environ.get("PTS", "1,2,4,8,12,20,30,50").split(",")]
model = wg.model_defaults(); model.N = int(os.environ.get("PN", "16"))
wg.mpc_configure(model)
rng = np.random.default_rng(20100)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
st = torch.frombuffer(bytearray(bytes(memoryview(s0).cast("B")) * B), dtype=torch.uint8).cuda()
def vref():
    return torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)).cuda()
wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1); wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 19)
v = vref(); wg.mpc_set_velref_dev(B, st.data_ptr(), v.data_ptr())
wg.mpc_run_batch_dev(B, st.data_ptr(), 60, 20, None, None)
torch.cuda.synchronize()
rows = []
for rep in range(3):
    for T in TS:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); wg.mpc_run_batch_dev(B, st.data_ptr(), T, 20, None, None); e1.record()
        torch.cuda.synchronize()
        rows.append((T, e0.elapsed_time(e1)))
r = np.array(rows)
for T in TS:
    ms = r[r[:, 0] == T, 1]
    print("T=%3d: %8.3f ms (min of %d; max %.3f)  %.0f ticks/s" % (T, ms.min(), len(ms), ms.max(), B * T / ms.min() * 1e3))
mins = np.array([r[r[:, 0] == T, 1].min() for T in TS])
A = np.stack([np.ones(len(TS)), np.array(TS, float)], 1)
sel = np.array(TS) >= 4
coef, *_ = np.linalg.lstsq(A[sel], mins[sel], rcond=None)
print("B=%d N=%d: duration = %.3f ms + %.4f ms per tick (T >= 4): steady %.0f ticks/s, the per-launch part = %.2f ticks" %
      (B, model.N, coef[0], coef[1], B / coef[1] * 1e3, coef[0] / coef[1]))
