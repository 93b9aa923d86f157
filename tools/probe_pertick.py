"""One launch per tick (wg_mpc_tick_batch_dev), the closed-loop mode of a fleet whose references change every tick: B gaits,
PT timed launches after a pre-roll; WG_TICK_LPT=0 starts the gaits in index order, the default longest-solve-first.  Prints the
state checksum (the order is scheduling only)."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "4096")); T = int(os.environ.get("PT", "100"))
model = wg.model_defaults(); model.N = int(os.environ.get("PN", "16"))
wg.mpc_configure(model)
rng = np.random.default_rng(20100)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
st = torch.frombuffer(bytearray(bytes(memoryview(s0).cast("B")) * B), dtype=torch.uint8).cuda()
v = torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)).cuda()
wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1); wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 19)
wg.mpc_set_velref_dev(B, st.data_ptr(), v.data_ptr())
for _ in range(60): wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 20)
torch.cuda.synchronize()
diag = torch.zeros(T, B, 6, dtype=torch.int32, device="cuda")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(T): wg.mpc_tick_batch_dev(B, st.data_ptr(), None, diag[t].data_ptr(), 20)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = diag.cpu().numpy()
print("N=%d B=%d: %d launches, %.3f ms per launch -> %.0f ticks/s; mean QL iterations %.1f, failed %d, WG_TICK_LPT=%s, state checksum %016x"
      % (model.N, B, T, ms / T, B * T / ms * 1e3, d[:, :, 1].mean(), int((d[:, :, 0] != 0).sum()), os.environ.get("WG_TICK_LPT", "default"),
         int(st.cpu().numpy().view(np.uint64).sum(dtype=np.uint64))))
