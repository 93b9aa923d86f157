"""Element-view (N = 32, BASELINE configs[4]) throughput probe: B gaits, multi-tick launches of wg_mpc_run_batch_dev only, timed
with events on the launch stream.  Knobs: PB (8192), PT ticks per launch (50), PR launches (3), WG_TICK_LDS_PAD (bytes of LDS
added per gait: lowers the residency), WG_LIB_PATH (an experiment build of the library), PQT (QP sampling period).  Run under rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE passes for the traffic per gait-tick (tools/elem_sweep.sh)."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "8192")); T = int(os.environ.get("PT", "50")); REPS = int(os.environ.get("PR", "3"))
model = wg.model_defaults(); model.N = int(os.environ.get("PN", "32"))
if "PQT" in os.environ:                                   # QP sampling period (short horizons need a longer one to preview a step)
    model.T = float(os.environ["PQT"]); model.t_double = model.T; model.Tctrl = model.T / 20.0
wg.mpc_configure(model)
lds = wg.lib().wg_mpc_tick_lds_bytes() + int(os.environ.get("WG_TICK_LDS_PAD", "0"))
per_cu = min(int(os.environ.get("PMAXW", "8" if model.N == 16 and "WG_TICK_VIEW" not in os.environ else "12")), 128 // ((lds + 1279) // 1280))
rng = np.random.default_rng(20100)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
st = torch.frombuffer(bytearray(bytes(memoryview(s0).cast("B")) * B), dtype=torch.uint8).cuda()
def vref():
    return torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)).cuda()
diag = torch.zeros(REPS * T, B, 6, dtype=torch.int32, device="cuda")
wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1); wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 19)
v = vref(); wg.mpc_set_velref_dev(B, st.data_ptr(), v.data_ptr())
wg.mpc_run_batch_dev(B, st.data_ptr(), 10, 20, None, None)
torch.cuda.synchronize()
ms = []
for r in range(REPS):
    v = vref(); wg.mpc_set_velref_dev(B, st.data_ptr(), v.data_ptr())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); wg.mpc_run_batch_dev(B, st.data_ptr(), T, 20, None, diag[r * T].data_ptr()); e1.record()
    torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
d = diag.cpu().numpy().reshape(-1, 6)
na = d[:, 2]
print("final active-set sizes: mean %.1f, p50 %d, p90 %d, p99 %d, max %d" % (na.mean(), np.percentile(na, 50), np.percentile(na, 90), np.percentile(na, 99), na.max()), flush=True)
print("N=%d B=%d T=%d lds/gait %d B -> %d gaits per CU: launches %s ms -> %.0f ticks/s (best %.0f); mean QL iterations %.1f, failed %d, state checksum %016x"
      % (model.N, B, T, lds, per_cu, ["%.1f" % m for m in ms], B * T * REPS / sum(ms) * 1e3, B * T / min(ms) * 1e3, d[:, 1].mean(),
         int((d[:, 0] != 0).sum()), int(st.cpu().numpy().view(np.uint64).sum(dtype=np.uint64))), flush=True)
