"""The ql0001_ boundary on the Herdt workload's real QPs (bench_kernels.ql_dense_on_real_qps) alone: PB QPs, placements by
WG_QL_A_IN_LDS / WG_QL_G_IN_LDS / WG_QL_W_IN_LDS; WG_QL_LPT=0: index order instead of longest-solve-first; PSAME=1: the same batch
solved three times (rounds 1 - 4's form) instead of three consecutive ticks' QPs."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_kernels as bk
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "4096"))
model = wg.model_defaults(); wg.mpc_configure(model)
rng = np.random.default_rng(20100)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
st = torch.frombuffer(bytearray(bytes(memoryview(s0).cast("B")) * B), dtype=torch.uint8).cuda()
v = torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)).cuda()
wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1); wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 19)
wg.mpc_set_velref_dev(B, st.data_ptr(), v.data_ptr())
wg.mpc_run_batch_dev(B, st.data_ptr(), 120, 20, None, None)
torch.cuda.synchronize()
stream = torch.cuda.current_stream()
alg = lambda n, m: 8.0 * (n * n + n + (m + 1) * n + (m + 1) + 2 * n) + 8.0 * (n + m + 2 * n)
r = bk.ql_dense_on_real_qps(wg, torch.device("cuda:0"), stream, B, st.data_ptr() if os.environ.get("PSAME") else st, 16, alg)
print("B=%d: %.0f QPs/s, %.3f ms per launch, %.1f iterations, failed %d (W_IN_LDS=%s, LPT=%s, %s)" %
      (B, r["value"], r["kernel_ms"], r["mean_iterations"], r["failed_qps"], os.environ.get("WG_QL_W_IN_LDS", "default"),
       os.environ.get("WG_QL_LPT", "default"), "same batch x 3" if os.environ.get("PSAME") else "three consecutive ticks"))
