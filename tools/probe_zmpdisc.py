"""Diagnostic: throughput of the batched ZMPDiscretization kernel, alone and chained with the preview kernel
(step sequences in, CoM trajectories out, nothing leaves the device)."""
import importlib, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
m = wg.zmpdisc_defaults()
g, F = wg.preview_gains(0.005, 0.814, 1.6)
wg.preview_configure(g, F)
S = int(os.environ.get("PS", "16"))
rng = np.random.default_rng(1)
for B in [int(b) for b in os.environ.get("PB", "4096,32768").split(",")]:
    tr = np.zeros((B, S, 3))
    side = rng.choice([-1.0, 1.0], B)
    for i in range(S):
        tr[:, i, 0] = 0.0 if i == 0 else rng.uniform(0.1, 0.25, B)
        tr[:, i, 1] = side * (0.105 if i == 0 else 0.21)
        tr[:, i, 2] = 0.0 if i == 0 else rng.uniform(-5, 5, B)
        side = -side
    rec = np.zeros((B, S), dtype=[("sx", "f8"), ("sy", "f8"), ("theta", "f8"), ("ss", "f8"), ("ds", "f8"), ("ty", "i4"), ("pad", "i4")])
    rec["sx"], rec["sy"], rec["theta"] = tr[:, :, 0], tr[:, :, 1], tr[:, :, 2]
    rec["ss"], rec["ds"], rec["ty"] = m.t_single, m.t_double, 1
    steps = (wg.RelStep * S).from_buffer_copy(rec[0].tobytes())
    L = wg.zmpdisc_length(m, steps)
    d_steps = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda()
    d_ns = torch.full((B,), S, dtype=torch.int32, device="cuda")
    init = np.tile(np.array([0.0, 0.095, 0.0, 0.0, -0.095, 0.0]), (B, 1))
    d_init = torch.from_numpy(init).cuda()
    zx = torch.zeros(L, B, dtype=torch.float64, device="cuda"); zy = torch.zeros_like(zx)
    d_len = torch.zeros(B, dtype=torch.int32, device="cuda")
    Lrun = L - g.nl + 1
    st = torch.zeros(B, 8, dtype=torch.float64, device="cuda")
    com = torch.zeros(Lrun, 6, B, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        st.zero_()
        e0.record()
        wg.zmpdisc_batch_dev(m, B, S, d_steps.data_ptr(), d_ns.data_ptr(), d_init.data_ptr(), L, zx.data_ptr(), zy.data_ptr(),
                             d_len.data_ptr(), stream)
        e1.record()
        wg.preview_run_batch_dev(B, Lrun, zx.data_ptr(), zy.data_ptr(), st.data_ptr(), com.data_ptr(), None, stream=stream)
        e2.record()
        torch.cuda.synchronize()
    tz, tp = e0.elapsed_time(e1) * 1e-3, e1.elapsed_time(e2) * 1e-3
    assert int(d_len.min()) == L == int(d_len.max())
    print(f"B={B} S={S} L={L}: zmpdisc {tz*1e3:.2f} ms = {B*L/tz/1e9:.2f} G gait-samples/s, {B*L*16/tz/1e9:.0f} GB/s of queue "
          f"written ({B*L*16/tz/8e12*100:.1f} % of HBM peak); preview {tp*1e3:.1f} ms; steps -> CoM {B/(tz+tp):.0f} gaits/s, "
          f"final CoM x mean {float(com[-1, 0].mean()):.3f}", flush=True)
