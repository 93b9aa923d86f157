"""Throughput probe of the fused Dimitrov tick kernel: B receding-horizon gaits, device-resident states, polytope
windows gathered on the host per tick (the only per-tick input: 248 B x N per gait)."""
import ctypes as C, importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
import dimitrov as dv
B = int(os.environ.get("PB", "4096")); TICKS = int(os.environ.get("PT", "40")); NPLAN = 128
model = wg.dimitrov_defaults(); model.solver = int(os.environ.get("PSOLVER", "0")); wg.dimitrov_configure(model); N = model.N   # PSOLVER=2: the QL back-end (QLDANDLQ)
PT = np.dtype([("nrows", "i4"), ("pad", "i4"), ("similar", "i4", 8), ("A", "f8", (8, 2)), ("B", "f8", 8), ("centre", "f8", 2)])
assert PT.itemsize == C.sizeof(wg.ZmpPolytope)
L = 260
table = np.zeros((NPLAN, L), PT)
for p in range(NPLAN):
    slots = dv.plan(np.random.default_rng(20100 + p), n_steps=24)
    for k in range(L):
        A, Bv, c, sim = slots[min(k, len(slots) - 1)]
        r = len(Bv); e = table[p, k]
        e["nrows"] = r; e["similar"][:r] = sim; e["A"][:r] = A; e["B"][:r] = Bv; e["centre"] = c
plan_id = np.arange(B) % NPLAN; offs = (7 * np.arange(B)) % 23
ST = np.dtype([("xk", "f8", 6), ("pldp", "u1", C.sizeof(wg.PldpState)), ("n_removed", "i4"), ("starting", "i4")])
assert ST.itemsize == C.sizeof(wg.DimitrovState)
st = np.zeros(B, ST); st["starting"] = 1
dst = torch.from_numpy(st.view(np.uint8)).cuda()
OUT = C.sizeof(wg.DimitrovOut)
dout = torch.zeros(B * OUT, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
tot_ms = 0.0; n_t = 0
for it in range(TICKS):
    win = table[plan_id[:, None], (it + offs)[:, None] + np.arange(N)[None, :]]          # B x N
    dpoly = torch.from_numpy(np.ascontiguousarray(win).view(np.uint8)).cuda()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        rc = wg.lib().wg_dimitrov_tick_batch_dev(B, dpoly.data_ptr(), dst.data_ptr(), dout.data_ptr(), 0, stream.cuda_stream)
        e1.record(stream)
    assert rc == 0
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    o = np.frombuffer(dout.cpu().numpy().tobytes(), dtype=np.dtype([("jerk", "f8", 2), ("ret", "i4"), ("n_iter", "i4"), ("n_active", "i4"), ("m", "i4"), ("rest", "u1", OUT - 32)]))
    if it >= 3: tot_ms += ms; n_t += 1
    bad = o["ret"] != 0
    print("tick %2d  %.3f ms  iters mean %.2f max %d  active mean %.1f max %d  m mean %.0f  ret!=0 %d" %
          (it, ms, o["n_iter"].mean(), o["n_iter"].max(), o["n_active"].mean(), o["n_active"].max(), o["m"].mean(), bad.sum()))
    if bad.any():                                     # the reference would have exited: restart those gaits from rest
        h = np.frombuffer(dst.cpu().numpy().tobytes(), dtype=ST).copy()
        h["xk"][bad] = 0.0; h["starting"][bad] = 1; h["n_removed"][bad] = 0; h["pldp"][bad] = 0
        offs[bad] = -it - 1 + (offs[bad] % 5)
        dst = torch.from_numpy(h.view(np.uint8)).cuda()
print("Dimitrov ticks/s %.0f (%.3f ms per batch of %d); LDS/gait %d B" % (1e3 * B * n_t / tot_ms, tot_ms / max(1, n_t), B, 0))
