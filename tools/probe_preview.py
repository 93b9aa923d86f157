"""Diagnostic: throughput of the batched Kajita stage-1 preview kernel (resident, time-major inputs)."""
import importlib, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
g, F = wg.preview_gains(0.005, 0.814, 1.6)
wg.preview_configure(g, F)
L = int(os.environ.get("PL", "200"))
print("kernel:", os.environ.get("WG_PREVIEW_KERNEL", "default"))
for B in (4096, 32768, 131072):
    Lz = L + g.nl - 1
    t = torch.arange(Lz, device="cuda", dtype=torch.float64)[:, None]
    ph = torch.rand(1, B, device="cuda", dtype=torch.float64)
    zx = (0.2 * torch.floor(t * 0.005 / 0.8 + ph)).contiguous(); zy = (0.1 * torch.sign(torch.sin(t * 0.005 * 3.9 + 6.28 * ph))).contiguous()
    st = torch.zeros(B, 8, device="cuda", dtype=torch.float64)
    com = torch.zeros(L, 6, B, device="cuda", dtype=torch.float64); z2 = torch.zeros(L, 2, B, device="cuda", dtype=torch.float64)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        wg.preview_run_batch_dev(B, L, zx.data_ptr(), zy.data_ptr(), st.data_ptr(), com.data_ptr(), z2.data_ptr())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flops = B * L * 2 * (2 * g.nl + 40)
    print(f"B={B} L={L}: {dt*1e3:.2f} ms  {B*L/dt/1e9:.3f} G gait-steps/s  {flops/dt/1e12:.2f} TFLOP/s (non-fused mul+add)  "
          f"out {B*L*64/dt/1e9:.1f} GB/s", flush=True)
