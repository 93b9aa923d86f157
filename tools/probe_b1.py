"""One robot (B = 1): 200 walking ticks through wg_mpc_tick_batch_dev, one launch per tick of ONE wave -- for the rocprofv3 passes of
tools/prof.sh (what a wave that is alone on its CU spends its cycles on: issuing, or waiting)."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
model = wg.model_defaults(); wg.mpc_configure(model)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
s0.vref[0], s0.vref[1], s0.vref[2] = 0.2, 0.02, 0.05
st = torch.frombuffer(bytearray(bytes(memoryview(s0).cast("B"))), dtype=torch.uint8).cuda()
diag = torch.zeros(1, 6, dtype=torch.int32, device="cuda")
T = int(os.environ.get("PT", "200"))
its = []
ev = []
for t in range(T):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); wg.mpc_tick_batch_dev(1, st.data_ptr(), None, diag.data_ptr(), 1 if t == 0 else (19 if t == 1 else 20)); e1.record()
    torch.cuda.synchronize(); ev.append(e0.elapsed_time(e1) * 1e3); its.append(int(diag[0, 1].item()))
print("B=1: %d ticks, kernel (HIP events) median %.1f us, mean QL iterations %.1f" % (T, float(np.median(ev[20:])), float(np.mean(its[20:]))), flush=True)
