"""Why the outs-on launch of 20 ticks costs more than its bytes: launches of T ticks after a common warm-up, with outs NULL / a fresh
buffer / the same buffer again / a buffer cleared beforehand; time per QL iteration so that different tick ranges compare."""
import ctypes as C, importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "4096")); T = int(os.environ.get("PT", "20"))
model = wg.model_defaults(); wg.mpc_configure(model)
rng = np.random.default_rng(20100)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
st = torch.frombuffer(bytearray(bytes(memoryview(s0).cast("B")) * B), dtype=torch.uint8).cuda()
v = torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)).cuda()
wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1); wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 19)
wg.mpc_set_velref_dev(B, st.data_ptr(), v.data_ptr())
wg.mpc_run_batch_dev(B, st.data_ptr(), 100, 20, None, None)
osz = C.sizeof(wg.TickOut)
torch.cuda.synchronize()
def run(name, outs):
    diag = torch.zeros(T, B, 6, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); wg.mpc_run_batch_dev(B, st.data_ptr(), T, 20, outs.data_ptr() if outs is not None else None, diag.data_ptr()); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); its = float(diag[:, :, 1].double().sum().item())
    print("%-44s %8.3f ms  %5.2f M ticks/s  %6.2f its/tick  %7.3f ns per iteration" % (name, ms, B * T / ms / 1e3, its / (B * T), ms * 1e6 / its))
for rep in range(2):
    run("outs NULL", None)
    buf = torch.empty(T * B * osz, dtype=torch.uint8, device="cuda")
    run("outs: fresh torch.empty buffer", buf)
    run("outs: the same buffer again", buf)
    run("outs NULL", None)
    buf2 = torch.empty(T * B * osz, dtype=torch.uint8, device="cuda"); buf2.zero_(); torch.cuda.synchronize()
    run("outs: another buffer, cleared beforehand", buf2)
    run("outs NULL", None)
