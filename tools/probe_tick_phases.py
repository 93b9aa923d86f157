"""Diagnostic: per-phase shader cycles of the tick kernel on the bench workload (libwg_mpc_prof.so)."""
import ctypes as C, importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WG_LIB_PATH"] = os.environ.get("WG_PROF_LIB", os.path.join(ROOT, "jrl-walkgen_amd", "lib", "libwg_mpc_prof.so"))
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "1024")); WARM = 60; MEAS = 20
model = wg.model_defaults()
if os.environ.get("PN"): model.N = int(os.environ["PN"])
wg.mpc_configure(model)
rng = np.random.default_rng(20100)
states = (wg.GaitState * B)()
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
for g in range(B): C.memmove(C.byref(states[g]), C.byref(s0), C.sizeof(wg.GaitState))
dev = torch.frombuffer(bytearray(bytes(memoryview(states).cast("B"))), dtype=torch.uint8).cuda()
diag = torch.zeros(B, 6, dtype=torch.int32, device="cuda")
buf = (C.c_ulonglong * 48)()
its = 0
for tick in range(WARM + MEAS):
    if tick % 50 == 0:
        v = torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)).cuda()
        wg.mpc_set_velref_dev(B, dev.data_ptr(), v.data_ptr())
    if tick == WARM:
        torch.cuda.synchronize(); wg.lib().wg_prof_read(buf)
    wg.mpc_tick_batch_dev(B, dev.data_ptr(), None, diag.data_ptr(), 1 if tick == 0 else (19 if tick == 1 else 20))
    if tick >= WARM:
        torch.cuda.synchronize(); its += int(diag[:, 1].sum().item())
torch.cuda.synchronize(); wg.lib().wg_prof_read(buf)
v = np.array(list(buf), dtype=np.float64); n = B * MEAS
names = ["norms", "diagchk", "chol", "inverse", "resid/reset+shift", "ZT*ww(resid)", "x-shift", "backsub+lam(resid)", "xmag(resid)",
         "scan", "fdiff/wx", "newnormal ZTa", "sweep", "route sums", "step-pre", "backsub(step)", "pickdrop", "step/upd/drop", "add",
         "xmag(add)", "tail", "TICK pre (lane0)", "TICK assembly", "TICK post", "(reset body)", "(resid: gradient+s)", "(resid: forward subst)", "(resid: d + G x)"]
tot = v[:28].sum()
print(f"gait-ticks={n} mean QL iters={its/n:.1f} cycles/tick={tot/n:.0f}  route decisions/tick={v[28]/n:.1f} of which coordinate checks {v[29]/n:.2f}, dependent routes {v[30]/n:.2f}")
print(f"   of 'inverse' (factor): constant blocks into LDS {v[2]/n:.0f} cyc/tick, border rows of R {v[31]/n:.0f} cyc/tick")
for k, nme in enumerate(names):
    print(f"{k:2d} {nme:22s} {v[k]/n:12.0f} cyc/tick  {100*v[k]/tot:5.1f}%")
    if k == 21 and v[35:39].sum() > 0:
        for j, ph in enumerate(["state HBM -> LDS", "lane 0: support FSM, preview of the support states", "lane 0: orientation preview",
                                "one instant per lane: selection, rotated references, hull edges"]):
            print(f"      pre {ph:60s} {v[35+j]/n:10.0f} cyc/tick  {100*v[35+j]/tot:5.1f}%")
    if k == 23 and v[39] > 0:
        print(f"      post {'state fetched back, jerk, CoM samples, LIPM step':59s} {v[39]/n:10.0f} cyc/tick  {100*v[39]/tot:5.1f}%")
        for j, ph in enumerate(["lane 0: trunk", "feet: polynomials, one lane per sample", "samples into the state's queue (LDS)"]):
            print(f"      post {ph:59s} {v[40+j]/n:10.0f} cyc/tick  {100*v[40+j]/tot:5.1f}%")
        rest = v[23] - v[39] - v[40:43].sum()
        print(f"      post {'state LDS -> HBM':59s} {rest/n:10.0f} cyc/tick  {100*rest/tot:5.1f}%")
    if k == 12 and v[32:35].sum() > 0:                       # the compact view's sweep by phase (sweep_flat)
        for j, ph in enumerate(["phase 1: chain of rotation norms", "phase 2: ga / gb of every rotation, one lane each", "phase 3: lane i carries row i of Z through the rotations"]):
            print(f"      sweep {ph:58s} {v[32+j]/n:10.0f} cyc/tick  {100*v[32+j]/tot:5.1f}%")
