"""Short horizons whose QPs fail (the reference goes on with whatever x ql0001_ left; so do the oracle and the kernels): the tick
of each view against the oracle, gait by gait and tick by tick.  PN horizon, PB gaits, PT ticks; WG_TICK_DENSE=1 for the dense view."""
import ctypes as C, importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
N = int(os.environ.get("PN", "8")); B = int(os.environ.get("PB", "64")); TICKS = int(os.environ.get("PT", "80"))
pt = C.CDLL(os.path.join(ol.ORACLE_DIR, "libwg_oracle_ptrig.so"))
model = wg.model_defaults(); model.N = N
wg.mpc_configure(model)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
st = (wg.GaitState * B)(); ref = (wg.GaitState * B)()
for g in range(B):
    C.memmove(C.byref(st[g]), C.byref(s0), C.sizeof(wg.GaitState)); C.memmove(C.byref(ref[g]), C.byref(s0), C.sizeof(wg.GaitState))
rng = np.random.default_rng(20100)
sz = C.sizeof(wg.GaitState)
bad_total = 0; fails = 0
for t in range(TICKS):
    if t % 25 == 0:
        for g in range(B):
            v = [rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2)]
            for s in (st[g], ref[g]):
                s.vref[0], s.vref[1], s.vref[2] = v
    adv = 1 if t == 0 else (19 if t == 1 else 20)
    _, diag, _, _ = wg.mpc_tick_batch(st, want_out=False, advance_calls=adv)
    fails += int((diag[:, 0] != 0).sum())
    odiag = []
    for g in range(B):
        c = ref[g].clock
        for _ in range(adv):
            c += model.Tctrl
        ref[g].clock = c
        out = wg.TickOut() if hasattr(wg, "TickOut") else None
        assert pt.wgo_mpc_tick(C.byref(model), C.byref(ref[g]), None, None) == 0
    a = np.frombuffer(bytes(memoryview(st).cast("B")), np.uint8).reshape(B, sz)
    b = np.frombuffer(bytes(memoryview(ref).cast("B")), np.uint8).reshape(B, sz)
    bad = np.nonzero((a != b).any(1))[0]
    if len(bad):
        bad_total += len(bad)
        g = int(bad[0])
        print("tick %d: %d gaits differ from the oracle; first: gait %d diag %s, first differing byte %d" %
              (t, len(bad), g, diag[g].tolist(), int(np.nonzero(a[g] != b[g])[0][0])), flush=True)
        if os.environ.get("PVERBOSE"):
            print("   gpu com_x %s com_y %s phase %d steps_left %d | oracle com_x %s com_y %s phase %d steps_left %d" %
                  (list(st[g].com_x), list(st[g].com_y), st[g].phase, st[g].nb_steps_left, list(ref[g].com_x), list(ref[g].com_y),
                   ref[g].phase, ref[g].nb_steps_left), flush=True)
        for g in bad:                                       # go on from the oracle's state so that later ticks are judged on their own
            C.memmove(C.byref(st[int(g)]), C.byref(ref[int(g)]), sz)
print("N=%d B=%d ticks=%d view=%s: failed QPs %d, gait-ticks that differ from the oracle %d" %
      (N, B, TICKS, "dense" if os.environ.get("WG_TICK_DENSE") else "default", fails, bad_total))
