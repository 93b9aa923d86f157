import ctypes as C, importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wg = importlib.import_module("jrl-walkgen_amd"); import oraclelib as ol
wg.init(0); ol.build_oracle()
pt = C.CDLL(os.path.join(ol.ORACLE_DIR, os.environ.get("ORC_SO", "libwg_oracle_ptrig.so")))
N, T = int(sys.argv[1]), float(sys.argv[2])
VX = float(sys.argv[3]) if len(sys.argv) > 3 else None
model = wg.model_defaults(); model.N = N; model.T = T; model.t_double = T; model.Tctrl = T / 20.0
wg.mpc_configure(model)
B = 6
def start():
    s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
    st = (wg.GaitState * B)()
    for g in range(B):
        C.memmove(C.byref(st[g]), C.byref(s0), C.sizeof(wg.GaitState))
        if VX is not None: st[g].com_x[1] = VX * (g + 1) / B; st[g].com_y[1] = 0.5 * VX * (g % 3 - 1)
    return st
st, ref = start(), start()
rng = np.random.default_rng(N)
for t in range(40):
    if t % 15 == 0:
        for g in range(B):
            v = [rng.uniform(-0.1, 0.3), rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2)] if VX is None else [0.0, 0.0, 0.0]
            for s in (st[g], ref[g]): s.vref[0], s.vref[1], s.vref[2] = v
    adv = 1 if t == 0 else (19 if t == 1 else 20)
    outs, diag, hist, hlen = wg.mpc_tick_batch(st, want_out=True, advance_calls=adv, hist_cap=256)
    if VX is not None and t < 6: print('tick', t, 'n', diag[:, 3].tolist(), 'nact', diag[:, 2].tolist(), 'ifail', diag[:, 0].tolist())
    for g in range(B):
        c = ref[g].clock
        for _ in range(adv): c += model.Tctrl
        ref[g].clock = c
        dump = __import__("herdt_replay").QpDump()
        assert pt.wgo_mpc_tick(C.byref(model), C.byref(ref[g]), None, C.byref(dump)) == 0
        a = bytes(memoryview(st)[g:g+1].cast("B")) if False else bytes(st[g]); b = bytes(ref[g])
        if a != b:
            i = next(k for k in range(len(a)) if a[k] != b[k])
            print("tick", t, "gait", g, "first diff byte", i, "diag", diag[g], "oracle n,m,ifail,iters,nact", dump.n, dump.m, dump.ifail, dump.n_iter, dump.nact)
            print(" gpu hist", hist[g][:hlen[g]].tolist()); print(" orc hist", list(dump.hist[:dump.hist_len]))
            print(" com gpu", st[g].com_x[0], st[g].com_y[0], "orc", ref[g].com_x[0], ref[g].com_y[0])
            sys.exit(0)
print("all equal")
