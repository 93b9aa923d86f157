"""Where the multi-tick kernel's wave-time goes (experiment build: make -C jrl-walkgen_amd lib/libwg_mpc_xs.so EXTRA=-DWG_XRUN_STATS,
WG_LIB_PATH pointing at it): every gait-tick reports when it ended and which block / XCD ran it.  Prints, per launch: ticks per XCD and
per block, when each XCD and block went idle relative to the end of the launch, and the gaps between a block's consecutive ticks."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
B = int(os.environ.get("PB", "4096")); T = int(os.environ.get("PT", "50"))
model = wg.model_defaults(); model.N = int(os.environ.get("PN", "16"))
wg.mpc_configure(model)
rng = np.random.default_rng(20100)
s0 = wg.gait_init(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0]); s0.nb_steps_left = 2
st = torch.frombuffer(bytearray(bytes(memoryview(s0).cast("B")) * B), dtype=torch.uint8).cuda()
def vref():
    return torch.from_numpy(np.stack([rng.uniform(-0.1, 0.3, B), rng.uniform(-0.1, 0.1, B), rng.uniform(-0.2, 0.2, B)], 1)).cuda()
wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 1); wg.mpc_tick_batch_dev(B, st.data_ptr(), None, None, 19)
v = vref(); wg.mpc_set_velref_dev(B, st.data_ptr(), v.data_ptr())
wg.mpc_run_batch_dev(B, st.data_ptr(), 60, 20, None, None)
torch.cuda.synchronize()
for rep in range(2):
    diag = torch.zeros(T, B, 6, dtype=torch.int32, device="cuda")
    v = vref(); wg.mpc_set_velref_dev(B, st.data_ptr(), v.data_ptr())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); wg.mpc_run_batch_dev(B, st.data_ptr(), T, 20, None, diag.data_ptr()); e1.record()
    torch.cuda.synchronize(); ms = e0.elapsed_time(e1)
    d = diag.cpu().numpy().reshape(-1, 6)
    end = d[:, 3].astype(np.uint32).astype(np.int64); blk = d[:, 4]; xcd = d[:, 5]; its = d[:, 1]
    if os.environ.get("PSAVE") and rep == 1:                  # [T, B] iteration counts, end times, XCDs: input of tools/xrun_sim.py
        np.savez_compressed(os.environ["PSAVE"], its=its.reshape(T, B).astype(np.int16), end=end.reshape(T, B), xcd=xcd.reshape(T, B).astype(np.int8),
                            blk=blk.reshape(T, B).astype(np.int16), ms=ms)
    t_last = end.max(); t_first = end.min()
    span_ms = (t_last - t_first) / 1e5
    print("launch %d: B=%d T=%d  events %.2f ms, first tick end -> last tick end %.2f ms, %.0f ticks/s" % (rep, B, T, ms, span_ms, B * T / ms * 1e3))
    nblk = blk.max() + 1
    per_blk = np.bincount(blk, minlength=nblk)
    print("  blocks that ran ticks: %d of %d; ticks per block min %d median %d max %d" % ((per_blk > 0).sum(), nblk, per_blk[per_blk > 0].min(), np.median(per_blk[per_blk > 0]), per_blk.max()))
    blk_xcd = np.zeros(nblk, int); blk_xcd[blk] = xcd
    for x in range(8):
        sel = xcd == x
        bl = np.unique(blk[sel])
        print("  XCD %d: %4d blocks, %6d ticks, %.1f iterations per tick, last tick ends %.2f ms before the launch's last" %
              (x, len(bl), sel.sum(), its[sel].mean(), (t_last - end[sel].max()) / 1e5))
    # when did each block stop: idle time between its last tick and the launch's last tick
    last_of_blk = np.zeros(nblk, np.int64); np.maximum.at(last_of_blk, blk, end)
    idle_tail = (t_last - last_of_blk[per_blk > 0]) / 1e5
    print("  idle tail per block (ms): mean %.3f, p50 %.3f, p90 %.3f, max %.3f  => %.2f %% of the launch" %
          (idle_tail.mean(), np.percentile(idle_tail, 50), np.percentile(idle_tail, 90), idle_tail.max(), 100 * idle_tail.mean() / ms))
    # a block's tick-to-tick period against its iterations: the per-tick overhead outside the solve
    order = np.lexsort((end, blk))
    eb = end[order]; bb = blk[order]; ib = its[order]
    same = bb[1:] == bb[:-1]
    per = (eb[1:] - eb[:-1])[same] / 1e2                      # microseconds
    itn = ib[1:][same]
    A = np.stack([np.ones_like(itn, float), itn.astype(float)], 1)
    coef, *_ = np.linalg.lstsq(A, per, rcond=None)
    print("  tick period of a block: mean %.1f us = %.1f + %.2f per iteration (least squares); p99 %.1f us, max %.1f us" %
          (per.mean(), coef[0], coef[1], np.percentile(per, 99), per.max()))
    first_of_blk = np.full(nblk, np.iinfo(np.int64).max); np.minimum.at(first_of_blk, blk, end)
    f = (first_of_blk[per_blk > 0] - t_first) / 1e5
    print("  first tick end per block relative to the earliest (ms): p50 %.3f p90 %.3f max %.3f" % (np.percentile(f, 50), np.percentile(f, 90), f.max()))
