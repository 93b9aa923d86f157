"""N > 1 plumbing on CPU: two processes over gloo exercise exactly what bench.py does between GPUs --
broadcast of the constant model block from rank 0, contiguous gait shards, max-over-ranks timing, whole-job
tick count -- with the CPU oracle standing in for the device step (no GPU here)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    shard = importlib.import_module("jrl-walkgen_amd.shard")
    wg = importlib.import_module("jrl-walkgen_amd")
    import herdt_replay as hr
    r, lr, w = shard.init_process_group("gloo")
    dev = torch.device("cpu")
    model = hr.default_model() if r == 0 else wg.Model()
    if r == 0:
        model.sole_w = 0.2468            # something only rank 0 knows
    shard.broadcast_struct(model, dev, src=0)
    total = 11
    lo, hi = shard.shard_range(total, r, w)
    com = []
    for g in range(lo, hi):
        s = hr.init_state(model, [0.0316055, 0.0, 0.7116911], [0.0, 0.09, 0.0], [0.0, -0.09, 0.0])
        s.nb_steps_left = 2
        s.vref[0] = 0.1 + 0.01 * g
        s.clock = 0.005
        hr.oracle_tick(model, s)
        com.append((g, s.com_x[0]))
    shard.barrier()
    tmax = shard.max_over_ranks(1.0 + r, dev)
    nsum = shard.sum_over_ranks(hi - lo, dev)
    q.put((r, model.sole_w, model.N, (lo, hi), tmax, nsum, com))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_gloo_plumbing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0.2468, 0.2468]          # the broadcast reached rank 1
    assert res[0][2] == res[1][2] == 16
    assert res[0][3] == (0, 6) and res[1][3] == (6, 11)      # contiguous shards cover every gait once
    assert res[0][4] == res[1][4] == 2.0                     # max over ranks
    assert res[0][5] == res[1][5] == 11.0                    # whole-job count
    gaits = sorted(res[0][6] + res[1][6])
    assert [g for g, _ in gaits] == list(range(11))
    assert len({v for _, v in gaits}) == 11                  # different references -> different CoM, no mix-up


def test_shard_range_properties():
    shard = importlib.import_module("jrl-walkgen_amd.shard")
    for total in (0, 1, 7, 4096, 32768, 32771):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_range_is_the_c_abi_rule():
    """shard.shard_range is wg_shard_range of include/wg_mpc.h (what host/fleet_bench.cpp calls); config 4's split"""
    import ctypes as C
    wg = importlib.import_module("jrl-walkgen_amd")
    shard = importlib.import_module("jrl-walkgen_amd.shard")
    assert [shard.shard_range(32768, r, 8) for r in range(8)] == [(4096 * r, 4096 * (r + 1)) for r in range(8)]
    lo, hi = C.c_longlong(), C.c_longlong()
    for bad in ((10, 2, 2), (10, -1, 2), (-1, 0, 1), (10, 0, 0)):
        assert wg.lib().wg_shard_range(C.c_longlong(bad[0]), bad[1], bad[2], C.byref(lo), C.byref(hi)) == -2
