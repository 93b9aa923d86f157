"""bench.py's host-side bookkeeping (no GPU): the launch plan covers exactly the timed ticks, never crosses a change of the
velocity references, keeps the control loop's first two ticks apart; the algorithmic-bytes formula is SURVEY 8(d)'s; the
synthetic references are the same for a gait whatever shard it lands in."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("wg_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["wg_bench"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_launch_plan_covers_ticks_and_respects_redraws():
    b = _bench()
    for t0, t1 in ((0, 50), (50, 250), (20, 220), (0, 3), (7, 8), (49, 51), (100, 100)):
        for per_tick in (False, True):
            for staged in (False, True):
                plan = b.launch_plan(t0, t1, per_tick, staged=staged)
                ticks = [t for s, n in plan for t in range(s, s + n)]
                assert ticks == list(range(t0, t1))
                for s, n in plan:
                    one_segment = s // b.REDRAW_TICKS == (s + n - 1) // b.REDRAW_TICKS
                    # a launch crosses a change of the references only if it starts on one (then they are staged on the device)
                    assert n >= 1 and (one_segment or (staged and s % b.REDRAW_TICKS == 0))
                    assert n == 1 or (s >= 2 and not per_tick)
    assert b.launch_plan(50, 250, staged=False) == [(50, 50), (100, 50), (150, 50), (200, 50)]
    assert b.launch_plan(50, 250) == [(50, 200)]
    assert b.launch_plan(20, 220) == [(20, 30), (50, 170)]
    assert b.launch_plan(0, 50)[:3] == [(0, 1), (1, 1), (2, 48)]


def test_algorithmic_bytes_is_the_ql0001_boundary():
    b = _bench()
    # n = 36, m = 75 (mmax = 76): 8 (n^2 + n + mmax n + mmax + 2n) read + 8 (n + m + 2n) written = 33 728 + 1 464
    assert b.algorithmic_bytes(36, 75) == 33728 + 1464
    assert b.algorithmic_bytes(32, 65) == 8 * (32 * 32 + 32 + 66 * 32 + 66 + 64) + 8 * (32 + 65 + 64)     # no previewed step


def test_velocity_table_is_per_gait_deterministic():
    b = _bench()
    a = b.velocity_table(0, 8, 3)
    c = b.velocity_table(4, 12, 3)
    assert np.array_equal(a[:, 4:8], c[:, 0:4])                   # gait g draws the same references in any shard
    assert (a[..., 0] >= -0.1).all() and (a[..., 0] <= 0.3).all() and (np.abs(a[..., 1]) <= 0.1).all() and (np.abs(a[..., 2]) <= 0.2).all()


def test_cpu_baseline_worker_count_is_what_the_process_may_use(monkeypatch):
    """"1 core and all host cores; the harness prints the core count" -- the count is the affinity mask
    cut down by the cgroup quota (or, with no limit visible on a many-core host, one GPU lease's share), never os.cpu_count()
    of a shared host; the line reports what was seen, what was used and the parallel efficiency."""
    b = _bench()
    used, seen = b.usable_cores()
    assert 1 <= used <= len(os.sched_getaffinity(0)) <= seen["os_cpu_count"]
    assert used <= max(b.BOX_CPU_SHARE, int(seen["cgroup_quota"] or 0)) or seen["affinity"] < seen["os_cpu_count"]
    monkeypatch.setenv("WG_BENCH_CPU_CORES", "2")
    assert b.usable_cores()[0] == min(2, used)
    one, allc = b.cpu_baseline(8, 4)                                # a token sample: the bookkeeping, not a measurement
    assert allc["cores_used"] == min(2, used) and allc["value"] > 0 and 0 < allc["parallel_efficiency"]
    assert allc["worker_rate_min"] <= allc["worker_rate_max"] and allc["cores_visible"] == seen["os_cpu_count"]
