"""Host-side logic that needs no GPU: the step-stack generators of the facade (StepStackHandler is plain host code), the
footstep-plan geometry bench.py's `kernels` legs are fed from, bench.py's launch plan with its pre-roll, and the ISA audit's
parser on a synthetic listing."""
import importlib.util
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DRIVER = r"""
#include <cstdio>
#include <deque>
#include <sstream>
#include <string>
#include "wg_walkgen.hh"
using namespace PatternGeneratorJRL;
int main(int argc, char **argv) {
  StepStackHandler ssh;
  for (int i = 1; i < argc; i++) {                       // each argument: one command line, e.g. ":arccentered 0.75 30 -1"
    std::istringstream s(argv[i]);
    std::string m;
    s >> m;
    ssh.CallMethod(m, s);
  }
  std::deque<RelativeFootPosition> q;
  ssh.CopyRelativeFootPosition(q, true);
  for (size_t i = 0; i < q.size(); i++) printf("%.17g %.17g %.17g %.17g %.17g\n", q[i].sx, q[i].sy, q[i].theta, q[i].SStime, q[i].DStime);
  return 0;
}
"""


def _steps(tmp_path, *cmds):
    src = tmp_path / "ssh_driver.cpp"
    exe = tmp_path / "ssh_driver"
    if not exe.exists():
        src.write_text(DRIVER)
        lib = os.path.join(ROOT, "jrl-walkgen_amd", "lib")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", lib,
                               "-lwg_walkgen", "-lwg_mpc", "-Wl,-rpath," + lib])
    out = subprocess.run([str(exe), *cmds], capture_output=True, text=True, check=True).stdout
    return np.array([[float(v) for v in ln.split()] for ln in out.splitlines()])


def _chain(rel):
    """absolute support-foot poses from relative steps the way ZMPDiscretization::UpdateCurrentSupportFootPosition chains them
    (ZMPDiscretization.cpp:515-558): rotate by the step's theta first, then move by (sx, sy) in the rotated frame"""
    x = y = th = 0.0
    out = []
    for sx, sy, dth in rel[:, :3]:
        th += math.radians(dth)
        x += math.cos(th) * sx - math.sin(th) * sy
        y += math.sin(th) * sx + math.cos(th) * sy
        out.append((x, y, th))
    return np.array(out)


def test_arccentered_steps_lie_on_circles_about_one_centre(tmp_path):
    """":arccentered R arc support" (StepStackHandler.cpp:459-752): pairs of (arc step, closing step); every arc step turns by
    0.10 m / R, the last by the remainder; each foot's landings are one rotation about one centre apart"""
    R, arc = 0.75, 30.0
    rel = _steps(tmp_path, ":supportfoot -1", ":arccentered %g %g -1" % (R, arc), ":lastsupport")
    assert rel.shape[0] == 2 + 2 * 4 + 1                         # support step, side flip, 3 whole + 1 last pair, last support
    assert np.allclose(rel[:, 3], 0.78) and np.allclose(rel[:, 4], 0.02)
    turns = rel[2:-1:2, 2]
    step = math.degrees(0.10 / R)
    assert np.allclose(turns[:-1], step, atol=1e-12) and abs(turns.sum() - arc) < 1e-9 and 0 < turns[-1] < step
    assert np.allclose(rel[3:-1:2, :3], [0.0, 0.19, 0.0])        # the closing steps
    P = _chain(rel)
    arcs, closes = P[2:-1:2], P[3:-1:2]
    for pts in (arcs, closes):
        # consecutive landings of one foot: the same rigid motion (rotation by `step` about the common centre)
        rels = []
        for a, b in zip(pts[:-2], pts[1:-1]):
            c, s = math.cos(a[2]), math.sin(a[2])
            dx, dy = b[0] - a[0], b[1] - a[1]
            rels.append((c * dx + s * dy, -s * dx + c * dy, b[2] - a[2]))
        rels = np.array(rels)
        assert np.abs(rels - rels[0]).max() < 1e-12
    # the centre: both feet keep their distance to it (the feet face it: it lies R ahead of the line between them)
    a0 = arcs[0]
    centre = None
    for guess_sign in (+1, -1):
        cx = a0[0] + R * math.cos(a0[2]) - guess_sign * 0.095 * -math.sin(a0[2])
        cy = a0[1] + R * math.sin(a0[2]) - guess_sign * 0.095 * math.cos(a0[2])
        d = np.hypot(arcs[:, 0] - cx, arcs[:, 1] - cy)
        if np.ptp(d) < 1e-12:
            centre = (cx, cy)
    assert centre is not None
    assert np.ptp(np.hypot(closes[:, 0] - centre[0], closes[:, 1] - centre[1])) < 1e-12


def test_arc_generator_unchanged_by_the_new_one(tmp_path):
    """":arc" (TestKajita2003's TurningOnTheCircle sequence) still produces 0.15 m steps whose headings add up to the arc"""
    rel = _steps(tmp_path, ":supportfoot 1", ":arc 0.0 0.75 30.0 -1", ":lastsupport")
    assert abs(rel[1:-1, 2].sum() - 30.0) < 1e-9 and rel.shape[0] >= 4


def test_footplans_are_valid_polytopes():
    import footplans as fp
    slots = fp.plan(np.random.default_rng(3), n_steps=6)
    assert len(slots) > 60
    for A, B, c, sim in slots:
        assert len(B) in (4, 6) and A.shape == (len(B), 2)
        assert (A @ np.array(c) + B > 0).all()                   # the centre is strictly inside
        for i, s in enumerate(sim):                              # SimilarConstraints: an earlier row with the opposite normal
            if s:
                assert s < 0 and np.array_equal(A[i + s], -A[i])
    win = fp.polys_at(slots, len(slots) - 3, 16)
    assert len(win) == 16 and win[-1] is slots[-1]


def test_bench_plan_with_preroll_covers_every_tick_once():
    spec = importlib.util.spec_from_file_location("wg_bench_plan", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.PREROLL_TICKS == 100
    for W0, K in ((50, 200), (5, 20), (0, 1)):
        W = W0 + b.PREROLL_TICKS
        plan = b.launch_plan(0, W) + b.launch_plan(W, W + K)
        t = 0
        for t0, n in plan:
            assert t0 == t and n >= 1
            if n > 1 and t0 % b.REDRAW_TICKS:
                assert (t0 + n - 1) // b.REDRAW_TICKS == t0 // b.REDRAW_TICKS      # an unstaged launch never crosses a redraw
            t += n
        assert t == W + K
    assert b.launch_plan(105, 125) == [(105, 20)]                # the driver's window: ONE wg_mpc_run_batch_dev launch


def test_isa_audit_places_spill_code_by_loop_depth():
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    ia = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ia)
    listing = """
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	scratch_store_dword off, v1, off offset:4
.LBB0_1:                                ; =>This Loop Header: Depth=1
                                        ;     Child Loop BB0_2 Depth 2
	v_writelane_b32 v255, s12, 3
	scratch_load_dword v2, off, off offset:4
.LBB0_2:                                ;   Parent Loop BB0_1 Depth=1
                                        ; =>  This Inner Loop Header: Depth=2
	v_add_f64 v[4:5], v[4:5], v[6:7]
	v_readlane_b32 s13, v255, 3
	s_add_i32 s2, s2, -1
	s_cbranch_scc1 .LBB0_2
; %bb.3:                                ;   in Loop: Header=BB0_1 Depth=1
	ds_read_b64 v[8:9], v10
	s_cbranch_scc1 .LBB0_1
; %bb.4:
	s_endpgm
""".splitlines()
    a = ia.audit_kernel(listing)
    assert a["hist"]["scratch"] == {0: 1, 1: 1}
    assert a["hist"]["sgpr_spill_write"] == {1: 1} and a["hist"]["sgpr_spill_read"] == {2: 1}
    assert a["max_depth"] == 2 and a["total"]["VALU"] == 3 and a["total"]["SALU"] == 5
    assert a["inner_mix"] == {}                                  # nothing deeper than 2 here
    # the fall-through block after the inner loop is back at depth 1, the epilogue at depth 0
    spec_lines = [ln for ln in listing]
    a2 = ia.audit_kernel(spec_lines + ["\tscratch_store_dword off, v1, off offset:8"])
    assert a2["hist"]["scratch"] == {0: 2, 1: 1}


def test_isa_audit_tells_spill_reloads_from_the_solvers_own_broadcasts():
    """tools/isa_audit.py: an SGPR spill lives in a lane of a VGPR that some `v_writelane_b32 vX, sN, <lane>` writes; only
    `v_readlane_b32 sN, vX, <lane>` from THOSE registers are reloads.  The solver's own constant-lane broadcasts (the register
    Cholesky: rl(r[k], i)) read data registers and must not be counted (rounds 3 - 4 did: 1 779 "reloads" of which 1 371 were not)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(root, "tools", "isa_audit.py"))
    ia = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ia)
    asm = """
	s_mov_b32 s4, 0
	v_writelane_b32 v237, s12, 5
	v_writelane_b32 v237, s13, 6
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
	v_readlane_b32 s12, v237, 5
	v_readlane_b32 s20, v40, 7
	v_readlane_b32 s21, v41, 7
	v_add_f64 v[2:3], v[2:3], v[4:5]
	s_cbranch_scc1 .LBB0_1
; %bb.2:
	v_readlane_b32 s13, v237, 6
	s_endpgm
""".splitlines()
    a = ia.audit_kernel(asm)
    assert sum(a["hist"]["sgpr_spill_write"].values()) == 2
    assert dict(a["hist"]["sgpr_spill_read"]) == {1: 1, 0: 1}
    assert dict(a["hist"]["readlane_const_lane_other"]) == {1: 2}
