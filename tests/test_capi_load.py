"""CPU-side checks of the product library: it builds, loads, and exports every
symbol that include/wg_mpc.h declares.  No compute call is made (no GPU here)."""
import ctypes
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "wg_mpc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    wg = importlib.import_module("jrl-walkgen_amd")
    lib = wg.lib()
    syms = _declared_symbols()
    assert "wg_qp_solve_batch" in syms and "wg_init" in syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in wg_mpc.h but not exported"
    assert lib.wg_abi_version() >= 1


def test_pod_layouts_of_the_binding_match_the_header():
    """ABI 5: wg_tick_out_t is 61 whole cache lines (7808 B; the library static_asserts the same), the gait state 1208 B; the ctypes
    mirrors must agree or every array handed over is mis-strided."""
    wg = importlib.import_module("jrl-walkgen_amd")
    assert ctypes.sizeof(wg.TickOut) == 61 * 128 == 7808
    assert ctypes.sizeof(wg.GaitState) == 1208
    assert wg.lib().wg_abi_version() == 5
    hdr = open(os.path.join(ROOT, "include", "wg_mpc.h")).read()
    assert "double pad_[15];" in hdr and "ABI 5" in hdr


def test_lds_footprint_matches_design():
    wg = importlib.import_module("jrl-walkgen_amd")
    # Herdt N=16, two previewed steps: n = 36, m = 75 -> three QPs per 160 KiB CU with A staged in LDS (the launcher
    # reads A in place instead, see wg_qp_solve_batch_dev)
    b = wg.qp_lds_bytes(36, 75)
    assert 3 * b <= 160 * 1024 < 4 * b


def test_tick_residency_budget():
    """The tick kernel's residency is set by its LDS footprint: a gfx950 CU hands out 128 granules of
    1280 B.  The benchmark model must stay within 16 granules (8 gaits per CU = two waves per SIMD: wa, b and the border
    block live in a global slot), config 5's N = 32 within 32 (4 per CU: Z lives in a global slot)."""
    wg = importlib.import_module("jrl-walkgen_amd")
    gran = lambda b: -(-b // 1280)  # noqa: E731
    m = wg.Model()
    wg.lib().wg_model_defaults(ctypes.byref(m))
    b16 = wg.mpc_tick_lds_bytes_for(m)
    assert 0 < b16 and 128 // gran(b16) >= 8, b16
    m.N = 32
    b32 = wg.mpc_tick_lds_bytes_for(m)
    assert 128 // gran(b32) >= 4, b32
    m.N = 20                                                       # dense view
    assert 0 < wg.mpc_tick_lds_bytes_for(m) <= 160 * 1024
    m.N = 48
    assert wg.mpc_tick_lds_bytes_for(m) == 0


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    wg = importlib.import_module("jrl-walkgen_amd")
    lib = wg.lib()
    rc = lib.wg_init(0)
    assert rc != 0
    assert b"no CPU path" in lib.wg_last_error() or b"HIP" in lib.wg_last_error()


def test_cpp_facade_builds_and_fails_loudly_without_gpu(tmp_path):
    """libwg_walkgen.so (PatternGeneratorInterface / SimplePlugin API) exports the factory; without a HIP device the
    TestHerdt2010 driver stops with an error instead of producing a trace from some fall-back."""
    import subprocess
    import torch
    lib = os.path.join(ROOT, "jrl-walkgen_amd", "lib", "libwg_walkgen.so")
    exe = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "test_herdt2010")
    assert os.path.exists(lib) and os.path.exists(exe), "run __graft_entry__.build()"
    syms = subprocess.run(["nm", "-DC", lib], capture_output=True, text=True).stdout
    for s in ("PatternGeneratorJRL::patternGeneratorInterfaceFactory", "PatternGeneratorJRL::SimplePluginManager::CallMethod",
              "PatternGeneratorJRL::SimplePlugin::RegisterMethod", "PatternGeneratorJRL::ZMPVelocityReferencedQP::OnLine",
              "PatternGeneratorJRL::ZMPVelocityReferencedQP::InitOnLine"):
        assert s in syms, s
    fleet = os.path.join(ROOT, "jrl-walkgen_amd", "bin", "fleet_bench")
    assert os.path.exists(fleet), "run __graft_entry__.build()"
    if torch.cuda.is_available():
        return
    r = subprocess.run([fleet, "--batch", "8", "--ticks", "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "FAILED" in r.stderr          # plain C++ over the C ABI: no device, no result
    out = tmp_path / "t.dat"
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "FAILED" in r.stderr
    assert (not out.exists()) or out.stat().st_size == 0
