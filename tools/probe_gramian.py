"""The one GEMM-shaped step of the path on the matrix cores: wg_gramian_batch_dev for many models (build_invariant_part,
generator-vel-ref.cpp:587-614).  Prints per-launch time and the MFMA rate against the dense peak of the dtype; under
rocprofv3 --pmc this is the run whose SQ_INSTS_VALU_MFMA_MOPS_* / SQ_VALU_MFMA_BUSY_CYCLES counters are filed in profiles/."""
import importlib, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
import ctypes as C
B = int(os.environ.get("PB", "262144")); REP = int(os.environ.get("PR", "20"))
rng = np.random.default_rng(1)
T = torch.from_numpy(rng.uniform(0.05, 0.2, B)).cuda(); h = torch.from_numpy(rng.uniform(0.6, 0.9, B)).cuda()
lib = wg.lib()
v = lambda t: C.c_void_p(t.data_ptr())
PEAK = {0: 78.6, 1: 157.3}          # TFLOP/s dense MFMA peak: f64 (= the fp64 vector rate, spec), f32 (MI355X_MICROARCH.md)
for N in (16, 32):
    Q = torch.zeros(B, N, N, dtype=torch.float64, device="cuda")
    for prec, name in ((0, "f64 v_mfma_f64_16x16x4_f64"), (1, "f32 v_mfma_f32_16x16x4_f32")):
        for _ in range(3):
            assert lib.wg_gramian_batch_dev(B, N, v(T), v(h), 1.0, 1e-5, 1e-6, prec, v(Q), None) == 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(REP):
            lib.wg_gramian_batch_dev(B, N, v(T), v(h), 1.0, 1e-5, 1e-6, prec, v(Q), None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / REP
        flop = 2.0 * 2 * N ** 3 * B                       # two N x N x N products per model (Uv'Uv, Uz'Uz); beta I is an add
        print("N=%d %-28s B=%d: %.3f ms per launch, %.2f TFLOP/s = %.1f %% of the %.1f TFLOP/s dense MFMA peak; output %.1f GB/s"
              % (N, name, B, dt * 1e3, flop / dt / 1e12, 100 * flop / dt / 1e12 / PEAK[prec], PEAK[prec], B * N * N * 8 / dt / 1e9))
