// wg_capi.hip -- C ABI (include/wg_mpc.h) over the HIP kernels.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wg_mpc.h"
#include "wg_ql_device.hpp"
#include "wg_tick_device.hpp"
#include "wg_tick_kernels.hpp"
#include "wg_pldp_device.hpp"
#include "wg_dimitrov_device.hpp"
#include "wg_preview_device.hpp"
#include "wg_gramian_device.hpp"
#include "wg_zmpdisc_device.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                            \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) return fail(WG_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// grow-only device buffer.  Growing never frees: a launch enqueued earlier may still be using the old allocation (and
// hipFree would make the host wait for the whole device), so the old one is retired and freed with the context.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  std::vector<void *> retired;
  int reserve(size_t bytes) {
    if (bytes <= cap) return WG_OK;
    if (p && bytes < cap + cap / 2) bytes = cap + cap / 2;   // grow geometrically: what is retired stays below twice what is live
    void *fresh = nullptr;
    hipError_t e = hipMalloc(&fresh, bytes);
    if (e != hipSuccess) return fail(WG_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    if (p) retired.push_back(p);
    p = fresh; cap = bytes;
    return WG_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    for (void *r : retired) (void)hipFree(r);
    retired.clear();
    p = nullptr; cap = 0;
  }
};

}  // namespace

// Everything the library keeps between calls: configured models (device copies of their tables), the workspaces its
// kernels need besides the caller's arrays, and the staging buffers of the host-pointer entry points.  The entry points
// without a context argument work on one process-wide default context.
struct wg_ctx {
  int device = 0;
  int num_cu = 256;
  std::mutex mu;
  // held from the ordering test of a launch that uses per-context device state (queue, solver slots, scratch states) through the
  // launch itself and the event record behind it: two host threads cannot both pass the test.  Lock order: mu -> launch_mu
  std::mutex launch_mu;
  // Herdt-2010 tick
  wg_model_t model;
  bool model_set = false;
  wg::TickTables *tables_dev = nullptr;
  wg_model_t *model_dev = nullptr;     // the model in device memory (the multi-tick kernels read it through a pointer)
  DevBuf tick_state, tick_out, tick_aux, run_buf, tick_z, asm_state;
  // one launch per tick: the gaits are started longest-solve-first, by the iteration counts of their previous tick
  // dense ql0001_ boundary: wa | b of every QP in a slot of global memory when that buys the eighth QP per CU; one launch at a
  // time uses the slots (a launch that arrives on another stream while one is pending keeps wa | b in LDS instead)
  DevBuf qp_slot;
  struct SlotOrder { hipEvent_t ev = nullptr; hipStream_t stream = nullptr; bool armed = false; };
  SlotOrder qp_order;
  DevBuf lpt_buf;                        // [iterations of the last tick (B) | start order (B)]
  const wg_gait_state_t *lpt_states = nullptr;
  int lpt_B = 0;
  // the same for the dense QP boundary and the Dimitrov tick's QL back-ends: a batch that follows another one of the same size on
  // the same arrays (an MPC loop: problem k of consecutive calls is the same robot a tick later) starts longest-solve-first
  DevBuf qlpt_buf, dlpt_buf;
  const void *qlpt_key = nullptr, *dlpt_key = nullptr;
  int qlpt_B = 0, dlpt_B = 0;
  // The tick / run kernels keep their queue and per-block solver slots in run_buf / tick_z: launches of one context must
  // not overlap ON THE DEVICE.  Every such launch leaves an event behind; a launch that arrives on ANOTHER stream while that
  // event is still pending is made to wait for it (hipStreamWaitEvent: ordered, not refused -- a double-buffered pipeline that
  // orders its streams with events of its own is enqueued ahead of time and must be accepted).  WG_OVERLAP_STRICT=1 refuses such a
  // launch instead (WG_ERR_BUSY, nothing launched): a way to find serialisation one did not intend.
  SlotOrder guard_order;
  // wg_mpc_assemble_batch_dev runs the tick on scratch copies of the states kept per context (asm_state): same ordering
  SlotOrder asm_order;
  // launches of the other back-ends (PLDP, Dimitrov tick, preview): they read constants a re-configuration overwrites, so they
  // leave an event for it to wait on (marked, never claimed: they keep nothing per launch in the context)
  SlotOrder aux_order;
  // The host-pointer entry points stage, launch and copy back on THIS stream (non-blocking: no implicit ordering against the
  // legacy stream or anybody else's) and wait for it alone -- never for the device: another context's launches, or a caller's
  // own streams, keep running while this context's host call waits for its own work.
  hipStream_t host_stream = nullptr;
  bool overlap_strict = false;           // WG_OVERLAP_STRICT at wg_ctx_create / wg_init, wg_set_overlap_strict afterwards
  long long serialised = 0;              // launches that were ordered behind one of another stream (wg_overlap_serialised)
  // one-robot path (wg_mpc_tick_pinned): its own stream, a completion counter in host-mapped memory
  hipStream_t pin_stream = nullptr;
  int *pin_flag = nullptr;               // host-mapped; the kernel adds 1 per gait when its outputs are visible
  int pin_seq = 0;
  // PLDP / Dimitrov
  wg::PldpModel *pldp_dev = nullptr;
  int pldp_N = 0;
  DevBuf pldp_buf;
  wg::DimitrovConst *dim_dev = nullptr;
  std::unique_ptr<wg::DimitrovConst> dim_host;
  bool dim_set = false;
  DevBuf dim_buf;
  // preview control
  wg::PreviewConst prev;
  double *prev_F = nullptr;            // device copy of the window gains
  bool prev_set = false;
  DevBuf prev_buf;
  // staging of the host-pointer entry points
  DevBuf in, out, gram_buf, zd_buf;
  void release_all() {
    if (tables_dev) (void)hipFree(tables_dev);
    if (model_dev) (void)hipFree(model_dev);
    if (pldp_dev) (void)hipFree(pldp_dev);
    if (dim_dev) (void)hipFree(dim_dev);
    if (prev_F) (void)hipFree(prev_F);
    tables_dev = nullptr; model_dev = nullptr; pldp_dev = nullptr; dim_dev = nullptr; prev_F = nullptr;
    model_set = false; pldp_N = 0; dim_set = false; prev_set = false;
    qlpt_key = dlpt_key = nullptr; qlpt_B = dlpt_B = 0;
    for (DevBuf *b : {&tick_state, &tick_out, &tick_aux, &run_buf, &tick_z, &asm_state, &lpt_buf, &qlpt_buf, &dlpt_buf, &qp_slot, &pldp_buf, &dim_buf, &prev_buf, &in, &out, &gram_buf, &zd_buf})
      b->release();
    for (SlotOrder *o : {&guard_order, &qp_order, &asm_order, &aux_order}) {
      if (o->ev) (void)hipEventDestroy(o->ev);
      o->ev = nullptr; o->armed = false; o->stream = nullptr;
    }
    lpt_states = nullptr; lpt_B = 0;
    if (pin_stream) (void)hipStreamDestroy(pin_stream);
    if (host_stream) (void)hipStreamDestroy(host_stream);
    host_stream = nullptr;
    if (pin_flag) (void)hipHostFree(pin_flag);
    pin_stream = nullptr; pin_flag = nullptr; pin_seq = 0;
  }
};

namespace {

std::mutex g_default_mu;
wg_ctx *g_default = nullptr;          // created by wg_init() or by the first call that needs it

int make_ctx(int device_ordinal, wg_ctx **out) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(WG_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device_ordinal < 0 || device_ordinal >= count)
    return fail(WG_ERR_BAD_ARG, "device ordinal %d out of range [0,%d)", device_ordinal, count);
  HIP_TRY(hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
  wg_ctx *c = new wg_ctx();
  c->device = device_ordinal;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&c->host_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return fail(WG_ERR_HIP, "hipStreamCreateWithFlags failed for the context's host stream");
  }
  if (const char *e = getenv("WG_OVERLAP_STRICT")) c->overlap_strict = atoi(e) != 0;   // read once per context, not per launch
  *out = c;
  return WG_OK;
}

int default_ctx(wg_ctx **out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (!g_default)
    if (int rc = make_ctx(0, &g_default)) return rc;
  *out = g_default;
  return WG_OK;
}

// every entry point starts here: the context's device becomes the calling thread's current device
int use_ctx(wg_ctx *ctx) {
  if (!ctx) return fail(WG_ERR_BAD_ARG, "null context");
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) HIP_TRY(hipSetDevice(ctx->device));
  return WG_OK;
}

// Both with ctx->launch_mu held, around the launch.
// before a launch that uses device state of the context (`o`): behind the previous such launch, whatever stream that was on
inline bool slot_pending(const wg_ctx::SlotOrder &o) { return o.armed && hipEventQuery(o.ev) == hipErrorNotReady; }
inline bool slot_pending_elsewhere(const wg_ctx::SlotOrder &o, hipStream_t st) { return o.stream != st && slot_pending(o); }
int slot_claim(wg_ctx *ctx, wg_ctx::SlotOrder &o, hipStream_t st, const char *what) {
  if (!slot_pending(o)) return WG_OK;
  // the handle alone does not identify a stream (one destroyed and re-created at the same address counts as the same): the wait
  // is issued whenever the previous launch is pending -- behind a launch of the same stream it costs nothing
  if (o.stream != st) {
    if (ctx->overlap_strict)
      return fail(WG_ERR_BUSY, "a %s launch of this context is still in flight on another stream (WG_OVERLAP_STRICT: launches of one "
                               "context are refused instead of ordered; give each stream its own wg_ctx to overlap them)", what);
    if (ctx->serialised++ == 0 && getenv("WG_OVERLAP_NOTE"))
      fprintf(stderr, "wg_mpc: a %s launch was ordered behind one of another stream of the same context (first occurrence; "
                      "wg_overlap_serialised() counts them, one wg_ctx per stream overlaps them)\n", what);
  }
  HIP_TRY(hipStreamWaitEvent(st, o.ev, 0));
  return WG_OK;
}
// a re-configuration overwrites tables that launches of THIS context may still be reading: wait for those launches (their events,
// the context's own streams) -- not for the device
int ctx_wait_own(wg_ctx *ctx) {
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  for (wg_ctx::SlotOrder *o : {&ctx->guard_order, &ctx->qp_order, &ctx->asm_order, &ctx->aux_order})
    if (o->armed) HIP_TRY(hipEventSynchronize(o->ev));
  if (ctx->host_stream) HIP_TRY(hipStreamSynchronize(ctx->host_stream));
  if (ctx->pin_stream) HIP_TRY(hipStreamSynchronize(ctx->pin_stream));
  return WG_OK;
}
// staging copies of the host-pointer entry points: on the context's stream; the caller of these waits for that stream before it
// returns (pageable host memory: the runtime stages the copy, the stream synchronise below covers both directions)
#define WG_H2D(dst, src, bytes) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, ctx->host_stream))
#define WG_D2H(dst, src, bytes) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, ctx->host_stream))
#define WG_ZERO(dst, bytes) HIP_TRY(hipMemsetAsync((dst), 0, (bytes), ctx->host_stream))
#define WG_HOST_WAIT() HIP_TRY(hipStreamSynchronize(ctx->host_stream))
// after it: the event later launches are ordered behind
int slot_mark(wg_ctx::SlotOrder &o, hipStream_t st) {
  if (!o.ev) HIP_TRY(hipEventCreateWithFlags(&o.ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(o.ev, st));
  o.stream = st; o.armed = true;
  return WG_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// Dense batched QP kernel: one wavefront (= one workgroup) per QP.
// Replaces ql0001_ (qld.hh:27-31) for B problems at once.
// ---------------------------------------------------------------------------
// kFixN / kFixM > 0: the Herdt-sized boundary (nmax == kFixN, mmax == kFixM, A, G and wa | b out of the LDS) with the strides, the
// LDS layout and the solver's loop bounds as compile-time constants (QlView::carve_fixed_dense, DenseProbT<false, kFixN>)
template <bool kALds, bool kGLds, bool kWLds = true, int kFixN = 0, int kFixM = 0>   // where A / G / wa | b live is known at compile time: ds_ or global_ accesses,
// Left to itself the compiler takes 256 VGPRs plus 3 AGPRs -- 259 registers, one wave per SIMD, four QPs per CU where the LDS
// would admit five at n = 36, m = 75.  Forced to two waves per SIMD (-DWG_QLD_WPE=2: 256 registers, 2-3 spilled, 12-16 B of
// scratch) it measured 5 % SLOWER on the Herdt workload's real QPs (1.73 against 1.82 M QPs/s, B = 4096): the fifth QP per CU
// does not pay for the tighter allocation.  The default stays.
// With G read in place as well (21.7 KB of LDS at n = 36, m = 75: seven QPs per CU) the residency is worth the 256-register
// build: that instantiation is compiled for two waves per SIMD.
#ifdef WG_QLD_WPE
#define WG_QLD_ATTR __attribute__((amdgpu_waves_per_eu(WG_QLD_WPE, WG_QLD_WPE)))
#else
#define WG_QLD_ATTR __attribute__((amdgpu_waves_per_eu(kGLds ? 1 : 2, kGLds ? 8 : 2)))
#endif
__global__ __launch_bounds__(64) WG_QLD_ATTR void wg_ql_dense_kernel(   // never flat_ (those also count on lgkmcnt and stall the LDS waits)
    int B, int nmax_arg, int mmax_arg, const int *__restrict__ n_arr, const int *__restrict__ m_arr,
    const int *__restrict__ me_arr, const double *__restrict__ C, const double *__restrict__ dvec,
    const double *__restrict__ A, const double *__restrict__ bvec, const double *__restrict__ xl,
    const double *__restrict__ xu, double eps, double *__restrict__ x, double *__restrict__ u,
    int *__restrict__ ifail, int *__restrict__ n_iter, int *__restrict__ iact, int *__restrict__ nact,
    int *__restrict__ hist, int hist_cap, int *__restrict__ hist_len, double *__restrict__ wab_slots,
    const int *__restrict__ order, int *__restrict__ iters_out) {
  extern __shared__ __attribute__((aligned(16))) double wg_lds[];
  const int lane = threadIdx.x & 63;
  // one QP per block (grid == B): nothing lane-dependent lives across QPs.  Blocks start in index order: `order`
  // (wg_lpt_order_kernel) makes that the order of decreasing solve length, as far as the previous batch predicts it
  const int qp = order ? wg::uni(order[blockIdx.x]) : (int)blockIdx.x;
  const int nmax = kFixN > 0 ? kFixN : nmax_arg, mmax = kFixM > 0 ? kFixM : mmax_arg;   // the host checks the match
  if (qp < B) {
    const int n = n_arr ? n_arr[qp] : nmax;
    const int m = m_arr ? m_arr[qp] : mmax - 1;
    const int me = me_arr ? me_arr[qp] : 0;
    wg::QlDims D(n, m, m, true, kALds, 0, true, true, kWLds, true, 0, kGLds);
    wg::QlView q;
    // kWLds = false: the constraint weights wa (m + n) and b (m) -- read lane-parallel once per iteration -- live in this
    // block's slot of global memory [wa (mmax + nmax) | b (mmax)]: 1.5 KB less LDS, the eighth QP on the CU at n = 36, m = 75
    if constexpr (kFixN > 0) q.template carve_fixed_dense<kFixN, kFixM>(wg_lds, n, m, me, wab_slots + (size_t)qp * (2 * kFixM + kFixN));
    else if constexpr (kWLds) q.carve(wg_lds, D, me);
    else q.template carve<true, false, true>(wg_lds, D, me, wab_slots + (size_t)qp * (2 * (size_t)mmax + nmax), mmax + nmax);

    // ---- stage the problem into LDS (coalesced 8-byte lanes) ----
    const double *Cg = C + (size_t)qp * nmax * nmax;
    const double *Ag = A + (size_t)qp * mmax * nmax;
    if constexpr (kGLds) {
      for (int j = 0; j < n; ++j)
        for (int i = lane; i < n; i += 64) q.G[i + j * q.ldg] = Cg[i + (size_t)j * nmax];
    } else {                                   // G is cold after the factorisation: in place (L2), its diagonal in LDS
      q.G = const_cast<double *>(Cg);
      q.ldg = nmax;
      for (int i = lane; i < n; i += 64) q.Gdiag[i] = Cg[i + (size_t)i * nmax];
    }
    if constexpr (kALds) {
      for (int i = 0; i < n; ++i)
        for (int k = lane; k < m; k += 64) q.A[k + i * q.lda] = Ag[k + (size_t)i * mmax];
    } else {                                   // too large for LDS next to G, Z, R: the solver only reads A -> in place (L2)
      q.A = const_cast<double *>(Ag);
      q.lda = mmax;
    }
    for (int i = lane; i < n; i += 64) {
      q.d[i] = dvec[(size_t)qp * nmax + i];
      q.xl[i] = xl[(size_t)qp * nmax + i];
      q.xu[i] = xu[(size_t)qp * nmax + i];
    }
    for (int k = lane; k < m; k += 64) q.b[k] = -bvec[(size_t)qp * mmax + k];   // qld.cpp:469-475
    WG_WSYNC();
    // qld.cpp:442-444: c(nmax,nmax) == 0 -> eps (inside the n x n block only if nmax == n)
    typename std::conditional<(kFixN > 0), wg::DenseRegProb<(kFixN > 0 ? kFixN : 1), (kFixM > 0 ? kFixM : 1)>, wg::DenseProbT<kGLds, kFixN>>::type prob;
    // (kFixN > 0: A's rows go into registers inside ql_solve, once R and Z exist -- m <= kFixM <= 128: two rows per lane)
    if (nmax == n && lane == 0 && fabs(prob.Gd(q, n - 1)) == 0.0) prob.setGd(q, n - 1, eps);
    WG_WSYNC();

    int *hq = hist ? hist + (size_t)qp * hist_cap : nullptr;
    wg::QlResult r = wg::ql_solve(q, prob, eps, hq, hist_cap);

    // ---- results ----
    for (int i = lane; i < n; i += 64) x[(size_t)qp * nmax + i] = q.x[i];
    if (u) {
      double *uq = u + (size_t)qp * (mmax + 2 * nmax);
      if (r.ifail == 0) {                                   // qld.cpp:520-536
        for (int j = lane; j < m + 2 * n; j += 64) uq[j] = 0.0;
        WG_WSYNC();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        for (int i = lane; i < r.nact; i += 64) uq[q.iact[i] - 1] = q.lam[i];
      }
    }
    if (iact)
      for (int i = lane; i < nmax; i += 64) iact[(size_t)qp * nmax + i] = (i < r.nact) ? q.iact[i] : 0;
    if (lane == 0) {
      ifail[qp] = r.ifail;
      if (n_iter) n_iter[qp] = r.n_iter;
      if (nact) nact[qp] = r.nact;
      if (hist_len) hist_len[qp] = r.hist_len;
      if (iters_out) iters_out[qp] = r.n_iter;
    }
    WG_WSYNC();
  }
}

extern "C" {

static_assert(sizeof(wg_tick_out_t) == 61 * 128, "wg_tick_out_t: whole 128-byte lines (include/wg_mpc.h)");
int wg_abi_version(void) { return 5; }

const char *wg_last_error(void) { return g_err.c_str(); }

int wg_ctx_create(int device_ordinal, wg_ctx_t **out) {
  if (!out) return fail(WG_ERR_BAD_ARG, "null out pointer");
  *out = nullptr;
  return make_ctx(device_ordinal, out);
}

void wg_ctx_destroy(wg_ctx_t *ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (ctx == g_default) g_default = nullptr;
  }
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur != ctx->device) (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  ctx->release_all();
  delete ctx;
}

int wg_ctx_device(const wg_ctx_t *ctx) { return ctx ? ctx->device : -1; }

int wg_set_overlap_strict_ctx(wg_ctx_t *ctx, int on) {
  if (!ctx) return fail(WG_ERR_BAD_ARG, "null context");
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  ctx->overlap_strict = on != 0;
  return WG_OK;
}

long long wg_overlap_serialised_ctx(wg_ctx_t *ctx) {
  if (!ctx) return -1;
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  return ctx->serialised;
}

int wg_shard_range(long long total, int rank, int world, long long *lo, long long *hi) {
  if (total < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi)
    return fail(WG_ERR_BAD_ARG, "wg_shard_range: need total >= 0, 0 <= rank < world");
  const long long base = total / world, rem = total % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
  return WG_OK;
}

int wg_init(int device_ordinal) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (g_default && g_default->device == device_ordinal) return use_ctx(g_default);
  wg_ctx *fresh = nullptr;
  if (int rc = make_ctx(device_ordinal, &fresh)) return rc;
  if (g_default) { g_default->release_all(); delete g_default; }
  g_default = fresh;
  return WG_OK;
}

void wg_shutdown(void) {
  wg_ctx *c = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_default_mu);
    c = g_default;
    g_default = nullptr;
  }
  if (c) { c->release_all(); delete c; }
}

#ifdef WG_PROFILE
// diagnostic build only: read-and-reset the in-kernel phase timers (shader cycles)
int wg_prof_read(unsigned long long *out48) {          // 48 counters (wg_ql_device.hpp, g_prof)
  if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(wg::g_prof), 48 * sizeof(unsigned long long)) != hipSuccess) return -1;
  unsigned long long z[48] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(wg::g_prof), z, sizeof z) != hipSuccess) return -1;
  return 0;
}
#endif

size_t wg_qp_lds_bytes(int n, int m) { return wg::QlDims(n, m, m).bytes(); }

int wg_qp_solve_batch_dev_ctx(wg_ctx_t *ctx, int B, int nmax, int mmax, const int *n, const int *m, const int *me, const double *C, const double *d, const double *A, const double *b, const double *xl, const double *xu, double eps, double *x, double *u, int *ifail, int *n_iter, int *iact, int *nact, int *hist, int hist_cap, int *hist_len, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  if (B < 0 || nmax <= 0 || mmax <= 0) return fail(WG_ERR_BAD_ARG, "bad sizes B=%d nmax=%d mmax=%d", B, nmax, mmax);
  if (!C || !d || !A || !b || !xl || !xu || !x || !ifail) return fail(WG_ERR_BAD_ARG, "null required pointer");
  if (hist && (!hist_len || hist_cap <= 0)) return fail(WG_ERR_BAD_ARG, "hist needs hist_len and hist_cap > 0");
  if (B == 0) return WG_OK;
  const int m_cap = m ? mmax : mmax - 1;
  size_t lds = wg::QlDims(nmax, m_cap, m_cap).bytes();
  // A (m x n, the largest operand) is only ever read: staged in LDS it caps the residency (n = 36, m = 75: 54 KB, three QPs
  // per CU), read in place it comes from L2 and the CU holds five -- measured 274 k vs 223 k QPs/s on the probe's QPs.  It
  // goes to LDS only while that does not cost a resident QP (8 per CU = two waves per SIMD is the useful maximum).
  int a_in_lds = 1, g_in_lds = 1;
  auto lds_for = [&](bool a, bool g) { return wg::QlDims(nmax, m_cap, m_cap, true, a, 0, true, true, true, true, 0, g).bytes(); };
  const size_t lds_noa = lds_for(false, true), lds_noag = lds_for(false, false);
  // residency each placement reaches: LDS, and the registers -- the kernels with G in LDS take 259 registers (one wave per
  // SIMD, four QPs per CU), the one with G in place is compiled for two waves per SIMD
  auto per_cu = [](size_t l, size_t reg_cap) { const size_t k = (160 * 1024) / (l ? l : 1); return k > reg_cap ? reg_cap : k; };
  if (per_cu(lds_noa, 4) > per_cu(lds, 4)) a_in_lds = 0;
  // G follows A out of the LDS when that buys at least two more resident QPs (G is cold after the factorisation; measured on the
  // Herdt workload's real QPs, n = 36, m = 75: 7 per CU against 4)
  if (!a_in_lds && per_cu(lds_noag, 8) >= per_cu(lds_noa, 4) + 2) g_in_lds = 0;
  if (const char *e = getenv("WG_QL_A_IN_LDS")) a_in_lds = atoi(e) != 0;   // tests force either path
  if (const char *e = getenv("WG_QL_G_IN_LDS")) g_in_lds = atoi(e) != 0;
  if (lds > 160 * 1024) a_in_lds = 0;
  if (a_in_lds) g_in_lds = 1;                                              // G leaves only after A
  if (!a_in_lds && lds_noa > 160 * 1024) g_in_lds = 0;
  lds = lds_for(a_in_lds, g_in_lds);
  if (lds > 160 * 1024) return fail(WG_ERR_TOO_LARGE, "QP (n=%d, m=%d) needs %zu B of LDS > 160 KiB", nmax, m_cap, lds);
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  // wa | b follow A and G out of the LDS when that buys one more resident QP (n = 36, m = 75: 21.7 -> 20.2 KB, the eighth QP
  // of the CU -- a batch of 4096 is then exactly two rounds of the 2048 resident waves).  WG_QL_W_IN_LDS=0/1 forces either.
  // The slots are used by one launch at a time: the test "are they free", the launch and the event behind it happen under one
  // lock (two host threads on two streams cannot both find them free); a launch that finds them in use by another stream keeps
  // wa | b in LDS instead of waiting for them.
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  double *wab = nullptr;
  if (!a_in_lds && !g_in_lds) {
    const size_t lds_now = wg::QlDims(nmax, m_cap, m_cap, true, false, 0, true, true, false, true, 0, false).bytes();
    auto gran_per_cu = [](size_t l) { const size_t k = 128 / ((l + 1279) / 1280); return k > 8 ? (size_t)8 : k; };
    bool w_out = gran_per_cu(lds_now) > gran_per_cu(lds);
    if (const char *e = getenv("WG_QL_W_IN_LDS")) w_out = atoi(e) == 0;
    if (w_out && !slot_pending_elsewhere(ctx->qp_order, st)) {
      if (int rc = ctx->qp_slot.reserve((size_t)B * (2 * (size_t)mmax + nmax) * 8)) return rc;
      wab = static_cast<double *>(ctx->qp_slot.p);
      lds = lds_now;
    }
  }
  // more QPs than resident waves: start them longest-solve-first by the iteration counts of the previous batch on the same
  // arrays (scheduling only; a caller that interleaves unrelated batches merely loses the benefit).  WG_QL_LPT=0: index order
  int *order = nullptr, *iters_out = nullptr;
  {
    size_t per_cu = 128 / ((lds + 1279) / 1280);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    // the order lives in a buffer of the context: only a launch that also owns the context's wa | b slots uses it (those are
    // handed to one launch at a time: see above), so no other launch rewrites the order under this one's blocks
    bool lpt = wab != nullptr && (size_t)B > (size_t)ctx->num_cu * per_cu;
    if (const char *e = getenv("WG_QL_LPT")) lpt = lpt && atoi(e) != 0;
    if (lpt) {
      const bool known = ctx->qlpt_key == C && ctx->qlpt_B == B && ctx->qlpt_buf.p;
      if (int rc = ctx->qlpt_buf.reserve((size_t)B * 2 * sizeof(int))) return rc;
      iters_out = static_cast<int *>(ctx->qlpt_buf.p);
      if (known) {
        order = iters_out + B;
        hipLaunchKernelGGL(wg_lpt_order_kernel, dim3(1), dim3(1024), 0, st, B, iters_out, order);
      }
      ctx->qlpt_key = C; ctx->qlpt_B = B;
    }
  }
  // the Herdt-sized boundary (what QPProblem::solve hands over at N = 16: nmax = 36, mmax = 76) has its own instantiation
  bool fixed36 = wab && nmax == 36 && mmax == 76;
  if (const char *e = getenv("WG_QL_FIXED")) fixed36 = fixed36 && atoi(e) != 0;
  if (fixed36) {
    lds = wg::QlView::fixed_dense_bytes<36>();
    hipLaunchKernelGGL((wg_ql_dense_kernel<false, false, false, 36, 76>), dim3(B), dim3(64), lds, st, B, nmax, mmax, n, m, me, C, d, A, b,
                       xl, xu, eps, x, u, ifail, n_iter, iact, nact, hist, hist_cap, hist_len, wab, order, iters_out);
    if (int rc = slot_mark(ctx->qp_order, st)) return rc;
    HIP_TRY(hipGetLastError());
    return WG_OK;
  }
  const void *kfn = a_in_lds ? reinterpret_cast<const void *>(wg_ql_dense_kernel<true, true>)
                             : (g_in_lds ? reinterpret_cast<const void *>(wg_ql_dense_kernel<false, true>)
                                         : (wab ? reinterpret_cast<const void *>(wg_ql_dense_kernel<false, false, false>)
                                                : reinterpret_cast<const void *>(wg_ql_dense_kernel<false, false>)));
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = B;
  if (a_in_lds)
    hipLaunchKernelGGL((wg_ql_dense_kernel<true, true>), dim3(grid), dim3(64), lds, st, B, nmax, mmax, n, m, me, C, d, A, b,
                       xl, xu, eps, x, u, ifail, n_iter, iact, nact, hist, hist_cap, hist_len, wab, order, iters_out);
  else if (g_in_lds)
    hipLaunchKernelGGL((wg_ql_dense_kernel<false, true>), dim3(grid), dim3(64), lds, st, B, nmax, mmax, n, m, me, C, d, A, b,
                       xl, xu, eps, x, u, ifail, n_iter, iact, nact, hist, hist_cap, hist_len, wab, order, iters_out);
  else if (!wab)
    hipLaunchKernelGGL((wg_ql_dense_kernel<false, false>), dim3(grid), dim3(64), lds, st, B, nmax, mmax, n, m, me, C, d, A, b,
                       xl, xu, eps, x, u, ifail, n_iter, iact, nact, hist, hist_cap, hist_len, wab, order, iters_out);
  else {
    hipLaunchKernelGGL((wg_ql_dense_kernel<false, false, false>), dim3(grid), dim3(64), lds, st, B, nmax, mmax, n, m, me, C, d, A, b,
                       xl, xu, eps, x, u, ifail, n_iter, iact, nact, hist, hist_cap, hist_len, wab, order, iters_out);
    if (int rc = slot_mark(ctx->qp_order, st)) return rc;
  }
  HIP_TRY(hipGetLastError());
  return WG_OK;
}

int wg_qp_solve_batch_ctx(wg_ctx_t *ctx, int B, int nmax, int mmax, const int *n, const int *m, const int *me, const double *C, const double *d, const double *A, const double *b, const double *xl, const double *xu, double eps, double *x, double *u, int *ifail, int *n_iter, int *iact, int *nact, int *hist, int hist_cap, int *hist_len) {
  if (int rc = use_ctx(ctx)) return rc;
  if (B < 0 || nmax <= 0 || mmax <= 0) return fail(WG_ERR_BAD_ARG, "bad sizes B=%d nmax=%d mmax=%d", B, nmax, mmax);
  if (!C || !d || !A || !b || !xl || !xu || !x || !ifail) return fail(WG_ERR_BAD_ARG, "null required pointer");
  if (B == 0) return WG_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t sB = (size_t)B;
  // input arena
  struct Seg { const void *h; size_t bytes; size_t off; };
  std::vector<Seg> in = {
      {C, sB * nmax * nmax * 8, 0}, {d, sB * nmax * 8, 0},  {A, sB * mmax * nmax * 8, 0},
      {b, sB * mmax * 8, 0},        {xl, sB * nmax * 8, 0}, {xu, sB * nmax * 8, 0},
      {n, n ? sB * 4 : 0, 0},       {m, m ? sB * 4 : 0, 0}, {me, me ? sB * 4 : 0, 0}};
  size_t tot = 0;
  for (auto &s : in) { s.off = tot; tot += (s.bytes + 255) & ~(size_t)255; }
  if (int rc = ctx->in.reserve(tot)) return rc;
  char *din = static_cast<char *>(ctx->in.p);
  for (auto &s : in)
    if (s.bytes) WG_H2D(din + s.off, s.h, s.bytes);
  struct OSeg { void *h; size_t bytes; size_t off; };
  std::vector<OSeg> out = {{x, sB * nmax * 8, 0},
                           {u, u ? sB * (mmax + 2 * (size_t)nmax) * 8 : 0, 0},
                           {ifail, sB * 4, 0},
                           {n_iter, n_iter ? sB * 4 : 0, 0},
                           {iact, iact ? sB * nmax * 4 : 0, 0},
                           {nact, nact ? sB * 4 : 0, 0},
                           {hist, hist ? sB * hist_cap * 4 : 0, 0},
                           {hist_len, hist_len ? sB * 4 : 0, 0}};
  size_t otot = 0;
  for (auto &s : out) { s.off = otot; otot += (s.bytes + 255) & ~(size_t)255; }
  if (int rc = ctx->out.reserve(otot)) return rc;
  char *dout = static_cast<char *>(ctx->out.p);
  WG_ZERO(dout, otot);
  auto ip = [&](int k) { return in[k].bytes ? din + in[k].off : nullptr; };
  auto op = [&](int k) { return out[k].bytes ? dout + out[k].off : nullptr; };
  int rc = wg_qp_solve_batch_dev_ctx(ctx, B, nmax, mmax, (const int *)ip(6), (const int *)ip(7), (const int *)ip(8),
                                 (const double *)ip(0), (const double *)ip(1), (const double *)ip(2),
                                 (const double *)ip(3), (const double *)ip(4), (const double *)ip(5), eps,
                                 (double *)op(0), (double *)op(1), (int *)op(2), (int *)op(3), (int *)op(4),
                                 (int *)op(5), (int *)op(6), hist_cap, (int *)op(7), ctx->host_stream);
  if (rc) return rc;
  for (auto &s : out)
    if (s.bytes) WG_D2H(s.h, dout + s.off, s.bytes);
  WG_HOST_WAIT();
  return WG_OK;
}

}  // extern "C"

// ===========================================================================
// Herdt-2010 MPC tick, batched (include/wg_mpc.h, second half)
// ===========================================================================

namespace {
inline bool tick_compact(const wg_model_t &m);
// at most two step changes fit in the preview window when N*T <= 2*step_period (each change is one step period
// after the previous one and the first previewed change is at pi >= 1): the compact kernel is sized for that
inline int tick_smax(const wg_model_t &m) { return tick_compact(m) ? 2 : wg::kSMax; }
inline int tick_max_n(const wg_model_t &m) { return 2 * m.N + 2 * tick_smax(m); }
inline int tick_max_m(const wg_model_t &m) { return 1 + 4 * m.N + 5 * tick_smax(m); }
// Problem views of the tick kernel (template argument of wg_mpc_tick_kernel):
//   16  compact  rows in registers, no G / A anywhere: N == 16 with at most two previewed steps (the benchmark model)
//   -1  element  G / A regenerated per element from the compact tables, Z in a global slot: every other model
//    0  dense    G and A as LDS matrices: on request (WG_TICK_DENSE=1, while they fit the CU's 160 KiB) and for
//                wg_mpc_assemble_batch, which writes the QP out
// WG_TICK_VIEW=element sends N == 16 through the element view as well (tests).
inline bool tick_compact(const wg_model_t &m) {
  const char *e = getenv("WG_TICK_DENSE");
  const char *v = getenv("WG_TICK_VIEW");
  return m.N == 16 && m.N * m.T <= 2.0 * m.step_period + 1e-12 && !(e && atoi(e) != 0) && !(v && *v);
}
// element view (-1: any horizon; 32: BASELINE config 5's horizon as a compile-time constant -- same LDS bytes, same slot, a fixed
// layout): Z in a per-block slot of global memory instead of LDS (decided at compile time: mpc_tick<-1>, mpc_tick<32>)
inline bool tick_elem(int view) { return view == -1 || view == 32 || view == 33; }   // 33: N = 32 with Z in registers
inline int tick_waves_per_simd(int view) { return view == 33 ? WG_ZR_WPS : (tick_elem(view) ? WG_TICK32_WPE : WG_TICK_WPE_MAX); }
inline bool tick_z_global(int view) { return tick_elem(view); }
// compact view (N = 16): wa, b and the border block Gv in a per-block slot of global memory (decided at compile time: mpc_tick<16>)
inline bool tick16_ext(int view) { return view == 16; }
// the solver area of a wave's LDS (the tick's own arrays follow it)
inline size_t tick_ql_bytes_for(const wg_model_t &m, int view, int r_cols) {
  const bool ext = tick16_ext(view) || tick_z_global(view);
  return (wg::QlDims(tick_max_n(m), tick_max_m(m), tick_max_m(m), view == 0, true, 0, view == 0, !tick_z_global(view),
                           !ext, !tick_elem(view), r_cols).bytes() + 15) & ~(size_t)15;
}
inline size_t tick_lds_with_cap(const wg_model_t &m, int view, int r_cols) {
  const bool ext = tick16_ext(view) || tick_z_global(view);        // wa, b, Gv (element view: the rows too) in the global slot
  const size_t ql = tick_ql_bytes_for(m, view, r_cols);
  const int gvld = view == 16 ? wg::kGvLd : (tick_elem(view) ? wg::kGvLdElem : 0);
  size_t tick = wg::TickLds::bytes(m.N, tick_smax(m), gvld, view == 16, !ext, !tick_elem(view), tick_elem(view));
  tick = (tick + 15) & ~(size_t)15;                        // the fixed element view puts the solver area behind it
  // element view, short horizons: the pre-solve overlay does not fit over R; it gets its own bytes behind the tick's arrays
  if (view == -1 && wg::TickLds::elem_overlay_apart(m.N, sizeof(wg_gait_state_t)))
    tick = ((tick + 15) & ~(size_t)15) + wg::TickLds::elem_overlay_need(m.N, sizeof(wg_gait_state_t));
  if (view == 33) tick += ((size_t)wg::kZrTile * wg::kZrTs + (size_t)wg::kZrTail * tick_max_n(m)) * 8;   // Z^T a tile + Z's tail rows (behind R)
  return ql + tick;
}
// Element view: how many columns of R the LDS holds (0: all of them).  R is the operand that decides the residency at N = 32
// (21.6 KB of the 26.8): the LDS keeps the first c columns and one working column, and a solve whose active set outgrows them
// moves its R to the per-block global slot and GOES ON there (mpc_tick<-1>, QlResume: no repeat; measured: even a cap most solves
// outgrow costs a few per cent).  So the cap is simply the largest one that reaches the best residency the kernel's register
// budget admits: 41 columns at N = 32 = 12 640 B of LDS = twelve gaits per CU, three on every SIMD.  WG_ELEM_NACT_CAP forces a
// value (tests run with tiny caps so that every solve takes the second route).
inline int tick_elem_cap(const wg_model_t &m, int view) {
  if (!tick_elem(view)) return 0;
  const int n = tick_max_n(m);
  const size_t overlay = wg::TickLds::pre_bytes(m.N, tick_smax(m)) + sizeof(wg_gait_state_t) + 32;
  // the pre-solve overlay -- the parked state copy at its end is fetched back right after the solve, while x still holds the
  // solution -- must lie within R and the four scratch vectors behind it (QlView::carve, lean layout); sized for the smallest
  // problem of the model (no previewed step: n = 2N: working column and scratch vectors of 2N entries each)
  auto fits = [&](int c) { return (size_t)8 * ((size_t)c * (c + 1) / 2 + (size_t)(2 * m.N) + (size_t)(4 * 2 * m.N)) >= overlay; };
  if (wg::TickLds::elem_overlay_apart(m.N, sizeof(wg_gait_state_t))) return 0;     // short horizons: R whole, the overlay apart
  if (const char *e = getenv("WG_ELEM_NACT_CAP")) {
    int c = atoi(e);
    if (c <= 0 || c >= n) return 0;
    while (c < n - 1 && !fits(c)) ++c;
    return (c < n && fits(c)) ? c : 0;
  }
  // waves a CU holds by the registers the element view's kernels are compiled for (WG_TICK32_WPE per SIMD, four SIMDs)
  const size_t wcap = 4 * (size_t)tick_waves_per_simd(view);
  auto per_cu = [&](int c) { const size_t g = (tick_lds_with_cap(m, view, c) + 1279) / 1280; size_t k = 128 / g; return k > wcap ? wcap : k; };
  const size_t full = per_cu(0);
  int best = 0;
  size_t best_k = full;
  for (int c = n - 1; c >= n / 4; --c)                   // descending: the first cap that reaches a residency is the largest
    if (fits(c) && per_cu(c) > best_k) { best = c; best_k = per_cu(c); }
  return best;
}
// what the kernels receive: the column cap in the low 16 bits; tests may ask the solver to give up EARLIER than the layout
// requires (WG_ELEM_ABORT_AT: active-set size at which the first attempt stops), so that the second route is taken often
inline int tick_elem_cap_arg(const wg_model_t &m, int view) {
  const int c = tick_elem_cap(m, view);
  if (!c) return 0;
  int a = c;
  if (const char *e = getenv("WG_ELEM_ABORT_AT")) { const int v = atoi(e); if (v >= 1 && v < c) a = v; }
  return c | (a << 16);
}
inline size_t tick_z_slot_doubles(const wg_model_t &m, int view) {
  const size_t n = (size_t)tick_max_n(m), mm = (size_t)tick_max_m(m);
  if (view == 16) return (n + 2 * mm) + n * wg::kGvLd;       // wa | b | Gv
  // element view: Z | wa | b | Gv | rowA | rowB | rowK | gd | d | wd | wx | R in full (mpc_tick<-1>)
  // (the fixed N = 32 view keeps Z with leading dimension n: whole cache lines per column; slots are multiples of 64 bytes)
  const size_t zd = (view == 32 || view == 33) ? n * n : n * (n | 1);
  return (zd + (n + 2 * mm) + n * wg::kGvLdElem + 2 * mm + (mm + 1) / 2 + 2 + 4 * n + (n * (n + 1) / 2 + n) + 7) & ~(size_t)7;
}
inline size_t tick_lds_for(const wg_model_t &m, int view) { return tick_lds_with_cap(m, view, tick_elem_cap(m, view)); }
inline int tick_view(const wg_model_t &m) {
  if (tick_compact(m)) return 16;
  const bool dense_fits = tick_lds_for(m, 0) <= 160 * 1024;
  const char *d = getenv("WG_TICK_DENSE");
  if (d && atoi(d) != 0 && dense_fits) return 0;          // tests: the dense view where the element view would be taken
  // Everywhere else the element view: 5 - 12.6 KB of LDS per gait (twelve per CU, three on every SIMD) against the dense view's
  // G and A as LDS matrices (N = 20: 100 KB, ONE gait per CU -- measured 1.65 M against 0.39 M ticks/s; N = 24: 1.15 M against
  // 0.27 M; same bits).  Its pre-solve group lies over R (short horizons: in bytes of its own, TickLds::elem_overlay_apart)
  // N = 32 (BASELINE config 5) has an instantiation with the horizon as a compile-time constant (mpc_tick<32>: every slot and
  // LDS offset a constant, only the two-rows-per-lane forms of the solver); WG_TICK_ELEM_GENERIC=1 keeps the any-horizon
  // kernel there too (tests run both: same bytes)
  if (m.N == 32) {
    const char *g = getenv("WG_TICK_ELEM_GENERIC");
    if (g && atoi(g) != 0) return -1;
#ifdef WG_WITH_REGZ
    const char *z = getenv("WG_TICK_REGZ");              // experiment builds: Z in registers, four gaits per CU (mpc_tick<33>)
    if (z && atoi(z) != 0) return 33;
#endif
    return 32;
  }
  return -1;
}
inline size_t tick_ql_bytes(const wg_model_t &m) { return tick_ql_bytes_for(m, tick_view(m), tick_elem_cap(m, tick_view(m))); }
}  // namespace

extern "C" {

void wg_model_defaults(wg_model_t *m) {
  memset(m, 0, sizeof *m);
  m->N = 16; m->flags = 0; m->T = 0.1; m->Tctrl = 0.005; m->com_height_qp = 0.814;
  m->alpha = 1.0; m->beta = 0.00001; m->gamma = 0.000001;
  m->sole_w = 0.25; m->sole_h = 0.14;
  m->margin_x = 0.04; m->margin_y = 0.04; m->ds_feet_distance = 0.2;
  m->hip_l_lo = -30.0 / 180.0 * wg::kPi; m->hip_l_hi = 45.0 / 180.0 * wg::kPi;
  m->hip_r_lo = -30.0 / 180.0 * wg::kPi; m->hip_r_hi = 45.0 / 180.0 * wg::kPi;
  m->hip_vmax = 0.0; m->hip_amax = 0.1; m->feet_cross_max = 5.0 / 180.0 * wg::kPi;
  m->step_period = 0.8; m->ds_period = 1e9; m->dsss_period = 0.8;
  m->t_single = 0.7; m->t_double = 0.1; m->step_height = 0.05; m->feet_distance = 0.2;
}

void wg_gait_init(const wg_model_t *, wg_gait_state_t *s, const double com0[3], const double left_xyt[3],
                  const double right_xyt[3]) {
  memset(s, 0, sizeof *s);
  s->online = 1; s->time_to_stop = -1.0;
  for (int k = 0; k < 3; k++) {
    s->lf[k].x = left_xyt[0]; s->lf[k].y = left_xyt[1]; s->lf[k].theta = left_xyt[2];
    s->rf[k].x = right_xyt[0]; s->rf[k].y = right_xyt[1]; s->rf[k].theta = right_xyt[2];
  }
  s->phase = WG_DS; s->foot = WG_LEFT; s->time_limit = 1000000000; s->nb_steps_left = 1;
  s->sup_x = left_xyt[0]; s->sup_y = left_xyt[1]; s->sup_yaw = left_xyt[2] * wg::kPi / 180;
  s->com_x[0] = com0[0]; s->com_y[0] = com0[1]; s->com_z = com0[2];
  s->front_com_x[0] = com0[0]; s->front_com_y[0] = com0[1];
  s->nb_steps_ssds = 2; s->rot_support_foot = WG_LEFT;
}

int wg_mpc_configure_ctx(wg_ctx_t *ctx, const wg_model_t *model) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!model) return fail(WG_ERR_BAD_ARG, "null model");
  if (model->N < 2 || model->N > wg::kNMaxH) return fail(WG_ERR_BAD_ARG, "N=%d outside [2,%d]", model->N, wg::kNMaxH);
  if ((int)(model->T / model->Tctrl) != WG_SAMPLES_PER_TICK)
    return fail(WG_ERR_BAD_ARG, "T/Tctrl must be %d", WG_SAMPLES_PER_TICK);
  size_t lds = tick_lds_for(*model, tick_view(*model));
  if (lds > 160 * 1024) return fail(WG_ERR_TOO_LARGE, "tick needs %zu B of LDS > 160 KiB", lds);
  std::vector<double> qb;                              // Q_b from the matrix cores, when the model asks for it
  if (model->flags & (WG_FLAG_GRAMIAN_MFMA_F64 | WG_FLAG_GRAMIAN_MFMA_F32)) {
    qb.resize((size_t)model->N * model->N);
    const int prec = (model->flags & WG_FLAG_GRAMIAN_MFMA_F32) ? WG_GRAMIAN_F32 : WG_GRAMIAN_F64;
    if (int rc = wg_gramian_batch_ctx(ctx, 1, model->N, &model->T, &model->com_height_qp, model->alpha, model->beta, model->gamma,
                                  prec, qb.data()))
      return rc;
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  std::unique_ptr<wg::TickTables> host_tables_p(new wg::TickTables);
  wg::TickTables &host_tables = *host_tables_p;
  wg::build_tables(*model, host_tables, qb.empty() ? nullptr : qb.data());
  if (!host_tables.blocks_ok && !qb.empty())
    return fail(WG_ERR_BAD_ARG, "the matrix-core Gramian is not positive definite enough for ql0002's factorisation");
  if (!ctx->tables_dev) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->tables_dev), sizeof(wg::TickTables)));
  // a launch of an earlier configuration may still be reading the tables: this context's own launches, on whatever stream they
  // went (every tick / run / assemble launch leaves an event) -- other contexts and other work on the device are not waited for
  if (int rc = ctx_wait_own(ctx)) return rc;
  WG_H2D(ctx->tables_dev, &host_tables, sizeof host_tables);
  if (!ctx->model_dev) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->model_dev), sizeof(wg_model_t)));
  WG_H2D(ctx->model_dev, model, sizeof(wg_model_t));
  WG_HOST_WAIT();
  ctx->model = *model;
  ctx->model_set = true;
  // The queue and the per-block solver slots of the tick / run kernels are sized by the launches themselves, for the grid they have
  // (a one-robot facade object -- B = 1 through wg_mpc_tick_pinned -- pays for ONE slot, 74 KB at N = 32, not for a fleet's
  // 230 MB); wg_mpc_reserve sizes them ahead of time for a fleet that wants no allocation on its first launch.
  return WG_OK;
}

size_t wg_mpc_tick_lds_bytes_for(const wg_model_t *model) {   // host arithmetic only: no device needed
  if (!model || model->N < 2 || model->N > wg::kNMaxH) return 0;
  return tick_lds_for(*model, tick_view(*model));
}

size_t wg_mpc_tick_lds_bytes_ctx(wg_ctx_t *ctx) {
  if (!ctx || !ctx->model_set) return 0;
  return tick_lds_for(ctx->model, tick_view(ctx->model));
}

/* The tick / run kernels' per-block solver slots and queue for fleets of up to max_gaits, allocated now instead of by the first
 * launch that needs them (the launches size them for their own grid otherwise). */
int wg_mpc_reserve_ctx(wg_ctx_t *ctx, int max_gaits) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->model_set) return fail(WG_ERR_BAD_ARG, "wg_mpc_configure() has not been called on this context");
  if (max_gaits < 1) return fail(WG_ERR_BAD_ARG, "max_gaits = %d", max_gaits);
  const int view = tick_view(ctx->model);
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  if (tick_z_global(view) || tick16_ext(view))
    if (int rc = ctx->tick_z.reserve((size_t)max_gaits * tick_z_slot_doubles(ctx->model, view) * 8)) return rc;
  int cap = 1;
  while (cap < 2 * max_gaits) cap <<= 1;
  return ctx->run_buf.reserve(sizeof(wg_xrun_ctl) + (size_t)kXcds * cap * 8 + (size_t)max_gaits * 4);
}

}  // extern "C"
namespace {
int tick_launch(wg_ctx_t *ctx, int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag, int advance_calls, int *hist, int hist_cap, int *hist_len, void *hip_stream, wg_gait_state_t *host_states, int *host_done) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->model_set) return fail(WG_ERR_BAD_ARG, "wg_mpc_configure() has not been called on this context");
  if (B < 0 || !states) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (hist && (!hist_len || hist_cap <= 0)) return fail(WG_ERR_BAD_ARG, "hist needs hist_len and hist_cap > 0");
  if (B == 0) return WG_OK;
  const size_t qlb = tick_ql_bytes(ctx->model);
  const int view = tick_view(ctx->model);
  size_t lds = tick_lds_for(ctx->model, view);
  if (const char *pad = getenv("WG_TICK_LDS_PAD")) lds += (size_t)atoi(pad);   // experiments: lower the residency
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(WG_KERNEL_BY_VIEW(wg_mpc_tick_kernel, view), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = B;                             // one gait per block; the dispatcher balances uneven iteration counts
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);   // ordering test, launch and event record are one critical section
  if (int rc = slot_claim(ctx, ctx->guard_order, st, "tick / run")) return rc;
  const int ecap = tick_elem_cap_arg(ctx->model, view);
  double *zs = nullptr;
  const size_t zslot = tick_z_slot_doubles(ctx->model, view);
  if (tick_z_global(view) || tick16_ext(view)) {
    // sized here, for the grid this launch has (a one-robot facade pays for one slot, a fleet for its resident waves); growing
    // never frees what a launch in flight may be using (DevBuf); wg_mpc_reserve sizes it ahead of time
    if (int rc = ctx->tick_z.reserve((size_t)grid * zslot * 8)) return rc;
    zs = static_cast<double *>(ctx->tick_z.p);
  }
  // more gaits than resident waves: start them longest-solve-first (see wg_lpt_order_kernel); the iteration counts are those of
  // the previous call on the same state array -- a prediction, so a caller that interleaves batches merely loses the benefit
  int *order = nullptr, *iters_out = nullptr;
  {
    int per_cu = 128 / (int)((lds + 1279) / 1280);
    const int max_waves = 4 * tick_waves_per_simd(view);
    if (per_cu > max_waves) per_cu = max_waves;
    bool lpt = B > ctx->num_cu * per_cu && !host_states;
    if (const char *e = getenv("WG_TICK_LPT")) lpt = lpt && atoi(e) != 0;
    if (lpt) {
      const bool known = ctx->lpt_states == states && ctx->lpt_B == B && ctx->lpt_buf.p;
      if (int rc = ctx->lpt_buf.reserve((size_t)B * 2 * sizeof(int))) return rc;
      iters_out = static_cast<int *>(ctx->lpt_buf.p);
      if (known) {
        order = iters_out + B;
        hipLaunchKernelGGL(wg_lpt_order_kernel, dim3(1), dim3(1024), 0, st, B, iters_out, order);
      }
      ctx->lpt_states = states; ctx->lpt_B = B;
    }
  }
  WG_LAUNCH_BY_VIEW(wg_mpc_tick_kernel, view, grid, lds, st, B, ctx->model, ctx->tables_dev, states, outs, diag, advance_calls, hist, hist_cap,
                    hist_len, (unsigned)qlb, zs, (unsigned)zslot, ecap, host_states, host_done, order, iters_out);
  HIP_TRY(hipGetLastError());
  return slot_mark(ctx->guard_order, st);
}
}  // namespace
extern "C" {

int wg_mpc_tick_batch_dev_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag, int advance_calls, int *hist, int hist_cap, int *hist_len, void *hip_stream) {
  return tick_launch(ctx, B, states, outs, diag, advance_calls, hist, hist_cap, hist_len, hip_stream, nullptr, nullptr);
}

/* ---- one robot (BASELINE configs[1]): state and outputs in host-mapped memory, no copies, no device synchronisation ---- */
int wg_host_alloc(void **out, size_t bytes) {
  if (!out || !bytes) return fail(WG_ERR_BAD_ARG, "wg_host_alloc: null pointer or zero size");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(WG_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocPortable));   // visible to every device's contexts
  memset(*out, 0, bytes);
  return WG_OK;
}

void wg_host_free(void *p) { if (p) (void)hipHostFree(p); }

int wg_mpc_tick_pinned_ctx(wg_ctx_t *ctx, wg_gait_state_t *state, wg_tick_out_t *out, int *diag, int advance_calls) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->model_set) return fail(WG_ERR_BAD_ARG, "wg_mpc_configure() has not been called on this context");
  if (!state) return fail(WG_ERR_BAD_ARG, "null state");
  void *d_state = nullptr, *d_out = nullptr, *d_diag = nullptr;
  if (hipHostGetDevicePointer(&d_state, state, 0) != hipSuccess || (out && hipHostGetDevicePointer(&d_out, out, 0) != hipSuccess) ||
      (diag && hipHostGetDevicePointer(&d_diag, diag, 0) != hipSuccess)) {
    (void)hipGetLastError();
    return fail(WG_ERR_BAD_ARG, "wg_mpc_tick_pinned: state / out / diag must come from wg_host_alloc (host-mapped memory)");
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->pin_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->pin_stream, hipStreamNonBlocking));
  if (!ctx->pin_flag) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->pin_flag), 64, hipHostMallocMapped));
    *ctx->pin_flag = 0; ctx->pin_seq = 0;
  }
  if (int rc = ctx->tick_state.reserve(sizeof(wg_gait_state_t))) return rc;
  void *d_flag = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&d_flag, ctx->pin_flag, 0));
  const int want = ++ctx->pin_seq;
  int rc = tick_launch(ctx, 1, static_cast<wg_gait_state_t *>(ctx->tick_state.p), static_cast<wg_tick_out_t *>(d_out),
                       static_cast<int *>(d_diag), advance_calls, nullptr, 0, nullptr, ctx->pin_stream,
                       static_cast<wg_gait_state_t *>(d_state), static_cast<int *>(d_flag));
  if (rc) { --ctx->pin_seq; return rc; }
  // the kernel's last act is a system-scope release on the counter: spin on it (a stream synchronise costs more than the
  // copies this path avoids); fall back to the runtime if it does not move for a long time (a fault, a hung device)
  volatile int *flag = ctx->pin_flag;
  for (long spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != want; ++spins) {
    if (spins > 200000000L) { HIP_TRY(hipStreamSynchronize(ctx->pin_stream)); break; }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) return fail(WG_ERR_HIP, "wg_mpc_tick_pinned: the kernel ended without signalling");
  return WG_OK;
}

/* ---- the QP of every gait's next tick at the ql0001_ boundary (QPProblem::dump_problem) -------------------------------- */
int wg_mpc_assemble_batch_dev_ctx(wg_ctx_t *ctx, int B, const wg_gait_state_t *states, int advance_calls, int nmax, int mmax, double *C, double *d, double *A, double *b, double *xl, double *xu, int *n, int *m, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->model_set) return fail(WG_ERR_BAD_ARG, "wg_mpc_configure() has not been called on this context");
  if (B < 0 || !states || !C || !d || !A || !b || !xl || !xu || !n || !m) return fail(WG_ERR_BAD_ARG, "bad arguments");
  const wg_model_t &M = ctx->model;
  const int need_n = 2 * M.N + 2 * wg::kSMax, need_m = 1 + 4 * M.N + 5 * wg::kSMax;
  const int need_n2 = tick_compact(M) ? 2 * M.N + 4 : need_n, need_m2 = tick_compact(M) ? 1 + 4 * M.N + 10 : need_m;
  if (nmax < need_n2 || mmax < need_m2 + 1)
    return fail(WG_ERR_BAD_ARG, "nmax >= %d and mmax >= %d needed for this model (mmax = m + 1, qp-problem.cpp:250)", need_n2, need_m2 + 1);
  if (B == 0) return WG_OK;
  // the dense view, laid out for the largest problem any model of this horizon can pose
  const size_t qlb = (wg::QlDims(need_n, need_m, need_m).bytes() + 15) & ~(size_t)15;
  const size_t lds = qlb + wg::TickLds::bytes(M.N, wg::kSMax, 0, false, true, true, false);
  if (lds > 160 * 1024) return fail(WG_ERR_TOO_LARGE, "the dense view of this model needs %zu B of LDS > 160 KiB", lds);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wg_mpc_assemble_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // the scratch copies of the states are the context's: assemble launches of one context are ordered like its tick / run launches
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  if (int rc = slot_claim(ctx, ctx->asm_order, st, "assemble")) return rc;
  if (int rc = ctx->asm_state.reserve((size_t)B * sizeof(wg_gait_state_t))) return rc;
  hipLaunchKernelGGL(wg_mpc_assemble_kernel, dim3(B), dim3(64), lds, st, B, ctx->model,
                     ctx->tables_dev, states, static_cast<wg_gait_state_t *>(ctx->asm_state.p), advance_calls, (unsigned)qlb, nmax, mmax,
                     C, d, A, b, xl, xu, n, m);
  HIP_TRY(hipGetLastError());
  return slot_mark(ctx->asm_order, st);
}

int wg_mpc_assemble_batch_ctx(wg_ctx_t *ctx, int B, const wg_gait_state_t *states, int advance_calls, int nmax, int mmax, double *C, double *d, double *A, double *b, double *xl, double *xu, int *n, int *m) {
  if (int rc = use_ctx(ctx)) return rc;
  if (B < 0 || !states || !C || !d || !A || !b || !xl || !xu || !n || !m || nmax <= 0 || mmax <= 0) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (B == 0) return WG_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t sB = (size_t)B, sn = (size_t)nmax, sm = (size_t)mmax;
  const size_t off_C = (sB * sizeof(wg_gait_state_t) + 255) & ~(size_t)255, off_d = off_C + sB * sn * sn * 8, off_A = off_d + sB * sn * 8,
               off_b = off_A + sB * sm * sn * 8, off_xl = off_b + sB * sm * 8, off_xu = off_xl + sB * sn * 8, off_n = off_xu + sB * sn * 8,
               off_m = off_n + sB * 4, tot = off_m + sB * 4;
  if (int rc = ctx->in.reserve(tot)) return rc;
  char *base = static_cast<char *>(ctx->in.p);
  WG_H2D(base, states, sB * sizeof(wg_gait_state_t));
  int rc = wg_mpc_assemble_batch_dev_ctx(ctx, B, reinterpret_cast<const wg_gait_state_t *>(base), advance_calls, nmax, mmax,
                                         reinterpret_cast<double *>(base + off_C), reinterpret_cast<double *>(base + off_d),
                                         reinterpret_cast<double *>(base + off_A), reinterpret_cast<double *>(base + off_b),
                                         reinterpret_cast<double *>(base + off_xl), reinterpret_cast<double *>(base + off_xu),
                                         reinterpret_cast<int *>(base + off_n), reinterpret_cast<int *>(base + off_m), ctx->host_stream);
  if (rc) return rc;
  WG_D2H(C, base + off_C, sB * sn * sn * 8);
  WG_D2H(d, base + off_d, sB * sn * 8);
  WG_D2H(A, base + off_A, sB * sm * sn * 8);
  WG_D2H(b, base + off_b, sB * sm * 8);
  WG_D2H(xl, base + off_xl, sB * sn * 8);
  WG_D2H(xu, base + off_xu, sB * sn * 8);
  WG_D2H(n, base + off_n, sB * 4);
  WG_D2H(m, base + off_m, sB * 4);
  WG_HOST_WAIT();
  return WG_OK;
}

int wg_mpc_run_batch_dev_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, int n_ticks, int advance_calls, wg_tick_out_t *outs, int *diag, void *hip_stream) {
  return wg_mpc_run_sched_dev_ctx(ctx, B, states, n_ticks, advance_calls, nullptr, 1, outs, diag, hip_stream);
}

int wg_mpc_run_sched_dev_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, int n_ticks, int advance_calls, const double *vref_sched, int period, wg_tick_out_t *outs, int *diag, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->model_set) return fail(WG_ERR_BAD_ARG, "wg_mpc_configure() has not been called on this context");
  if (B < 0 || n_ticks < 0 || !states || period < 1) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (B == 0 || n_ticks == 0) return WG_OK;
  if (vref_sched) {
    bool xcd = true;
    if (const char *e = getenv("WG_RUN_QUEUE")) xcd = e[0] != 'g';
    if (!xcd) {                                      // the device-wide queue of round 1 has no staged form: one launch per stretch
      for (int t = 0; t < n_ticks; t += period) {
        const int n = n_ticks - t < period ? n_ticks - t : period;
        if (int rc = wg_mpc_set_velref_dev_ctx(ctx, B, states, vref_sched + (size_t)(t / period) * B * 3, hip_stream)) return rc;
        if (int rc = wg_mpc_run_sched_dev_ctx(ctx, B, states, n, advance_calls, nullptr, 1, outs ? outs + (size_t)t * B : nullptr,
                                              diag ? diag + (size_t)t * B * 6 : nullptr, hip_stream))
          return rc;
      }
      return WG_OK;
    }
  }
  if ((long long)B * n_ticks > 0x3fffffffLL) return fail(WG_ERR_TOO_LARGE, "B * n_ticks = %lld work items", (long long)B * n_ticks);
  const int total = B * n_ticks;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);   // ordering test, launch and event record are one critical section
  if (int rc = slot_claim(ctx, ctx->guard_order, st, "tick / run")) return rc;
  // hand-over inside one XCD (default) or through one device-wide queue (WG_RUN_QUEUE=global: A/B tests)
  bool xcd_mode = true;
  if (const char *e = getenv("WG_RUN_QUEUE")) xcd_mode = e[0] != 'g';
  int cap = 1;
  while (cap < 2 * B) cap <<= 1;                   // ring slots per XCD: a gait is in at most one ring, at most once
  {
    const size_t need = xcd_mode ? sizeof(wg_xrun_ctl) + (size_t)kXcds * cap * 8 + (size_t)B * 4
                                 : sizeof(wg_run_queue) + (size_t)(total + B) * 4;
    if (int rc = ctx->run_buf.reserve(need)) return rc;
  }
  wg_run_queue *q = static_cast<wg_run_queue *>(ctx->run_buf.p);
  int *ring = reinterpret_cast<int *>(q + 1), *done = ring + total;
  wg_xrun_ctl *xctl = static_cast<wg_xrun_ctl *>(ctx->run_buf.p);
  unsigned long long *xrings = reinterpret_cast<unsigned long long *>(xctl + 1);
  int *xdone = reinterpret_cast<int *>(xrings + (size_t)kXcds * cap);
  if (xcd_mode) {
    const int items = kXcds * cap > B ? kXcds * cap : B;
    hipLaunchKernelGGL(wg_xrun_init_kernel, dim3((items + 255) / 256), dim3(256), 0, st, B, xctl, xrings, cap, xdone);
  } else
    hipLaunchKernelGGL(wg_run_queue_init_kernel, dim3((total + 255) / 256), dim3(256), 0, st, B, total, q, ring, done);
  const size_t qlb = tick_ql_bytes(ctx->model);
  const int view = tick_view(ctx->model);
  size_t lds = tick_lds_for(ctx->model, view);
  if (const char *pad = getenv("WG_TICK_LDS_PAD")) lds += (size_t)atoi(pad);   // experiments: lower the residency
  if (lds > 64 * 1024) {
    HIP_TRY(hipFuncSetAttribute(WG_KERNEL_BY_VIEW(wg_mpc_run_kernel, view), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(WG_KERNEL_BY_VIEW(wg_mpc_run_xcd_kernel, view), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // as many blocks as the device keeps resident: LDS granules (1280 B, 128 per CU), at most 8 waves of 256 registers per CU
  int per_cu = 128 / (int)((lds + 1279) / 1280);
  const int max_waves = 4 * tick_waves_per_simd(view);   // what the kernel's register budget admits per CU
  if (per_cu > max_waves) per_cu = max_waves;
  if (per_cu < 1) per_cu = 1;
  int grid = ctx->num_cu * per_cu;
  if (grid > B) grid = B;
  const int ecap = tick_elem_cap_arg(ctx->model, view);
  double *zs = nullptr;
  const size_t zslot = tick_z_slot_doubles(ctx->model, view);
  if (tick_z_global(view) || tick16_ext(view)) {
    if (int rc = ctx->tick_z.reserve((size_t)grid * zslot * 8)) return rc;
    zs = static_cast<double *>(ctx->tick_z.p);
  }
  int keep_k = 0;                                    // a wave keeps a gait that is behind its XCD's mean progress (see the kernel)
  if (const char *e = getenv("WG_RUN_KEEP")) keep_k = (e[0] == 'o' || e[0] == '-') ? -1 : atoi(e);
  else if (B <= grid) keep_k = -1;                   // a wave per gait: nothing waits, the launch takes what its slowest gait takes
  if (xcd_mode)
    WG_LAUNCH_BY_VIEW(wg_mpc_run_xcd_kernel, view, grid, lds, st, B, n_ticks, ctx->model_dev, ctx->tables_dev, states, outs, diag, advance_calls,
                      xctl, xrings, cap, xdone, (unsigned)qlb, zs, (unsigned)zslot, vref_sched, period, ecap, keep_k);
  else
    WG_LAUNCH_BY_VIEW(wg_mpc_run_kernel, view, grid, lds, st, B, n_ticks, ctx->model_dev, ctx->tables_dev, states, outs, diag, advance_calls, q,
                      ring, done, (unsigned)qlb, zs, (unsigned)zslot, ecap);
  HIP_TRY(hipGetLastError());
  return slot_mark(ctx->guard_order, st);
}

int wg_mpc_tick_batch_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag, int advance_calls, int *hist, int hist_cap, int *hist_len) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->model_set) return fail(WG_ERR_BAD_ARG, "wg_mpc_configure() has not been called on this context");
  if (B < 0 || !states) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (B == 0) return WG_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t sB = (size_t)B;
  if (int rc = ctx->tick_state.reserve(sB * sizeof(wg_gait_state_t))) return rc;
  if (outs) if (int rc = ctx->tick_out.reserve(sB * sizeof(wg_tick_out_t))) return rc;
  const size_t aux_bytes = sB * 6 * 4 + (hist ? sB * hist_cap * 4 + sB * 4 : 0);
  if (int rc = ctx->tick_aux.reserve(aux_bytes)) return rc;
  WG_H2D(ctx->tick_state.p, states, sB * sizeof(wg_gait_state_t));
  int *d_diag = static_cast<int *>(ctx->tick_aux.p);
  int *d_hist = hist ? d_diag + sB * 6 : nullptr;
  int *d_hlen = hist ? d_hist + sB * hist_cap : nullptr;
  WG_ZERO(ctx->tick_aux.p, aux_bytes);
  int rc = wg_mpc_tick_batch_dev_ctx(ctx, B, static_cast<wg_gait_state_t *>(ctx->tick_state.p),
                                 outs ? static_cast<wg_tick_out_t *>(ctx->tick_out.p) : nullptr, d_diag, advance_calls,
                                 d_hist, hist_cap, d_hlen, ctx->host_stream);
  if (rc) return rc;
  WG_D2H(states, ctx->tick_state.p, sB * sizeof(wg_gait_state_t));
  if (outs) WG_D2H(outs, ctx->tick_out.p, sB * sizeof(wg_tick_out_t));
  if (diag) WG_D2H(diag, d_diag, sB * 6 * 4);
  if (hist) {
    WG_D2H(hist, d_hist, sB * hist_cap * 4);
    WG_D2H(hist_len, d_hlen, sB * 4);
  }
  WG_HOST_WAIT();
  return WG_OK;
}

int wg_mpc_set_velref_dev_ctx(wg_ctx_t *ctx, int B, wg_gait_state_t *states, const double *vref, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  if (B < 0 || !states || !vref) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (B == 0) return WG_OK;
  hipLaunchKernelGGL(wg_set_velref_kernel, dim3((B + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                     B, states, vref);
  HIP_TRY(hipGetLastError());
  return WG_OK;
}

}  // extern "C"

// ---- PLDP / OptCholesky back-end -----------------------------------------------------------------------------------

template <bool kALds>                                      // A's place known at compile time (see wg_ql_dense_kernel)
__global__ void __launch_bounds__(64)
wg_pldp_kernel(int B, int mcap, const wg::PldpModel *__restrict__ model, const int *__restrict__ m,
               const double *__restrict__ D, const double *__restrict__ A, const double *__restrict__ b,
               const double *__restrict__ zmpref, const double *__restrict__ xkyk, const int *__restrict__ similar,
               const int *__restrict__ n_removed, const int *__restrict__ starting, int max_iter,
               wg_pldp_state_t *states, double *X, int *ret, int *n_iter, int *active, int *n_active) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pldp_lds[];
  const wg::PldpModel &M = *model;
  const int n = 2 * M.N;
  const size_t aslot = (size_t)(mcap + 1) * n;
  const int p = blockIdx.x;                        // one problem per block (grid == B)
  if (p < B) {
    int mp = m[p];
    if (mp < 0 || mp > mcap) {                      // refuse rather than index out of the slot
      if (threadIdx.x == 0) { ret[p] = WG_PLDP_BAD_INPUT; if (n_iter) n_iter[p] = 0; if (n_active) n_active[p] = 0; }
      return;
    }
    wg::pldp_problem<kALds>(M, pldp_lds, mcap, mp, D + (size_t)p * n, A + p * aslot, b + (size_t)p * mcap,
                     zmpref + (size_t)p * n, xkyk + (size_t)p * 6, similar + (size_t)p * mcap, n_removed[p], starting[p],
                     max_iter, states + p, X + (size_t)p * n, ret + p, n_iter ? n_iter + p : nullptr,
                     active ? active + (size_t)p * mcap : nullptr, n_active ? n_active + p : nullptr);
  }
}

extern "C" {

size_t wg_pldp_lds_bytes(void) { return wg::PldpLds::bytes(WG_PLDP_MMAX); }

int wg_pldp_configure_ctx(wg_ctx_t *ctx, int N, const double *iPu, const double *Px, const double *Pu) {
  if (int rc = use_ctx(ctx)) return rc;
  if (N < 1 || N > WG_PLDP_N || !iPu || !Px || !Pu) return fail(WG_ERR_BAD_ARG, "wg_pldp_configure: 1 <= N <= %d", WG_PLDP_N);
  std::unique_ptr<wg::PldpModel> host_p(new wg::PldpModel);
  wg::PldpModel &host = *host_p;
  std::lock_guard<std::mutex> lk(ctx->mu);
  memset(&host, 0, sizeof host);
  host.N = N;
  memcpy(host.iPu, iPu, sizeof(double) * N * N);
  memcpy(host.Pu, Pu, sizeof(double) * N * N);
  memcpy(host.Px, Px, sizeof(double) * N * 3);
  // PLDPSolver::PrecomputeiPuPx, PLDPSolver.cpp:263-283 (block diagonal, k ascending)
  for (int i = 0; i < N; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < N; k++) s += iPu[k * N + i] * Px[k * 3 + j];
      host.iPuPx[i * 6 + j] = s;
      host.iPuPx[(i + N) * 6 + j + 3] = s;
    }
  if (!ctx->pldp_dev) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->pldp_dev), sizeof(wg::PldpModel)));
  if (int rc = ctx_wait_own(ctx)) return rc;             // a solve of the previous model may still be reading it
  WG_H2D(ctx->pldp_dev, &host, sizeof host);
  WG_HOST_WAIT();
  ctx->pldp_N = N;
  return WG_OK;
}

int wg_pldp_solve_batch_dev_ctx(wg_ctx_t *ctx, int B, int mcap, const int *m, const double *D, const double *A, const double *b, const double *zmpref, const double *xkyk, const int *similar, const int *n_removed, const int *starting, int max_iter, wg_pldp_state_t *states, double *X, int *ret, int *n_iter, int *active, int *n_active, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->pldp_N) return fail(WG_ERR_BAD_ARG, "wg_pldp_configure() has not been called on this context");
  if (B < 0 || mcap < 1 || mcap > WG_PLDP_MMAX) return fail(WG_ERR_BAD_ARG, "need 1 <= mcap <= %d", WG_PLDP_MMAX);
  if (!m || !D || !A || !b || !zmpref || !xkyk || !similar || !n_removed || !starting || !states || !X || !ret)
    return fail(WG_ERR_BAD_ARG, "null argument");
  if (B == 0) return WG_OK;
  // like the dense QP kernel: A in LDS only while that does not cost a resident problem (8 per CU is the useful maximum)
  size_t lds = wg::PldpLds::bytes(mcap);
  const size_t lds_noa = wg::PldpLds::bytes(mcap, WG_PLDP_ACTIVE_CAP, false, false);
  auto per_cu = [](size_t l) { const size_t k = (160 * 1024) / (l ? l : 1); return k > 8 ? (size_t)8 : k; };
  int a_in_lds = per_cu(lds_noa) > per_cu(lds) ? 0 : 1;
  if (const char *e = getenv("WG_PLDP_A_IN_LDS")) a_in_lds = atoi(e) != 0;   // tests force either path
  if (!a_in_lds) lds = lds_noa;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(a_in_lds ? reinterpret_cast<const void *>(wg_pldp_kernel<true>)
                                         : reinterpret_cast<const void *>(wg_pldp_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = B;
  if (a_in_lds)
    hipLaunchKernelGGL(wg_pldp_kernel<true>, dim3(grid), dim3(64), lds, reinterpret_cast<hipStream_t>(hip_stream), B, mcap,
                       ctx->pldp_dev, m, D, A, b, zmpref, xkyk, similar, n_removed, starting, max_iter, states, X, ret, n_iter,
                       active, n_active);
  else
    hipLaunchKernelGGL(wg_pldp_kernel<false>, dim3(grid), dim3(64), lds, reinterpret_cast<hipStream_t>(hip_stream), B, mcap,
                       ctx->pldp_dev, m, D, A, b, zmpref, xkyk, similar, n_removed, starting, max_iter, states, X, ret, n_iter,
                       active, n_active);
  HIP_TRY(hipGetLastError());
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  return slot_mark(ctx->aux_order, reinterpret_cast<hipStream_t>(hip_stream));
}

int wg_pldp_solve_batch_ctx(wg_ctx_t *ctx, int B, int mcap, const int *m, const double *D, const double *A, const double *b, const double *zmpref, const double *xkyk, const int *similar, const int *n_removed, const int *starting, int max_iter, wg_pldp_state_t *states, double *X, int *ret, int *n_iter, int *active, int *n_active) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->pldp_N) return fail(WG_ERR_BAD_ARG, "wg_pldp_configure() has not been called on this context");
  if (B < 0 || mcap < 1 || mcap > WG_PLDP_MMAX) return fail(WG_ERR_BAD_ARG, "need 1 <= mcap <= %d", WG_PLDP_MMAX);
  if (!m || !D || !A || !b || !zmpref || !xkyk || !similar || !n_removed || !starting || !states || !X || !ret)
    return fail(WG_ERR_BAD_ARG, "null argument");
  if (B == 0) return WG_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t sB = (size_t)B, n = 2 * (size_t)ctx->pldp_N;
  const size_t aslot = (size_t)(mcap + 1) * n;
  // one arena: doubles first, then the state structs (8-byte aligned), then ints
  const size_t nd = sB * (n /*D*/ + aslot + mcap /*b*/ + n /*zmpref*/ + 6 + n /*X*/);
  const size_t ni = sB * (1 /*m*/ + mcap /*similar*/ + 1 + 1 /*n_removed starting*/ + 1 + 1 /*ret n_iter*/ + mcap + 1 /*active n_active*/);
  const size_t bytes = nd * 8 + sB * sizeof(wg_pldp_state_t) + ni * 4;
  if (int rc = ctx->pldp_buf.reserve(bytes)) return rc;
  double *dD = static_cast<double *>(ctx->pldp_buf.p), *dA = dD + sB * n, *db = dA + sB * aslot, *dz = db + sB * mcap,
         *dx = dz + sB * n, *dX = dx + sB * 6;
  wg_pldp_state_t *dst = reinterpret_cast<wg_pldp_state_t *>(dX + sB * n);
  int *dm = reinterpret_cast<int *>(dst + sB), *dsim = dm + sB, *dnr = dsim + sB * mcap, *dstart = dnr + sB,
      *dret = dstart + sB, *dit = dret + sB, *dact = dit + sB, *dnact = dact + sB * mcap;
  WG_H2D(dD, D, sB * n * 8);
  WG_H2D(dA, A, sB * aslot * 8);
  WG_H2D(db, b, sB * mcap * 8);
  WG_H2D(dz, zmpref, sB * n * 8);
  WG_H2D(dx, xkyk, sB * 6 * 8);
  WG_H2D(dst, states, sB * sizeof(wg_pldp_state_t));
  WG_H2D(dm, m, sB * 4);
  WG_H2D(dsim, similar, sB * mcap * 4);
  WG_H2D(dnr, n_removed, sB * 4);
  WG_H2D(dstart, starting, sB * 4);
  WG_ZERO(dret, sB * (3 + mcap) * 4);
  int rc = wg_pldp_solve_batch_dev_ctx(ctx, B, mcap, dm, dD, dA, db, dz, dx, dsim, dnr, dstart, max_iter, dst, dX, dret, dit, dact,
                                   dnact, ctx->host_stream);
  if (rc) return rc;
  WG_D2H(states, dst, sB * sizeof(wg_pldp_state_t));
  WG_D2H(X, dX, sB * n * 8);
  WG_D2H(ret, dret, sB * 4);
  if (n_iter) WG_D2H(n_iter, dit, sB * 4);
  if (active) WG_D2H(active, dact, sB * mcap * 4);
  if (n_active) WG_D2H(n_active, dnact, sB * 4);
  WG_HOST_WAIT();
  return WG_OK;
}

}  // extern "C"

// ---- Dimitrov-2008 tick around PLDP ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(64)
wg_dimitrov_tick_kernel(int B, const wg::DimitrovConst *__restrict__ K, const wg_zmp_polytope_t *__restrict__ polys,
                        wg_dimitrov_state_t *states, wg_dimitrov_out_t *outs, int max_iter) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dim_lds[];
  const int N = K->N;
  const int g = blockIdx.x;                       // one gait per block (grid == B), like the Herdt tick
  // (a longest-solve-first start order as in the QL back-ends below was measured here and gave nothing: 5.19 against 5.20 M
  // ticks/s -- PLDP's four iterations per tick leave nothing to order)
  if (g < B) (void)wg::dimitrov_tick(*K, dim_lds, polys + (size_t)g * N, states + g, outs ? outs + g : nullptr, max_iter);
}

// modes QLD / QLDANDLQ: the same tick with the in-wave ql0002 as its back-end (wg_dimitrov_device.hpp, dimitrov_qld_tick)
template <bool kLQ>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
wg_dimitrov_qld_tick_kernel(int B, const wg::DimitrovConst *__restrict__ K, const wg_zmp_polytope_t *__restrict__ polys,
                            wg_dimitrov_state_t *states, wg_dimitrov_out_t *outs, const int *__restrict__ order,
                            int *__restrict__ iters_out) {
  extern __shared__ __attribute__((aligned(16))) double dimq_lds[];
  const int N = K->N;
  const int g = order ? wg::uni(order[blockIdx.x]) : (int)blockIdx.x;     // longest-solve-first by the previous tick (scheduling only)
  if (g < B) {
    const int it = wg::dimitrov_qld_tick<kLQ>(*K, dimq_lds, polys + (size_t)g * N, states + g, outs ? outs + g : nullptr);
    if (iters_out && (threadIdx.x & 63) == 0) iters_out[g] = it;
  }
}

namespace {
inline size_t dimitrov_lds_bytes() {
  return wg::PldpLds::bytes(WG_PLDP_MMAX, wg::kDimitrovActiveCap, true) + (4 * 2 * WG_PLDP_N + 8) * 8 +
         ((WG_PLDP_N + 1) * 4 + 15) / 16 * 16;
}
}  // namespace

extern "C" {

void wg_dimitrov_defaults(wg_dimitrov_model_t *m) {      // ZMPConstrainedQPFastFormulation.cpp:81-97
  if (!m) return;
  memset(m, 0, sizeof *m);
  m->N = 16; m->T = 0.1; m->Tctrl = 0.005; m->com_height = 0.80; m->alpha = 200.0; m->beta = 1000.0;
}

int wg_dimitrov_configure_ctx(wg_ctx_t *ctx, const wg_dimitrov_model_t *model) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!model) return fail(WG_ERR_BAD_ARG, "null model");
  if (model->N < 1 || model->N > WG_PLDP_N) return fail(WG_ERR_BAD_ARG, "N=%d outside [1,%d]", model->N, WG_PLDP_N);
  if (!(model->T > 0.0) || !(model->Tctrl > 0.0) || (int)(model->T / model->Tctrl) != WG_SAMPLES_PER_TICK)
    return fail(WG_ERR_BAD_ARG, "T/Tctrl must be %d", WG_SAMPLES_PER_TICK);
  if (model->solver != WG_DIMITROV_PLDP && model->solver != WG_DIMITROV_QLD && model->solver != WG_DIMITROV_QLDANDLQ)
    return fail(WG_ERR_BAD_ARG, "solver = %d: WG_DIMITROV_PLDP (0), WG_DIMITROV_QLD (1) or WG_DIMITROV_QLDANDLQ (2)", model->solver);
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->dim_host) ctx->dim_host.reset(new wg::DimitrovConst);
    ctx->dim_set = false;
    if (!wg::DimitrovHost::build(*model, (*ctx->dim_host)))
      return fail(WG_ERR_BAD_ARG, "the LQ factor or the inverse of Pu does not exist for this model");
    if (!ctx->dim_dev) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->dim_dev), sizeof(wg::DimitrovConst)));
    if (int rc = ctx_wait_own(ctx)) return rc;           // a tick of the previous model may still be reading the constants
    WG_H2D(ctx->dim_dev, &(*ctx->dim_host), sizeof (*ctx->dim_host));
    WG_HOST_WAIT();
    ctx->dim_set = true;
  }
  // the PLDPSolver constructor of the reference (:104-109): same iPu, Px, Pu
  return wg_pldp_configure_ctx(ctx, model->N, (*ctx->dim_host).pldp.iPu, (*ctx->dim_host).pldp.Px, (*ctx->dim_host).pldp.Pu);
}

int wg_dimitrov_get_constants_ctx(wg_ctx_t *ctx, double *iLQ, double *OptB, double *OptC, double *Pu, double *iPu, double *Px) {
  if (!ctx) return fail(WG_ERR_BAD_ARG, "null context");
  if (!ctx->dim_set) return fail(WG_ERR_BAD_ARG, "wg_dimitrov_configure() has not been called on this context");
  const size_t N = (size_t)(*ctx->dim_host).N, n = 2 * N;
  if (iLQ) memcpy(iLQ, (*ctx->dim_host).iLQ, 8 * n * n);
  if (OptB) memcpy(OptB, (*ctx->dim_host).OptB, 8 * n * 6);
  if (OptC) memcpy(OptC, (*ctx->dim_host).OptC, 8 * n * n);
  if (Pu) memcpy(Pu, (*ctx->dim_host).pldp.Pu, 8 * N * N);
  if (iPu) memcpy(iPu, (*ctx->dim_host).pldp.iPu, 8 * N * N);
  if (Px) memcpy(Px, (*ctx->dim_host).pldp.Px, 8 * N * 3);
  return WG_OK;
}

int wg_dimitrov_get_qld_constants_ctx(wg_ctx_t *ctx, double *Q, double *OptB, double *OptC, double *PuT) {
  if (!ctx) return fail(WG_ERR_BAD_ARG, "null context");
  if (!ctx->dim_set) return fail(WG_ERR_BAD_ARG, "wg_dimitrov_configure() has not been called on this context");
  const size_t N = (size_t)(*ctx->dim_host).N, n = 2 * N;
  if (Q) memcpy(Q, (*ctx->dim_host).Qq, 8 * n * n);
  if (OptB) memcpy(OptB, (*ctx->dim_host).OptBq, 8 * n * 6);
  if (OptC) memcpy(OptC, (*ctx->dim_host).OptCq, 8 * n * n);
  if (PuT) memcpy(PuT, (*ctx->dim_host).PuTq, 8 * N * N);
  return WG_OK;
}

int wg_dimitrov_tick_batch_dev_ctx(wg_ctx_t *ctx, int B, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *states, wg_dimitrov_out_t *outs, int max_iter, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->dim_set) return fail(WG_ERR_BAD_ARG, "wg_dimitrov_configure() has not been called on this context");
  if (B < 0 || !polys || !states) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (B == 0) return WG_OK;
  hipStream_t stq = reinterpret_cast<hipStream_t>(hip_stream);
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  // more gaits than resident waves: longest-solve-first by the previous tick on the same state array (scheduling only)
  int *order = nullptr, *iters_out = nullptr;
  auto lpt_setup = [&](size_t lds_bytes, size_t wave_cap) -> int {
    size_t per_cu = 128 / ((lds_bytes + 1279) / 1280);
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    // the order lives in a buffer of the context: not while another stream's launch of this context may still be reading it
    bool lpt = (size_t)B > (size_t)ctx->num_cu * per_cu && !slot_pending_elsewhere(ctx->aux_order, stq);
    if (const char *e = getenv("WG_QL_LPT")) lpt = lpt && atoi(e) != 0;
    if (!lpt) return WG_OK;
    const bool known = ctx->dlpt_key == states && ctx->dlpt_B == B && ctx->dlpt_buf.p;
    if (int rc = ctx->dlpt_buf.reserve((size_t)B * 2 * sizeof(int))) return rc;
    iters_out = static_cast<int *>(ctx->dlpt_buf.p);
    if (known) {
      order = iters_out + B;
      hipLaunchKernelGGL(wg_lpt_order_kernel, dim3(1), dim3(1024), 0, stq, B, iters_out, order);
    }
    ctx->dlpt_key = states; ctx->dlpt_B = B;
    return WG_OK;
  };
  if ((*ctx->dim_host).solver != WG_DIMITROV_PLDP) {
    const size_t ldsq = wg::dimitrov_qld_lds_bytes();
    if (int rc = lpt_setup(ldsq, 8)) return rc;            // eight gaits per CU (256 registers: two waves per SIMD)
    if ((*ctx->dim_host).solver == WG_DIMITROV_QLDANDLQ)
      hipLaunchKernelGGL(wg_dimitrov_qld_tick_kernel<true>, dim3(B), dim3(64), ldsq, stq, B, ctx->dim_dev, polys, states, outs, order, iters_out);
    else
      hipLaunchKernelGGL(wg_dimitrov_qld_tick_kernel<false>, dim3(B), dim3(64), ldsq, stq, B, ctx->dim_dev, polys, states, outs, order, iters_out);
    HIP_TRY(hipGetLastError());
    return slot_mark(ctx->aux_order, stq);
  }
  const size_t lds = dimitrov_lds_bytes();
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wg_dimitrov_tick_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = B;
  hipLaunchKernelGGL(wg_dimitrov_tick_kernel, dim3(grid), dim3(64), lds, stq, B, ctx->dim_dev, polys, states, outs, max_iter);
  HIP_TRY(hipGetLastError());
  return slot_mark(ctx->aux_order, stq);
}

int wg_dimitrov_tick_batch_ctx(wg_ctx_t *ctx, int B, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *states, wg_dimitrov_out_t *outs, int max_iter) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->dim_set) return fail(WG_ERR_BAD_ARG, "wg_dimitrov_configure() has not been called on this context");
  if (B < 0 || !polys || !states) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (B == 0) return WG_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t sB = (size_t)B, N = (size_t)(*ctx->dim_host).N;
  const size_t pb = sB * N * sizeof(wg_zmp_polytope_t), sb = sB * sizeof(wg_dimitrov_state_t),
               ob = outs ? sB * sizeof(wg_dimitrov_out_t) : 0;
  if (int rc = ctx->dim_buf.reserve(pb + sb + ob)) return rc;
  unsigned char *base = static_cast<unsigned char *>(ctx->dim_buf.p);
  wg_zmp_polytope_t *dp = reinterpret_cast<wg_zmp_polytope_t *>(base);
  wg_dimitrov_state_t *ds = reinterpret_cast<wg_dimitrov_state_t *>(base + pb);
  wg_dimitrov_out_t *dout = outs ? reinterpret_cast<wg_dimitrov_out_t *>(base + pb + sb) : nullptr;
  WG_H2D(dp, polys, pb);
  WG_H2D(ds, states, sb);
  if (dout) WG_ZERO(dout, ob);
  int rc = wg_dimitrov_tick_batch_dev_ctx(ctx, B, dp, ds, dout, max_iter, ctx->host_stream);
  if (rc) return rc;
  WG_D2H(states, ds, sb);
  if (dout) WG_D2H(outs, dout, ob);
  WG_HOST_WAIT();
  return WG_OK;
}

}  // extern "C"

// ---- Kajita stage-1 preview control -------------------------------------------------------------------------------------

extern "C" {

int wg_preview_configure_ctx(wg_ctx_t *ctx, const wg_preview_gains_t *gains, const double *F) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!gains || !F) return fail(WG_ERR_BAD_ARG, "null argument");
  if (gains->nl < 1 || gains->nl > WG_PREVIEW_NL_MAX) return fail(WG_ERR_BAD_ARG, "need 1 <= nl <= %d", WG_PREVIEW_NL_MAX);
  if (!(gains->T > 0.0)) return fail(WG_ERR_BAD_ARG, "sampling period must be positive");
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->prev_F) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->prev_F), sizeof(double) * WG_PREVIEW_NL_MAX));
  if (int rc = ctx_wait_own(ctx)) return rc;             // a run with the previous window may still be reading the gains
  WG_H2D(ctx->prev_F, F, sizeof(double) * gains->nl);
  WG_HOST_WAIT();
  const double T = gains->T;                                   // PreviewControl.cpp:203-214
  ctx->prev.A01 = T; ctx->prev.A02 = T * T / 2.0; ctx->prev.A12 = T;
  ctx->prev.B0 = T * T * T / 6.0; ctx->prev.B1 = T * T / 2.0; ctx->prev.B2 = T;
  ctx->prev.C2 = -gains->zc / 9.81;
  ctx->prev.Kx0 = gains->Kx[0]; ctx->prev.Kx1 = gains->Kx[1]; ctx->prev.Kx2 = gains->Kx[2]; ctx->prev.Ks = gains->Ks;
  ctx->prev.nl = gains->nl;
  ctx->prev_set = true;
  return WG_OK;
}

int wg_preview_window_ctx(wg_ctx_t *ctx) { return (ctx && ctx->prev_set) ? ctx->prev.nl : 0; }

int wg_preview_run_batch_dev_ctx(wg_ctx_t *ctx, int B, int L, const double *zmp_x_tm, const double *zmp_y_tm, double *state, double *com_tm, double *zmp2_tm, int simulation, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->prev_set) return fail(WG_ERR_BAD_ARG, "wg_preview_configure() has not been called on this context");
  if (B < 0 || L < 0 || !zmp_x_tm || !zmp_y_tm || !state) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (B == 0 || L == 0) return WG_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  const char *force = getenv("WG_PREVIEW_KERNEL");            // "l2" / "ring" / "split": tests hold each to the oracle
  // The ring kernel keeps one wave per CU (its window fills the LDS): it wins while the batch is too small to give every
  // SIMD several waves of the L2 kernel (measured: B = 4096: 0.62 vs 0.43 G gait-steps/s; B = 32768: 1.18 vs 1.50), and
  // a few steps do not repay filling the ring.
  const bool ring = force ? force[0] == 'r' : (L >= 8 && (long long)B * 2 <= (long long)ctx->num_cu * 64 * 2);
  // The split-chain kernel (eight lanes per gait-axis, nothing re-read) covers the standard window sizes and wins at every
  // batch size measured; other windows, and runs too short to repay filling its rings, use the kernels below.
  constexpr int kSplitK = 8;
  int splitT = 0;                                              // smallest instantiated T with K T >= nl
  for (int t : {16, 24, 32, 40, 48})
    if (!splitT && kSplitK * t >= ctx->prev.nl) splitT = t;
  const bool can_split = splitT != 0 && ctx->prev.nl >= 64;
  const bool split = force ? (force[0] == 's' && can_split) : (can_split && L >= 4);
  if (split) {
    const int per_wave = 64 / kSplitK;
    const dim3 grid((B + per_wave - 1) / per_wave, 2);
    const size_t lds = (size_t)splitT * 64 * 8;
    const bool full = ctx->prev.nl % splitT == 0;
#define WG_SPLIT_LAUNCH(TT)                                                                                              \
    do {                                                                                                                 \
      if (full)                                                                                                          \
        hipLaunchKernelGGL((wg::wg_preview_split_kernel<TT, kSplitK, true>), grid, dim3(64), lds, st, B, L, ctx->prev,      \
                           ctx->prev_F, zmp_x_tm, zmp_y_tm, state, com_tm, zmp2_tm, simulation);                            \
      else                                                                                                               \
        hipLaunchKernelGGL((wg::wg_preview_split_kernel<TT, kSplitK, false>), grid, dim3(64), lds, st, B, L, ctx->prev,     \
                           ctx->prev_F, zmp_x_tm, zmp_y_tm, state, com_tm, zmp2_tm, simulation);                            \
    } while (0)
    switch (splitT) {
      case 16: WG_SPLIT_LAUNCH(16); break;
      case 24: WG_SPLIT_LAUNCH(24); break;
      case 32: WG_SPLIT_LAUNCH(32); break;
      case 40: WG_SPLIT_LAUNCH(40); break;
      default: WG_SPLIT_LAUNCH(48); break;
    }
#undef WG_SPLIT_LAUNCH
  } else if (ring) {
    int R = ctx->prev.nl < 288 ? ctx->prev.nl : 288;                 // 288 x 512 B = 144 KB of the CU's 160 KB
    if (R < 1) R = 1;
    const size_t lds = (size_t)R * 64 * 8;
    if (lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wg::wg_preview_ring_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(wg::wg_preview_ring_kernel, dim3((B + 63) / 64, 2), dim3(64), lds, st, B, L, ctx->prev, R, ctx->prev_F,
                       zmp_x_tm, zmp_y_tm, state, com_tm, zmp2_tm, simulation);
  } else {
    const int threads = B >= 4096 ? 256 : 64;                  // small batches: more blocks, one wave each
    hipLaunchKernelGGL(wg::wg_preview_kernel, dim3((B + threads - 1) / threads, 2), dim3(threads), 0, st, B, L, ctx->prev,
                       ctx->prev_F, zmp_x_tm, zmp_y_tm, state, com_tm, zmp2_tm, simulation);
  }
  HIP_TRY(hipGetLastError());
  std::lock_guard<std::mutex> launch_lk(ctx->launch_mu);
  return slot_mark(ctx->aux_order, st);
}

int wg_preview_run_batch_ctx(wg_ctx_t *ctx, int B, int L, const double *zmp_x, const double *zmp_y, double *state, double *com, double *zmp2, int simulation) {
  if (int rc = use_ctx(ctx)) return rc;
  if (!ctx->prev_set) return fail(WG_ERR_BAD_ARG, "wg_preview_configure() has not been called on this context");
  if (B < 0 || L < 0 || !zmp_x || !zmp_y || !state) return fail(WG_ERR_BAD_ARG, "bad arguments");
  if (B == 0 || L == 0) return WG_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t sB = (size_t)B, sL = (size_t)L, Lz = sL + ctx->prev.nl - 1;
  // arena: gait-major staging (largest user: com, B x L x 6), time-major zx, zy, com, zmp2, state
  const size_t stage = sB * (sL * 6 > Lz ? sL * 6 : Lz);
  const size_t nd = stage + 2 * sB * Lz + sB * sL * 6 + sB * sL * 2 + sB * 8;
  if (int rc = ctx->prev_buf.reserve(nd * 8)) return rc;
  double *d_stage = static_cast<double *>(ctx->prev_buf.p), *d_zx = d_stage + stage, *d_zy = d_zx + sB * Lz,
         *d_com = d_zy + sB * Lz, *d_z2 = d_com + sB * sL * 6, *d_st = d_z2 + sB * sL * 2;
  auto transpose = [&](int rows, int cols, const double *in, double *out) {
    hipLaunchKernelGGL(wg::wg_transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, ctx->host_stream, rows,
                       cols, in, out);
  };
  // everything below is in order on the context's stream: the staging buffer is reused only behind the transpose that read it
  WG_H2D(d_stage, zmp_x, sB * Lz * 8);
  transpose(B, (int)Lz, d_stage, d_zx);
  WG_H2D(d_stage, zmp_y, sB * Lz * 8);
  transpose(B, (int)Lz, d_stage, d_zy);
  WG_H2D(d_st, state, sB * 8 * 8);
  int rc = wg_preview_run_batch_dev_ctx(ctx, B, L, d_zx, d_zy, d_st, com ? d_com : nullptr, zmp2 ? d_z2 : nullptr, simulation,
                                    ctx->host_stream);
  if (rc) return rc;
  WG_D2H(state, d_st, sB * 8 * 8);
  if (com) {                                                   // [L*6][B] -> [B][L*6]
    transpose(L * 6, B, d_com, d_stage);
    WG_D2H(com, d_stage, sB * sL * 6 * 8);
  }
  if (zmp2) {
    transpose(L * 2, B, d_z2, d_stage);
    WG_D2H(zmp2, d_stage, sB * sL * 2 * 8);
  }
  HIP_TRY(hipGetLastError());
  WG_HOST_WAIT();
  return WG_OK;
}

}  // extern "C"

// ---- invariant Hessian block on the matrix cores (fleets with per-gait models) -----------------------------------------

extern "C" {

int wg_gramian_batch_dev_ctx(wg_ctx_t *ctx, int B, int N, const double *T, const double *h, double alpha, double beta, double gamma, int precision, double *Qb, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  if (B < 0 || N < 1 || N > 32 || !T || !h || !Qb) return fail(WG_ERR_BAD_ARG, "need B >= 0, 1 <= N <= 32, non-null arrays");
  if (precision != WG_GRAMIAN_F64 && precision != WG_GRAMIAN_F32) return fail(WG_ERR_BAD_ARG, "unknown precision %d", precision);
  if (B == 0) return WG_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  if (precision == WG_GRAMIAN_F32)
    hipLaunchKernelGGL(wg::wg_gramian_kernel<true>, dim3(B), dim3(64), 0, st, B, N, T, h, alpha, beta, gamma, Qb);
  else
    hipLaunchKernelGGL(wg::wg_gramian_kernel<false>, dim3(B), dim3(64), 0, st, B, N, T, h, alpha, beta, gamma, Qb);
  HIP_TRY(hipGetLastError());
  return WG_OK;
}

int wg_gramian_batch_ctx(wg_ctx_t *ctx, int B, int N, const double *T, const double *h, double alpha, double beta, double gamma, int precision, double *Qb) {
  if (int rc = use_ctx(ctx)) return rc;
  if (B < 0 || N < 1 || N > 32 || !T || !h || !Qb) return fail(WG_ERR_BAD_ARG, "need B >= 0, 1 <= N <= 32, non-null arrays");
  if (B == 0) return WG_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t sB = (size_t)B, nq = sB * N * N;
  if (int rc = ctx->gram_buf.reserve((2 * sB + nq) * 8)) return rc;
  double *dT = static_cast<double *>(ctx->gram_buf.p), *dh = dT + sB, *dQ = dh + sB;
  WG_H2D(dT, T, sB * 8);
  WG_H2D(dh, h, sB * 8);
  if (int rc = wg_gramian_batch_dev_ctx(ctx, B, N, dT, dh, alpha, beta, gamma, precision, dQ, ctx->host_stream)) return rc;
  WG_D2H(Qb, dQ, nq * 8);
  WG_HOST_WAIT();
  return WG_OK;
}

}  // extern "C"

// ---- Kajita stage-1 inputs: ZMPDiscretization, batched (one lane per gait) ----------------------------------------------

namespace {
// InitializeFilter, ZMPDiscretization.cpp:240-262 (sin from include/wg_trig.h: same bits on host and device)
int zd_make_const(const wg_zmpdisc_model_t *model, wg::ZdConst *K) {
  if (!model) return fail(WG_ERR_BAD_ARG, "null model");
  if (!(model->T > 0.0)) return fail(WG_ERR_BAD_ARG, "sampling period must be positive");
  const int n = (int)floor(0.05 / model->T);
  if (n < 1 || n + 1 > WG_ZD_WIN_MAX)
    return fail(WG_ERR_BAD_ARG, "filter window of %d taps unsupported (1 < taps <= %d)", n + 1, WG_ZD_WIN_MAX);
  K->M = *model;
  K->nwin = n + 1;
  K->pad_ = 0;
  double sum = 0;
  for (int i = 0; i < n + 1; i++) {
    const double tmp = wg_sin((WG_ZD_PI * i) / n);
    K->win[i] = tmp * tmp;
  }
  for (int i = 0; i < n + 1; i++) sum += K->win[i];
  for (int i = 0; i < n + 1; i++) K->win[i] /= sum;
  for (int i = n + 1; i < WG_ZD_WIN_MAX; i++) K->win[i] = 0.0;
  return WG_OK;
}

int zd_launch(const wg::ZdConst &K, int B, int smax, const wg_rel_step_t *steps, const int *n_steps,
              const double *init_feet, int lcap, const wg::ZdOut &O, int *length, hipStream_t st) {
  const size_t lds = (size_t)K.nwin * 3 * 2 * 64 * 8;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wg::wg_zmpdisc_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(wg::wg_zmpdisc_kernel, dim3((B + 63) / 64), dim3(64), lds, st, K, B, smax, steps, n_steps, init_feet,
                     lcap, O, length);
  HIP_TRY(hipGetLastError());
  return WG_OK;
}
}  // namespace

extern "C" {

void wg_zmpdisc_defaults(wg_zmpdisc_model_t *m) {
  if (!m) return;
  memset(m, 0, sizeof *m);
  m->T = 0.005;                    // ZMPRefTrajectoryGeneration's members as PatternGeneratorInterfacePrivate.cpp sets them
  m->preview_time = 1.6;
  m->t_single = 0.78;
  m->t_double = 0.02;
  m->step_height = 0.07;
  m->omega = 0.0;
  m->modulation = 0.9;             // ZMPDiscretization.cpp:99
}

int wg_zmpdisc_length(const wg_zmpdisc_model_t *model, const wg_rel_step_t *steps, int n_steps) {
  if (!model || !steps) return WG_ZMPDISC_BAD_INPUT;
  return wg::zd_length(*model, steps, n_steps);
}

int wg_zmpdisc_batch_dev_ctx(wg_ctx_t *ctx, const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps, const int *n_steps, const double *init_feet, int lcap, double *zmp_x_tm, double *zmp_y_tm, int *length, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  wg::ZdConst K;
  if (int rc = zd_make_const(model, &K)) return rc;
  if (B < 0 || smax < 2 || smax > WG_ZMPDISC_MAX_STEPS || lcap < 1 || !steps || !n_steps || !init_feet || !zmp_x_tm || !zmp_y_tm)
    return fail(WG_ERR_BAD_ARG, "need B >= 0, 2 <= smax <= %d, lcap >= 1, non-null arrays", WG_ZMPDISC_MAX_STEPS);
  if (B == 0) return WG_OK;
  wg::ZdOut O;
  memset(&O, 0, sizeof O);
  O.zx = zmp_x_tm;
  O.zy = zmp_y_tm;
  return zd_launch(K, B, smax, steps, n_steps, init_feet, lcap, O, length, reinterpret_cast<hipStream_t>(hip_stream));
}

int wg_zmpdisc_full_batch_dev_ctx(wg_ctx_t *ctx, const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps, const int *n_steps, const double *init_feet, int lcap, double *zmp_x_tm, double *zmp_y_tm, double *zmp_theta_tm, int *zmp_type_tm, double *left_tm, int *left_type_tm, double *right_tm, int *right_type_tm, int *length, void *hip_stream) {
  if (int rc = use_ctx(ctx)) return rc;
  wg::ZdConst K;
  if (int rc = zd_make_const(model, &K)) return rc;
  if (B < 0 || smax < 2 || smax > WG_ZMPDISC_MAX_STEPS || lcap < 1 || !steps || !n_steps || !init_feet)
    return fail(WG_ERR_BAD_ARG, "need B >= 0, 2 <= smax <= %d, lcap >= 1, non-null inputs", WG_ZMPDISC_MAX_STEPS);
  if ((zmp_x_tm == nullptr) != (zmp_y_tm == nullptr)) return fail(WG_ERR_BAD_ARG, "zmp_x_tm and zmp_y_tm go together");
  if (B == 0) return WG_OK;
  wg::ZdOut O;
  O.zx = zmp_x_tm; O.zy = zmp_y_tm; O.ztheta = zmp_theta_tm; O.ztype = zmp_type_tm;
  O.left = left_tm; O.ltype = left_type_tm; O.right = right_tm; O.rtype = right_type_tm;
  return zd_launch(K, B, smax, steps, n_steps, init_feet, lcap, O, length, reinterpret_cast<hipStream_t>(hip_stream));
}

int wg_zmpdisc_batch_ctx(wg_ctx_t *ctx, const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps, const int *n_steps, const double *init_feet, int lcap, double *zmp, double *zmp_theta, int *zmp_type, double *left, int *left_type, double *right, int *right_type, int *length) {
  if (int rc = use_ctx(ctx)) return rc;
  wg::ZdConst K;
  if (int rc = zd_make_const(model, &K)) return rc;
  if (B < 0 || smax < 2 || smax > WG_ZMPDISC_MAX_STEPS || lcap < 1 || !steps || !n_steps || !init_feet || !length)
    return fail(WG_ERR_BAD_ARG, "need B >= 0, 2 <= smax <= %d, lcap >= 1, non-null arrays", WG_ZMPDISC_MAX_STEPS);
  if (B == 0) return WG_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t sB = (size_t)B, sL = (size_t)lcap, row = sB * sL;
  // arena (8-byte units): steps | init_feet | zx zy ztheta | left right (6 rows each) | ints: n_steps length ztype ltype rtype
  const size_t n_step_d = (sB * smax * sizeof(wg_rel_step_t) + 7) / 8;
  const size_t nd = n_step_d + sB * 6 + 3 * row + 12 * row, ni = 2 * sB + 3 * row;
  if (int rc = ctx->zd_buf.reserve(nd * 8 + ni * 4 + 64)) return rc;
  double *d0 = static_cast<double *>(ctx->zd_buf.p);
  wg_rel_step_t *d_steps = reinterpret_cast<wg_rel_step_t *>(d0);
  double *d_feet = d0 + n_step_d, *d_zx = d_feet + sB * 6, *d_zy = d_zx + row, *d_zt = d_zy + row, *d_l = d_zt + row,
         *d_r = d_l + 6 * row;
  int *d_ns = reinterpret_cast<int *>(d_r + 6 * row), *d_len = d_ns + sB, *d_zty = d_len + sB, *d_lty = d_zty + row,
      *d_rty = d_lty + row;
  WG_H2D(d_steps, steps, sB * smax * sizeof(wg_rel_step_t));
  WG_H2D(d_feet, init_feet, sB * 6 * 8);
  WG_H2D(d_ns, n_steps, sB * 4);
  wg::ZdOut O;
  memset(&O, 0, sizeof O);
  if (zmp) { O.zx = d_zx; O.zy = d_zy; }
  if (zmp_theta) O.ztheta = d_zt;
  if (zmp_type) O.ztype = d_zty;
  if (left) O.left = d_l;
  if (left_type) O.ltype = d_lty;
  if (right) O.right = d_r;
  if (right_type) O.rtype = d_rty;
  if (int rc = zd_launch(K, B, smax, d_steps, d_ns, d_feet, lcap, O, d_len, ctx->host_stream)) return rc;
  WG_D2H(length, d_len, sB * 4);
  WG_HOST_WAIT();
  // time-major device arrays -> the caller's gait-major arrays, samples below each gait's length only
  std::vector<double> hd;
  std::vector<int> hi;
  auto fetch_d = [&](const double *dev, int comps, double *dst, int dst_stride, int dst_off) -> int {
    hd.resize(row * comps);
    WG_D2H(hd.data(), dev, row * comps * 8);
    WG_HOST_WAIT();
    for (size_t b = 0; b < sB; b++)
      for (int l = 0; l < length[b]; l++)
        for (int c = 0; c < comps; c++)
          dst[(b * sL + l) * dst_stride + dst_off + c] = hd[((size_t)l * comps + c) * sB + b];
    return WG_OK;
  };
  auto fetch_i = [&](const int *dev, int *dst) -> int {
    hi.resize(row);
    WG_D2H(hi.data(), dev, row * 4);
    WG_HOST_WAIT();
    for (size_t b = 0; b < sB; b++)
      for (int l = 0; l < length[b]; l++) dst[b * sL + l] = hi[(size_t)l * sB + b];
    return WG_OK;
  };
  if (zmp) {
    if (int rc = fetch_d(d_zx, 1, zmp, 2, 0)) return rc;
    if (int rc = fetch_d(d_zy, 1, zmp, 2, 1)) return rc;
  }
  if (zmp_theta) if (int rc = fetch_d(d_zt, 1, zmp_theta, 1, 0)) return rc;
  if (left) if (int rc = fetch_d(d_l, 6, left, 6, 0)) return rc;
  if (right) if (int rc = fetch_d(d_r, 6, right, 6, 0)) return rc;
  if (zmp_type) if (int rc = fetch_i(d_zty, zmp_type)) return rc;
  if (left_type) if (int rc = fetch_i(d_lty, left_type)) return rc;
  if (right_type) if (int rc = fetch_i(d_rty, right_type)) return rc;
  return WG_OK;
}

}  // extern "C"

// ---- the same entry points on the process-wide default context -----------------------------------------------------
extern "C" {

int wg_qp_solve_batch_dev(int B, int nmax, int mmax, const int *n, const int *m, const int *me, const double *C, const double *d, const double *A, const double *b, const double *xl, const double *xu, double eps, double *x, double *u, int *ifail, int *n_iter, int *iact, int *nact, int *hist, int hist_cap, int *hist_len, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_qp_solve_batch_dev_ctx(c, B, nmax, mmax, n, m, me, C, d, A, b, xl, xu, eps, x, u, ifail, n_iter, iact, nact, hist, hist_cap, hist_len, hip_stream);
}

int wg_qp_solve_batch(int B, int nmax, int mmax, const int *n, const int *m, const int *me, const double *C, const double *d, const double *A, const double *b, const double *xl, const double *xu, double eps, double *x, double *u, int *ifail, int *n_iter, int *iact, int *nact, int *hist, int hist_cap, int *hist_len) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_qp_solve_batch_ctx(c, B, nmax, mmax, n, m, me, C, d, A, b, xl, xu, eps, x, u, ifail, n_iter, iact, nact, hist, hist_cap, hist_len);
}

int wg_set_overlap_strict(int on) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_set_overlap_strict_ctx(c, on);
}

long long wg_overlap_serialised(void) {
  wg_ctx *c = nullptr;
  if (default_ctx(&c)) return -1;
  return wg_overlap_serialised_ctx(c);
}

int wg_mpc_configure(const wg_model_t *model) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_configure_ctx(c, model);
}

int wg_mpc_reserve(int max_gaits) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_reserve_ctx(c, max_gaits);
}

size_t wg_mpc_tick_lds_bytes(void) {                 // a query: does not create the default context
  std::lock_guard<std::mutex> lk(g_default_mu);
  return wg_mpc_tick_lds_bytes_ctx(g_default);
}

int wg_mpc_tick_batch_dev(int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag, int advance_calls, int *hist, int hist_cap, int *hist_len, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_tick_batch_dev_ctx(c, B, states, outs, diag, advance_calls, hist, hist_cap, hist_len, hip_stream);
}

int wg_mpc_run_sched_dev(int B, wg_gait_state_t *states, int n_ticks, int advance_calls, const double *vref_sched, int period, wg_tick_out_t *outs, int *diag, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_run_sched_dev_ctx(c, B, states, n_ticks, advance_calls, vref_sched, period, outs, diag, hip_stream);
}

int wg_mpc_run_batch_dev(int B, wg_gait_state_t *states, int n_ticks, int advance_calls, wg_tick_out_t *outs, int *diag, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_run_batch_dev_ctx(c, B, states, n_ticks, advance_calls, outs, diag, hip_stream);
}

int wg_mpc_tick_batch(int B, wg_gait_state_t *states, wg_tick_out_t *outs, int *diag, int advance_calls, int *hist, int hist_cap, int *hist_len) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_tick_batch_ctx(c, B, states, outs, diag, advance_calls, hist, hist_cap, hist_len);
}

int wg_mpc_set_velref_dev(int B, wg_gait_state_t *states, const double *vref, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_set_velref_dev_ctx(c, B, states, vref, hip_stream);
}

int wg_pldp_configure(int N, const double *iPu, const double *Px, const double *Pu) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_pldp_configure_ctx(c, N, iPu, Px, Pu);
}

int wg_pldp_solve_batch_dev(int B, int mcap, const int *m, const double *D, const double *A, const double *b, const double *zmpref, const double *xkyk, const int *similar, const int *n_removed, const int *starting, int max_iter, wg_pldp_state_t *states, double *X, int *ret, int *n_iter, int *active, int *n_active, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_pldp_solve_batch_dev_ctx(c, B, mcap, m, D, A, b, zmpref, xkyk, similar, n_removed, starting, max_iter, states, X, ret, n_iter, active, n_active, hip_stream);
}

int wg_pldp_solve_batch(int B, int mcap, const int *m, const double *D, const double *A, const double *b, const double *zmpref, const double *xkyk, const int *similar, const int *n_removed, const int *starting, int max_iter, wg_pldp_state_t *states, double *X, int *ret, int *n_iter, int *active, int *n_active) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_pldp_solve_batch_ctx(c, B, mcap, m, D, A, b, zmpref, xkyk, similar, n_removed, starting, max_iter, states, X, ret, n_iter, active, n_active);
}

int wg_dimitrov_configure(const wg_dimitrov_model_t *model) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_dimitrov_configure_ctx(c, model);
}

int wg_dimitrov_get_constants(double *iLQ, double *OptB, double *OptC, double *Pu, double *iPu, double *Px) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_dimitrov_get_constants_ctx(c, iLQ, OptB, OptC, Pu, iPu, Px);
}

int wg_dimitrov_get_qld_constants(double *Q, double *OptB, double *OptC, double *PuT) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_dimitrov_get_qld_constants_ctx(c, Q, OptB, OptC, PuT);
}

int wg_dimitrov_tick_batch_dev(int B, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *states, wg_dimitrov_out_t *outs, int max_iter, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_dimitrov_tick_batch_dev_ctx(c, B, polys, states, outs, max_iter, hip_stream);
}

int wg_dimitrov_tick_batch(int B, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *states, wg_dimitrov_out_t *outs, int max_iter) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_dimitrov_tick_batch_ctx(c, B, polys, states, outs, max_iter);
}

int wg_mpc_tick_pinned(wg_gait_state_t *state, wg_tick_out_t *out, int *diag, int advance_calls) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_tick_pinned_ctx(c, state, out, diag, advance_calls);
}

int wg_mpc_assemble_batch_dev(int B, const wg_gait_state_t *states, int advance_calls, int nmax, int mmax, double *C, double *d, double *A, double *b, double *xl, double *xu, int *n, int *m, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_assemble_batch_dev_ctx(c, B, states, advance_calls, nmax, mmax, C, d, A, b, xl, xu, n, m, hip_stream);
}

int wg_mpc_assemble_batch(int B, const wg_gait_state_t *states, int advance_calls, int nmax, int mmax, double *C, double *d, double *A, double *b, double *xl, double *xu, int *n, int *m) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_mpc_assemble_batch_ctx(c, B, states, advance_calls, nmax, mmax, C, d, A, b, xl, xu, n, m);
}

int wg_preview_configure(const wg_preview_gains_t *gains, const double *F) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_preview_configure_ctx(c, gains, F);
}

int wg_preview_window(void) {                        // a query: does not create the default context
  std::lock_guard<std::mutex> lk(g_default_mu);
  return wg_preview_window_ctx(g_default);
}

int wg_preview_run_batch_dev(int B, int L, const double *zmp_x_tm, const double *zmp_y_tm, double *state, double *com_tm, double *zmp2_tm, int simulation, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_preview_run_batch_dev_ctx(c, B, L, zmp_x_tm, zmp_y_tm, state, com_tm, zmp2_tm, simulation, hip_stream);
}

int wg_preview_run_batch(int B, int L, const double *zmp_x, const double *zmp_y, double *state, double *com, double *zmp2, int simulation) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_preview_run_batch_ctx(c, B, L, zmp_x, zmp_y, state, com, zmp2, simulation);
}

int wg_gramian_batch_dev(int B, int N, const double *T, const double *h, double alpha, double beta, double gamma, int precision, double *Qb, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_gramian_batch_dev_ctx(c, B, N, T, h, alpha, beta, gamma, precision, Qb, hip_stream);
}

int wg_gramian_batch(int B, int N, const double *T, const double *h, double alpha, double beta, double gamma, int precision, double *Qb) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_gramian_batch_ctx(c, B, N, T, h, alpha, beta, gamma, precision, Qb);
}

int wg_zmpdisc_batch_dev(const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps, const int *n_steps, const double *init_feet, int lcap, double *zmp_x_tm, double *zmp_y_tm, int *length, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_zmpdisc_batch_dev_ctx(c, model, B, smax, steps, n_steps, init_feet, lcap, zmp_x_tm, zmp_y_tm, length, hip_stream);
}

int wg_zmpdisc_full_batch_dev(const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps, const int *n_steps, const double *init_feet, int lcap, double *zmp_x_tm, double *zmp_y_tm, double *zmp_theta_tm, int *zmp_type_tm, double *left_tm, int *left_type_tm, double *right_tm, int *right_type_tm, int *length, void *hip_stream) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_zmpdisc_full_batch_dev_ctx(c, model, B, smax, steps, n_steps, init_feet, lcap, zmp_x_tm, zmp_y_tm, zmp_theta_tm, zmp_type_tm, left_tm, left_type_tm, right_tm, right_type_tm, length, hip_stream);
}

int wg_zmpdisc_batch(const wg_zmpdisc_model_t *model, int B, int smax, const wg_rel_step_t *steps, const int *n_steps, const double *init_feet, int lcap, double *zmp, double *zmp_theta, int *zmp_type, double *left, int *left_type, double *right, int *right_type, int *length) {
  wg_ctx *c = nullptr;
  if (int rc = default_ctx(&c)) return rc;
  return wg_zmpdisc_batch_ctx(c, model, B, smax, steps, n_steps, init_feet, lcap, zmp, zmp_theta, zmp_type, left, left_type, right, right_type, length);
}

}  // extern "C"
