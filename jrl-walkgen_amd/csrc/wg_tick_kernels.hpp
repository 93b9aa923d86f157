// wg_tick_kernels.hpp -- the __global__ kernels of the Herdt-2010 tick (one launch per tick, many ticks per launch through a
// device-wide queue, many ticks per launch with the hand-over kept inside one XCD) and their small helpers.  Included by
// wg_capi.hip, which launches them; a translation unit of its own can instantiate ONE of them (tools/one_kernel.sh: resource
// usage and ISA of a single kernel in seconds instead of the whole library).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/wg_mpc.h"
#include "wg_ql_device.hpp"
#include "wg_tick_device.hpp"

#ifndef WG_TICK32_WPE
#define WG_TICK32_WPE 3                                    // element view (N = 32): 168 registers, three gaits on a SIMD
#endif
#ifndef WG_TICK_WPE_MIN
#define WG_TICK_WPE_MIN 2
#endif
#ifndef WG_TICK_WPE_MAX
#define WG_TICK_WPE_MAX 2
#endif

// Waves per SIMD the tick kernel is compiled for: 2 => at most 256 registers per lane.  With 26.2 KB of LDS per gait six
// gaits fit a CU (SIMDs hold 2,2,1,1 waves); the second wave of a SIMD hides the first one's dependent fp64 chains
// (measured: 4 -> 6 resident gaits per CU = 1.78 -> 2.23 M ticks/s).  -DWG_TICK_WPE_MIN=1 -DWG_TICK_WPE_MAX=1 gives the
// 512-register build (lib/libwg_mpc_w1.so, tools/bench_variants.sh).
// The element view (NH == -1, N = 32) keeps 26.8 KB of LDS per gait: six gaits per CU, two SIMDs of a CU hold two of them, so
// its kernels are compiled for 256 registers as well (WG_TICK32_WPE = 2; = 1 gives the 512-register build, four per CU).
// The run kernels pin their pointer arguments in scalar registers with an opaque asm once per tick (so that nothing derived
// from them stays alive across the tick).  Pinned as GENERIC pointers they would come back with no address space and every
// access through them would be a flat_ instruction: those count on both vmcnt and lgkmcnt and may return out of order with
// LDS reads, so each LDS wait behind one turns into lgkmcnt(0) -- a global round trip.  Pinned as address-space-1 pointers
// they stay global_ accesses (kernel arguments are global memory).
constexpr bool wg_tick_is_elem(int NH) { return NH == -1 || NH == 32; }
// view 33 (N = 32 with Z in registers): one wave per SIMD, 512 registers
#define WG_TICK_WAVES(NH) __attribute__((amdgpu_waves_per_eu((NH) == 33 ? WG_ZR_WPS : wg_tick_is_elem(NH) ? WG_TICK32_WPE : WG_TICK_WPE_MIN, (NH) == 33 ? WG_ZR_WPS : wg_tick_is_elem(NH) ? WG_TICK32_WPE : WG_TICK_WPE_MAX)))
// View 33 (N = 32 with Z in registers, docs/HISTORY.md 3.2 "Z on chip": measured, slower than view 32) is an EXPERIMENT: its three kernels are
// only compiled into builds made with -DWG_WITH_REGZ (make lib/libwg_mpc_xregz.so EXTRA=-DWG_WITH_REGZ; tools/regz_probe.sh)
#ifdef WG_WITH_REGZ
#define WG_REGZ_KERNEL(KERN, view) (view) == 33 ? reinterpret_cast<const void *>(KERN<33>) :
#define WG_REGZ_CASE(KERN, grid, lds, st, ...) case 33: hipLaunchKernelGGL(KERN<33>, dim3(grid), dim3(64), lds, st, __VA_ARGS__); break;
#else
#define WG_REGZ_KERNEL(KERN, view)
#define WG_REGZ_CASE(KERN, grid, lds, st, ...)
#endif
// one kernel instantiation per problem view of the tick (16 compact, 0 dense, -1 element, 32 element with N = 32 fixed)
#define WG_KERNEL_BY_VIEW(KERN, view)                                                                                              \
  ((view) == 16 ? reinterpret_cast<const void *>(KERN<16>) : (view) == 0 ? reinterpret_cast<const void *>(KERN<0>)                 \
   : (view) == 32 ? reinterpret_cast<const void *>(KERN<32>) : WG_REGZ_KERNEL(KERN, view) reinterpret_cast<const void *>(KERN<-1>))
#define WG_LAUNCH_BY_VIEW(KERN, view, grid, lds, st, ...)                                                                           \
  do {                                                                                                                             \
    switch (view) {                                                                                                                \
      case 16: hipLaunchKernelGGL(KERN<16>, dim3(grid), dim3(64), lds, st, __VA_ARGS__); break;                                     \
      case 0: hipLaunchKernelGGL(KERN<0>, dim3(grid), dim3(64), lds, st, __VA_ARGS__); break;                                       \
      case 32: hipLaunchKernelGGL(KERN<32>, dim3(grid), dim3(64), lds, st, __VA_ARGS__); break;                                     \
      WG_REGZ_CASE(KERN, grid, lds, st, __VA_ARGS__)                                                                                \
      default: hipLaunchKernelGGL(KERN<-1>, dim3(grid), dim3(64), lds, st, __VA_ARGS__); break;                                     \
    }                                                                                                                              \
  } while (0)
#define WG_PIN_GLOBAL(p)                                                                          \
  do {                                                                                            \
    auto gp_ = (__attribute__((address_space(1))) std::remove_pointer_t<decltype(p)> *)(p);       \
    asm volatile("" : "+s"(gp_));                                                                 \
    (p) = (decltype(p))gp_;                                                                       \
  } while (0)

template <int NH>
__global__ __launch_bounds__(64) WG_TICK_WAVES(NH) void wg_mpc_tick_kernel(int B, wg_model_t model, const wg::TickTables *__restrict__ tb,
                                                         wg_gait_state_t *__restrict__ states,
                                                         wg_tick_out_t *__restrict__ outs, int *__restrict__ diag,
                                                         int advance_calls, int *__restrict__ hist, int hist_cap,
                                                         int *__restrict__ hist_len, unsigned ql_bytes, double *zscratch,
                                                         unsigned zslot, int elem_cap,
                                                         wg_gait_state_t *__restrict__ host_states, int *host_done,
                                                         const int *__restrict__ order, int *__restrict__ iters_out) {
  extern __shared__ __attribute__((aligned(16))) double wg_lds[];
  const int lane = threadIdx.x & 63;
  // one block = one gait (grid == B): no grid-stride loop, so nothing lane-dependent is hoisted out of it and kept
  // alive (in registers) across the whole tick.  Blocks start in index order: `order` (wg_lpt_order_kernel) makes that the
  // order of decreasing solve length, as far as the previous tick predicts it
  const int g = order ? wg::uni(order[blockIdx.x]) : (int)blockIdx.x;
  if (g < B) {
    if (host_states) {
      // one-robot path (wg_mpc_tick_pinned): the caller's state lives in host-mapped memory; the tick works on a device copy
      const double *src = reinterpret_cast<const double *>(host_states + g);
      double *dst = reinterpret_cast<double *>(states + g);
      for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      WG_WSYNC();
    }
    if (advance_calls > 0) {
      if (lane == 0) {
        double c = states[g].clock;
        for (int k = 0; k < advance_calls; ++k) c += model.Tctrl;   // PatternGeneratorInterfacePrivate.cpp:1256
        states[g].clock = c;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      WG_WSYNC();
    }
    wg::TickDiag dg = wg::mpc_tick<NH>(model, tb, states + g, outs ? outs + g : nullptr, wg_lds,
                                   reinterpret_cast<char *>(wg_lds) + ql_bytes, hist ? hist + (size_t)g * hist_cap : nullptr,
                                   hist_cap, hist_len ? hist_len + g : nullptr,
                                   zscratch ? zscratch + (size_t)blockIdx.x * zslot : nullptr, elem_cap);
    if (diag && lane == 0) {
      int *dq = diag + (size_t)g * 6;
      dq[0] = dg.ifail; dq[1] = dg.n_iter; dq[2] = dg.nact; dq[3] = dg.n; dq[4] = dg.m; dq[5] = dg.ns;
    }
    if (iters_out && lane == 0) iters_out[g] = dg.n_iter;
    WG_WSYNC();
    if (host_states) {
      // state back to the caller's memory; outs / diag were written there directly.  Every store of this wave is performed
      // (system-scope release) before the completion counter moves: the host spins on it instead of synchronising.
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double *src = reinterpret_cast<const double *>(states + g);
      double *dst = reinterpret_cast<double *>(host_states + g);
      for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64) dst[i] = src[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      WG_WSYNC();
      if (lane == 0) __hip_atomic_fetch_add(host_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Start order of a one-launch-per-tick batch: gaits by decreasing iteration count of their previous tick (a counting sort by one
// block; the order inside a count is whatever the atomics make it -- scheduling only, no result depends on it).  A batch of
// several rounds of resident waves ends when its last gait ends: started in index order the long solves (15 - 37 iterations, and a
// gait's difficulty persists from tick to tick) land anywhere, the last round included; started longest-first the tail is made of
// the short ones.  Replayed on measured solve times (B = 4096, 2048 resident waves): 884 us per tick in index order, 795 us
// with this order, 738 us with perfect foresight, 709 us the work bound.
constexpr int kLptBins = 128;
__global__ __launch_bounds__(1024) void wg_lpt_order_kernel(int B, const int *__restrict__ iters, int *__restrict__ order) {
  __shared__ int cnt[kLptBins], base[kLptBins];
  const int tid = threadIdx.x;
  if (tid < kLptBins) cnt[tid] = 0;
  __syncthreads();
  for (int g = tid; g < B; g += 1024) { int k = iters[g]; k = k < 0 ? 0 : (k >= kLptBins ? kLptBins - 1 : k); atomicAdd(&cnt[k], 1); }
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int k = kLptBins - 1; k >= 0; --k) { base[k] = acc; acc += cnt[k]; } }
  __syncthreads();
  for (int g = tid; g < B; g += 1024) { int k = iters[g]; k = k < 0 ? 0 : (k >= kLptBins ? kLptBins - 1 : k); order[atomicAdd(&base[k], 1)] = g; }
}

// The assembled QP of every gait's NEXT tick, without advancing anything: the dense view of the tick run on a scratch copy
// of the state up to the point where QPProblem::solve would call ql0001_ (qp-problem.cpp:245-279), the arrays written in
// ql0001_'s layout instead (QPProblem::dump_problem, qp-problem.cpp:639-653: what the reference writes to
// /tmp/Problem_<time>.dat when a solve fails, ZMPVelocityReferencedQP.cpp:399-402).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WG_TICK_WPE_MIN, WG_TICK_WPE_MAX))) void wg_mpc_assemble_kernel(
    int B, wg_model_t model, const wg::TickTables *__restrict__ tb, const wg_gait_state_t *__restrict__ states,
    wg_gait_state_t *__restrict__ scratch, int advance_calls, unsigned ql_bytes, int nmax, int mmax, double *__restrict__ C,
    double *__restrict__ d, double *__restrict__ A, double *__restrict__ b, double *__restrict__ xl, double *__restrict__ xu,
    int *__restrict__ n_out, int *__restrict__ m_out) {
  extern __shared__ __attribute__((aligned(16))) double wg_lds[];
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x;
  if (g < B) {
    {
      const double *src = reinterpret_cast<const double *>(states + g);
      double *dst = reinterpret_cast<double *>(scratch + g);
      for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64) dst[i] = src[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      WG_WSYNC();
    }
    if (advance_calls > 0) {
      if (lane == 0) {
        double c = scratch[g].clock;
        for (int k = 0; k < advance_calls; ++k) c += model.Tctrl;   // PatternGeneratorInterfacePrivate.cpp:1256
        scratch[g].clock = c;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      WG_WSYNC();
    }
    wg::QpDumpOut o;
    o.C = C + (size_t)g * nmax * nmax; o.d = d + (size_t)g * nmax; o.A = A + (size_t)g * mmax * nmax; o.b = b + (size_t)g * mmax;
    o.xl = xl + (size_t)g * nmax; o.xu = xu + (size_t)g * nmax; o.n = n_out + g; o.m = m_out + g; o.nmax = nmax; o.mmax = mmax;
    (void)wg::mpc_tick<0>(model, tb, scratch + g, nullptr, wg_lds, reinterpret_cast<char *>(wg_lds) + ql_bytes, nullptr, 0, nullptr,
                          nullptr, 0, &o);
  }
}

// ---- many ticks per launch: a work queue of (gait, next tick) ---------------------------------------------------------------
// A gait's tick t+1 depends only on its own tick t, so the batch need not synchronise between ticks.  One launch of
// wg_mpc_run_kernel advances every gait by n_ticks: resident waves pull gait ids from a ring in arrival order; a wave that
// finishes a tick appends its gait again (until the gait has done n_ticks).  All wave slots stay busy until the very end
// of the launch, instead of draining at the end of every tick (B = 4096 is 2.3 rounds of the 1792 resident gaits: with
// one launch per tick a quarter of the machine idles in the last round).  Results are those of n_ticks single-tick
// launches, bit for bit: a tick only ever reads its own gait's state.
struct wg_run_queue {
  int head, tail;      // next ring position to take / to fill
  int pad_[2];
};

__global__ void wg_run_queue_init_kernel(int B, int total, wg_run_queue *q, int *ring, int *done) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { q->head = 0; q->tail = B; }
  if (i < total) ring[i] = i < B ? i : -1;
  if (i < B) done[i] = 0;
}

template <int NH>
__global__ __launch_bounds__(64) WG_TICK_WAVES(NH) void wg_mpc_run_kernel(
    int B, int n_ticks, const wg_model_t *__restrict__ model_p, const wg::TickTables *__restrict__ tb_p,
    wg_gait_state_t *__restrict__ states_p, wg_tick_out_t *__restrict__ outs_p, int *__restrict__ diag_p, int advance_calls,
    wg_run_queue *__restrict__ q, int *__restrict__ ring, int *__restrict__ done, unsigned ql_bytes, double *zscratch,
    unsigned zslot, int elem_cap) {
  extern __shared__ __attribute__((aligned(16))) double wg_lds[];
  const int total = B * n_ticks;
  for (;;) {
    // Everything the tick reads through is made opaque per item, so that nothing loop-invariant (lane-derived values,
    // model constants, table addresses) is hoisted out of the loop and kept alive across whole ticks: the register
    // budget is the single tick's.
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const wg_model_t *mp = model_p; const wg::TickTables *tb = tb_p; wg_gait_state_t *states = states_p;
    wg_tick_out_t *outs = outs_p; int *diag = diag_p;
    WG_PIN_GLOBAL(mp); WG_PIN_GLOBAL(tb); WG_PIN_GLOBAL(states); WG_PIN_GLOBAL(outs); WG_PIN_GLOBAL(diag);
    const wg_model_t &model = *mp;
    int idx = 0, g = 0, t = 0;
    if (lane == 0) idx = atomicAdd(&q->head, 1);
    idx = wg::uni(idx);
    if (idx >= total) break;                     // every wave reaches this: head only grows
    if (lane == 0) {
      // positions are filled in order; every position below `total` is filled eventually by a wave holding an earlier one
      while ((g = __hip_atomic_load(ring + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < 0) __builtin_amdgcn_s_sleep(16);
      t = __hip_atomic_load(done + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    g = wg::uni(g); t = wg::uni(t);
    // the gait's previous tick may have run on another CU / XCD: its state is read with agent-scope loads (mpc_tick), the
    // ring entry was read with acquire -- no further invalidate here
    if (advance_calls > 0) {
      if (lane == 0) {
        double c = __hip_atomic_load(&states[g].clock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < advance_calls; ++k) c += model.Tctrl;   // PatternGeneratorInterfacePrivate.cpp:1256
        __hip_atomic_store(&states[g].clock, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      WG_WSYNC();
    }
    wg::TickDiag dg = wg::mpc_tick<NH>(model, tb, states + g, outs ? outs + (size_t)t * B + g : nullptr, wg_lds,
                                   reinterpret_cast<char *>(wg_lds) + ql_bytes, nullptr, 0, nullptr,
                                   zscratch ? zscratch + (size_t)blockIdx.x * zslot : nullptr, elem_cap);
    if (diag && lane == 0) {
      int *dq = diag + ((size_t)t * B + g) * 6;
      dq[0] = dg.ifail; dq[1] = dg.n_iter; dq[2] = dg.nact; dq[3] = dg.n; dq[4] = dg.m; dq[5] = dg.ns;
    }
    WG_WSYNC();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // the state is in L2 before the gait is offered again
    if (lane == 0) {
      __hip_atomic_store(done + g, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1 < n_ticks) {
        const int pos = atomicAdd(&q->tail, 1);
        __hip_atomic_store(ring + pos, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ---- the same, with the hand-over kept inside one XCD ------------------------------------------------------------------------
// The queue above lets a gait's next tick run anywhere, so every tick ends with an agent-scope release: on gfx950 that is a
// write-back of ALL dirty lines of the XCD's L2 (buffer_wbl2) -- the parked state copies and solver slots of the other 255
// resident gaits included -- and every tick starts with an invalidate of the CU's L1.  Here each XCD has its own ring: a gait
// is adopted by the XCD that runs its first tick of the launch and stays there, so its state goes from one CU to the next
// through that XCD's L2 alone (stores are written through L1, loads of state / ring / counters bypass L1): no L2 write-back,
// no L1 invalidate, the constant tables stay cached.
//   * gaits are dealt to the XCDs in eight contiguous ranges; a wave first adopts the fresh gaits of its own XCD's range,
//     then serves its XCD's ring, and only when that is empty adopts fresh gaits of other ranges (untouched in this launch,
//     so visible everywhere: an XCD that received no wave leaves no gait behind);
//   * a wave that finds nothing to do exits: a gait in flight is always held by a live wave, which offers it to its own
//     ring and takes it back if nobody else does -- every gait reaches n_ticks whatever the placement of the waves;
//   * ring entries carry their position as a tag, so slots are reused without being cleared;
//   * a wave KEEPS its gait for the next tick when the gait is behind its XCD's mean progress (prog counts the XCD's finished
//     ticks; behind: (t + 1 + keep_k) * gaits of the XCD <= prog).  A ring alone serves first-in first-out, so a gait's period
//     is the ring's revolution PLUS its own solve: gaits with long solves fall behind (ticks per gait after a 200-tick launch
//     spread by tens), the ring drains with them still far from done, and the waves leave early -- 4.3 - 4.9 % of the wave-time
//     of a B = 4096 launch was that tail (tools/xrun_stats.py on an experiment build; tools/xrun_sim.py replays the policies on
//     the measured solve times: FIFO +4.2 % over the work bound, keep +0.2 %).  Keeping also saves the laggards' hand-overs.
//     WG_RUN_KEEP=k sets keep_k (default 0), WG_RUN_KEEP=off restores the plain ring.
// Counters and ring entries are only touched by read-modify-write atomics (one coherence point whatever the hardware does
// with them); the XCD a wave runs on is read from the hardware register, not inferred from blockIdx.
constexpr int kXcds = 8;
struct wg_xrun_ctl {
  struct alignas(64) { int fresh, fresh_end, head, tail, prog, count; } x[kXcds];
};

__global__ void wg_xrun_init_kernel(int B, wg_xrun_ctl *ctl, unsigned long long *rings, int cap, int *done) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kXcds) {
    ctl->x[i].fresh = (int)((long long)B * i / kXcds);
    ctl->x[i].fresh_end = (int)((long long)B * (i + 1) / kXcds);
    ctl->x[i].head = 0; ctl->x[i].tail = 0;
    ctl->x[i].prog = 0; ctl->x[i].count = ctl->x[i].fresh_end - ctl->x[i].fresh;
  }
  if (i < kXcds * cap) rings[i] = 0ull;
  if (i < B) done[i] = 0;
}

// reads of the counters and ring entries are read-modify-writes with an operand the compiler cannot see through (a literal 0
// is folded into an atomic LOAD, which the vector L1 may serve: a stale tail reads as "ring empty" and the wave leaves early --
// measured: 59 % average wave residency and a third of the throughput once these became global_ instead of flat_ loads)
__device__ __forceinline__ int xrun_opaque_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }
__device__ __forceinline__ int xrun_rmw_load(int *p) { return __hip_atomic_fetch_add(p, xrun_opaque_zero(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long xrun_rmw_load(unsigned long long *p) {
  return __hip_atomic_fetch_or(p, (unsigned long long)(unsigned)xrun_opaque_zero(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int xrun_take_fresh(wg_xrun_ctl *ctl, int y) {
  const int end = ctl->x[y].fresh_end;                     // written before the launch, never changed
  if (xrun_rmw_load(&ctl->x[y].fresh) >= end) return -1;
  const int f = __hip_atomic_fetch_add(&ctl->x[y].fresh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return f < end ? f : -1;
}
// every store of this wave is in L2 (vector L1 is write-through; vmcnt counts a store down when L2 has it)
__device__ __forceinline__ void xrun_stores_done() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

template <int NH>
__global__ __launch_bounds__(64) WG_TICK_WAVES(NH) void wg_mpc_run_xcd_kernel(
    int B, int n_ticks, const wg_model_t *__restrict__ model_p, const wg::TickTables *__restrict__ tb_p,
    wg_gait_state_t *__restrict__ states_p, wg_tick_out_t *__restrict__ outs_p, int *__restrict__ diag_p, int advance_calls,
    wg_xrun_ctl *__restrict__ ctl_p, unsigned long long *__restrict__ rings_p, int cap, int *__restrict__ done_p,
    unsigned ql_bytes, double *zscratch, unsigned zslot, const double *__restrict__ vsched, int vperiod, int elem_cap, int keep_k) {
  extern __shared__ __attribute__((aligned(16))) double wg_lds[];
  unsigned fresh_gone = 0;                                 // bit y: range y was seen exhausted (the counters only grow)
  int kept_g = -1, kept_t = 0;                             // the gait this wave goes on with (it was behind its XCD's mean progress)
  for (;;) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const wg_model_t *mp = model_p; const wg::TickTables *tb = tb_p; wg_gait_state_t *states = states_p;
    wg_tick_out_t *outs = outs_p; int *diag = diag_p; wg_xrun_ctl *ctl = ctl_p; unsigned long long *rings = rings_p; int *done = done_p;
    // data pointers come back as global-memory pointers; the queue's stay generic (measured: 2.5 % faster than global_ atomics)
    WG_PIN_GLOBAL(mp); WG_PIN_GLOBAL(tb); WG_PIN_GLOBAL(states); WG_PIN_GLOBAL(outs); WG_PIN_GLOBAL(diag);
    asm volatile("" : "+s"(ctl), "+s"(rings), "+s"(done));
    const wg_model_t &model = *mp;
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc &= kXcds - 1;
    int g = kept_g, t = kept_t;
    kept_g = -1;
    if (g < 0 && lane == 0) {
      if (!((fresh_gone >> xcc) & 1u)) {
        g = xrun_take_fresh(ctl, xcc);
        if (g < 0) fresh_gone |= 1u << xcc;
      }
      if (g < 0) {
        for (;;) {
          const int h = xrun_rmw_load(&ctl->x[xcc].head), tl = xrun_rmw_load(&ctl->x[xcc].tail);
          if (h >= tl) break;
          int expect = h;
          if (!__hip_atomic_compare_exchange_strong(&ctl->x[xcc].head, &expect, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT))
            continue;
          unsigned long long *slot = rings + (size_t)xcc * cap + (h & (cap - 1));
          unsigned long long e;
          // the pusher reserved position h (tail) before writing the entry: a short wait at most
          while ((unsigned)((e = xrun_rmw_load(slot)) >> 32) != (unsigned)(h + 1))
            __builtin_amdgcn_s_sleep(4);
          g = (int)(e & 0xffffffffull);
          t = xrun_rmw_load(done + g);                       // written (L2) before the entry was
          break;
        }
      }
      for (int y = 1; g < 0 && y < kXcds; ++y) {
        const int z = (xcc + y) & (kXcds - 1);
        if ((fresh_gone >> z) & 1u) continue;
        g = xrun_take_fresh(ctl, z);
        if (g < 0) fresh_gone |= 1u << z;
      }
    }
    g = wg::uni(g); t = wg::uni(t); fresh_gone = (unsigned)wg::uni((int)fresh_gone);
    if (g < 0) break;
    if (vsched && t % vperiod == 0) {                      // staged references: what wg_set_velref_kernel writes between launches
      if (lane < 3) states[g].vref[lane] = vsched[((size_t)(t / vperiod) * B + g) * 3 + lane];
      xrun_stores_done();                                  // mpc_tick reads the state back from L2
      WG_WSYNC();
    }
    if (advance_calls > 0) {
      if (lane == 0) {
        double c = __hip_atomic_load(&states[g].clock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < advance_calls; ++k) c += model.Tctrl;   // PatternGeneratorInterfacePrivate.cpp:1256
        states[g].clock = c;
      }
      xrun_stores_done();                                  // mpc_tick reads the state back from L2
      WG_WSYNC();
    }
    // (the tick body as a real function call instead of inlined code -- no hoisting across ticks, 50 spilled SGPRs in the loop
    // instead of 276 -- measured: N = 16 unchanged (-0.7 %), N = 32 a third slower; the inlined body stays)
    wg::TickDiag dg = wg::mpc_tick<NH>(model, tb, states + g, outs ? outs + (size_t)t * B + g : nullptr, wg_lds,
                                   reinterpret_cast<char *>(wg_lds) + ql_bytes, nullptr, 0, nullptr,
                                   zscratch ? zscratch + (size_t)blockIdx.x * zslot : nullptr, elem_cap);
    if (diag && lane == 0) {
      int *dq = diag + ((size_t)t * B + g) * 6;
      dq[0] = dg.ifail; dq[1] = dg.n_iter; dq[2] = dg.nact; dq[3] = dg.n; dq[4] = dg.m; dq[5] = dg.ns;
#ifdef WG_XRUN_STATS
      // experiment build (tools/xrun_stats.py): when this tick ended (100 MHz counter), where it ran (block, XCD)
      dq[3] = (int)(unsigned)wall_clock64(); dq[4] = (int)blockIdx.x; dq[5] = xcc;
#endif
    }
    WG_WSYNC();
    if (t + 1 < n_ticks) {
      if (keep_k >= 0) {
        int keep = 0;
        if (lane == 0) {
          const int p = __hip_atomic_fetch_add(&ctl->x[xcc].prog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          keep = (long long)(t + 1 + keep_k) * ctl->x[xcc].count <= (long long)p;
        }
        if (wg::uni(keep)) {
          xrun_stores_done();                              // the state is read back from L2 by the next tick
          kept_g = g; kept_t = t + 1;
          continue;
        }
      }
      if (lane == 0) done[g] = t + 1;
      xrun_stores_done();                                  // state and tick count are in this XCD's L2 before the gait is offered
      if (lane == 0) {
        const int pos = __hip_atomic_fetch_add(&ctl->x[xcc].tail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_exchange(rings + (size_t)xcc * cap + (pos & (cap - 1)), ((unsigned long long)(unsigned)(pos + 1) << 32) | (unsigned)g,
                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

__global__ void wg_set_velref_kernel(int B, wg_gait_state_t *states, const double *vref) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < B) {
    states[g].vref[0] = vref[3 * g + 0];
    states[g].vref[1] = vref[3 * g + 1];
    states[g].vref[2] = vref[3 * g + 2];
  }
}

