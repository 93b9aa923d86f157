// wg_ql_device.hpp -- one dense convex QP per wavefront, all factors in LDS.
//
// Device-side replacement for the solver on the reference's Herdt-2010 path:
//   ql0001_ / ql0002_   /root/reference/src/Mathematics/qld.cpp:378-612, 621-2091
// (Powell / Schittkowski dual active-set method; called from
//  QPProblem::solve, src/ZMPRefTrajectoryGeneration/qp-problem.cpp:245-294.)
//
// Design (gfx950 / CDNA4):
//   * one 64-lane wavefront owns one QP; a workgroup is exactly one wave, so
//     no s_barrier is ever needed -- LDS traffic of one wave is processed in
//     order, only the compiler has to be fenced (WG_WSYNC);
//   * Hessian G, Z (= R^-1, later rotated), packed R, the constraint matrix A
//     and all vectors live in the wave's LDS slice; leading dimensions are odd
//     so that both row and column sweeps are bank-conflict free for 8-byte
//     accesses;
//   * lanes parallelise over *independent outputs* only (rows of Z, columns of
//     R, constraint rows of A); every inner sum runs sequentially inside one
//     lane in the reference's order, so every double is bit-identical to the
//     CPU solver and the active-set add/drop sequence is reproduced exactly;
//   * order-insensitive reductions (max, arg-max with first-index tie-break,
//     "any") use wave shuffles;
//   * the long scalar chains (Givens sweep norms, triangular solves) are
//     executed redundantly by all lanes on LDS-broadcast operands.
// Compile with -ffp-contract=off: the reference build has no FMA contraction.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace wg {

#ifndef WG_UNROLL_N
#define WG_UNROLL_N 4
#endif
// Prefetch group size of the wide (64 < n <= 128) forms: entries of Z requested together ahead of the add chains / rotations.
// Eight is what a 256-register kernel carries (the dense boundary kernel); the element view is compiled for 168 registers --
// three waves per SIMD -- and takes groups of four: measured 7 % slower per wave and, with twelve gaits on a CU instead of eight,
// 7 % faster overall.
#ifndef WG_ELEM_ZT_GRP
#define WG_ELEM_ZT_GRP 8                // rows of the Z^T a column walk requested together (its own knob: that walk is bound by the address path)
#endif
#ifndef WG_ELEM_GRP
#define WG_ELEM_GRP 4
#endif
#define WG_PRAGMA(x) _Pragma(#x)
#define WG_UNROLL_(n) WG_PRAGMA(unroll n)
#define WG_UNROLL WG_UNROLL_(WG_UNROLL_N)

#define WG_WSYNC()                                          \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)

// Z of the wide views (n > 64) may live in global memory and is streamed: -DWG_Z_NT=1 marks those accesses non-temporal (an
// experiment knob, tools/probe_elem.py; the default build uses plain accesses)
#if defined(WG_Z_NT) && WG_Z_NT
#define WG_ZLD(p) __builtin_nontemporal_load(p)
#define WG_ZST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define WG_ZLD(p) (*(p))
#define WG_ZST(p, v) (*(p) = (v))
#endif

// lane index, opaque to the optimiser: inside a persistent loop (wg_mpc_run_kernel) nothing derived from it can be hoisted
// out of the loop and kept alive across a whole tick (that hoisting costs ~180 spilled registers)
__device__ __forceinline__ int wg_lane() { int l = threadIdx.x & 63; asm volatile("" : "+v"(l)); return l; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}
// A wave-uniform predicate as a scalar: every lane computes the same value redundantly, so taking lane 0's copy is the
// identity -- but it tells the compiler the branch is uniform (s_cbranch on SCC instead of exec-mask juggling), which
// also keeps everything assigned under it (nact, knext, loop counters, LDS addresses) in scalar registers.
#define WG_UBOOL(c) (uni((int)(c)) != 0)
// value of `v` in lane `src` (src must be wave-uniform): two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double rl(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// sum_{j=lo}^{hi-1} v[lane j] in index order, starting from 0.0.  `v` must be 0.0 in every lane that is not in
// [lo, hi): adding +0.0 never changes a running sum, so the loop can run in chunks of four without a remainder
// loop (v_readlane is convergent and the compiler will not unroll it itself).  Needs hi <= 61.
__device__ __forceinline__ double lane_sum_ordered(double v, int lo, int hi) {
  double sum = 0.0;
  for (int j = lo; j < hi; j += 4) {
    sum += rl(v, j);
    sum += rl(v, j + 1);
    sum += rl(v, j + 2);
    sum += rl(v, j + 3);
  }
  return sum;
}
// A double constant materialised where it is used (two s_mov_b32), opaque to the optimiser: left to itself the compiler hoists
// 64-bit literals (0.1, 0.2, 1e-8, 0.01, 1.5 ...) out of the persistent loop into VGPR pairs at kernel entry, runs out of
// registers, SPILLS them and reloads them from scratch memory inside the active-set loop -- seen in the 256-register tick kernel:
// every scratch_ instruction at loop depth 2 was the reload of such a constant.
__device__ __forceinline__ double wg_kconst(double c) {
  int lo = __double2loint(c), hi = __double2hiint(c);
  asm volatile("" : "+s"(lo), "+s"(hi));
  return __hiloint2double(hi, lo);
}
// f2c.h max/min (qld.cpp:269-270)
__device__ __forceinline__ double maxd(double a, double b) { return a >= b ? a : b; }
__device__ __forceinline__ double mind(double a, double b) { return a <= b ? a : b; }

// LDS footprint of one QP (in doubles, then ints).  Host and device agree on it.
struct QlDims {
  int n, m, mmax;
  int ldg, ldz, lda;
  bool dense;   // G and A held as LDS matrices (false: the problem view regenerates them)
  bool a_lds;   // dense only: A staged in LDS (false: read in place from global memory -- large QPs)
  int nsc;      // length of each of the four scratch vectors: n, or the static length of the compact view's ordered sums
  bool bounds;  // xl / xu held in LDS (false: the problem view supplies them -- constants for the Herdt QP)
  bool z_lds;   // Z held in LDS (false: the caller points QlView::Z at a per-problem slot in global memory -- large n, where
                // Z is the operand that caps the residency; it is streamed lane-parallel, never on a serial chain)
  bool wab_lds; // wa and b held in LDS (false: in a per-block slot of global memory -- they are read lane-parallel once per
                // iteration, early enough for an L2 round trip to hide; freeing them is what lets an eighth gait onto the CU)
  bool cold_lds; // d, wd, wx held in LDS (false: in the global slot too -- the gradient, the saved diagonal and the saved
                 // iterate are read lane-parallel, d once per iteration, the others a few times per solve)
  int r_cols;    // > 0: the LDS holds only the first r_cols columns of R plus one working column (QlView::nact_cap): a solve
                 // whose active set would grow past r_cols stops with kQlCapHit and is repeated with R in global memory
  bool g_lds;    // dense only: G staged in LDS (false: read in place from global memory, only its diagonal -- the one part ql0002
                 // writes, :814-854 -- is kept in LDS.  G is cold once Z = R^-1 exists: the residual refresh and the
                 // objective-increase test read it a few times per solve)
  __host__ __device__ QlDims(int n_, int m_, int mmax_, bool dense_ = true, bool a_lds_ = true, int nsc_ = 0,
                             bool bounds_ = true, bool z_lds_ = true, bool wab_lds_ = true, bool cold_lds_ = true, int r_cols_ = 0,
                             bool g_lds_ = true)
      : n(n_), m(m_), mmax(mmax_), ldg(n_ | 1), ldz(n_ | 1), lda(mmax_ | 1), dense(dense_), a_lds(a_lds_),
        nsc(nsc_ > n_ ? nsc_ : n_), bounds(bounds_), z_lds(z_lds_), wab_lds(wab_lds_), cold_lds(cold_lds_),
        r_cols(r_cols_ > 0 && r_cols_ < n_ ? r_cols_ : 0), g_lds(g_lds_) {}
  __host__ __device__ int r_tail() const { return r_cols ? r_cols * (r_cols + 1) / 2 : n * (n + 1) / 2; }
  __host__ __device__ int r_len() const { return r_tail() + n; }
  __host__ __device__ int n_doubles() const {
    return (dense ? (g_lds ? n * ldg : n) + (a_lds ? n * lda : 0) : 0) + (z_lds ? n * ldz : 0) + r_len()   // [G | diag(G), A,] [Z,] R
           + ((bounds ? 8 : 6) - (cold_lds ? 0 : 3)) * n   // x [d] ww [wd wx] lam [xl xu]
           + (wab_lds ? (m + n) + m : 0)            // wa, b (inner)
           + 4 * nsc + 8;                           // scratch + scalar slots
  }
  __host__ __device__ size_t bytes() const {
    return (size_t)n_doubles() * 8 + (size_t)((n + 1) & ~1) * 4;
  }
};

constexpr int kQlCapHit = -7777;                          // QlResult::ifail of a solve stopped by QlView::nact_cap
struct QlView {
  int n, m, me, mn, ldg, ldz, lda;
  int r_tail = 0;                                          // offset of the n scratch entries behind R's columns
  int nact_cap = 0;                                        // > 0: stop (kQlCapHit) when the active set would exceed it
  double *G, *Z, *R, *A;
  double *Gdiag = nullptr;                                 // dense view with G read in place: the diagonal's LDS copy
  double *Rf;                                              // where the Cholesky factor of G is formed on the way to Z = R^-1 (dead
                                                           // afterwards): R itself, or a full-size array when R is capped
  double *x, *d, *ww, *wd, *wx, *lam, *xl, *xu, *wa, *b;
  double *sc0, *sc1, *sc2, *sc3, *slot;
  double *ztile = nullptr;                                 // Z in registers (ZRegs): the LDS tile of zr_zt_times_ww
  int *iact;
  // kBounds / kWabLds mirror QlDims::bounds / wab_lds at compile time (a run-time choice between an LDS and a global array
  // would make the pointer generic and every access through it a flat_ instruction); ext_wab: [wa (mmax + nmax) | b (mmax)]
  // kColdLds mirrors QlDims::cold_lds: false puts d | wd | wx at ext_cold, ext_cold_ld doubles apart
  template <bool kBounds = true, bool kWabLds = true, bool kColdLds = true>
  __device__ __forceinline__ void carve(double *base, const QlDims &D, int me_, double *ext_wab = nullptr, int ext_b_off = 0,
                                        double *ext_cold = nullptr, int ext_cold_ld = 0) {
    n = D.n; m = D.m; me = me_; mn = D.m + D.n; ldg = D.ldg; ldz = D.ldz; lda = D.lda;
    r_tail = D.r_tail(); nact_cap = D.r_cols;
    double *p = base;
    G = nullptr; A = nullptr;
    if (D.dense) {
      if (D.g_lds) { G = p; p += n * ldg; }
      else { Gdiag = p; p += n; }                          // the caller points G (and ldg) at the problem's own array
    }
    Z = nullptr;
    if (D.z_lds) { Z = p; p += n * ldz; }
    R = p; p += D.r_len(); Rf = R;
    if (D.dense && D.a_lds) { A = p; p += n * lda; }
    if constexpr (kColdLds) { x = p; p += n;  d = p; p += n;  ww = p; p += n;  wd = p; p += n;  wx = p; p += n; lam = p; p += n; }
    else {
      // lean layout (element view): the four scratch vectors come right behind R -- like R they are dead outside the solve, so the
      // tick's pre-solve overlay may run over both (the smaller R's LDS part, the more gaits fit a CU) -- and x, which must
      // survive the solve, after them
      sc0 = p; p += D.nsc; sc1 = p; p += D.nsc; sc2 = p; p += D.nsc; sc3 = p; p += D.nsc;
      x = p; p += n; ww = p; p += n; lam = p; p += n; d = ext_cold; wd = ext_cold + ext_cold_ld; wx = ext_cold + 2 * ext_cold_ld;
    }
    if constexpr (kBounds) { xl = p; p += n; xu = p; p += n; } else { xl = nullptr; xu = nullptr; }
    if constexpr (kWabLds) { wa = p; p += m + n; b = p; p += m; } else { wa = ext_wab; b = ext_wab + ext_b_off; }
    if constexpr (kColdLds) { sc0 = p; p += D.nsc; sc1 = p; p += D.nsc; sc2 = p; p += D.nsc; sc3 = p; p += D.nsc; }
    slot = p; p += 8;
    iact = reinterpret_cast<int *>(p);
  }
  // Same partition laid out for the compile-time maxima (NMAX, MMAX), whatever the actual n, m: every array then sits at
  // a constant offset from the wave's LDS base (immediate offsets in the ds instructions, no address registers), and Z's
  // leading dimension is the constant NMAX|1.  No G / A matrices (compact views only).
  template <int NMAX, int MMAX, int NSC, bool kExt = false>   // kExt: wa / b live in ext_wab (global memory), known at compile time
  __device__ void carve_fixed(double *base, int n_, int m_, int me_, double *ext_wab = nullptr) {
    n = n_; m = m_; me = me_; mn = m_ + n_; ldg = NMAX | 1; ldz = NMAX | 1; lda = MMAX | 1;
    r_tail = n_ * (n_ + 1) / 2; nact_cap = 0;
    double *p = base;
    G = nullptr; A = nullptr;
    Z = p; p += NMAX * (NMAX | 1);
    R = p; p += NMAX * (NMAX + 1) / 2 + NMAX; Rf = R;
    x = p; p += NMAX;  d = p; p += NMAX;  ww = p; p += NMAX;  wd = p; p += NMAX;
    wx = p; p += NMAX; lam = p; p += NMAX; xl = nullptr; xu = nullptr;     // bounds come from the problem view
    if constexpr (kExt) { wa = ext_wab; b = ext_wab + (MMAX + NMAX); }
    else { wa = p; p += MMAX + NMAX; b = p; p += MMAX; }
    sc0 = p; p += NSC; sc1 = p; p += NSC; sc2 = p; p += NSC; sc3 = p; p += NSC;
    slot = p; p += 8;
    iact = reinterpret_cast<int *>(p);
  }
  // Element view with the horizon known at compile time (NMAX = 2N + 2 kSMax, MMAX = 1 + 4N + 5 kSMax): the lean partition of
  // carve<false, false, false> laid out for the model's LARGEST problem whatever n and m the tick has, so that every LDS array
  // sits at a constant offset from the wave's base (immediates in the ds instructions instead of address registers) and Z's
  // leading dimension is the constant NMAX:   x | ww | lam | slot | iact | sc0 sc1 sc2 sc3 | R.
  // R comes last: its length -- r_cols columns and one working column, or all n -- is the one thing the column cap decides.  The
  // tick's pre-solve overlay lies over sc0 .. R (dead outside the solve).  Z, wa | b and d | wd | wx live in the per-block global
  // slot (the caller passes them: constants behind one base).  Same bytes as QlDims(NMAX, MMAX, ...).bytes().
  template <int NMAX, int MMAX>
  __device__ __forceinline__ void carve_fixed_elem(double *base, int n_, int m_, int me_, int r_cols, double *z_ext, double *wa_ext,
                                                    double *b_ext, double *d_ext, double *wd_ext, double *wx_ext, double *rf_ext) {
    n = n_; m = m_; me = me_; mn = m_ + n_; ldg = NMAX | 1; ldz = NMAX; lda = MMAX | 1;   // Z is global here: whole cache lines per column
    const bool capped = r_cols > 0 && r_cols < n_;
    nact_cap = capped ? r_cols : 0;
    r_tail = capped ? r_cols * (r_cols + 1) / 2 : n_ * (n_ + 1) / 2;
    G = nullptr; A = nullptr; xl = nullptr; xu = nullptr;
    double *p = base;
    x = p; p += NMAX; ww = p; p += NMAX; lam = p; p += NMAX;
    slot = p; p += 8;
    iact = reinterpret_cast<int *>(p); p += ((NMAX + 1) & ~1) / 2;
    sc0 = p; p += NMAX; sc1 = p; p += NMAX; sc2 = p; p += NMAX; sc3 = p; p += NMAX;
    R = p;
    Z = z_ext; wa = wa_ext; b = b_ext; d = d_ext; wd = wd_ext; wx = wx_ext; Rf = rf_ext;
  }
  // The dense ql0001_ boundary at a size known at compile time (the Herdt QP: NMAX = 36, MMAX = 76): G and A are read in place
  // (the caller points G / A at the problem's own arrays, leading dimensions NMAX / MMAX), wa | b live in the block's global
  // slot; Z, R and the vectors sit at constant LDS offsets:  Z | R | x d ww wd wx lam | xl xu | diag(G) | sc0..sc3 | slot | iact
  template <int NMAX, int MMAX>
  __device__ __forceinline__ void carve_fixed_dense(double *base, int n_, int m_, int me_, double *ext_wab) {
    n = n_; m = m_; me = me_; mn = m_ + n_; ldg = NMAX; ldz = NMAX | 1; lda = MMAX;
    r_tail = n_ * (n_ + 1) / 2; nact_cap = 0;
    double *p = base;
    Z = p; p += NMAX * (NMAX | 1);
    R = p; p += NMAX * (NMAX + 1) / 2 + NMAX; Rf = R;
    x = p; p += NMAX; d = p; p += NMAX; ww = p; p += NMAX; wd = p; p += NMAX; wx = p; p += NMAX; lam = p; p += NMAX;
    xl = p; p += NMAX; xu = p; p += NMAX;
    Gdiag = p; p += NMAX;
    wa = ext_wab; b = ext_wab + (MMAX + NMAX);
    sc0 = p; p += NMAX; sc1 = p; p += NMAX; sc2 = p; p += NMAX; sc3 = p; p += NMAX;
    slot = p; p += 8;
    iact = reinterpret_cast<int *>(p);
    G = nullptr; A = nullptr;
  }
  template <int NMAX> static constexpr size_t fixed_dense_bytes() {
    return 8 * (size_t)(NMAX * (NMAX | 1) + NMAX * (NMAX + 1) / 2 + NMAX + 9 * NMAX + 4 * NMAX + 8) + 4 * (size_t)((NMAX + 1) & ~1);
  }
  // doubles in front of sc0 in that layout (where the tick's overlay starts)
  template <int NMAX> static constexpr int fixed_elem_head() { return 3 * NMAX + 8 + ((NMAX + 1) & ~1) / 2; }
};

#define Zm(i, j) q.Z[(i) + (j) * q.ldz]
// G and A go through the problem view `prob` (DenseProb: LDS matrices; HerdtProb: regenerated on the fly)
#define Gm(i, j) prob.G(q, (i), (j))
#define Am(k, i) prob.A(q, (k), (i))

struct QlView;
template <bool kGLds, int kNMc = 0>         // where G lives is known at compile time (ds_ or global_ accesses, never flat_)
struct DenseProbT {
  static constexpr bool kCompact = false;
  static constexpr bool kNanExact = true;   // the ql0001_ boundary takes anybody's QP: NaN iterates end the way the reference ends them (scan_nan_exact)
  static constexpr bool kHasFactor = false;    // no structure to exploit: ql0002's own Cholesky and inverse
  static constexpr bool kRowOps = false;   // no structured row products: rows are read element by element
  static constexpr int kNM = kNMc;     // 0: no compile-time bound on n; > 0: n <= kNM (the Herdt-sized boundary kernel: the
                                       // compile-time-bounded forms of the sweep, the back substitution and the ordered sums)
  static constexpr bool kWideN = false;  // 64 <= n <= 128 is not known at compile time: the wide (two rows / columns per lane) forms by test
  static constexpr int kFixedLdz = kNMc > 0 ? (kNMc | 1) : 0;   // > 0: Z in LDS with this leading dimension (carve_fixed_dense)
  __device__ __forceinline__ double G(const QlView &q, int i, int j) const;
  __device__ __forceinline__ double A(const QlView &q, int k, int i) const;
  __device__ __forceinline__ double Gd(const QlView &q, int i) const;
  __device__ __forceinline__ void setGd(const QlView &q, int i, double v) const;
  __device__ __forceinline__ double xl(const QlView &q, int i) const;
  __device__ __forceinline__ double xu(const QlView &q, int i) const;
};
typedef DenseProbT<true> DenseProb;
// The dense boundary at a size known at compile time (n <= NM, m <= MM <= 128) with the constraint matrix kept as REGISTER ROWS:
// lane k carries row k (ar0) and row 64 + k (ar1) of A, loaded once per QP -- the violation scan, which walks every row in every
// iteration, then reads no memory but x (LDS broadcasts); the new normal is written out by the lane that owns the row.  Every
// other access to A (once per solve, or in the residual refresh) still reads it in place.
template <int NM, int MM>
struct DenseRegProb : DenseProbT<false, NM> {
  static constexpr bool kRegRows = true;
  double ar0[NM], ar1[NM];
  __device__ __forceinline__ void load_rows(const QlView &q, int lane) {
    const int m = q.m;
    const int mc = m > 0 ? m - 1 : 0;          // a bounds-only QP (m == 0) must not read row -1: row 0 of the caller's buffer exists (mmax >= 1)
    const int k0 = lane < m ? lane : mc, k1 = lane + 64 < m ? lane + 64 : mc;
#pragma unroll
    for (int i = 0; i < NM; ++i) { ar0[i] = q.A[k0 + i * q.lda]; ar1[i] = q.A[k1 + i * q.lda]; }
  }
  // ww[i] = A(k, i), i < n: the owner of row k writes it (compile-time column indices: the rows stay in registers)
  __device__ __forceinline__ void row_to(const QlView &q, int k, double *dst, int lane) const {
    const int n = q.n;
    if (lane == (k & 63)) {
#pragma unroll
      for (int i = 0; i < NM; ++i)
        if (i < n) dst[i] = k < 64 ? ar0[i] : ar1[i];
    }
  }
};
template <class P, class = void> struct HasRegRows { static constexpr bool value = false; };
template <class P> struct HasRegRows<P, typename std::enable_if<P::kRegRows>::type> { static constexpr bool value = true; };
#define Rp(i, j) q.R[(j) * ((j) + 1) / 2 + (i)]
#define Rf(i, j) q.Rf[(j) * ((j) + 1) / 2 + (i)]        // the same packing, in the factorisation's array

// ---- attribution builds (never shipped): -DWG_REPEAT_PHASE=k executes the idempotent phase k of every active-set iteration
// TWICE (same results: each of these phases only reads the solver state it does not write); the difference of the hardware
// counters against the plain build is that phase's share (tools/phase_attribution.sh).  The memory clobber makes the second
// pass reload its operands instead of being folded into the first.
#ifdef WG_REPEAT_PHASE
#define WG_REP(id) for (int wg_rep_ = 0; wg_rep_ < (((WG_REPEAT_PHASE) == (id)) ? 2 : 1); ++wg_rep_, ({ asm volatile("" ::: "memory"); }))
// a phase whose results live in registers only would lose its first pass to dead-code elimination: the sink "uses" them
#define WG_SINK(x) asm volatile("" ::"v"(x))
#else
#define WG_REP(id)
#define WG_SINK(x) do {} while (0)
#endif

// ---- optional in-kernel phase timers (diagnostic build only: -DWG_PROFILE) ----
#ifdef WG_PROFILE
__device__ unsigned long long g_prof[48];               // 32..34: the sweep's three phases (norm chain, coefficients, row rotations)
#define PT_DECL unsigned long long pt_acc[28] = {0}; unsigned long long pt_cnt[4] = {0}; unsigned long long pt_sw[3] = {0}; unsigned long long pt_last = clock64();
#define PT(k) do { unsigned long long t_ = clock64(); pt_acc[k] += t_ - pt_last; pt_last = t_; } while (0)
#define PT_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 28; ++k_) if (k_ < 21 || k_ > 23) atomicAdd(&g_prof[k_], pt_acc[k_]); \
                                                      for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&g_prof[28 + k_], pt_cnt[k_]); \
                                                      for (int k_ = 0; k_ < 3; ++k_) atomicAdd(&g_prof[32 + k_], pt_sw[k_]); } } while (0)
#define PT_SW_PARAM , unsigned long long *ptsw = nullptr
#define PT_SW_ARG , pt_sw
#define PT_SW(k) do { if (ptsw) { unsigned long long t_ = clock64(); ptsw[k] += t_ - ptsw_last; ptsw_last = t_; } } while (0)
#define PT_SW_BEGIN unsigned long long ptsw_last = clock64();
// event counters 28..31: kept in registers and flushed once (a global atomic per event would show up in the phase it sits in)
#define PT_COUNT(k) do { pt_cnt[(k) - 28]++; } while (0)
#else
#define PT_COUNT(k) do {} while (0)
#define PT_SW_PARAM
#define PT_SW_ARG
#define PT_SW(k) do {} while (0)
#define PT_SW_BEGIN
#define PT_DECL
#define PT(k) do {} while (0)
#define PT_FLUSH do {} while (0)
#endif

// per-lane parameters of the active constraints, for problem views that supply a fast residual refresh
struct NoActiveParams {};
template <class P, class = void> struct ActiveParamsOf { typedef NoActiveParams type; };
template <class P> struct ActiveParamsOf<P, typename std::enable_if<P::kCompact>::type> { typedef typename P::ActiveParams type; };

struct QlResult {
  int ifail, n_iter, nact, hist_len;
};
// The scalar state of ql0002's main loop between two iterations: what a solve stopped by QlView::nact_cap (kQlCapHit) hands to
// its continuation.  The arrays (x, multipliers, active set, Z, R, wa) stay where they are; the caller moves R to its larger
// home, clears the cap and calls ql_solve again with `valid` set: the solve goes on where it stopped, same arithmetic.
struct QlResume {
  int valid = 0;
  int nact, info, iterc, itref, iflag, jfinc, knext, st, hist_len;
  double xmag, vfact, res, ratio, diag;
};

template <bool kGLds, int kNMc> __device__ __forceinline__ double DenseProbT<kGLds, kNMc>::G(const QlView &q, int i, int j) const {
  if constexpr (kGLds) return q.G[i + j * q.ldg];
  else {
    const double g = q.G[i + j * q.ldg], dg = q.Gdiag[i];  // both requested: the select costs no round trip
    return i == j ? dg : g;
  }
}
template <bool kGLds, int kNMc> __device__ __forceinline__ double DenseProbT<kGLds, kNMc>::A(const QlView &q, int k, int i) const { return q.A[k + i * q.lda]; }
template <bool kGLds, int kNMc> __device__ __forceinline__ double DenseProbT<kGLds, kNMc>::Gd(const QlView &q, int i) const {
  if constexpr (kGLds) return q.G[i + i * q.ldg]; else return q.Gdiag[i];
}
template <bool kGLds, int kNMc> __device__ __forceinline__ void DenseProbT<kGLds, kNMc>::setGd(const QlView &q, int i, double v) const {
  if constexpr (kGLds) q.G[i + i * q.ldg] = v; else q.Gdiag[i] = v;
}
template <bool kGLds, int kNMc> __device__ __forceinline__ double DenseProbT<kGLds, kNMc>::xl(const QlView &q, int i) const { return q.xl[i]; }
template <bool kGLds, int kNMc> __device__ __forceinline__ double DenseProbT<kGLds, kNMc>::xu(const QlView &q, int i) const { return q.xu[i]; }

// ---- wave reductions on the DPP data path (gfx9 row shifts / row broadcasts: one VALU move per 32-bit half and step, no
// LDS crossbar, no exec-mask branching).  max / min are idempotent, so lanes without a partner just keep their own value
// (update_dpp's `old` operand): after row_shr 1,2,4,8 lane 15 of every row holds the row's result, row_bcast:15 and
// row_bcast:31 fold the rows into lane 63.  The __shfl_xor butterflies these replace cost ~1400 cycles per arg-max
// (three ds_bpermute per round plus divergent selects); this is ~250.
template <int CTRL>
__device__ __forceinline__ int dpp_keep(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_keep(double v) {
  const int lo = dpp_keep<CTRL>(__double2loint(v)), hi = dpp_keep<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {
  v = __builtin_fmax(v, dpp_keep<0x111>(v));   // row_shr:1
  v = __builtin_fmax(v, dpp_keep<0x112>(v));   // row_shr:2
  v = __builtin_fmax(v, dpp_keep<0x114>(v));   // row_shr:4
  v = __builtin_fmax(v, dpp_keep<0x118>(v));   // row_shr:8
  v = __builtin_fmax(v, dpp_keep<0x142>(v));   // row_bcast:15
  v = __builtin_fmax(v, dpp_keep<0x143>(v));   // row_bcast:31
  return rl(v, 63);
}
__device__ __forceinline__ int wave_min_int(int v) {
  { const int o = dpp_keep<0x111>(v); v = o < v ? o : v; }
  { const int o = dpp_keep<0x112>(v); v = o < v ? o : v; }
  { const int o = dpp_keep<0x114>(v); v = o < v ? o : v; }
  { const int o = dpp_keep<0x118>(v); v = o < v ? o : v; }
  { const int o = dpp_keep<0x142>(v); v = o < v ? o : v; }
  { const int o = dpp_keep<0x143>(v); v = o < v ? o : v; }
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_int(int v) {
  { const int o = dpp_keep<0x111>(v); v = o > v ? o : v; }
  { const int o = dpp_keep<0x112>(v); v = o > v ? o : v; }
  { const int o = dpp_keep<0x114>(v); v = o > v ? o : v; }
  { const int o = dpp_keep<0x118>(v); v = o > v ? o : v; }
  { const int o = dpp_keep<0x142>(v); v = o > v ? o : v; }
  { const int o = dpp_keep<0x143>(v); v = o > v ? o : v; }
  return __builtin_amdgcn_readlane(v, 63);
}
// arg-max over the wave: larger v wins, equal v -> smaller idx.  idx < 0 = no candidate (then idx stays < 0).
// Candidates must be finite.  Result is wave-uniform.
__device__ __forceinline__ void wave_argmax_first(double &v, int &idx) {
  const double vv = idx >= 0 ? v : -__builtin_huge_val();
  const double vmax = wave_max(vv);
  const int key = (idx >= 0 && v == vmax) ? idx : 0x7fffffff;
  const int kmin = wave_min_int(key);
  v = vmax;
  idx = kmin == 0x7fffffff ? -1 : kmin;
}

// norm of a rotation, qld.cpp:1921-1926 / 2005-2010:  t = max(|p|,|q|);  t * sqrt((p/t)^2 + (q/t)^2).
// This sits on the sequential chain of every sweep.  One of the two quotients is x/|x| = +-1 EXACTLY (IEEE division is
// exact there), its square is exactly 1.0, and a + b == b + a: so only the other quotient is a real division.  Same
// bits as the reference's two divisions for finite operands (0/0 stays NaN); half the divide latency on the chain.
// sqrt(x) for 1 <= x <= 2: the compiler's own correctly-rounded f64 expansion (v_rsq_f64 + two coupled Newton steps,
// same operations in the same order) without its range scaling (ldexp in, ldexp out) and special-value select, which
// are exact no-ops on this interval.  Five dependent instructions shorter; bit-identical result.
__device__ __forceinline__ double sqrt_1to2(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}
__device__ __forceinline__ double givens_norm(double p, double qq) {
  const double ap = fabs(p), aq = fabs(qq);
  const bool pbig = ap >= aq;                      // maxd(): a >= b ? a : b
  // the value of the select (finite operands) in one instruction; __builtin_fmax would first canonicalise both operands
  // (two more v_max_f64) -- NaN operands give NaN either way, and that result is discarded wherever it can arise
  double t;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(t) : "v"(p), "v"(qq));
  const double d = (pbig ? qq : p) / t;            // |d| <= 1
  const double x = 1.0 + d * d;
#ifdef WG_GENERIC_SQRT
  return t * sqrt(x);
#else
  return t * sqrt_1to2(x);                         // x is in [1, 2], or NaN (which stays NaN)
#endif
}
// The same norm for operands whose non-zero magnitudes lie in [2^-400, 2^404] (sweep_range_ok), five instructions shorter:
//   * min / max of the magnitudes in one instruction each: the quotient enters only through its square, and
//     (|x| / |y|)^2 == (x / y)^2 bit for bit;
//   * the division is the compiler's own f64 expansion (v_rcp_f64, two Newton steps, quotient, residual, correction) without
//     v_div_scale_f64 / v_div_fixup_f64: with 0 <= a <= t and t in that range both scalings are the identity and the fix-up
//     passes the quotient through whenever it is >= 2^-27; below that d * d < 2^-54 and 1 + d * d is exactly 1 whatever the
//     last bits of d.  t == 0 (both operands zero) gives NaN here as 0/0 does there: the caller discards it.
__device__ __forceinline__ double givens_norm_fast(double p, double qq) {
  double t, a;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(t) : "v"(p), "v"(qq));
  asm("v_min_f64 %0, |%1|, |%2|" : "=v"(a) : "v"(p), "v"(qq));
  double r = __builtin_amdgcn_rcp(t);
  double e = __builtin_fma(-t, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-t, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q0 = a * r;
  const double rem = __builtin_fma(-t, q0, a);
  const double d = __builtin_fma(rem, r, q0);
  const double x = 1.0 + d * d;
  return t * sqrt_1to2(x);
}
// every entry of s[lo, hi) is zero or has its magnitude in [2^-400, 2^400] (then every norm of a sweep over them, being at
// least its larger operand and at most sqrt(n) times the largest entry, is zero or in [2^-400, 2^404]); wave-uniform
template <bool kOnePass = false>                          // kOnePass: hi - lo <= 64 (the caller's n <= 64)
__device__ __forceinline__ bool sweep_range_ok(const double *s, int lo, int hi, int lane) {
  if constexpr (kOnePass) {                                 // one entry per lane: no lane-dependent loop, no exec-mask juggling
    const int j = lo + lane;
    const bool in = j < hi;
    const double v = fabs(s[in ? j : lo]);
    return __ballot(in && !(v == 0.0 || (v >= 0x1p-400 && v <= 0x1p400))) == 0ull;
  }
  bool bad = false;
  for (int j = lo + lane; j < hi; j += 64) {
    const double v = fabs(s[j]);
    bad = bad || !(v == 0.0 || (v >= 0x1p-400 && v <= 0x1p400));
  }
  return __ballot(bad) == 0ull;
}
// qld.cpp:1921-1930 / 2005-2014
__device__ __forceinline__ void givens(double p, double qq, double &ga, double &gb, double &nrm) {
  const double sum = givens_norm(p, qq);
  ga = p / sum;
  gb = qq / sum;
  nrm = sum;
}
__device__ __forceinline__ bool significant(double base, double delta_abs) {
  double temp = base + delta_abs * wg_kconst(.1);
  double tempa = base + delta_abs * wg_kconst(.2);
  if (temp <= base) return false;
  if (tempa <= temp) return false;
  return true;
}

// s[i] = sum_j Z(j,i) * ww[j]   (qld.cpp:2071-2085); lane i owns s[i]
template <int NM = 0, int GRP = 8, bool kWide = false>   // NM > 0: n <= NM known at compile time (the wide form is left out)
__device__ __forceinline__ void zt_times_ww(const QlView &q, double *s, int lane) {   // kWide: 64 <= n <= 128 known at compile time
  const int n = q.n;
  if (kWide || ((NM == 0 || NM > 64) && n > 64 && n <= 128)) {
    // two columns per lane in ONE pass (the second pass of the strided form has n - 64 useful lanes), loads in groups of
    // eight ahead of the two add chains: at this size Z may live in global memory (L2), where every exposed round trip
    // costs hundreds of cycles
    // surplus lanes all shadow column 0: one address per load instruction (it coalesces to a single request on a line lane 0 has
    // just fetched) instead of 56 more scattered ones -- the column walk is bound by the address path, not by the bytes
    const int i0 = lane, i1 = lane + 64 < n ? lane + 64 : 0;
    const double *z0 = q.Z + (size_t)i0 * q.ldz, *z1 = q.Z + (size_t)i1 * q.ldz;
    double a0 = 0.0, a1 = 0.0;
    int j = 0;
    for (; j + GRP <= n; j += GRP) {
      double u0[GRP], u1[GRP], w[GRP];
#pragma unroll
      for (int e = 0; e < GRP; ++e) { u0[e] = WG_ZLD(z0 + j + e); u1[e] = WG_ZLD(z1 + j + e); w[e] = q.ww[j + e]; }
#pragma unroll
      for (int e = 0; e < GRP; ++e) { a0 += u0[e] * w[e]; a1 += u1[e] * w[e]; }
    }
    for (; j < n; ++j) { const double w = q.ww[j]; a0 += WG_ZLD(z0 + j) * w; a1 += WG_ZLD(z1 + j) * w; }
    s[i0] = a0;
    if (lane + 64 < n) s[i1] = a1;
    WG_WSYNC();
    return;
  }
  for (int i = lane; i < n; i += 64) {
    double acc = 0.0;
    WG_UNROLL
    for (int j = 0; j < n; ++j) acc += Zm(j, i) * q.ww[j];
    s[i] = acc;
  }
  WG_WSYNC();
}

// The same product for a constraint normal whose entries are known to be exact zeros outside the row ranges [0, r], [NH, NH + r]
// and [2 NH, n) (a CoP row of instant r of the Herdt QP; r = -1: a foot-placement row, nothing in the jerk columns): the rows of
// Z in between are not read.  Their products are +-0.0 and the sums -- started from +0.0, never -0.0 -- do not change when they
// are left out: the same bits for 2 (NH - 1 - r) / n fewer bytes of Z, on average 46 % of the walk at NH = 32, which is what
// this kernel is bound by (Z lives in global memory).  Ranges are wave-uniform and end on multiples of GRP (NH is one): no
// ragged tails; entries of a chunk beyond r meet their zero coefficients.
template <int NH, int GRP>
__device__ __forceinline__ void zt_times_ww_cop(const QlView &q, double *s, int lane, int r) {
  static_assert(NH % GRP == 0, "the row ranges must end on whole chunks");
  const int n = q.n;
  const int i0 = lane, i1 = lane + 64 < n ? lane + 64 : 0;         // surplus lanes all shadow column 0 (one coalesced request: see zt_times_ww)
  const double *z0 = q.Z + (size_t)i0 * q.ldz, *z1 = q.Z + (size_t)i1 * q.ldz;
  double a0 = 0.0, a1 = 0.0;
  const int len = r + 1;                                        // rows of each jerk block that carry an entry
#pragma unroll 1
  for (int blk = 0; blk < 2; ++blk) {
    const int base = blk * NH;
    for (int j = base; j < base + len; j += GRP) {
      double u0[GRP], u1[GRP], w[GRP];
#pragma unroll
      for (int e = 0; e < GRP; ++e) { u0[e] = WG_ZLD(z0 + j + e); u1[e] = WG_ZLD(z1 + j + e); w[e] = q.ww[j + e]; }
#pragma unroll
      for (int e = 0; e < GRP; ++e) { a0 += u0[e] * w[e]; a1 += u1[e] * w[e]; }
    }
  }
  for (int j = 2 * NH; j < n; ++j) { const double w = q.ww[j]; a0 += WG_ZLD(z0 + j) * w; a1 += WG_ZLD(z1 + j) * w; }
  s[i0] = a0;
  if (lane + 64 < n) s[i1] = a1;
  WG_WSYNC();
}

#ifndef WG_ZT_TILED
#define WG_ZT_TILED 2                  // 0: one column per lane (zt_times_ww_cop), 1: four lanes per column, 2: that with the next block requested ahead
#endif
// The same product for the fixed N = 32 view (Z global, leading dimension NH * 2 + 8 = 72) with FOUR lanes per column: lane L
// owns column 16 p + L / 4 in pass p and the two rows j0 + 2 (L & 3), + 1 of every eight-row block -- the four lanes of a column
// read 64 contiguous bytes, so a load instruction touches 16 cache lines instead of 64 and an eight-row block of all 72 columns
// costs 80 line requests instead of 288 (the column walk is bound by its requests).  The eight products of a block
// reach every lane of the quad by DPP (quad_perm broadcasts, no LDS) and are added in row order: the same adds in the same
// order as zt_times_ww_cop with groups of eight (whole blocks: the rows past the row's instant carry exact zeros; rows past n --
// block 8 only -- are masked to +0.0, which never changes a sum that cannot be -0.0).
template <int K>
__device__ __forceinline__ double wg_quad_bcast(double v) {   // lane K of every quad to the whole quad
  constexpr int ctrl = K * 0x55;                              // quad_perm:[K,K,K,K]
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int NH>
__device__ __forceinline__ void zt_times_ww_cop_tiled(const QlView &q, double *s, int lane, int r, bool tail) {
  constexpr int L = 2 * NH + 8;                               // q.ldz of carve_fixed_elem
  constexpr int NP = (L + 15) / 16;                           // passes of sixteen columns
  const int n = q.n;
  const int h = lane & 3, cq = lane >> 2;
  const double2 *zb[NP];
  bool colok[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c = 16 * p + cq;
    colok[p] = c < n;
    zb[p] = reinterpret_cast<const double2 *>(q.Z + (size_t)(colok[p] ? c : 0) * L + 2 * h);   // lanes without a column: column 0
  }
  double acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) acc[p] = 0.0;
  struct Blk { double2 u[NP]; double w0, w1; };
  auto load = [&](Blk &B, int j0) {                           // rows j0 .. j0 + 7 (j0 a multiple of 8): requested, not waited for
#pragma unroll
    for (int p = 0; p < NP; ++p) B.u[p] = zb[p][j0 >> 1];
    B.w0 = q.ww[j0 + 2 * h]; B.w1 = q.ww[j0 + 2 * h + 1];
  };
  auto sum = [&](const Blk &B, bool last) {
    const int jr = 2 * NH + 2 * h;                            // the last block's rows (the only one that can reach past n)
    const bool ok0 = !last || jr < n, ok1 = !last || jr + 1 < n;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      double p0 = B.u[p].x * B.w0, p1 = B.u[p].y * B.w1;
      if (last) { p0 = ok0 ? p0 : 0.0; p1 = ok1 ? p1 : 0.0; }
      double a = acc[p];
      a += wg_quad_bcast<0>(p0); a += wg_quad_bcast<0>(p1);
      a += wg_quad_bcast<1>(p0); a += wg_quad_bcast<1>(p1);
      a += wg_quad_bcast<2>(p0); a += wg_quad_bcast<2>(p1);
      a += wg_quad_bcast<3>(p0); a += wg_quad_bcast<3>(p1);
      acc[p] = a;
    }
  };
  const int nb = (r >> 3) + 1;                                // blocks of each jerk range that carry entries (r = -1: none)
  // x blocks, y blocks, then the step columns' rows 2 NH .. n - 1 -- unless the normal has no entry there (a CoP row of an instant in
  // the current support phase: exact zeros, whose products leave the sums unchanged)
  const int total = 2 * nb + (tail ? 1 : 0);
  auto start = [&](int idx) { return idx < nb ? 8 * idx : (idx < 2 * nb ? NH + 8 * (idx - nb) : 2 * NH); };
#if WG_ZT_TILED == 2
  // the next block is requested before the current one is summed (two register sets taking turns)
  Blk A, B;
  load(A, start(0));
  for (int idx = 0;;) {
    if (idx + 1 >= total) { sum(A, tail); break; }
    load(B, start(idx + 1)); sum(A, false); ++idx;
    if (idx + 1 >= total) { sum(B, tail); break; }
    load(A, start(idx + 1)); sum(B, false); ++idx;
  }
#else
  for (int idx = 0; idx < total; ++idx) { Blk A; load(A, start(idx)); sum(A, tail && idx == total - 1); }
#endif
#pragma unroll
  for (int p = 0; p < NP; ++p)
    if (h == 0 && colok[p]) s[16 * p + cq] = acc[p];
  WG_WSYNC();
}

// r0 = sum_{j0 <= j < j1} Z(i0, j) * s[j], r1 the same for row i1 (j ascending, from +0.0): rows i0 = lane and i1 = lane + 64
// of a matrix of 64 < n <= 128 rows in ONE pass, the entries of eight columns requested together ahead of the two add chains
// (with Z in global memory an exposed entry is an L2 round trip; one register set only: this sits where many values are live).
// Surplus lanes shadow a real row.
template <int GRP = 8>
__device__ __forceinline__ void z_rows_times(const QlView &q, const double *s, int j0, int j1, int lane, double &r0, double &r1) {
  const int n = q.n, ldz = q.ldz;
  // lanes without a second row all shadow row 64 (n > 64 here; one coalesced request per load instead of a second copy of the first set's)
  const int i0 = lane < n ? lane : n - 1, i1 = lane + 64 < n ? lane + 64 : (n > 64 ? 64 : i0);
  const double *z0 = q.Z + i0, *z1 = q.Z + i1;
  constexpr int kG = GRP;
  double a0 = 0.0, a1 = 0.0;
  int j = j0;
  for (; j + kG <= j1; j += kG) {
    double u0[kG], u1[kG], w[kG];
#pragma unroll
    for (int e = 0; e < kG; ++e) { u0[e] = WG_ZLD(z0 + (j + e) * ldz); u1[e] = WG_ZLD(z1 + (j + e) * ldz); w[e] = s[j + e]; }
#pragma unroll
    for (int e = 0; e < kG; ++e) { a0 += u0[e] * w[e]; a1 += u1[e] * w[e]; }
  }
  if (j < j1) {                                             // the odd columns: requested together (clamped), added in order
    double u0[kG - 1], u1[kG - 1];
#pragma unroll
    for (int e = 0; e < kG - 1; ++e) { const int jj = j + e < j1 ? j + e : j1 - 1; u0[e] = WG_ZLD(z0 + jj * ldz); u1[e] = WG_ZLD(z1 + jj * ldz); }
#pragma unroll
    for (int e = 0; e < kG - 1; ++e)
      if (j + e < j1) { const double w = s[j + e]; a0 += u0[e] * w; a1 += u1[e] * w; }
  }
  r0 = a0; r1 = a1;
}

// ww[0..nact) = R^-1 s[0..nact)   (qld.cpp:1824-1851): rows from the bottom up, inner sums ascending in j.
// nact <= 64: lane j keeps ww[j] in a register; row i's products R(i,j)*ww[j] are formed lane-parallel and
// summed in index order through v_readlane (no LDS round trip on the dependent chain).
__device__ __forceinline__ void backsub(const QlView &q, const double *s, int nact, int lane) {
  if (nact <= 60) {
    const bool mine = lane < nact;
    const double sreg = mine ? s[lane] : 0.0;
    const double dreg = mine ? Rp(lane, lane) : 1.0;
    double w = 0.0;
    double rrow = (nact >= 2 && lane == nact - 1) ? Rp(nact - 2, lane) : 0.0;   // R(i, lane) of the next row to do
    for (int i = nact - 1; i >= 0; --i) {
      double sum = 0.0;
      if (i < nact - 1) {
        const double p = (lane > i && mine) ? rrow * w : 0.0;
        sum = lane_sum_ordered(p, i + 1, nact);
      }
      const double v = (rl(sreg, i) - sum) / rl(dreg, i);
      if (lane == i) w = v;
      if (i >= 1) rrow = (lane > i - 1 && mine) ? Rp(i - 1, lane) : 0.0;          // prefetch row i-1
    }
    if (mine) q.ww[lane] = w;
    WG_WSYNC();
    return;
  }
  for (int i = nact - 1; i >= 0; --i) {
    double sum = 0.0;
    WG_UNROLL
    for (int j = i + 1; j < nact; ++j) sum += Rp(i, j) * q.ww[j];
    double v = (s[i] - sum) / Rp(i, i);
    if (lane == 0) q.ww[i] = v;
    WG_WSYNC();
  }
}

// Back substitution for n <= 36 (the compact view), restructured around its dependent chain.
// Row j needs sum_{k>j} R(j,k) w_k summed ascending in k, and its FIRST term carries the value produced last (w_{j+1}):
// the additions of a row are one chain, (nact-j-1) x 8 cycles plus the divide, and nothing else may sit on it.
//   * lane k keeps w_k and forms the products R(j-1,k) w_k for the NEXT row while the current row's chain runs; they
//     go to a double-buffered LDS vector (entries outside (j, nact) are written as +0.0: adding them changes nothing);
//   * every lane then runs the row's chain redundantly on LDS-broadcast operands, the first chunk prefetched one row
//     ahead, the first term formed in registers from R(j,j+1) and the w just computed.
// The v_readlane form above costs ~40 cycles per term (two readlanes + add, serialised); this one ~8.
// buf: 2 * kBsLen doubles of LDS (the four scratch vectors are contiguous).
// one row of backsub_lds: P = {terms k = j+2 .. j+9, R(j, j+1)} prefetched by the previous row, Nx receives the same for
// row j-1.  Two copies of this body with P / Nx swapped make the hand-over a renaming instead of nine register moves.
struct BsState { double w, wprev, rr, sreg, dreg, rsd; int col, nact, lane; bool mine; };   // rsd: R(lane, lane + 1), the first term's coefficient of row `lane`
// kHead: what is known about the row's length at compile time (the rows go from the bottom up, so the r-th row from the bottom has
// r terms): -1 nothing (three uniform branches per row), 0 no term, 4 / 8 at most so many (the prefetched head only: entries
// past the row are +0.0), 9 more than eight (head and tail loop, no test) -- backsub_lds unrolls the first nine rows that way
template <int kBsLen, int kHead = -1>                       // kBsLen: length of each of the two product buffers (>= nact + 12)
__device__ __forceinline__ void bs_row(const QlView &q, double *buf, int j, BsState &S, const double (&P)[9], double (&Nx)[9]) {
  const double *bj = buf + (j & 1) * kBsLen;
  double *bn = buf + ((j & 1) ^ 1) * kBsLen;
  const int jn = j >= 1 ? j - 1 : 0, jnn = j >= 2 ? j - 2 : 0;
  const int nact = S.nact, lane = S.lane;
  // products of the next row (j-1) with the multipliers known so far (k >= j+1); nothing here waits on LDS: R(j-1, .)
  // was fetched one row ahead, and a wave's LDS operations execute in order (the barrier only pins the compiler)
  {
    const double val = (lane >= j + 1 && S.mine) ? S.rr * S.w : 0.0;
    // no exec-mask juggling on the chain's path: lanes past the buffer hold +0.0 (they are beyond nact) and write it to the
    // last slot, which is +0.0 anyway (kBsLen >= nact + 12)
    if constexpr (kBsLen <= 64) bn[lane < kBsLen ? lane : kBsLen - 1] = val;
    else bn[lane] = val;
    S.rr = Rp(jnn, S.col);
  }
  __builtin_amdgcn_wave_barrier();
  {
    // the eight terms through ONE address register with constant offsets (the scalar address arithmetic and the move into a vector
    // register were repeated for every pair), the superdiagonal entry from the register its row's lane loaded before the first row
    typedef __attribute__((address_space(3))) double lds_f64;
    const lds_f64 *np = (const lds_f64 *)(bn + j + 1);
    asm volatile("" : "+v"(np));
#pragma unroll
    for (int e = 0; e < 8; ++e) Nx[e] = np[e];              // prefetch: the next row's first terms
    Nx[8] = rl(S.rsd, jn);
  }
  const double sj = rl(S.sreg, j), dj = rl(S.dreg, j);
  double sum = 0.0;
  if constexpr (kHead >= 4) {
    sum += P[8] * S.wprev;
    sum += P[0]; sum += P[1]; sum += P[2]; sum += P[3];
    if constexpr (kHead >= 8) { sum += P[4]; sum += P[5]; sum += P[6]; sum += P[7]; }
    if constexpr (kHead >= 9) {
      // the tail through one walking address register: entries up to nact + 6 <= kBsLen - 6 are read (nact <= kBsLen - 12), no clamp
      typedef __attribute__((address_space(3))) double lds_f64;
      const lds_f64 *tp = (const lds_f64 *)(bj + j + 10);
      asm volatile("" : "+v"(tp));
      double a0 = tp[0], a1 = tp[1], a2 = tp[2], a3 = tp[3];
      for (int k = j + 10; k < nact; k += 4) {
        const double b0 = tp[4], b1 = tp[5], b2 = tp[6], b3 = tp[7];
        sum += a0; sum += a1; sum += a2; sum += a3;
        a0 = b0; a1 = b1; a2 = b2; a3 = b3;
        tp += 4;
      }
    }
  } else if constexpr (kHead == 0) {
  } else
  if (j + 1 < nact) {
    sum += P[8] * S.wprev;
    sum += P[0]; sum += P[1]; sum += P[2]; sum += P[3];
    if (j + 6 < nact) {
      sum += P[4]; sum += P[5]; sum += P[6]; sum += P[7];
      if (j + 10 < nact) {
        typedef __attribute__((address_space(3))) double lds_f64;
        const lds_f64 *tp = (const lds_f64 *)(bj + j + 10);
        asm volatile("" : "+v"(tp));
        double a0 = tp[0], a1 = tp[1], a2 = tp[2], a3 = tp[3];
        for (int k = j + 10; k < nact; k += 4) {
          const double b0 = tp[4], b1 = tp[5], b2 = tp[6], b3 = tp[7];
          sum += a0; sum += a1; sum += a2; sum += a3;
          a0 = b0; a1 = b1; a2 = b2; a3 = b3;
          tp += 4;
        }
      }
    }
  }
  const double v = (sj - sum) / dj;
  if (lane == j) S.w = v;
  S.wprev = v;
}
template <int kBsLen = 48>                                  // nact <= 64 (one multiplier per lane) and nact + 12 <= kBsLen
__device__ __forceinline__ void backsub_lds(const QlView &q, const double *s, int nact, int lane, double *buf) {
  BsState S;
  S.mine = lane < nact; S.nact = nact; S.lane = lane;
  if constexpr (kBsLen <= 64) {
    // loads from clamped addresses and selects: a load under a lane-dependent condition is an exec-mask save / restore
    const int ml = S.mine ? lane : 0;
    const double sv = s[ml], dv = Rp(ml, ml);
    S.sreg = S.mine ? sv : 0.0;
    S.dreg = S.mine ? dv : 1.0;
  } else {                                                  // the 168-register kernels (N = 32) keep the predicated form: measured
    S.sreg = S.mine ? s[lane] : 0.0;
    S.dreg = S.mine ? Rp(lane, lane) : 1.0;
  }
  if constexpr (kBsLen <= 64) {
    const int zl = lane < kBsLen ? lane : kBsLen - 1; buf[zl] = 0.0; buf[kBsLen + zl] = 0.0;
    const int rlc = lane + 1 < nact ? lane : 0;              // rows without a first term (the last one, lanes past it) read R(0, 1): unused
    S.rsd = Rp(rlc, rlc + 1);
  }
  else {
    for (int e = lane; e < kBsLen; e += 64) { buf[e] = 0.0; buf[kBsLen + e] = 0.0; }
    const int rlc = lane + 1 < nact ? lane : 0;
    S.rsd = Rp(rlc, rlc + 1);
  }
  S.col = S.mine ? lane : 0;
  S.w = 0.0; S.wprev = 0.0;
  S.rr = Rp(nact >= 2 ? nact - 2 : 0, S.col);               // R(j-1, lane) of the row whose products are formed next
  double A9[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, B9[9];
  int j = nact - 1;
#ifndef WG_BS_UNROLL_WIDE
#define WG_BS_UNROLL_WIDE 0
#endif
  if constexpr (kBsLen <= 64 || WG_BS_UNROLL_WIDE) {
    // the first nine rows unrolled with their lengths known (one uniform test per row instead of three), the rest in pairs
    if (nact > 0) { bs_row<kBsLen, 0>(q, buf, nact - 1, S, A9, B9);
    if (nact > 1) { bs_row<kBsLen, 4>(q, buf, nact - 2, S, B9, A9);
    if (nact > 2) { bs_row<kBsLen, 4>(q, buf, nact - 3, S, A9, B9);
    if (nact > 3) { bs_row<kBsLen, 4>(q, buf, nact - 4, S, B9, A9);
    if (nact > 4) { bs_row<kBsLen, 4>(q, buf, nact - 5, S, A9, B9);
    if (nact > 5) { bs_row<kBsLen, 8>(q, buf, nact - 6, S, B9, A9);
    if (nact > 6) { bs_row<kBsLen, 8>(q, buf, nact - 7, S, A9, B9);
    if (nact > 7) { bs_row<kBsLen, 8>(q, buf, nact - 8, S, B9, A9);
    if (nact > 8) { bs_row<kBsLen, 8>(q, buf, nact - 9, S, A9, B9);
      for (j = nact - 10; j >= 1; j -= 2) { bs_row<kBsLen, 9>(q, buf, j, S, B9, A9); bs_row<kBsLen, 9>(q, buf, j - 1, S, A9, B9); }
      if (j == 0) bs_row<kBsLen, 9>(q, buf, 0, S, B9, A9);
    }}}}}}}}}
  } else {
    for (; j >= 1; j -= 2) { bs_row<kBsLen>(q, buf, j, S, A9, B9); bs_row<kBsLen>(q, buf, j - 1, S, B9, A9); }
    if (j == 0) bs_row<kBsLen>(q, buf, 0, S, A9, B9);
  }
  if constexpr (kBsLen <= 64) {
    if (nact > 0) {                                         // lanes past nact shadow lane nact - 1: its value to its address
      const double wl = rl(S.w, nact - 1);
      q.ww[S.mine ? lane : nact - 1] = S.mine ? S.w : wl;
    }
  } else if (S.mine) q.ww[lane] = S.w;
  WG_WSYNC();
}

// ---------------------------------------------------------------------------------------------------
// Compile-time-bounded versions for n <= NM (the Herdt QP: NM = 36).  Measured on MI355X (tools/micro/lat.hip):
// a dependent fp64 add/mul costs ~8 cycles, an add fed by v_readlane ~40, a divide 71, sqrt ~100.  With static
// trip counts every LDS address is an immediate offset, so the compiler issues all loads ahead of the dependent
// chain and the ordered sums run at the 8-cycle floor.
// ---------------------------------------------------------------------------------------------------

// sum of term[0..cnt) in index order (term lives one-per-lane); scratch: NM doubles of LDS, 16-byte aligned.
#ifndef WG_OS_CHUNK
#define WG_OS_CHUNK 12
#endif
constexpr int kOsChunk = WG_OS_CHUNK;
template <int NM>
__device__ __forceinline__ double ordered_sum_lds(double term, double *scratch, int cnt, int lane) {
  {
    const double v = (lane < cnt) ? term : 0.0;
    const double vl = rl(v, NM - 1);                        // lanes past NM shadow lane NM - 1
    scratch[lane < NM ? lane : NM - 1] = lane < NM ? v : vl;
  }
  WG_WSYNC();
  // loads in groups of kOsChunk ahead of the add chain: enough to cover the LDS latency, few enough live registers
  double sum = 0.0;
#pragma unroll
  for (int i0 = 0; i0 < NM; i0 += kOsChunk) {
    double t[kOsChunk];
#pragma unroll
    for (int i = 0; i < kOsChunk; ++i) t[i] = (i0 + i < NM) ? scratch[i0 + i] : 0.0;
#pragma unroll
    for (int i = 0; i < kOsChunk; ++i) if (i0 + i < NM) sum += t[i];   // entries >= cnt are +0.0: they leave the sum unchanged
  }
  WG_WSYNC();
  return sum;
}

// Givens sweep (qld.cpp:1992-2030) for n <= 64, written without data-dependent control flow: on a single
// resident wave every taken branch costs tens of cycles, so predicates become selects, inactive lanes shadow
// lane n-1 (same addresses, same values), and LDS operands are fetched one or two steps ahead of the
// dependent chain.
//   phase 1  chain of rotation norms (all lanes redundantly; s[c-1] prefetched); lane c records its rotation;
//   phase 2  lane c turns (p, q, norm) into (ga, gb) and publishes the pair in LDS;
//   phase 3  lane i carries row i of Z through the rotations.
template <int kLdzC = 0>                                     // > 0: Z lives in LDS with this leading dimension (every compile-time-bounded view)
__device__ __forceinline__ void sweep_flat(const QlView &q, double *s, int nu, int nact, int lane PT_SW_PARAM) {
  const int n = q.n;
  if (nu - 1 <= nact) return;
  PT_SW_BEGIN
  // Phase 1 leaves ONE value per rotation behind -- `cur` as it leaves rotation c, in chain[c - 1] -- from which lane c
  // rebuilds its rotation afterwards: q = the value that entered (chain[c], or s[nu-1] for the first one), p = s[c-1]
  // (untouched until phase 2), norm = chain[c-1] when q != 0 (then cur = norm), skipped when q == 0 (then cur = p).
  // givens_norm runs unguarded on q == 0: its result (|p|, or NaN for 0/0) is discarded by the select.
  double *chain = q.sc2;                                    // nu <= n entries
  WG_REP(3)
  if (sweep_range_ok<true>(s, nact, nu, lane)) {
    // the usual case: the shorter norm; unrolled by two so that handing the prefetched operand on is a renaming.  The operand
    // and the record are reached through two walking pointers kept in vector registers (constant offsets in the ds instructions,
    // one v_add per pair of rotations) instead of clamped indices rebuilt from the scalar counter for every access; the operand
    // fetched ahead of the LAST rotation may lie one or two entries below s (nact = 0): in the wave's LDS (s is never the first
    // array of a view; an out-of-range LDS read returns zero anyway), and never used.
    double cur = s[nu - 1];
    double pa = s[nu - 2], pb;
    int c = nu - 1;
    typedef __attribute__((address_space(3))) double lds_f64;  // s (R's working column) and the scratch vectors are LDS in every view that sweeps here
    const lds_f64 *sp = (const lds_f64 *)(s + (nu - 4));      // sp[1] = s[c - 2], sp[0] = s[c - 3]
    lds_f64 *cp = (lds_f64 *)(chain + (nu - 3));              // cp[1] = chain[c - 1], cp[0] = chain[c - 2]
    asm volatile("" : "+v"(sp), "+v"(cp));
    // pairs in a counted loop with ONE exit (the two-exit form cost a flag and a trampoline block per rotation), then the odd one
    const int rots = c - nact;                              // >= 1
    if (WG_UBOOL(cur != 0.0)) {
      // a norm is at least its larger operand: once cur is non-zero it stays non-zero, no rotation is skipped and the
      // "cur == 0 ? p : norm" select of the general form below always takes the norm
      for (int k = rots >> 1; k > 0; --k) {
        pb = sp[1];                                         // operand of the next rotation, off the chain
        cur = givens_norm_fast(pa, cur); cp[1] = cur;
        pa = sp[0];
        cur = givens_norm_fast(pb, cur); cp[0] = cur;
        sp -= 2; cp -= 2;
      }
      if (rots & 1) { cur = givens_norm_fast(pa, cur); cp[1] = cur; }
    } else {
      for (int k = rots >> 1; k > 0; --k) {
        pb = sp[1];
        { const double nrmc = givens_norm_fast(pa, cur); cur = (cur == 0.0) ? pa : nrmc; cp[1] = cur; }
        pa = sp[0];
        { const double nrmc = givens_norm_fast(pb, cur); cur = (cur == 0.0) ? pb : nrmc; cp[0] = cur; }
        sp -= 2; cp -= 2;
      }
      if (rots & 1) { const double nrmc = givens_norm_fast(pa, cur); cur = (cur == 0.0) ? pa : nrmc; cp[1] = cur; }
    }
  } else {
    double cur = s[nu - 1];
    double p = s[nu - 2];
    for (int c = nu - 1; c > nact; --c) {
      const int nx = (c - 2 >= 0) ? c - 2 : 0;
      const double p_next = s[nx];                          // operand of the next rotation, off the chain
      const double nrmc = givens_norm(p, cur);
      cur = (cur == 0.0) ? p : nrmc;
      chain[c - 1] = cur;
      p = p_next;
    }
  }
  WG_WSYNC();
  PT_SW(0);
  double myP = 0.0, myQ = 0.0, myN = 0.0;
  {
    const bool mine = lane > nact && lane < nu;
    const int c = mine ? lane : nu - 1;
    myP = s[c - 1];
    myQ = (c == nu - 1) ? s[nu - 1] : chain[c];
    const double chl = chain[c - 1];
    myN = (myQ == 0.0) ? 0.0 : chl;
  }
  WG_WSYNC();
  double *gab = q.sc0;                                      // pairs {ga, gb}; sc0 and sc1 are adjacent (2n doubles)
  bool any_skip;
  {
    const bool mine = lane > nact && lane < nu;
    const bool rot = mine && myN != 0.0;
    const double den = rot ? myN : 1.0;
    // ga == 2 marks a skipped rotation (q was 0; a rotation's |ga| = |p| / norm <= 1).  NOT gb == 0: a denormal q under a large p
    // gives gb = q / norm = 0 by underflow and ga = -1 for p < 0 -- a rotation the reference carries out (both columns change sign)
    const double ga = rot ? myP / den : 2.0;
    const double gb = rot ? myQ / den : 0.0;
    // a skipped rotation is rare: when the sweep has none -- one ballot -- phase 3 runs without the selects
    any_skip = __ballot(mine && !rot) != 0ull;
    const int cl = mine ? lane : nu - 1;                    // lanes without a rotation shadow lane nu-1 ... with ITS values
    const double ga_l = rl(ga, nu - 1), gb_l = rl(gb, nu - 1);
    const double ga_w = mine ? ga : ga_l, gb_w = mine ? gb : gb_l;
    gab[2 * cl] = ga_w; gab[2 * cl + 1] = gb_w;
    if (rot) s[lane - 1] = myN;
  }
  WG_WSYNC();
  PT_SW(1);
  {
    // phase 3: lane i carries row i of Z through the rotations.  Operands of rotation c -- Z(i, c-1) and the pair
    // (ga, gb) -- are fetched three rotations ahead into one of three register sets used in turn (an unroll by three, so
    // that handing a set on is a renaming, not a move).
    const int i = lane < n ? lane : n - 1;                  // surplus lanes shadow row n-1
    const int ldz = q.ldz;
    double *zp = q.Z + i + (nu - 1) * ldz;                  // Z(i, c)
    double carry = zp[0];
    struct Op { double zl, ga, gb; };
    auto fetch = [&](int c) -> Op {                         // operands of rotation c (clamped: unused past the end)
      const int cc = c > nact ? c : nact + 1;
      Op o; o.zl = q.Z[i + (cc - 1) * ldz]; o.ga = gab[2 * cc]; o.gb = gab[2 * cc + 1];
      return o;
    };
    auto rotate = [&](const Op &o) {
      const bool skip = (o.ga == 2.0);
      const double t_r = o.ga * o.zl + o.gb * carry;
      const double z_r = o.ga * carry - o.gb * o.zl;
      zp[0] = skip ? carry : z_r;
      carry = skip ? o.zl : t_r;
      zp -= ldz;
    };
    auto rotate_all = [&](const Op &o) {
      const double t_r = o.ga * o.zl + o.gb * carry;
      zp[0] = o.ga * carry - o.gb * o.zl;
      carry = t_r;
      zp -= ldz;
    };
    Op s0 = fetch(nu - 1), s1 = fetch(nu - 2), s2 = fetch(nu - 3);
    int c = nu - 1;
    if (any_skip) {
      for (;;) {
        { const Op nx = fetch(c - 3); rotate(s0); s0 = nx; }
        if (--c <= nact) break;
        { const Op nx = fetch(c - 3); rotate(s1); s1 = nx; }
        if (--c <= nact) break;
        { const Op nx = fetch(c - 3); rotate(s2); s2 = nx; }
        if (--c <= nact) break;
      }
    } else {
      // Groups of three while every rotation fetched ahead exists (c - 5 > nact): ONE exit test per three rotations, the
      // operands through two walking pointers with constant offsets (no clamp, no index arithmetic), the three register sets
      // handed on where the loop closes -- the stepping loop below (clamped fetches, a test per rotation, and the moves the
      // compiler needs to make its three exits agree) took 23 instructions per rotation for 6 of arithmetic and 3 of LDS
      if constexpr (kLdzC > 0) {
        // Z in LDS with a constant leading dimension: every operand and every store of a group through THREE address registers
        // (the group's lowest column of operands, of pairs, of stores) with constant offsets -- a v_add, or a scalar add and a move
        // into a vector register, per access otherwise
        typedef __attribute__((address_space(3))) double lds_f64;
        constexpr int L = kLdzC;
        while (c - 8 > nact) {                              // six at a time: the two register sets swap roles, no moves
          const lds_f64 *zlo = (const lds_f64 *)(q.Z + i + (c - 9) * L);      // Z(i, c - 9): operand of rotation c - 8
          const lds_f64 *glo = (const lds_f64 *)(gab + 2 * (c - 8));
          lds_f64 *zst = (lds_f64 *)(q.Z + i + (c - 5) * L);                   // Z(i, c - 5): the group's last store
          asm volatile("" : "+v"(zlo), "+v"(glo), "+v"(zst));
          Op n0, n1, n2;
          n0.zl = zlo[5 * L]; n0.ga = glo[10]; n0.gb = glo[11];
          n1.zl = zlo[4 * L]; n1.ga = glo[8];  n1.gb = glo[9];
          n2.zl = zlo[3 * L]; n2.ga = glo[6];  n2.gb = glo[7];
          { const double t = s0.ga * s0.zl + s0.gb * carry; zst[5 * L] = s0.ga * carry - s0.gb * s0.zl; carry = t; }
          { const double t = s1.ga * s1.zl + s1.gb * carry; zst[4 * L] = s1.ga * carry - s1.gb * s1.zl; carry = t; }
          { const double t = s2.ga * s2.zl + s2.gb * carry; zst[3 * L] = s2.ga * carry - s2.gb * s2.zl; carry = t; }
          s0.zl = zlo[2 * L]; s0.ga = glo[4]; s0.gb = glo[5];
          s1.zl = zlo[L];     s1.ga = glo[2]; s1.gb = glo[3];
          s2.zl = zlo[0];     s2.ga = glo[0]; s2.gb = glo[1];
          { const double t = n0.ga * n0.zl + n0.gb * carry; zst[2 * L] = n0.ga * carry - n0.gb * n0.zl; carry = t; }
          { const double t = n1.ga * n1.zl + n1.gb * carry; zst[L] = n1.ga * carry - n1.gb * n1.zl; carry = t; }
          { const double t = n2.ga * n2.zl + n2.gb * carry; zst[0] = n2.ga * carry - n2.gb * n2.zl; carry = t; }
          c -= 6;
        }
        while (c - 5 > nact) {
          const lds_f64 *zlo = (const lds_f64 *)(q.Z + i + (c - 6) * L);      // Z(i, c - 6): operand of rotation c - 5
          const lds_f64 *glo = (const lds_f64 *)(gab + 2 * (c - 5));
          lds_f64 *zst = (lds_f64 *)(q.Z + i + (c - 2) * L);
          asm volatile("" : "+v"(zlo), "+v"(glo), "+v"(zst));
          Op n0, n1, n2;
          n0.zl = zlo[2 * L]; n0.ga = glo[4]; n0.gb = glo[5];
          n1.zl = zlo[L];     n1.ga = glo[2]; n1.gb = glo[3];
          n2.zl = zlo[0];     n2.ga = glo[0]; n2.gb = glo[1];
          { const double t = s0.ga * s0.zl + s0.gb * carry; zst[2 * L] = s0.ga * carry - s0.gb * s0.zl; carry = t; }
          { const double t = s1.ga * s1.zl + s1.gb * carry; zst[L] = s1.ga * carry - s1.gb * s1.zl; carry = t; }
          { const double t = s2.ga * s2.zl + s2.gb * carry; zst[0] = s2.ga * carry - s2.gb * s2.zl; carry = t; }
          s0 = n0; s1 = n1; s2 = n2;
          c -= 3;
        }
        zp = q.Z + i + c * ldz;                              // where the stepping loop goes on
      } else {
        const double *zq = q.Z + i + (c - 4) * ldz;          // Z(i, c - 4): operand of rotation c - 3
        const double *gq = gab + 2 * (c - 3);
        while (c - 8 > nact) {                              // six at a time: the two register sets swap roles, no moves
          Op n0, n1, n2;
          n0.zl = zq[0];        n0.ga = gq[0];  n0.gb = gq[1];
          n1.zl = zq[-ldz];     n1.ga = gq[-2]; n1.gb = gq[-1];
          n2.zl = zq[-2 * ldz]; n2.ga = gq[-4]; n2.gb = gq[-3];
          rotate_all(s0); rotate_all(s1); rotate_all(s2);
          s0.zl = zq[-3 * ldz]; s0.ga = gq[-6];  s0.gb = gq[-5];
          s1.zl = zq[-4 * ldz]; s1.ga = gq[-8];  s1.gb = gq[-7];
          s2.zl = zq[-5 * ldz]; s2.ga = gq[-10]; s2.gb = gq[-9];
          rotate_all(n0); rotate_all(n1); rotate_all(n2);
          zq -= 6 * ldz; gq -= 12; c -= 6;
        }
        while (c - 5 > nact) {
          Op n0, n1, n2;
          n0.zl = zq[0];        n0.ga = gq[0];  n0.gb = gq[1];
          n1.zl = zq[-ldz];     n1.ga = gq[-2]; n1.gb = gq[-1];
          n2.zl = zq[-2 * ldz]; n2.ga = gq[-4]; n2.gb = gq[-3];
          rotate_all(s0); rotate_all(s1); rotate_all(s2);
          s0 = n0; s1 = n1; s2 = n2;
          zq -= 3 * ldz; gq -= 6; c -= 3;
        }
      }
      if constexpr (kLdzC > 0) {
        // the last one to five rotations, straight-line per count: columns relative to nact through two address registers with
        // constant offsets, the operands of the first three rotations are in s0 / s1 / s2 already, the others are requested
        // together before the first rotation; ends with Z(i, nact) = carry
        typedef __attribute__((address_space(3))) double lds_f64;
        constexpr int L = kLdzC;
        lds_f64 *zb = (lds_f64 *)(q.Z + i + nact * L);        // Z(i, nact)
        const lds_f64 *gb_ = (const lds_f64 *)(gab + 2 * nact);
        asm volatile("" : "+v"(zb), "+v"(gb_));
        auto rot = [&](const Op &o, int k) {                  // rotation nact + k: stores Z(i, nact + k)
          const double t = o.ga * o.zl + o.gb * carry;
          zb[k * L] = o.ga * carry - o.gb * o.zl;
          carry = t;
        };
        auto ld = [&](int k) -> Op { Op o; o.zl = zb[(k - 1) * L]; o.ga = gb_[2 * k]; o.gb = gb_[2 * k + 1]; return o; };
        switch (c - nact) {
          case 5: { const Op o2 = ld(2), o1 = ld(1); rot(s0, 5); rot(s1, 4); rot(s2, 3); rot(o2, 2); rot(o1, 1); break; }
          case 4: { const Op o1 = ld(1); rot(s0, 4); rot(s1, 3); rot(s2, 2); rot(o1, 1); break; }
          case 3: rot(s0, 3); rot(s1, 2); rot(s2, 1); break;
          case 2: rot(s0, 2); rot(s1, 1); break;
          default: rot(s0, 1); break;
        }
        zb[0] = carry;                                        // Z(i, nact)
      } else {
      for (;;) {                                            // the last (at most five) rotations
        { const Op nx = fetch(c - 3); rotate_all(s0); s0 = nx; }
        if (--c <= nact) break;
        { const Op nx = fetch(c - 3); rotate_all(s1); s1 = nx; }
        if (--c <= nact) break;
        { const Op nx = fetch(c - 3); rotate_all(s2); s2 = nx; }
        if (--c <= nact) break;
      }
      zp[0] = carry;                                         // Z(i, nact)
      }
    }
    if (any_skip) zp[0] = carry;                             // Z(i, nact)
  }
  WG_WSYNC();
  PT_SW(2);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Z in REGISTERS (the N = 32 element view's "Z on chip" form, docs/HISTORY.md 3.2): lane L carries row L of Z in z0[], every access with a
// compile-time column index (fully unrolled loops with wave-uniform predicates), so the array lives in the register file (144
// of the 256 registers a lane has at one wave per SIMD).  The rows beyond the 64th (n <= 72: at most eight) live in LDS (zt, 8 x
// NMAX, 4.6 KB): every lane works on tail row 64 + (L & 7) -- eight lanes compute the same values and store them to the same
// place, no lane-dependent branch.  Z never moves through global memory inside the active-set loop: the sweep and the deletions
// rotate registers (and the LDS tail), x and the dependence sums read the column the last sweep left, and the one operation that
// goes against the layout -- Z^T a, a sum over ROWS in row order for every column -- passes the per-lane products through an LDS
// tile of kZrTile columns and lets the column's owner add them in row order: the same additions in the same order as the column
// walk it replaces.
template <int NMAX>
struct ZRegs {
  double z0[NMAX];
  double *zt;                                               // LDS: rows 64 .. 71, [row - 64][NMAX]
  double c0, c1;                                            // column nact as the last sweep left it (rows L, 64 + (L & 7))
};
struct NoZRegs {};
#ifndef WG_ZR_TILE
#define WG_ZR_TILE 24
#endif
#ifndef WG_ZR_WPS
#define WG_ZR_WPS 1                                         // waves per SIMD the register-Z kernels are compiled for (experiment: 2)
#endif
constexpr int kZrTile = WG_ZR_TILE;                         // columns of Z^T a per pass through the tile (a divisor of NMAX)
constexpr int kZrTs = 73;                                   // tile row stride (doubles): [column][row]
constexpr int kZrTail = 8;                                  // rows kept in LDS

// f(integral_constant<K * STEP>) for the K with c0 == K * STEP (wave-uniform c0)
template <int NP, int STEP, class F>
__device__ __forceinline__ void zr_dispatch(int c0, F &&f) {
  if constexpr (NP == 1) f(std::integral_constant<int, 0>{});
  else {
    constexpr int kH = NP / 2;
    if (c0 < kH * STEP) zr_dispatch<kH, STEP>(c0, f);
    else zr_dispatch<NP - kH, STEP>(c0 - kH * STEP, [&](auto b) { f(std::integral_constant<int, decltype(b)::value + kH * STEP>{}); });
  }
}
// rows of Z from the global slot (where factor() / the generic inverse leave them) into the registers / the LDS tail
template <int NMAX>
__device__ __forceinline__ void zr_load(const QlView &q, ZRegs<NMAX> &zr, int lane) {
  const int n = q.n;
  const double *r0 = q.Z + lane;
#pragma unroll
  for (int j = 0; j < NMAX; ++j) {
    const int jc = j < n ? j : n - 1;                       // columns past n: a valid address, a value nobody uses
    zr.z0[j] = r0[(size_t)jc * q.ldz];
  }
  for (int e = lane; e < kZrTail * NMAX; e += 64) {
    const int r = 64 + e / NMAX, j = e % NMAX;
    const bool in = r < n && j < n;
    const double v = q.Z[(size_t)(in ? r : 0) + (size_t)(in ? j : 0) * q.ldz];
    zr.zt[e] = in ? v : 0.0;
  }
  zr.c0 = 0.0; zr.c1 = 0.0;
  WG_WSYNC();
}
// column j (wave-uniform, < n) of the rows
template <int NMAX>
__device__ __forceinline__ void zr_col(const ZRegs<NMAX> &zr, int j, int lane, double &c0, double &c1) {
  c0 = 0.0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k)
    if (k == j) c0 = zr.z0[k];
  c1 = zr.zt[(lane & 7) * NMAX + j];
}
// s[j] = sum_i Z(i, j) ww[i], i ascending from +0.0 (qld.cpp:2071-2085): products per lane, sums per column through the tile
template <int NMAX>
__device__ __forceinline__ void zr_zt_times_ww(const QlView &q, const ZRegs<NMAX> &zr, double *s, int lane) {
  const int n = q.n;
  const int tr = 64 + (lane & 7);                           // the lane's tail row (rows >= n: products nobody adds)
  const double w0 = q.ww[lane], w1 = q.ww[tr < n ? tr : n - 1];
  double *tile = q.ztile;
  const double *ztr = zr.zt + (lane & 7) * NMAX;
#pragma unroll 1
  for (int c0 = 0; c0 < NMAX; c0 += kZrTile) {
    // which slice of the columns: a wave-uniform dispatch over compile-time register indices
    auto put = [&](auto base) {
      constexpr int kB = decltype(base)::value;
      double tv[kZrTile];
#pragma unroll
      for (int t = 0; t < kZrTile; ++t) tv[t] = ztr[kB + t];
#pragma unroll
      for (int t = 0; t < kZrTile; ++t) tile[t * kZrTs + lane] = zr.z0[kB + t] * w0;
#pragma unroll
      for (int t = 0; t < kZrTile; ++t) tile[t * kZrTs + tr] = tv[t] * w1;
    };
    static_assert(NMAX % kZrTile == 0, "whole passes");
    zr_dispatch<NMAX / kZrTile, kZrTile>(c0, put);
    WG_WSYNC();
    {
      const int t = lane < kZrTile ? lane : kZrTile - 1;    // surplus lanes shadow the last column of the pass
      const double *col = tile + t * kZrTs;
      double acc = 0.0;
      int i = 0;
      for (; i + 8 <= n; i += 8) {
        double v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = col[i + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += v[e];
      }
      for (; i < n; ++i) acc += col[i];
      const int j = c0 + t;
      if (lane < kZrTile && j < n) s[j] = acc;
    }
    WG_WSYNC();
  }
}
// r0 = sum_{j0 <= j < j1} Z(L, j) s[j] (j ascending from +0.0), r1 the same for row 64 + (L & 7)
template <int NMAX>
__device__ __forceinline__ void zr_rows_times(const QlView &q, const ZRegs<NMAX> &zr, const double *s, int j0, int j1, int lane, double &r0,
                                              double &r1) {
  const int n = q.n;
  double a0 = 0.0, a1 = 0.0;
  const double *ztr = zr.zt + (lane & 7) * NMAX;
#pragma unroll
  for (int jc = 0; jc < NMAX; jc += 8) {
    double sv[8], tv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sv[e] = s[jc + e < n ? jc + e : n - 1]; tv[e] = ztr[jc + e]; }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (jc + e >= j0 && jc + e < j1) { a0 += zr.z0[jc + e] * sv[e]; a1 += tv[e] * sv[e]; }
  }
  r0 = a0; r1 = a1;
}
// Givens sweep (qld.cpp:1992-2030) on register rows.  Phase 1 (the chain of norms) as in sweep(); phase 2: lane c & 63 turns
// its rotation into (ga, gb) and KEEPS the pair (ga == 2 marks a skipped rotation); phase 3: every lane carries its rows through
// the rotations, the coefficients of rotation c read from lane c's registers (v_readlane with a compile-time lane): no global
// memory access at all in the rotation loop.
template <int NMAX>
__device__ __forceinline__ void zr_sweep(const QlView &q, ZRegs<NMAX> &zr, double *s, int nu, int nact, int lane) {
  if (nu - 1 <= nact) { zr_col(zr, nact, lane, zr.c0, zr.c1); return; }
  double *chain = q.sc3;
  if (sweep_range_ok(s, nact, nu, lane)) {
    double cur = s[nu - 1];
    double pa = s[nu - 2], pb;
    int c = nu - 1;
    if (WG_UBOOL(cur != 0.0)) {
      for (;;) {
        pb = s[(c - 2 >= 0) ? c - 2 : 0];
        cur = givens_norm_fast(pa, cur); chain[c - 1] = cur;
        if (--c <= nact) break;
        pa = s[(c - 2 >= 0) ? c - 2 : 0];
        cur = givens_norm_fast(pb, cur); chain[c - 1] = cur;
        if (--c <= nact) break;
      }
    } else
    for (;;) {
      pb = s[(c - 2 >= 0) ? c - 2 : 0];
      { const double nrmc = givens_norm_fast(pa, cur); cur = (cur == 0.0) ? pa : nrmc; chain[c - 1] = cur; }
      if (--c <= nact) break;
      pa = s[(c - 2 >= 0) ? c - 2 : 0];
      { const double nrmc = givens_norm_fast(pb, cur); cur = (cur == 0.0) ? pb : nrmc; chain[c - 1] = cur; }
      if (--c <= nact) break;
    }
  } else {
    double cur = s[nu - 1];
    double p = s[nu - 2];
    for (int c = nu - 1; c > nact; --c) {
      const double p_next = s[(c - 2 >= 0) ? c - 2 : 0];
      const double nrmc = givens_norm(p, cur);
      cur = (cur == 0.0) ? p : nrmc;
      chain[c - 1] = cur;
      p = p_next;
    }
  }
  WG_WSYNC();
  // phase 2: rotation c lives in lane c & 63 (set 0: c < 64, set 1: c >= 64)
  double ga0 = 1.0, gb0 = 0.0, ga1 = 1.0, gb1 = 0.0;
  {
    const int ca = lane, cb = lane + 64;
    const bool ma = ca > nact && ca < nu, mb = cb > nact && cb < nu;
    const int cca = ma ? ca : nu - 1, ccb = mb ? cb : nu - 1;
    const double Pa = s[cca - 1], Qa = (cca == nu - 1) ? s[nu - 1] : chain[cca], Na = (Qa == 0.0) ? 0.0 : chain[cca - 1];
    const double Pb = s[ccb - 1], Qb = (ccb == nu - 1) ? s[nu - 1] : chain[ccb], Nb = (Qb == 0.0) ? 0.0 : chain[ccb - 1];
    const bool ra = ma && Na != 0.0, rb = mb && Nb != 0.0;
    const double da = ra ? Na : 1.0, db = rb ? Nb : 1.0;
    ga0 = ra ? Pa / da : 2.0; gb0 = ra ? Qa / da : 0.0;   // ga == 2: a skipped rotation (see sweep_flat)
    ga1 = rb ? Pb / db : 2.0; gb1 = rb ? Qb / db : 0.0;
    WG_WSYNC();                                             // every lane has read s[] before any lane rewrites it
    if (ra) s[ca - 1] = Na;
    if (rb) s[cb - 1] = Nb;
  }
  WG_WSYNC();
  // phase 3
  double carry0 = 0.0, carry1 = 0.0;
  double *ztr = zr.zt + (lane & 7) * NMAX;
#pragma unroll
  for (int c = NMAX - 1; c >= 1; --c) {
    if (c == nu - 1) { carry0 = zr.z0[c]; carry1 = ztr[c]; }
    if (c <= nu - 1 && c > nact) {
      const double ga = (c < 64) ? rl(ga0, c & 63) : rl(ga1, c & 63);
      const double gb = (c < 64) ? rl(gb0, c & 63) : rl(gb1, c & 63);
      const double zl0 = zr.z0[c - 1], zl1 = ztr[c - 1];
      if (ga == 2.0) {                                      // a skipped rotation (q was 0): wave-uniform
        zr.z0[c] = carry0; ztr[c] = carry1;
        carry0 = zl0; carry1 = zl1;
      } else {
        const double t0 = ga * zl0 + gb * carry0, t1 = ga * zl1 + gb * carry1;
        zr.z0[c] = ga * carry0 - gb * zl0; ztr[c] = ga * carry1 - gb * zl1;
        carry0 = t0; carry1 = t1;
      }
      if (c - 1 == nact) { zr.z0[c - 1] = carry0; ztr[c - 1] = carry1; }
    }
  }
  zr.c0 = carry0; zr.c1 = carry1;
  WG_WSYNC();
}
// the Z part of a constraint deletion (qld.cpp:1903-1982): columns k, k + 1 rotated for k = kdrop .. nact - 2 with the
// coefficients the R pass left in sc0 (ga) / sc1 (gb)
template <int NMAX>
__device__ __forceinline__ void zr_drop_rotations(const QlView &q, ZRegs<NMAX> &zr, int kdrop, int nact_old, int lane) {
  double *ztr = zr.zt + (lane & 7) * NMAX;
#pragma unroll
  for (int k = 0; k < NMAX - 1; ++k) {
    if (k >= kdrop && k < nact_old - 1) {
      const double ga = q.sc0[k], gb = q.sc1[k];
      const double a0 = zr.z0[k], b0 = zr.z0[k + 1], a1 = ztr[k], b1 = ztr[k + 1];
      zr.z0[k + 1] = ga * b0 - gb * a0; zr.z0[k] = ga * a0 + gb * b0;
      ztr[k + 1] = ga * b1 - gb * a1; ztr[k] = ga * a1 + gb * b1;
    }
  }
  WG_WSYNC();
}
// s[i] = sg * Z(row, i), i < n (the normal of a bound constraint, qld.cpp:1461-1470): the lane that owns the row writes it out
template <int NMAX>
__device__ __forceinline__ void zr_row_to(const QlView &q, const ZRegs<NMAX> &zr, int row, double sg, double *s, int lane) {
  const int n = q.n;
  if (row < 64) {
    if (lane == row) {
#pragma unroll
      for (int j = 0; j < NMAX; ++j)
        if (j < n) { const double z = zr.z0[j]; s[j] = (sg > 0.0) ? z : -z; }
    }
  } else {
    for (int j = lane; j < n; j += 64) { const double z = zr.zt[(row - 64) * NMAX + j]; s[j] = (sg > 0.0) ? z : -z; }
  }
}

// qld.cpp:1861-1889.  Returns kdrop (0-based) or -1; ratio updated when found.
// ---- the two SELECTIONS of an iteration once values stop being ordinary numbers (round 5) ------------------------------------
// The violation scan and the ratio test pick a row by a running comparison -- "skip unless strictly better than the best so far" --
// which the lane-parallel forms below replace by per-lane candidates and a wave arg-max (first index among equals).  The two agree
// while every compared value is an ordinary number.  Once the iterate holds a NaN they do not: the reference's `if (sumx <= cvmax)
// goto skip` does NOT skip a NaN, and with cvmax = NaN it skips nothing any more -- the LAST row that passes its other tests wins,
// which no arg-max reproduces (found on two of 2304 random Herdt-shaped QPs: the reference loops to maxit and reports ifail = 1,
// the arg-max form "converged" with ifail = 0 and a NaN solution).  So: when x (or the ratio test's operands) is not a number of
// sane size -- one compare per lane and one ballot per iteration -- the wave takes the forms below instead: scan_nan_exact
// (lane-parallel, with the reference's dense row sums and its NaN semantics) and the ratio test as the reference's own loop.
// Such a solve is lost and only has to end the way the reference's does -- but it runs maxit = 40 (m + n) iterations on the
// way, and a fleet waits for its slowest gait: the scan is lane-parallel for that reason (a first, fully serial form took 1 ms
// per iteration in the compact view, 4.5 s per lost tick).
#ifndef WG_TICK_NAN_EXACT
#define WG_TICK_NAN_EXACT 1                                // 0: the tick's views without these forms (A/B of their cost); see mpc_tick
#endif
#ifndef WG_NAN_REGIME
#define WG_NAN_REGIME 1                                    // 0: experiment builds without the NaN-regime tests (A/B of their cost)
#endif
// |v| < 2^332 (8.7e99): false for NaN, infinities and overflow-bound values.  On the exponent field of the high word -- an integer
// mask and a compare against a 32-bit literal: a 64-bit floating-point literal would be hoisted into a scalar register pair and
// kept alive (or spilled) across the active-set loop, which is what wg_kconst exists to avoid
__device__ __forceinline__ bool wg_sane(double v) { return ((unsigned)__double2hiint(v) & 0x7fffffffu) < 0x54b00000u; }

// qld.cpp:1255-1331 for an iterate that may hold NaNs and infinities, lane-parallel.  What the reference's loop does, restated:
// a row (or bound) is a CANDIDATE when it passes every test that does not involve cvmax (weight, significance of the residual;
// `sum != 0` for a bound) -- written below as the reference's own comparisons, negated, so that a NaN operand passes exactly
// where it passes there.  Candidates are visited in order (rows 1..m, then the bounds by variable) and taken unless
// `value <= cvmax`; a taken candidate's value becomes cvmax.  While every value is a number that is the first strict maximum
// above 0.  A candidate whose value is a NaN is taken (NaN <= cvmax is false) and leaves cvmax = NaN, so the NEXT candidate is
// taken whatever its value -- and if that value is a number, the running maximum starts again from it.  Hence: with L the last
// candidate whose value is a NaN, the winner is the first maximum among the candidates BEHIND L (no threshold: the first of them
// is always taken), or L itself when none follows.  Row sums are the reference's DENSE sums -- every x_i times every
// coefficient, the structural zeros included: 0 * inf and 0 * NaN are NaN there, which a view that skips its zeros would not
// produce -- one row per lane, terms in index order.
template <class P>
__device__ __forceinline__ void scan_nan_exact(const QlView &q, const P &prob, double onha, double &cvmax, double &res,
                                               double &wsel, int &knext, int lane) {
  const int n = q.n, m = q.m, me = q.me, mn = q.mn;
  struct Row { bool cand; double v, r, w; int code; };
  auto eval = [&](int pos) -> Row {
    Row o;
    if (pos < m) {
      const int k = pos;
      o.w = q.wa[k];
      const double bk = q.b[k];
      double sum = -bk, temp = fabs(bk);
      for (int i = 0; i < n; ++i) {
        double aki;
        if constexpr (P::kCompact) aki = prob.A_own(k, i); else aki = Am(k, i);
        const double t = q.x[i] * aki; sum += t; temp += fabs(t);
      }
      o.v = -sum * o.w;
      if (k + 1 <= me) o.v = fabs(o.v);
      const double tempa = temp + fabs(sum);
      const double temp2 = temp + onha * fabs(sum);
      o.cand = !(o.w <= 0.0) && !(tempa <= temp) && !(temp2 <= tempa);
      o.r = sum; o.code = k + 1;
    } else {
      const int k = pos - m;
      o.w = q.wa[m + k];
      const double xk = q.x[k], s1 = prob.xl(q, k) - xk;
      const bool upper = s1 < 0.0;
      o.v = upper ? xk - prob.xu(q, k) : s1;
      o.cand = !(o.w <= 0.0) && !(s1 == 0.0);
      o.r = -o.v; o.code = upper ? k + 1 + mn : k + 1 + m;
    }
    return o;
  };
  // the lane's j-th position, ascending in j: lane + 64 j -- or, in the compact view (whose rows' coefficients live in their
  // lanes' registers), CoP row lane + 1, foot-placement row 1 + 4N + lane, bound lane (row 0, all zeros, is never a
  // candidate).  -1: the lane has no j-th position
  auto position = [&](int j) -> int {
    if constexpr (P::kCompact) {
      if (j == 0) return (lane + 1 <= P::kCopRows && lane + 1 < m) ? lane + 1 : -1;
      if (j == 1) return (1 + P::kCopRows + lane < m) ? 1 + P::kCopRows + lane : -1;
      return lane < n ? m + lane : -1;
    } else {
      const int pos = lane + 64 * j;
      return pos < m + n ? pos : -1;
    }
  };
  const int slots = P::kCompact ? 3 : (m + n + 63) / 64;
  // pass 1: the all-numbers answer (running strict maximum above 0 = cvmax's start), and the lane's last NaN candidate
  double av = 0.0, ar = 0.0, aw = 0.0;
  int apos = -1, acode = 0;
  double nr = 0.0, nw = 0.0;
  int npos = -1, ncode = 0;
  for (int j = 0; j < slots; ++j) {
    const int pos = position(j);
    if (pos < 0) continue;
    const Row o = eval(pos);
    if (!o.cand) continue;
    if (o.v != o.v) { npos = pos; nr = o.r; nw = o.w; ncode = o.code; }
    else if (o.v > av) { av = o.v; ar = o.r; aw = o.w; apos = pos; acode = o.code; }
  }
  const int lastnan = uni(wave_max_int(npos));
  int src;
  if (lastnan >= 0) {
    // pass 2: the candidates behind the last NaN (numbers all of them): first maximum, the first one taken unconditionally
    bool any = false;
    av = 0.0; apos = -1;
    for (int j = 0; j < slots; ++j) {
      const int pos = position(j);
      if (pos <= lastnan) continue;                          // (covers pos == -1)
      const Row o = eval(pos);
      if (!o.cand) continue;
      if (!any || o.v > av) { av = o.v; ar = o.r; aw = o.w; apos = pos; acode = o.code; }
      any = true;
    }
    if (__ballot(any) == 0ull) {                             // none follows: the NaN candidate itself, cvmax = NaN
      src = __ffsll((long long)__ballot(npos == lastnan)) - 1;
      cvmax = __builtin_nan(""); res = rl(nr, src); wsel = rl(nw, src); knext = __builtin_amdgcn_readlane(ncode, src);
      return;
    }
  }
  double v = av;
  int kk = apos;
  wave_argmax_first(v, kk);
  kk = uni(kk);
  cvmax = 0.0;
  if (kk < 0) return;                                        // nothing above 0: knext / res / wsel stay what they were
  src = __ffsll((long long)__ballot(apos == kk)) - 1;
  cvmax = rl(av, src); res = rl(ar, src); wsel = rl(aw, src); knext = __builtin_amdgcn_readlane(acode, src);
}

// qld.cpp:1861-1889
__device__ __forceinline__ int pick_drop_serial_reference(const QlView &q, int nact, double res, double &ratio) {
  int kdrop = -1;
  for (int k = 0; k < nact; ++k) {
    if (q.iact[k] <= q.me) continue;
    const double w = q.ww[k];
    if (res * w >= 0.0) continue;
    const double temp = q.lam[k] / w;
    if (kdrop >= 0 && fabs(temp) >= fabs(ratio)) continue;
    kdrop = k; ratio = temp;
  }
  return kdrop;
}

template <bool kOnePass = false, bool kNan = false>       // kOnePass: nact <= 64 known at compile time (n <= 64); kNan: see above
__device__ __forceinline__ int pick_drop(const QlView &q, int nact, double res, double &ratio, int lane) {
  double best = 0.0, bestt = 0.0;
  int bidx = -1;
  // operands that are no ordinary numbers (tested on the values this form loads anyway, consumed after it: nothing waits for
  // the test): the reference's own loop decides then (see above)
  bool bad = !wg_sane(res);
  if constexpr (kOnePass) {                                 // one multiplier per lane: selects instead of a lane-dependent loop
    const bool in = lane < nact;
    const int kc = in ? lane : 0;
    const double w = q.ww[kc];
    const int ia = q.iact[kc];
    const bool cand = in & (ia > q.me) & !(res * w >= 0.0);
    const double lamv = q.lam[kc];
    const double temp = lamv / w;
    if constexpr (kNan) {
      const bool bw = !wg_sane(w), bl = !wg_sane(lamv);    // plain values: the || below have nothing to short-circuit
      bad = bad || (in && (bw || bl));
    }
    best = cand ? -fabs(temp) : 0.0; bestt = cand ? temp : 0.0; bidx = cand ? lane : -1;
  } else
  for (int k = lane; k < nact; k += 64) {
    const double w = q.ww[k], lamv = q.lam[k];
    if constexpr (kNan) {
      const bool bw = !wg_sane(w), bl = !wg_sane(lamv);
      bad = bad || bw || bl;
    }
    if (q.iact[k] <= q.me) continue;
    if (res * w >= 0.0) continue;
    double temp = lamv / w;
    double key = -fabs(temp);          // smaller |temp| wins, first index on ties
    if (bidx < 0 || key > best) { best = key; bestt = temp; bidx = k; }
  }
  if constexpr (kNan)
    if (__ballot(bad) != 0ull) return uni(pick_drop_serial_reference(q, nact, res, ratio));
  int idx = bidx;
  double v = best;
  wave_argmax_first(v, idx);
  idx = uni(idx);
  if (idx < 0) return -1;
  // fetch the winning lane's temp
  ratio = rl(bestt, idx & 63);
  return idx;
}

// qld.cpp:2039-2058 (lql): per-lane terms, then the sum in index order.
template <class P>
__device__ __forceinline__ double xmag_sum(const QlView &q, const P &prob, double vfact, int lane) {
  const int n = q.n;
  if constexpr (P::kNM > 0) {
    const int il = lane < n ? lane : n - 1;                 // surplus lanes compute lane n - 1's term and drop it
    const double xi = q.x[il];
    const double term = fabs(xi) * vfact * (fabs(q.d[il]) + fabs(prob.Gd(q, il) * xi));
    return ordered_sum_lds<P::kNM>(term, q.sc3, n, lane);
  }
  for (int i = lane; i < n; i += 64) {
    double xi = q.x[i];
    q.sc3[i] = fabs(xi) * vfact * (fabs(q.d[i]) + fabs(prob.Gd(q, i) * xi));
  }
  WG_WSYNC();
  double sum = 0.0;
  int i = 0;
  for (; i + 8 <= n; i += 8) {                              // loads in groups of eight ahead of the add chain
    double t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = q.sc3[i + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += t[e];
  }
  for (; i < n; ++i) sum += q.sc3[i];
  return sum;
}

// qld.cpp:1992-2030.  Three phases: (1) the chain of rotation norms; (2) ga/gb of every rotation, one lane
// each; (3) every lane carries its own row of Z through the whole rotation sequence.
// n <= 64: s[] and the rotation coefficients live in registers (lane c <-> column c) and are handed
// around with v_readlane, so the dependent chain of phase 1 contains no LDS access at all.
template <int GRP = 8, bool kWide = false>                 // kWide: 64 <= n <= 128 known at compile time (only that form is compiled)
__device__ __forceinline__ void sweep(const QlView &q, double *s, int nu, int nact, int lane) {
  const int n = q.n;
  if (nu - 1 <= nact) return;
  if (!kWide && n <= 64) {
    const double sreg = (lane < nu) ? s[lane] : 0.0;
    double myP = 0.0, myQ = 0.0, myN = 0.0;
    {
      double cur = rl(sreg, nu - 1);
      for (int c = nu - 1; c > nact; --c) {
        const double p = rl(sreg, c - 1);
        double nrm;
        if (cur == 0.0) { nrm = 0.0; cur = p; }
        else { nrm = givens_norm(p, cur); if (lane == c) { myP = p; myQ = cur; } cur = nrm; }
        if (lane == c) myN = nrm;
      }
    }
    double ga = 1.0, gb = 0.0;
    if (lane > nact && lane < nu && myN != 0.0) {
      ga = myP / myN;
      gb = myQ / myN;
      s[lane - 1] = myN;
    }
    {
      const int i = lane;
      const bool act = i < n;
      double carry = act ? Zm(i, nu - 1) : 0.0;
      double zl = act ? Zm(i, nu - 2) : 0.0;            // nu - 2 >= nact >= 0 here
      for (int c = nu - 1; c > nact; --c) {
        const double zn = (act && c - 2 >= nact) ? Zm(i, c - 2) : 0.0;   // prefetch for the next rotation
        const double nc = rl(myN, c);
        if (nc == 0.0) { if (act) Zm(i, c) = carry; carry = zl; }
        else {
          const double gac = rl(ga, c), gbc = rl(gb, c);
          const double t = gac * zl + gbc * carry;
          if (act) Zm(i, c) = gac * carry - gbc * zl;
          carry = t;
        }
        zl = zn;
      }
      if (act) Zm(i, nact) = carry;
    }
    WG_WSYNC();
    return;
  }
  // phase 1: the chain of norms, as in sweep_flat: ONE value per rotation is recorded (`cur` as it leaves rotation c, in
  // chain[c - 1]); p = s[c - 1] is fetched one rotation ahead, off the chain.  Phase 2 rebuilds each rotation from it:
  // sc0[c] = ga, sc1[c] = gb, sc2[c] = norm (0 marks "skipped").
  double *chain = q.sc3;
  WG_REP(3)
  if (sweep_range_ok(s, nact, nu, lane)) {                  // the usual case: the shorter norm, as in sweep_flat
    double cur = s[nu - 1];
    double pa = s[nu - 2], pb;
    int c = nu - 1;
    // operand and record through ONE index kept in a vector register (s may be LDS or, once a solve went on in the global slot,
    // global memory: the address space follows the caller, so an index, not a typed pointer): constant offsets in the loads and
    // stores, one v_add per pair of rotations instead of a clamp, a shift, an add and a move per access.  The operand fetched
    // ahead of the last rotation may lie one entry below s (nact = 0): inside the view's memory, never used.
    int iv = nu - 4;                                        // s[iv + 1] = s[c - 2], chain[iv + 2] = chain[c - 1]
    asm volatile("" : "+v"(iv));
    const int rots = c - nact;                              // >= 1: pairs in a counted loop with one exit, then the odd one (sweep_flat)
    if (WG_UBOOL(cur != 0.0)) {                             // cur stays non-zero: no select (see sweep_flat)
      for (int k = rots >> 1; k > 0; --k) {
        pb = s[iv + 1];
        cur = givens_norm_fast(pa, cur); chain[iv + 2] = cur;
        pa = s[iv];
        cur = givens_norm_fast(pb, cur); chain[iv + 1] = cur;
        iv -= 2;
      }
      if (rots & 1) { cur = givens_norm_fast(pa, cur); chain[iv + 2] = cur; }
    } else {
      for (int k = rots >> 1; k > 0; --k) {
        pb = s[iv + 1];
        { const double nrmc = givens_norm_fast(pa, cur); cur = (cur == 0.0) ? pa : nrmc; chain[iv + 2] = cur; }
        pa = s[iv];
        { const double nrmc = givens_norm_fast(pb, cur); cur = (cur == 0.0) ? pb : nrmc; chain[iv + 1] = cur; }
        iv -= 2;
      }
      if (rots & 1) { const double nrmc = givens_norm_fast(pa, cur); cur = (cur == 0.0) ? pa : nrmc; chain[iv + 2] = cur; }
    }
  } else {
    double cur = s[nu - 1];
    double p = s[nu - 2];
    for (int c = nu - 1; c > nact; --c) {
      const double p_next = s[(c - 2 >= 0) ? c - 2 : 0];
      const double nrmc = givens_norm(p, cur);
      cur = (cur == 0.0) ? p : nrmc;
      chain[c - 1] = cur;
      p = p_next;
    }
    WG_WSYNC();
  }
  bool any_skip = false;                                    // a skipped rotation (q == 0) is rare: phase 3 has a select-free form
  for (int c0 = nact + 1; c0 < nu; c0 += 64) {
    const int c = c0 + lane;
    const bool mine = c < nu;
    const int cc = mine ? c : nu - 1;
    const double P = s[cc - 1];
    const double Q = (cc == nu - 1) ? s[nu - 1] : chain[cc];
    const double Nn = (Q == 0.0) ? 0.0 : chain[cc - 1];
    any_skip = any_skip || (__ballot(mine && Nn == 0.0) != 0ull);
    WG_WSYNC();                                             // every lane has read s[] before any lane rewrites it
    if (mine) {
      q.sc2[c] = Nn;
      if (Nn != 0.0) {
        q.sc0[c] = P / Nn;   // ga
        q.sc1[c] = Q / Nn;   // gb
        s[c - 1] = Nn;
      }
    }
  }
  WG_WSYNC();
  if (kWide || n <= 128) {
    // phase 3 for 64 < n <= 128: two rows per lane in one pass.  Rotation c reads Z(i, c-1) BEFORE any rotation rewrites it,
    // so the row entries are independent of the carry chain: they are fetched a chunk of kSwC columns at a time, the next
    // chunk while the current one is rotated (two register sets, loop unrolled by two so that handing a set on is a renaming).
    // Everything inside a chunk is straight-line code -- no early exit between a load and its use, which is what lets the
    // compiler wait for exactly the loads it needs (with Z in global memory an exposed entry is an L2 round trip on the
    // carry chain; an earlier form with an exit test per rotation waited for ALL outstanding accesses at every step).
    // The first chunk takes the cnt % kSwC odd rotations; past-the-end addresses are clamped to column nact (loaded, unused).
    // Surplus lanes shadow their first row completely: same loads, same arithmetic, the same value stored to the same place.
    // The rotation coefficients of a chunk (LDS, broadcast reads) are fetched at the head of the chunk as well, so that no
    // step waits for an LDS round trip; when no rotation of the sweep is skipped (one ballot in phase 2) the steps run
    // without the selects.
    constexpr int kSwC = GRP;
    const int i0 = lane;
    // lanes without a second row all mirror row 64 when there is one (ONE address per load and per store -- the same value to the
    // same place as lane 0's second row -- instead of a second copy of the first set's 56); with n <= 64 they mirror their own first row
    const int i1 = lane + 64 < n ? lane + 64 : (n > 64 ? 64 : lane);
    const int ldz = q.ldz;
    double *z0 = q.Z + i0, *z1 = q.Z + i1;
    double carry0 = WG_ZLD(z0 + (nu - 1) * ldz), carry1 = WG_ZLD(z1 + (nu - 1) * ldz);
    struct Co { double ga[kSwC], gb[kSwC], nr[kSwC]; };
    auto coef = [&](Co &o, int c) {                          // coefficients of rotations c, c-1, .. (clamped: unused past the end)
#pragma unroll
      for (int k = 0; k < kSwC; ++k) {
        const int cc = (c - k) > nact ? (c - k) : nact + 1;
        o.ga[k] = q.sc0[cc]; o.gb[k] = q.sc1[cc]; o.nr[k] = q.sc2[cc];
      }
    };
    auto rows = [&](double (&p0)[kSwC], double (&p1)[kSwC], int c) {   // Z(i, c-1-k): the entries rotations c, c-1, .. read
#pragma unroll
      for (int k = 0; k < kSwC; ++k) {
        const int cc = (c - 1 - k) > nact ? (c - 1 - k) : nact;
        p0[k] = WG_ZLD(z0 + cc * ldz); p1[k] = WG_ZLD(z1 + cc * ldz);
      }
    };
    auto step = [&](auto may_skip, int c, double zl0, double zl1, double ga, double gb, double nrm) {
      const double t0 = ga * zl0 + gb * carry0, w0 = ga * carry0 - gb * zl0;
      const double t1 = ga * zl1 + gb * carry1, w1 = ga * carry1 - gb * zl1;
      if constexpr (decltype(may_skip)::value) {
        const bool skip = (nrm == 0.0);
        WG_ZST(z0 + c * ldz, skip ? carry0 : w0);
        WG_ZST(z1 + c * ldz, skip ? carry1 : w1);
        carry0 = skip ? zl0 : t0;
        carry1 = skip ? zl1 : t1;
      } else {
        WG_ZST(z0 + c * ldz, w0); WG_ZST(z1 + c * ldz, w1);
        carry0 = t0; carry1 = t1;
      }
    };
    auto run = [&](auto may_skip) {
      int c = nu - 1;                                        // the next rotation
      {
        const int rem = (nu - 1 - nact) % kSwC;
        if (rem) {
          double h0[kSwC], h1[kSwC];
          Co hc;
          rows(h0, h1, c); coef(hc, c);
#pragma unroll
          for (int k = 0; k < kSwC - 1; ++k)
            if (k < rem) step(may_skip, c - k, h0[k], h1[k], hc.ga[k], hc.gb[k], hc.nr[k]);
          c -= rem;
        }
      }
      if (c > nact) {                                        // a whole number of chunks is left
        double a0[kSwC], a1[kSwC], b0[kSwC], b1[kSwC];
        Co cc;                                               // one set: read at the head of its chunk (one LDS wait per chunk)
        rows(a0, a1, c);
        for (;;) {
          rows(b0, b1, c - kSwC); coef(cc, c);
#pragma unroll
          for (int k = 0; k < kSwC; ++k) step(may_skip, c - k, a0[k], a1[k], cc.ga[k], cc.gb[k], cc.nr[k]);
          c -= kSwC;
          if (c <= nact) break;
          rows(a0, a1, c - kSwC); coef(cc, c);
#pragma unroll
          for (int k = 0; k < kSwC; ++k) step(may_skip, c - k, b0[k], b1[k], cc.ga[k], cc.gb[k], cc.nr[k]);
          c -= kSwC;
          if (c <= nact) break;
        }
      }
    };
    if (any_skip) run(std::true_type{}); else run(std::false_type{});
    WG_ZST(z0 + nact * ldz, carry0);
    WG_ZST(z1 + nact * ldz, carry1);
    WG_WSYNC();
    return;
  }
  if constexpr (!kWide)
  for (int i = lane; i < n; i += 64) {
    double carry = Zm(i, nu - 1);
    for (int c = nu - 1; c > nact; --c) {
      if (q.sc2[c] == 0.0) { Zm(i, c) = carry; carry = Zm(i, c - 1); continue; }
      double ga = q.sc0[c], gb = q.sc1[c];
      double zl = Zm(i, c - 1);
      double t = ga * zl + gb * carry;
      Zm(i, c) = ga * carry - gb * zl;
      carry = t;
    }
    Zm(i, nact) = carry;
  }
  WG_WSYNC();
}

// qld.cpp:1903-1982.  nu = number of R columns taking part (nact, or nact+1
// when the S column rides along).  Returns the new nact.
// ZR = ZRegs: Z lives in registers -- the rotations are recorded (sc0 / sc1) while R is updated and applied to the register rows
// afterwards (the R updates never read Z: the same arithmetic, the Z part in one unrolled pass).
template <class ZR = NoZRegs>
__device__ __forceinline__ int drop_constraint(const QlView &q, int kdrop, int nu, int nact, int lane, ZR *zr = nullptr) {
  constexpr bool kRegs = !std::is_same<ZR, NoZRegs>::value;
  const int n = q.n;
  if (lane == 0) {
    int code = q.iact[kdrop];
    int ia = code - 1;
    if (code > q.mn) ia -= n;
    q.wa[ia] = -q.wa[ia];
  }
  WG_WSYNC();
  for (int k = kdrop; k < nact - 1; ++k) {
    double ga, gb, nrm;
    givens(Rp(k, k + 1), Rp(k + 1, k + 1), ga, gb, nrm);   // redundant on all lanes
    WG_WSYNC();
    for (int i = lane; i <= k; i += 64) {
      double t = Rp(i, k + 1);
      Rp(i, k + 1) = Rp(i, k);
      Rp(i, k) = t;
    }
    WG_WSYNC();
    if (lane == 0) { Rp(k + 1, k + 1) = 0.0; Rp(k, k) = nrm; }
    WG_WSYNC();
    for (int c = k + 1 + lane; c < nu; c += 64) {
      double rk = Rp(k, c), rk1 = Rp(k + 1, c);
      double t = ga * rk + gb * rk1;
      Rp(k + 1, c) = ga * rk1 - gb * rk;
      Rp(k, c) = t;
    }
    if constexpr (kRegs) { if (lane == 0) { q.sc0[k] = ga; q.sc1[k] = gb; } }
    else
    for (int i = lane; i < n; i += 64) {
      double zk = Zm(i, k), zk1 = Zm(i, k + 1);
      double t = ga * zk + gb * zk1;
      Zm(i, k + 1) = ga * zk1 - gb * zk;
      Zm(i, k) = t;
    }
    if (lane == 0) { q.iact[k] = q.iact[k + 1]; q.lam[k] = q.lam[k + 1]; }
    WG_WSYNC();
  }
  if constexpr (kRegs) zr_drop_rotations(q, *zr, kdrop, nact, lane);
  return nact - 1;
}

// qld.cpp:1547-1658
template <class P>
__device__ __forceinline__ bool independent_coordinate(const QlView &q, const P &prob, int knext, int nact, double vsmall, int lane) {
  const int n = q.n, m = q.m;
  int k1 = 0;
  if (knext > m) { k1 = knext - m; if (k1 > n) k1 -= n; }
  bool found = false;
  for (int i = 1 + lane; i <= n; i += 64) {
    double suma;
    if (knext <= m) suma = Am(knext - 1, i - 1);
    else { suma = 0.0; if (i == k1) suma = (knext > q.mn) ? -1.0 : 1.0; }
    double sumb = fabs(suma);
    WG_UNROLL
    for (int k = 0; k < nact; ++k) {
      int kk = q.iact[k];
      double temp;
      if (kk <= m) temp = q.ww[k] * Am(kk - 1, i - 1);
      else {
        kk -= m; temp = 0.0;
        if (kk == i) temp = q.ww[kk - 1];
        kk -= n;
        if (kk == i) temp = -q.ww[kk - 1];
      }
      suma -= temp;
      sumb += fabs(temp);
    }
    if (knext <= m && suma <= vsmall) continue;
    if (significant(sumb, fabs(suma))) found = true;
  }
  return __any(found) != 0;
}

// Cholesky factor of G (qld.cpp:859-890) and Z = R^-1 (:937-975) for n <= NM <= 64 with one COLUMN of R / one ROW of Z per
// lane, in registers, the loops over rows and columns unrolled (compile-time register indices): entry (i, j) of the factor is
// temp = G(i, j) - sum_{k < i} R(k, j) R(k, i), k ascending -- lane j holds R(., j), R(k, i) comes from lane i's registers
// (v_readlane, both indices known at compile time) -- then R(i, j) = temp / R(i, i); row i of the inverse is
// Z(i, c) = -(sum_{k < c} Z(i, k) R(k, c)) / R(c, c), k ascending, with the exact zeros Z(i, k < i) = +0.0 left in (their
// products are +-0.0 and the sum, started from +0.0, does not move).  The same operations in the same order as the loops they
// replace, without their two LDS hand-overs per row: 27 k cycles instead of 240 k at n = 36.
// Returns false -- R and Z untouched -- when a pivot falls below vsmall: the caller then takes the generic path, which finds
// the same pivot and applies ql0002's diagonal shift.
template <int NM, class P>
__device__ __forceinline__ bool chol_inverse_regs(const QlView &q, const P &prob, double vsmall, int lane) {
  const int n = q.n;
  const int j = lane < n ? lane : n - 1;                    // surplus lanes shadow the last column / row
  double r[NM];
#pragma unroll
  for (int k = 0; k < NM; ++k) r[k] = 0.0;
  bool ok = true;
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    if (i < n) {
      const int jj = j > i ? j : i;                         // finished columns (j < i) walk column i along: unused
      double temp = prob.G(q, i, jj);
#pragma unroll
      for (int k = 0; k < i; ++k) temp -= r[k] * rl(r[k], i);
      const double tpiv = rl(temp, i);
      ok = ok && !(tpiv < vsmall);
      const double rii = sqrt(tpiv);
      const double quo = temp / rii;
      r[i] = (j == i) ? rii : ((j > i) ? quo : r[i]);
    }
  }
  if (!WG_UBOOL(ok)) return false;
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < NM; ++k)
      if (k <= j) Rf(k, j) = r[k];
  }
  WG_WSYNC();
  // ---- Z = R^-1: lane i owns row i ----
  const int i = j;
  double z[NM];
#pragma unroll
  for (int k = 0; k < NM; ++k) z[k] = 0.0;
  {
    double rdiag = 0.0;
#pragma unroll
    for (int k = 0; k < NM; ++k) rdiag = (k == i) ? r[k] : rdiag;     // R(i, i): the lane's own diagonal
    const double zd = 1.0 / rdiag;
#pragma unroll
    for (int k = 0; k < NM; ++k) z[k] = (k == i) ? zd : z[k];
  }
#pragma unroll
  for (int c = 1; c < NM; ++c) {
    if (c < n) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < c; ++k) sum += z[k] * Rf(k, c);
      const double zc = -sum / Rf(c, c);
      z[c] = (i < c) ? zc : z[c];
    }
  }
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < NM; ++k)
      if (k < n) Zm(i, k) = z[k];
  }
  WG_WSYNC();
  return true;
}

// The solver.  Problem data must already be in LDS: G (copy of C, patched per
// qld.cpp:442-444), A, d, b (INNER sign: b = -b_user, qld.cpp:469-475), xl, xu.
// hist: optional global add(+code)/drop(-code) log written by lane 0.
#ifdef WG_BACKSUB_READLANE
#define WG_BACKSUB(q, s, nact, lane) backsub(q, s, nact, lane)
#else
// compact view (n <= 36): the LDS-pipelined form with two 48-entry buffers; other views whose four scratch vectors hold two
// 96-entry buffers (n >= 48) take it too while every multiplier has a lane of its own (nact <= 60), the generic forms otherwise
#define WG_BACKSUB(q, s, nact, lane)                                                                   \
  do {                                                                                                 \
    if constexpr (P::kNM > 0 && P::kNM + 12 <= 48) backsub_lds<48>(q, s, nact, lane, q.sc0);          \
    else if (P::kNM == 0 && (P::kWideN || q.n >= 48) && (nact) <= 60) backsub_lds<96>(q, s, nact, lane, q.sc0);     \
    else if (P::kNM == 0 && !P::kRowOps && q.n >= 24 && (nact) <= 36) backsub_lds<48>(q, s, nact, lane, q.sc0); \
    else backsub(q, s, nact, lane);                                                                    \
  } while (0)
#endif
// one row of Z per lane (n <= 64): the branch-free, prefetching form -- the compact view always, the dense view by size (the
// element view is built for n > 64: it keeps the one form it needs, its kernel is large enough as it is)
#define WG_SWEEP(q, s, nu, nact, lane) \
  do { if constexpr (kRegs) zr_sweep(q, *zr, s, nu, nact, lane); else \
       if (P::kNM > 0 || (!P::kRowOps && q.n <= 64)) sweep_flat<P::kFixedLdz>(q, s, nu, nact, lane PT_SW_ARG); else sweep<(P::kRowOps ? WG_ELEM_GRP : 8), P::kWideN>(q, s, nu, nact, lane); } while (0)

template <class P, class ZR = NoZRegs>
__device__ __forceinline__ QlResult ql_solve(const QlView &q, P &prob, double vsmall, int *hist, int hist_cap, QlResume *rs = nullptr,
                                             ZR *zr = nullptr) {
  constexpr bool kRegs = !std::is_same<ZR, NoZRegs>::value;   // Z in registers (ZRegs): see zr_* above
  // NaN iterates followed exactly (every policy of the shipped kernels; a policy may opt out for an A/B of what that costs)
  constexpr bool kNan = P::kNanExact && (WG_NAN_REGIME != 0);
  int lane = wg_lane();
  const int n = q.n, m = q.m, me = q.me, mn = q.mn;
  QlResult out;
  out.hist_len = 0;
  const bool resuming = rs != nullptr && rs->valid != 0;     // wave-uniform; a compile-time constant where rs is
  int nact = 0, info = 0, iterc = 1, itref = 0, iflag = 0;
  const int maxit = (m + n) * 40;                       // :459
  const double onha = 1.5, xmagr = .01, diagr = 2.0;
  const int ifinc = 3, kfinc = n > 10 ? n : 10;
  int jfinc = -kfinc;
  double xmag = 0.0, vfact = 1.0, res = 0.0, ratio = 0.0, diag = 0.0;
  double wsel = 1.0;                                        // wa[.] of the constraint knext, as the scan read it (positive)
  int knext = 0;
  const int s_tail = q.r_tail;                              // n (n + 1) / 2, or the working column of the last allowed nact
  double *s = q.R + s_tail;
  bool early_exit = false;
  bool cap_hit = false;
  // Has the iterate left the ordinary numbers (wave-uniform, sticky)?  x only changes in the residual refresh (once or twice per
  // solve: tested there with one compare per lane) and by `x += step z` (tested on the scalar step: free).  From then on the
  // violation scan is scan_nan_exact's -- the reference's dense row sums and its treatment of NaN values, exact for any x
  bool x_suspect = false;
  auto x_has_non_numbers = [&]() {
    bool b = false;
    for (int i = lane; i < n; i += 64) { const bool bi = !wg_sane(q.x[i]); b = b || bi; }
    return __ballot(b) != 0ull;
  };
  PT_DECL
  if (resuming) {
    nact = rs->nact; info = rs->info; iterc = rs->iterc; itref = rs->itref; iflag = rs->iflag; jfinc = rs->jfinc; knext = rs->knext;
    out.hist_len = rs->hist_len;
    xmag = rs->xmag; vfact = rs->vfact; res = rs->res; ratio = rs->ratio; diag = rs->diag;
    if constexpr (kNan) x_suspect = x_has_non_numbers();
  }

#define LOG_EVENT(code)                                                   \
  do {                                                                    \
    if (lane == 0 && hist && out.hist_len < hist_cap) hist[out.hist_len] = (code); \
    out.hist_len++;                                                       \
  } while (0)

  // ---- reciprocal lengths of the constraint normals, :769-807 ----
  if (!resuming) {
    int fatal = 0x7fffffff;
    if constexpr (P::kCompact) {
      fatal = prob.norms(q, lane);
    } else {
      for (int k = lane; k < m; k += 64) {
        double sum = 0.0;
        if constexpr (P::kRowOps) sum = prob.row_sqnorm(q, k);
        else {
          WG_UNROLL
          for (int i = 0; i < n; ++i) { double a = Am(k, i); sum += a * a; }
        }
        if (sum > 0.0) sum = 1.0 / sqrt(sum);
        else if (q.b[k] == 0.0) {}
        else if (k + 1 <= me || !(q.b[k] <= 0.0)) fatal = k + 1 < fatal ? k + 1 : fatal;   // :789, a NaN fails
        q.wa[k] = sum;
      }
    }
    for (int k = lane; k < n; k += 64) q.wa[m + k] = 1.0;
    fatal = uni(wave_min_int(fatal));
    if (fatal != 0x7fffffff) { info = -fatal; early_exit = true; }
  }
  PT(0);

  if (!early_exit && !resuming) {
    // ---- make the Hessian numerically positive definite, :814-854 ----
    for (int i = lane; i < n; i += 64) q.wd[i] = prob.Gd(q, i);
    WG_WSYNC();
    if constexpr (P::kCompact) {
      diag = prob.diag_check(q, vsmall, lane);
    } else {
      double dl = 0.0;
      for (int i = lane; i < n; i += 64) {
        double wdi = q.wd[i];
        dl = maxd(dl, vsmall - wdi);
        WG_UNROLL
        for (int j = i + 1; j < n; ++j) {
          double gjj = q.wd[j], gij = Gm(i, j);
          double ga = -mind(wdi, gjj);
          double gb = fabs(wdi - gjj) + fabs(gij);
          if (gb > 0.0) ga += gij * gij / gb;
          dl = maxd(dl, ga);
        }
      }
      diag = wave_max(dl);
    }
    diag = uni(diag);
    bool need_shift = !(diag <= 0.0);                       // :844 `if (diag <= 0) goto L90`: a NaN shifts
    PT(1);
    bool factored = false;
    if constexpr (P::kHasFactor) {
      WG_REP(8)
      if (!need_shift && prob.blocks_ok) factored = WG_UBOOL(prob.factor(q, vsmall, lane));
    }
    if constexpr (P::kNM > 0 && P::kNM <= 64 && !kRegs && !P::kHasFactor) {
      // compile-time-bounded views without a structured factor of their own (the dense boundary at a known size, the Dimitrov
      // tick's QL back-end): R and Z through registers.  The compact view keeps the generic loops for the rare tick whose blocks
      // do not apply: unrolled into the tick kernels this body doubled their spilled SGPRs (260 -> 552) for a path they almost never take
      if (!factored && !need_shift) factored = chol_inverse_regs<P::kNM>(q, prob, vsmall, lane);
    }
    if (!factored) {
    for (;;) {
      if (need_shift) {
        diag = diagr * diag;
        for (int i = lane; i < n; i += 64) prob.setGd(q, i, diag + q.wd[i]);
        WG_WSYNC();
      }
      // ---- Cholesky, row by row (same sums as the column order of :859-890) ----
      int jfail = -1;
      double tfail = 0.0;
      for (int i = 0; i < n; ++i) {
        for (int j = i + lane; j < n; j += 64) {
          double temp = Gm(i, j);
          WG_UNROLL
          for (int k = 0; k < i; ++k) temp -= Rf(k, j) * Rf(k, i);
          if (j == i) {
            if (temp < vsmall) { q.slot[0] = 1.0; q.slot[1] = temp; }
            else { q.slot[0] = 0.0; Rf(i, i) = sqrt(temp); }
          } else q.sc0[j] = temp;
        }
        WG_WSYNC();
        if (WG_UBOOL(q.slot[0] != 0.0)) { jfail = i; tfail = q.slot[1]; break; }
        double rii = Rf(i, i);
        for (int j = i + 1 + lane; j < n; j += 64) Rf(i, j) = q.sc0[j] / rii;
        WG_WSYNC();
      }
      jfail = uni(jfail);
      if (jfail < 0) break;
      // ---- :895-918 further diagonal shift (rare; lane 0, serial) ----
      if (lane == 0) {
        double dnew;
        if (jfail == 0) dnew = diag + vsmall - tfail;
        else {
          double *v = q.lam;
          double sumx = 1.0;
          v[jfail] = 1.0;
          for (int k = jfail; k >= 1; --k) {
            double sum = 0.0;
            WG_UNROLL
            for (int i = k; i <= jfail; ++i) sum -= Rf(k - 1, i) * v[i];
            v[k - 1] = sum / Rf(k - 1, k - 1);
            sumx += v[k - 1] * v[k - 1];
          }
          dnew = diag + vsmall - tfail / sumx;
        }
        q.slot[2] = dnew;
      }
      WG_WSYNC();
      diag = uni(q.slot[2]);
      need_shift = true;
    }

    PT(2);
    // ---- Z = R^-1, :937-975 ----
    for (int i = lane; i < n; i += 64) {
      WG_UNROLL
      for (int j = 0; j < i; ++j) Zm(i, j) = 0.0;
      Zm(i, i) = 1.0 / Rf(i, i);
    }
    {
      // aligned form: all lanes walk (c, k) together so R(k,c) is a broadcast
      const int i0 = lane, i1 = lane + 64;
      double sum0, sum1;
      for (int c = 1; c < n; ++c) {
        sum0 = 0.0; sum1 = 0.0;
        WG_UNROLL
        for (int k = 0; k < c; ++k) {
          double rkc = Rf(k, c);
          if (i0 <= k) sum0 += Zm(i0, k) * rkc;
          if (i1 <= k) sum1 += Zm(i1, k) * rkc;
        }
        double rcc = Rf(c, c);
        if (i0 < c) Zm(i0, c) = -sum0 / rcc;
        if (i1 < c) Zm(i1, c) = -sum1 / rcc;
      }
    }
    WG_WSYNC();
    }   // !factored
  }

  PT(3);
  // register rows of A (DenseRegProb) are loaded here, after the factorisation has given its registers back
  if constexpr (HasRegRows<P>::value) { if (!early_exit && !resuming) prob.load_rows(q, lane); }
  if constexpr (kRegs) {
    // Z = R^-1 as factor() (or the generic inverse above) left it in the global slot: its rows into the registers, where Z stays
    if (!early_exit && !resuming) { WG_WSYNC(); zr_load(q, *zr, lane); }
  }
  enum { ST_RESET, ST_RESID, ST_SCAN, ST_CONVERGED, ST_FINISH };
  int st = early_exit ? ST_FINISH : ST_RESET;
  if (resuming) st = rs->st;
  while (st != ST_FINISH) {
    // the lane index is made opaque once per iteration: otherwise everything derived from it alone (clamped indices, LDS
    // addresses of the lane's entries) is computed in front of the loop and kept alive -- in the 256-register kernel: spilled
    // there and reloaded from scratch memory in every iteration
    asm volatile("" : "+v"(lane));
    if (st == ST_RESET || st == ST_RESID) {
      s = q.R + s_tail;
      if (st == ST_RESET) {                                 // :989-1027
        iflag = 1;
        for (int i = lane; i < n; i += 64) {
          q.x[i] = 0.0;
          q.ww[i] = q.d[i];
          if (i >= nact) continue;
          q.lam[i] = 0.0;
          int k = q.iact[i];
          if (k <= m) s[i] = q.b[k - 1];
          else if (k > mn) s[i] = -prob.xu(q, k - mn - 1);
          else s[i] = prob.xl(q, k - m - 1);
        }
        xmag = 0.0;
        vfact = 1.0;
        WG_WSYNC();
        PT(24);
      } else {                                              // :1031-1099
        iflag = 2;
        WG_REP(11) {                                        // gradient and residuals of the refresh: reads x, lam; writes ww, s
        typename ActiveParamsOf<P>::type ap;
        if constexpr (P::kCompact) ap = prob.active_params(q, nact, lane);
        for (int i = lane; i < n; i += 64) {
          double acc = q.d[i];
          if constexpr (P::kCompact) acc = prob.gdot_acc(q, i, q.x, acc);
          else {
            WG_UNROLL
            for (int j = 0; j < n; ++j) acc += Gm(i, j) * q.x[j];
          }
#ifdef WG_PROFILE
          q.ww[i] = acc; PT(27); acc = q.ww[i];
#endif
          if constexpr (P::kCompact) acc = prob.grad_minus_active(q, ap, nact, i, acc);
          else {
            WG_UNROLL
            for (int k = 0; k < nact; ++k) {
              int kk = q.iact[k];
              if (kk <= m) acc -= q.lam[k] * Am(kk - 1, i);
              else if (kk <= mn) { if (kk - m - 1 == i) acc -= q.lam[k]; }
              else { if (kk - mn - 1 == i) acc += q.lam[k]; }
            }
          }
          q.ww[i] = acc;
        }
        if constexpr (P::kCompact) { prob.row_residuals(q, q.sc0, lane); WG_WSYNC(); }
        for (int k = lane; k < nact; k += 64) {
          int kk = q.iact[k];
          double sk;
          if (kk <= m) {
            if constexpr (P::kCompact) sk = q.sc0[kk - 1];
            else {
              sk = q.b[kk - 1];
              WG_UNROLL
              for (int i = 0; i < n; ++i) sk -= q.x[i] * Am(kk - 1, i);
            }
          } else if (kk <= mn) { int k1 = kk - m - 1; sk = prob.xl(q, k1) - q.x[k1]; }
          else { int k1 = kk - mn - 1; sk = -prob.xu(q, k1) + q.x[k1]; }
          s[k] = sk;
        }
        WG_WSYNC();
        }   // WG_REP(11)
        PT(25);
      }
      if (nact > 0) {                                       // :1104-1170
        // forward substitution with R^T, column oriented (sums ascend in j)
        {
          double sum0 = 0.0, sum1 = 0.0;
          const int i0 = lane, i1 = lane + 64;
          for (int j = 0; j < nact; ++j) {
            if (i0 == j) s[j] = (s[j] - sum0) / Rp(j, j);
            if (i1 == j) s[j] = (s[j] - sum1) / Rp(j, j);
            WG_WSYNC();
            double sj = s[j];
            if (i0 > j && i0 < nact) sum0 += Rp(j, i0) * sj;
            if (i1 > j && i1 < nact) sum1 += Rp(j, i1) * sj;
          }
        }
        PT(26);
        if (P::kWideN || (P::kNM == 0 && n > 64 && n <= 128)) {
          double r0, r1;
          if constexpr (kRegs) zr_rows_times(q, *zr, s, 0, nact, lane, r0, r1);
          else z_rows_times<(P::kRowOps ? WG_ELEM_GRP : 8)>(q, s, 0, nact, lane, r0, r1);
          q.x[lane] += r0; q.sc0[lane] = r0;
          if (lane + 64 < n) { q.x[lane + 64] += r1; q.sc0[lane + 64] = r1; }
        } else
        for (int i = lane; i < n; i += 64) {
          double sum = 0.0;
          WG_UNROLL
          for (int j = 0; j < nact; ++j) sum += s[j] * Zm(i, j);
          q.x[i] += sum;
          q.sc0[i] = sum;
        }
        WG_WSYNC();
        for (int j = lane; j < n; j += 64) {
          double acc = q.ww[j];
          if constexpr (P::kCompact) acc = prob.gdot_acc(q, j, q.sc0, acc);
          else {
            WG_UNROLL
            for (int i = 0; i < n; ++i) acc += q.sc0[i] * Gm(i, j);
          }
          q.ww[j] = acc;
        }
        WG_WSYNC();
      }
      PT(4);
      if constexpr (kRegs) zr_zt_times_ww(q, *zr, s, lane);
      else zt_times_ww<P::kNM, (P::kRowOps ? WG_ELEM_GRP : 8), P::kWideN>(q, s, lane);                      // :1175-1177
      PT(5);
      if (nact != n) {                                      // :1186-1201
        if (P::kWideN || (P::kNM == 0 && n > 64 && n <= 128)) {
          double r0, r1;
          if constexpr (kRegs) zr_rows_times(q, *zr, s, nact, n, lane, r0, r1);
          else z_rows_times<(P::kRowOps ? WG_ELEM_GRP : 8)>(q, s, nact, n, lane, r0, r1);
          q.x[lane] -= r0;
          if (lane + 64 < n) q.x[lane + 64] -= r1;
        } else
        for (int i = lane; i < n; i += 64) {
          double sum = 0.0;
          WG_UNROLL
          for (int j = nact; j < n; ++j) sum += Zm(i, j) * s[j];
          q.x[i] -= sum;
        }
        info = 0;
        WG_WSYNC();
      }
      PT(6);
      if (nact != 0) {                                      // :1208-1217
        WG_BACKSUB(q, s, nact, lane);
        for (int k = lane; k < nact; k += 64) q.lam[k] += q.ww[k];
        WG_WSYNC();
      }
      PT(7);
      { double sm = 0.0; WG_REP(6) { sm = uni(xmag_sum(q, prob, vfact, lane)); WG_SINK(sm); } xmag = maxd(xmag, sm); }
      PT(8);
      if (iflag == itref) { st = ST_RESID; continue; }      // :1226
      // first inequality with a negative multiplier, :1233-1249 (`if (w[kdrop] >= zero) goto next`: a NaN multiplier IS dropped)
      int kd = 0x7fffffff;
      for (int k = lane; k < nact; k += 64)
        if (!(q.lam[k] >= 0.0) && q.iact[k] > me) { kd = k < kd ? k : kd; }
      kd = uni(wave_min_int(kd));
      if (kd != 0x7fffffff) {
        LOG_EVENT(-q.iact[kd]);
        nact = drop_constraint<ZR>(q, kd, nact, nact, lane, zr);
        st = ST_RESID;
        continue;
      }
      if constexpr (kNan) x_suspect = x_suspect || x_has_non_numbers();       // the refresh rewrote x
      st = ST_SCAN;
    }

    if (st == ST_SCAN) {
      // ---- most violated normalised constraint, :1255-1331 ----
      // bestw: the weight wa[.] of the lane's candidate as the scan read it.  The winner's is kept (wsel): the linear-dependence
      // test divides by it and the activation stores its negative -- wa may live in global memory (L2), where reading it again on
      // the critical path is an exposed round trip per iteration (the same value: nothing writes wa between the scan and the add)
      double bestv = 0.0, bestres = 0.0, bestw = 0.0;
      int bidx = -1;
      if (kNan && x_suspect) {                              // the iterate may hold a NaN / an infinity: the NaN-exact form decides
        if constexpr (kNan) {
          double cv = 0.0;
          int kn = knext;
          scan_nan_exact(q, prob, onha, cv, res, wsel, kn, lane);
          knext = uni(kn);
          bestv = uni(cv); bidx = -1;                       // res / knext / wsel are already what the reference leaves
        }
      } else
      WG_REP(1) {
      bestv = 0.0; bestres = 0.0; bestw = 0.0; bidx = -1;
      if constexpr (P::kCompact) {
        constexpr int NH = sizeof(prob.ax) / sizeof(double);
        double xs[2 * NH];
#pragma unroll
        for (int c = 0; c < 2 * NH; ++c) xs[c] = q.x[c];
        // the foot-placement row's operands are requested here as well: they arrive while the CoP row is summed
        const bool has_foot = lane < 5 * prob.ns;
        const int kf = has_foot ? 1 + 4 * NH + lane : 0;
        const double wakf = q.wa[kf], bkf = q.b[kf];
        {
          const int k = lane + 1;                     // the lane's CoP row
          const double wak = q.wa[k], bk = prob.bcop;
          double sum = -bk, asum = fabs(bk);
          prob.cop_row_dot(xs, sum, asum);
          if (prob.fj >= 0) {
            double t = q.x[2 * NH + prob.fj] * prob.fa(); sum += t; asum += fabs(t);
            t = q.x[2 * NH + prob.ns + prob.fj] * prob.fb(); sum += t; asum += fabs(t);
          }
          const double sumx = -sum * wak;
          {                                           // the reference's tests in its order, as one predicate and three selects
            const double tempa = asum + fabs(sum);
            const double temp2 = asum + onha * fabs(sum);
            const bool take = (wak > 0.0) & !(sumx <= 0.0) & !(tempa <= asum) & !(temp2 <= tempa);
            bestv = take ? sumx : bestv; bestres = take ? sum : bestres; bestw = take ? wak : bestw; bidx = take ? k + 1 : bidx;
          }
        }
        {
          // the lane's foot-placement row; lanes without one walk the all-zero dummy row 0 (weight 0: never a candidate) on
          // valid columns with zero coefficients
          const int k = kf;
          const double wak = wakf, bk = bkf;
          double sum = -bk, asum = fabs(bk);
          const auto fr = prob.foot_row();
#pragma unroll
          for (int e = 0; e < 4; ++e) { const double t = q.x[fr.c[e]] * fr.v[e]; sum += t; asum += fabs(t); }
          const double sumx = -sum * wak;
          const double tempa = asum + fabs(sum);
          const double temp2 = asum + onha * fabs(sum);
          const bool take = has_foot & (wak > 0.0) & !(sumx <= 0.0) & !((bidx >= 0) & (sumx <= bestv)) & !(tempa <= asum) & !(temp2 <= tempa);
          bestv = take ? sumx : bestv; bestres = take ? sum : bestres; bestw = take ? wak : bestw; bidx = take ? k + 1 : bidx;
        }
      } else {
      if constexpr (P::kRowOps) {
        // structured rows: every lane of a pass walks its row the same number of steps, both sums at once (row_dot_both);
        // the tests below are the reference's, in its order
        // wa, b and the row tables may live in global memory (L2): the operands of every pass are requested before the first
        // pass starts (m <= 1 + 64 kScanPasses rows, checked where the view is chosen)
        constexpr int kScanPasses = 3;
        double wak_p[kScanPasses], bk_p[kScanPasses], ra_p[kScanPasses], rb_p[kScanPasses];
        int rk_p[kScanPasses];
#pragma unroll
        for (int pp = 0; pp < kScanPasses; ++pp) {
          const int k = 1 + 64 * pp + lane;
          const bool in = k < m;
          const int kc = in ? k : 0;                          // surplus lanes walk the all-zero dummy row
          wak_p[pp] = in ? q.wa[kc] : 0.0; bk_p[pp] = q.b[kc];
          ra_p[pp] = prob.rowA[kc]; rb_p[pp] = prob.rowB[kc]; rk_p[pp] = prob.rowK[kc];
        }
#pragma unroll
        for (int pp = 0; pp < kScanPasses; ++pp) {            // row 0 is the all-zero dummy row: never a candidate, not walked
          const int k0 = 1 + 64 * pp;
          if (k0 >= m) break;
          const int k = k0 + lane;
          const int kc = k < m ? k : 0;
          const double wak = wak_p[pp], bk = bk_p[pp];
          double sum = -bk, temp = fabs(bk);
          prob.row_dot_both(q, kc, k0, ra_p[pp], rb_p[pp], rk_p[pp], q.x, sum, temp);
          if (wak <= 0.0) continue;
          double sumx = -sum * wak;
          if (k + 1 <= me) sumx = fabs(sumx);
          if (sumx <= 0.0) continue;
          if (bidx >= 0 && sumx <= bestv) continue;
          double tempa = temp + fabs(sum);
          if (tempa <= temp) continue;
          temp += onha * fabs(sum);
          if (temp <= tempa) continue;
          bestv = sumx; bestres = sum; bestw = wak; bidx = k + 1;
        }
      } else if constexpr (HasRegRows<P>::value) {
        // dense rows in registers: no memory access but x (LDS broadcasts, eight at a time); both of the lane's rows (lane,
        // 64 + lane) are walked together, then the reference's tests in its order -- row lane first, row 64 + lane second
        constexpr int NMr = P::kNM;
        const int ka = lane, kb = lane + 64;
        const bool ina = ka < m, inb = kb < m;
        const int mc = m > 0 ? m - 1 : 0;      // m == 0: entry 0 of b exists (mmax >= 1), its value is masked below
        const int kca = ina ? ka : mc, kcb = inb ? kb : mc;
        const double waka = ina ? q.wa[kca] : 0.0, bka = q.b[kca], wakb = inb ? q.wa[kcb] : 0.0, bkb = q.b[kcb];
        double suma = -bka, tempa_ = fabs(bka), sumb = -bkb, tempb_ = fabs(bkb);
#pragma unroll
        for (int i0 = 0; i0 < NMr; i0 += 8) {
          double xs[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) xs[e] = q.x[(i0 + e < n) ? i0 + e : n - 1];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (i0 + e < NMr && i0 + e < n) {
              const double ta = xs[e] * prob.ar0[i0 + e]; suma += ta; tempa_ += fabs(ta);
              const double tb = xs[e] * prob.ar1[i0 + e]; sumb += tb; tempb_ += fabs(tb);
            }
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int k = pp == 0 ? ka : kb;
          const double wak = pp == 0 ? waka : wakb, sum = pp == 0 ? suma : sumb;
          double temp = pp == 0 ? tempa_ : tempb_;
          if (wak <= 0.0) continue;
          double sumx = -sum * wak;
          if (k + 1 <= me) sumx = fabs(sumx);
          if (sumx <= 0.0) continue;
          if (bidx >= 0 && sumx <= bestv) continue;
          double tempa = temp + fabs(sum);
          if (tempa <= temp) continue;
          temp += onha * fabs(sum);
          if (temp <= tempa) continue;
          bestv = sumx; bestres = sum; bestw = wak; bidx = k + 1;
        }
      } else
      // dense rows: ONE walk of the row for both sums (sum += x_i a_ki, temp += |x_i a_ki|, i ascending: the same values the
      // serial code forms in two walks, the second only for candidates), the row's entries of eight columns requested while the
      // previous eight are added -- with A read in place (L2) the 4-at-a-time walk exposed a round trip every four terms.
      // Every lane of a pass walks (surplus lanes a valid row); the tests below are the reference's, in its order.
      for (int k0 = 0; k0 < m; k0 += 64) {
        const int k = k0 + lane;
        const bool in = k < m;
        const int kc = in ? k : m - 1;
        const double wak = in ? q.wa[kc] : 0.0, bk = q.b[kc];
        double sum = -bk, temp = fabs(bk);
        {
          constexpr int kG = 8;
          double ua[kG], ub[kG];
          auto fetch = [&](double (&u)[kG], int i) {
#pragma unroll
            for (int e = 0; e < kG; ++e) u[e] = Am(kc, i + e < n ? i + e : n - 1);   // past the end: clamped (loaded, unused)
          };
          auto add = [&](const double (&u)[kG], int i) {
            double xs[kG];
#pragma unroll
            for (int e = 0; e < kG; ++e) xs[e] = q.x[i + e];
#pragma unroll
            for (int e = 0; e < kG; ++e) { const double t = xs[e] * u[e]; sum += t; temp += fabs(t); }
          };
          int i = 0;
          const int whole = n / kG * kG;
          if (whole > 0) {
            fetch(ua, 0);
            for (;;) {
              fetch(ub, i + kG);
              add(ua, i); i += kG;
              if (i >= whole) break;
              fetch(ua, i + kG);
              add(ub, i); i += kG;
              if (i >= whole) { 
#pragma unroll
                for (int e = 0; e < kG; ++e) ub[e] = ua[e];
                break;
              }
            }
            // ub holds columns i .. i + kG - 1 (clamped): the odd ones
#pragma unroll
            for (int e = 0; e < kG - 1; ++e)
              if (i + e < n) { const double t = q.x[i + e] * ub[e]; sum += t; temp += fabs(t); }
          } else {
            for (; i < n; ++i) { const double t = q.x[i] * Am(kc, i); sum += t; temp += fabs(t); }
          }
        }
        if (wak <= 0.0) continue;
        double sumx = -sum * wak;
        if (k + 1 <= me) sumx = fabs(sumx);
        if (sumx <= 0.0) continue;              // cvmax starts at 0 (:1256)
        if (bidx >= 0 && sumx <= bestv) continue;
        double tempa = temp + fabs(sum);
        if (tempa <= temp) continue;
        temp += onha * fabs(sum);
        if (temp <= tempa) continue;
        bestv = sumx; bestres = sum; bestw = wak; bidx = k + 1;
      }
      }
      if constexpr (P::kNM > 0) {                            // n <= 64: one bound pair per lane, selects instead of continues
        const bool in = lane < n;
        const int kc = in ? lane : n - 1;
        const double w = q.wa[m + kc], xk = q.x[kc];
        const double s1 = prob.xl(q, kc) - xk;
        const bool upper = s1 < 0.0;
        const double sum = upper ? xk - prob.xu(q, kc) : s1;
        const bool take = in && !(w <= 0.0) && !(s1 == 0.0) && !(sum <= 0.0) && !(bidx >= 0 && sum <= bestv);
        bestv = take ? sum : bestv; bestres = take ? -sum : bestres; bestw = take ? w : bestw; bidx = take ? (upper ? kc + 1 + mn : kc + 1 + m) : bidx;
      } else
      for (int k = lane; k < n; k += 64) {
        const double w = q.wa[m + k];
        if (w <= 0.0) continue;
        bool lower = true;
        double sum = prob.xl(q, k) - q.x[k];
        if (sum == 0.0) continue;
        if (sum < 0.0) { sum = q.x[k] - prob.xu(q, k); lower = false; }
        if (sum <= 0.0) continue;               // cvmax starts at 0
        if (bidx >= 0 && sum <= bestv) continue;
        bestv = sum; bestres = -sum; bestw = w; bidx = lower ? k + 1 + m : k + 1 + mn;
      }
      {
        // order key: general rows 1..m, then bounds by variable; lower/upper of one
        // variable never compete.  knext codes > mn (upper) must sort by variable.
        int key = bidx < 0 ? -1 : (bidx > mn ? bidx - n : bidx);
        double v = bestv;
        int kk = key;
        wave_argmax_first(v, kk);
        kk = uni(kk);
        if (kk < 0) { bestv = 0.0; bidx = -1; }
        else {
          int src = -1;
          // the lane that owns the winning key
          unsigned long long mask = __ballot(key == kk);
          src = __ffsll((long long)mask) - 1;
          bestv = rl(bestv, src);
          bestres = rl(bestres, src);
          bestw = rl(bestw, src);
          bidx = __builtin_amdgcn_readlane(bidx, src);
        }
      }
      WG_SINK(bestv); WG_SINK(bestres); WG_SINK(bestw); WG_SINK(bidx);
      }   // WG_REP(1)
      double cvmax = bestv;
      if (bidx >= 0) { res = bestres; knext = bidx; wsel = bestw; }
      PT(9);
      info = 0;
      if (WG_UBOOL(cvmax <= wg_kconst(vsmall))) { st = ST_CONVERGED; continue; }  // :1336

      // ---- has the objective stopped increasing?  :1343-1408 ----
      ++jfinc;
      if (jfinc == 0 || jfinc == ifinc) {
        if (jfinc == ifinc) {
          for (int i = lane; i < n; i += 64) {
            double sum = 2.0 * q.d[i];
            double sumx = fabs(sum);
            WG_UNROLL
            for (int j = 0; j < n; ++j) {
              double temp = Gm(i, j) * (q.wx[j] + q.x[j]);
              sum += temp;
              sumx += fabs(temp);
            }
            double dx = q.x[i] - q.wx[i];
            q.sc0[i] = sum * dx;
            q.sc1[i] = sumx * fabs(dx);
          }
          WG_WSYNC();
          double fdiff = 0.0, fdiffa = 0.0;
          WG_UNROLL
          for (int i = 0; i < n; ++i) { fdiff += q.sc0[i]; fdiffa += q.sc1[i]; }
          info = 2;
          double sum = fdiffa + fdiff;
          if (WG_UBOOL(sum <= fdiffa)) { st = ST_CONVERGED; continue; }
          double temp = fdiffa + onha * fdiff;
          if (WG_UBOOL(temp <= sum)) { st = ST_CONVERGED; continue; }
          jfinc = 0;
          info = 0;
        }
        for (int i = lane; i < n; i += 64) q.wx[i] = q.x[i];
        WG_WSYNC();
      }
      PT(10);
      ++iterc;                                              // :1415-1420
      if (iterc > maxit) { info = 1; st = ST_FINISH; continue; }

      // ---- new normal and its products with the columns of Z, :1422-1470 ----
      s = q.R + nact * (nact + 1) / 2;
      WG_REP(2)
      if (knext <= m) {
        if constexpr (P::kCompact) prob.fill_row(q, knext - 1, q.ww, lane);
        else if constexpr (HasRegRows<P>::value) prob.row_to(q, knext - 1, q.ww, lane);
        else for (int i = lane; i < n; i += 64) q.ww[i] = Am(knext - 1, i);
        WG_WSYNC();
        if constexpr (P::kCompact) prob.zt_row(q, s, knext - 1, lane);
        else if constexpr (kRegs) zr_zt_times_ww(q, *zr, s, lane);
        else if constexpr (P::kWideN && P::kRowOps) {
          // the Herdt QP at a horizon known at compile time: a CoP row of instant r has no entry in rows (r, N) and (N + r, 2N)
          constexpr int kNHc = P::kHorizon;
          const int k = knext - 1;
          if constexpr (WG_ZT_TILED != 0) {
            const bool cop = k >= 1 && k <= 4 * kNHc;
            const int rr = cop ? ((k - 1) >> 2) : -1;
            const int si = cop ? prob.stepidx[rr] : 1;       // the previewed step the instant belongs to (0: the current support phase)
            zt_times_ww_cop_tiled<kNHc>(q, s, lane, rr, !cop || (si >= 1 && si <= prob.ns));
          }
          else
          zt_times_ww_cop<kNHc, WG_ELEM_ZT_GRP>(q, s, lane, (k >= 1 && k <= 4 * kNHc) ? ((k - 1) >> 2) : -1);
        }
        else zt_times_ww<P::kNM, (P::kRowOps ? WG_ELEM_GRP : 8), P::kWideN>(q, s, lane);
      } else {
        int k1 = knext - m;
        double sg = 1.0;
        if (k1 > n) { k1 = knext - mn; sg = -1.0; }
        if constexpr (kRegs) {
          for (int i = lane; i < n; i += 64) q.ww[i] = (i == k1 - 1) ? sg : 0.0;
          zr_row_to(q, *zr, k1 - 1, sg, s, lane);
        } else
        for (int i = lane; i < n; i += 64) {
          q.ww[i] = (i == k1 - 1) ? sg : 0.0;
          double z = Zm(k1 - 1, i);
          s[i] = (sg > 0.0) ? z : -z;
        }
        WG_WSYNC();
      }
      PT(11);
      double parnew = 0.0, parinc = 0.0, step = 0.0, sumy;
      int kdrop = -1;
      int route;   // 0 step, 1 dependent (multipliers needed), 2 dependent (multipliers in ww)
      if (nact == n) route = 1;                             // :1477
      else {
        WG_SWEEP(q, s, n, nact, lane);                         // :1480-1482
        PT(12);
        if (nact == 0) route = 0;                           // :1488
        else {                                              // :1491-1532
          double suma = 0.0, sumb = 0.0, sumc = 0.0;
          WG_REP(5) {
          if constexpr (P::kNM > 0) {
            constexpr int NM = P::kNM;
            if (lane < NM) {
              const bool in = lane < n;
              const double zi = in ? Zm(lane, nact) : 0.0, wi = in ? q.ww[lane] : 0.0;
              q.sc0[lane] = wi * zi; q.sc1[lane] = fabs(wi * zi); q.sc2[lane] = zi * zi;
            }
            WG_WSYNC();
            // three ordered sums of NM terms: lane 0 adds the first vector, lane 1 the second, lane 2 the third (the
            // scratch vectors are contiguous) -- one 8-cycle add chain per lane instead of three interleaved ones in all
            const double *src = q.sc0 + (lane < 3 ? lane : 0) * (int)(q.sc1 - q.sc0);
            double acc = 0.0;
#pragma unroll
            for (int i0 = 0; i0 < NM; i0 += kOsChunk) {
              double t[kOsChunk];
#pragma unroll
              for (int i = 0; i < kOsChunk; ++i) if (i0 + i < NM) t[i] = src[i0 + i];
#pragma unroll
              for (int i = 0; i < kOsChunk; ++i) if (i0 + i < NM) acc += t[i];
            }
            suma = rl(acc, 0); sumb = rl(acc, 1); sumc = rl(acc, 2);
            WG_WSYNC();
          } else {
            // column nact of Z is read once, lane-parallel (with Z in global memory: two coalesced loads instead of n
            // broadcast ones in a row); the three ordered sums then run from LDS, one per lane, as in the compact view
            if constexpr (kRegs) {                            // column nact: what the sweep just left in zr.c0 / c1
              { const double zi = zr->c0, wi = q.ww[lane]; q.sc0[lane] = wi * zi; q.sc1[lane] = fabs(wi * zi); q.sc2[lane] = zi * zi; }
              if (lane + 64 < n) { const double zi = zr->c1, wi = q.ww[lane + 64]; q.sc0[lane + 64] = wi * zi; q.sc1[lane + 64] = fabs(wi * zi); q.sc2[lane + 64] = zi * zi; }
            } else
            for (int i = lane; i < n; i += 64) {
              const double zi = Zm(i, nact), wi = q.ww[i];
              q.sc0[i] = wi * zi; q.sc1[i] = fabs(wi * zi); q.sc2[i] = zi * zi;
            }
            WG_WSYNC();
            const double *src = q.sc0 + (lane < 3 ? lane : 0) * (int)(q.sc1 - q.sc0);
            double acc = 0.0;
            int i = 0;
            for (; i + 8 <= n; i += 8) {
              double t[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) t[e] = src[i + e];
#pragma unroll
              for (int e = 0; e < 8; ++e) acc += t[e];
            }
            for (; i < n; ++i) acc += src[i];
            suma = rl(acc, 0); sumb = rl(acc, 1); sumc = rl(acc, 2);
            WG_WSYNC();
          }
          WG_SINK(suma); WG_SINK(sumb); WG_SINK(sumc);
          }   // WG_REP(5)
#ifdef WG_DEBUG_ROUTE
          if (lane == 0 && blockIdx.x == 2 && iterc == 3) { for (int i = 0; i < n; i++) printf("GPUW %d %.17g %.17g\n", i, q.ww[i], Zm(i, nact)); }
          if (lane == 0) printf("GPU blk %d it %d knext %d nact %d suma %.17g sumb %.17g sumc %.17g wa %.17g\n", (int)blockIdx.x, iterc, knext, nact, suma, sumb, sumc, knext <= m ? q.wa[knext - 1] : 0.0);
#endif
          if (WG_UBOOL(!significant(sumb, fabs(suma)) || !(sumb > wg_kconst(vsmall)))) route = 1;
          else {
            sumc = sqrt(sumc);
            if (knext <= m) sumc /= wsel;                    // wa[knext - 1]: the value the scan read
            if (WG_UBOOL(significant(sumc, fabs(suma)))) route = 0;
            else {                                          // :1538-1540
              PT_COUNT(29);
              WG_BACKSUB(q, s, nact, lane);
              route = independent_coordinate(q, prob, knext, nact, vsmall, lane) ? 0 : 2;
            }
          }
        }
      }
      route = uni(route);
      PT(13);
      PT_COUNT(28);
      if (route != 0) PT_COUNT(30);
      if (route != 0) {
        if (route == 1) WG_BACKSUB(q, s, nact, lane);
        kdrop = pick_drop<(P::kNM > 0), kNan>(q, nact, res, ratio, lane);
        info = -knext;                                      // :1663
        if (kdrop < 0) { st = ST_CONVERGED; continue; }
        parinc = ratio;
        parnew = parinc;
      }

      // ---- partial steps, each ending in a deletion, :1673-1759 ----
      bool dual_only = (route != 0);
      for (;;) {
        if (!dual_only) {
          sumy = s[nact];                                   // :1718-1720
          step = -res / sumy;
          parinc = step / sumy;
          kdrop = -1;
          if (nact > 0) {
            PT(14);
            WG_REP(4) WG_BACKSUB(q, s, nact, lane);
            PT(15);
            WG_REP(7) { kdrop = pick_drop<(P::kNM > 0), kNan>(q, nact, res, ratio, lane); WG_SINK(kdrop); WG_SINK(ratio); }
            PT(16);
            if (kdrop >= 0) {                               // :1734-1743
              double temp = 1.0 - ratio / parinc;
              if (WG_UBOOL(temp <= 0.0)) kdrop = -1;
              else { step = ratio * sumy; parinc = ratio; res = temp * res; }
            }
          }
          if constexpr (P::kNM > 0) {                       // :1749-1755; surplus lanes shadow lane n - 1 (same address, same value)
            const int il = lane < n ? lane : n - 1;
            q.x[il] = q.x[il] + step * Zm(il, nact);
          } else if constexpr (kRegs) {
            q.x[lane] += step * zr->c0;
            if (lane + 64 < n) q.x[lane + 64] += step * zr->c1;
          } else
          for (int i = lane; i < n; i += 64) q.x[i] += step * Zm(i, nact);
          parnew += parinc;
          if constexpr (kNan) x_suspect = x_suspect || WG_UBOOL(!wg_sane(step) || !wg_sane(parinc));
          WG_WSYNC();
          if (nact < 1) break;
        }
        dual_only = false;
        if constexpr (P::kNM > 0) {                         // :1677-1687; surplus lanes shadow lane nact - 1
          if (nact > 0) {
            const int kl = lane < nact ? lane : nact - 1;
            const double l0 = q.lam[kl] - parinc * q.ww[kl];
            const int ia = q.iact[kl];
            q.lam[kl] = (ia > me) ? maxd(0.0, l0) : l0;
          }
        } else
        for (int k = lane; k < nact; k += 64) {             // :1677-1687
          double l = q.lam[k] - parinc * q.ww[k];
          if (q.iact[k] > me) l = maxd(0.0, l);
          q.lam[k] = l;
        }
        WG_WSYNC();
        if (kdrop < 0) break;
        {                                                   // :1697-1711
          int nu = nact + 1;
          LOG_EVENT(-q.iact[kdrop]);
          nact = drop_constraint<ZR>(q, kdrop, nu, nact, lane, zr);
          double *snew = s - (nact + 1);
          if (nu > n) nu = n;
          // ascending copy, source ahead of destination: lanes in index order
          for (int i0 = 0; i0 < nu; i0 += 64) {
            int i = i0 + lane;
            double v = (i < nu) ? s[i] : 0.0;
            WG_WSYNC();
            if (i < nu) snew[i] = v;
            WG_WSYNC();
          }
          s = snew;
          WG_SWEEP(q, s, nu, nact, lane);
        }
      }

      PT(17);
      // ---- add the new constraint, :1764-1771 ----
      if (lane == 0) {
        q.lam[nact] = parnew;
        q.iact[nact] = knext;
        int ia = knext - 1;
        if (knext > mn) ia -= n;
        q.wa[ia] = -wsel;                                   // = -wa[ia]: a store, not a read-modify-write
      }
      nact++;
      LOG_EVENT(knext);
      WG_WSYNC();
      PT(18);
      double sm = 0.0;
      WG_REP(6) { sm = uni(xmag_sum(q, prob, vfact, lane)); WG_SINK(sm); }   // :1776-1786
      xmag = maxd(xmag, sm);
      PT(19);
      if (WG_UBOOL(sm < wg_kconst(xmagr) * xmag)) st = ST_RESET;
      else if (itref <= 0) st = ST_SCAN;
      else st = ST_RESID;
      // R's LDS part is full (its columns and the working column hold nact finished columns): stop BETWEEN two iterations,
      // the loop's state goes to the caller, who moves R and resumes (or, without rs, repeats the solve from the start)
      if (q.nact_cap > 0 && nact > q.nact_cap) { cap_hit = true; break; }
      continue;
    }

    if (st == ST_CONVERGED) {                               // :1791-1799
      ++itref;
      jfinc = -1;
      if (itref == 1) { st = ST_RESID; continue; }
      st = ST_FINISH;
    }
  }
#undef LOG_EVENT

  PT(20);
  PT_FLUSH;
  if (cap_hit) {
    if (rs) {
      rs->nact = nact; rs->info = info; rs->iterc = iterc; rs->itref = itref; rs->iflag = iflag; rs->jfinc = jfinc; rs->knext = knext;
      rs->st = st; rs->hist_len = out.hist_len;
      rs->xmag = xmag; rs->vfact = vfact; rs->res = res; rs->ratio = ratio; rs->diag = diag;
    }
    out.ifail = kQlCapHit; out.n_iter = iterc; out.nact = nact;
    return out;
  }
  // ---- ql0001 epilogue, :497-608 ----
  out.ifail = 0;
  if (info == 1) out.ifail = 1;
  else if (info == 2) out.ifail = 2;
  else if (info < 0) out.ifail = -info + 10;
  out.n_iter = iterc;
  out.nact = nact;
  return out;
}

}  // namespace wg
