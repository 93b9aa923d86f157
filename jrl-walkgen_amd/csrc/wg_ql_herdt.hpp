// wg_ql_herdt.hpp -- problem view of the Herdt-2010 QP for the in-wave QL solver: nothing of the
// Hessian or of the constraint matrix is stored as a matrix.
//
// Structure used (all from the reference's assembly, generator-vel-ref.cpp:393-474, 587-674):
//   * C = blockdiag(Qb, Qb) + a 2ns-column border; Qb (N x N) is a constant of the model.
//   * general rows 1..4N ("CoP rows", 4 per previewed instant r) are
//         [ -(a*Uz[r,:]) | -(b*Uz[r,:]) | a*e_step | b*e_step ],   Uz[r,c] = u[r-c] (Toeplitz, c <= r)
//     i.e. three doubles (a, b, r) + the 16-entry vector u describe a row; rows 4N+1.. (foot placement)
//     have at most four non-zeros.
//   * 4N = 64 rows = one row per lane (N = 16): each lane keeps the 2N products of ITS row in registers, so
//     the per-iteration violation scan is pure register arithmetic on LDS-broadcast x.
//   * the leading 2N x 2N block of the Cholesky factor and of its inverse does not depend on the gait; it is
//     computed once per model (same recurrences, same order => same bits) and copied in; only the last 2ns
//     columns are factorised per tick.
// Skipped terms are structural zeros only: x*(+-0) added to a running sum never changes it, so every value
// that the reference's dense loops produce is reproduced bit for bit.
#pragma once
#include "wg_ql_device.hpp"

namespace wg {

constexpr int kSMaxQ = 4;   // == kSMax of the tick
constexpr int kGvLd = 4;    // row stride of the border block Gv: 2*ns <= 4 columns (at most two previewed steps)
constexpr int kQbLd = 32;   // row stride of TickTables::Qb (== kNMaxH)

// ------------------------------------------------------------------ border columns of R, one row per lane
// R(i,j) = (G(i,j) - sum_{k<i} R(k,j) R(k,i)) / R(i,i) for the 2 ns border columns j (:859-890 restricted to them) is a
// forward substitution per column: serial in i, and in the row-serial form each row's sum is itself a chain of i dependent
// multiply-subtracts carried by the few lanes that own a border column.  Here one ROW per lane: lane i keeps the running
// value of its entries; step k finalises row k (lane k divides), broadcasts it, and every row below subtracts its product --
// the same products in the same order k = 0, 1, ... for every entry, so the same bits, in n short steps.
//
// Structure used on top of that (C = blockdiag(Qb, Qb) + border, generator-vel-ref.cpp:617-674):
//   * the x half (jerk-x rows, x-foot rows / columns) and the y half never mix: every entry that pairs a row of one half
//     with a border column of the other is G = +0.0 minus products that each have a +0.0 factor, i.e. +0.0 - (+-0.0) = +0.0,
//     divided by a positive diagonal: exactly +0.0.  Those entries are written, not computed; where such a product enters
//     the sum of a non-cross entry it is (+0.0)(+0.0) = +0.0 and x - (+0.0) == x for every x;
//   * the two halves are the same numbers: the assembly writes the same -gamma Uz'V and gamma V'V into both (mpc_tick), and
//     the constant factor blocks are blockdiag(Rb, Rb).  The x half is computed (N + ns rows, ns columns), the y half copied.
// NH = horizon, NS = previewed steps (compile time); Gv rows are kGvLd wide with the row's own group first (both table views).
template <int NH, int NS>
__device__ __forceinline__ bool herdt_border_rows(const QlView &q, const double *gd, const double *Gv, double vsmall, int lane) {
  constexpr int M2 = 2 * NH;
  const int i = lane;                                     // x half: rows 0..N-1 jerk-x, N..N+NS-1 the x-foot rows
  const bool jerk = i < NH, foot = i >= NH && i < NH + NS, row = jerk || foot;
  const int fi = i - NH;                                  // foot row index
  const int gi = jerk ? i : M2 + (foot ? fi : 0);         // the row's index in the QP
  double acc[NS];
#pragma unroll
  for (int f = 0; f < NS; ++f)
    acc[f] = (jerk || (foot && f >= fi)) ? ((foot && f == fi) ? gd[gi] : Gv[gi * kGvLd + f]) : 0.0;
  const double dreg = jerk ? Rf(i, i) : 1.0;
  const int colbase = jerk ? i * (i + 1) / 2 : 0;         // R(k, i) = q.Rf[colbase + k], k <= i (first diagonal block)
  const int khi = jerk ? i : 0;
  // ---- steps k < N: the divisor is the constant diagonal ----
  double rnext = (0 < khi) ? q.Rf[colbase] : 0.0;
  for (int k = 0; k < NH; ++k) {
    const double rki_lds = rnext;                         // R(k, i), fetched one step ahead (clamped address, selected value)
    {
      const int kn = k + 1;
      const double v = q.Rf[colbase + (kn < khi ? kn : 0)];
      rnext = (kn < khi) ? v : 0.0;
    }
    double bk[NS];
#pragma unroll
    for (int f = 0; f < NS; ++f) {
      const double rk = acc[f] / dreg;                    // lane k's is R(k, 2N + f)
      bk[f] = rl(rk, k);
    }
    if (lane == k) {
#pragma unroll
      for (int f = 0; f < NS; ++f) {
        Rf(k, M2 + f) = bk[f]; Rf(NH + k, M2 + NS + f) = bk[f];     // x entry and its y twin
        Rf(k, M2 + NS + f) = 0.0; Rf(NH + k, M2 + f) = 0.0;         // the two cross entries
      }
    }
    double rki = rki_lds;
#pragma unroll
    for (int f = 0; f < NS; ++f) rki = (fi == f) ? bk[f] : rki;     // a foot row's own column entry R(k, i)
    const bool upd = row && i > k;
#pragma unroll
    for (int f = 0; f < NS; ++f) {
      const double nv = acc[f] - bk[f] * rki;
      acc[f] = (upd && (jerk || f >= fi)) ? nv : acc[f];
    }
  }
  // ---- the foot rows: pivot test (:868-872), square root, the entries to its right ----
  bool ok = true;
#pragma unroll
  for (int kb = 0; kb < NS; ++kb) {
    const int k = NH + kb;                                // lane that holds the row
    const double t = rl(acc[kb], k);                      // the pivot, wave-uniform
    ok = ok && !(t < wg_kconst(vsmall));                  // a failed pivot: the rest is computed and discarded (caller: generic path)
    const double rt = sqrt(t);
    double bk[NS];
#pragma unroll
    for (int f = 0; f < NS; ++f) {
      bk[f] = 0.0;
      if (f > kb) bk[f] = rl(acc[f] / rt, k);
    }
    if (lane == k) {
      Rf(M2 + kb, M2 + kb) = rt; Rf(M2 + NS + kb, M2 + NS + kb) = rt;
#pragma unroll
      for (int f = 0; f < NS; ++f) {
        if (f > kb) { Rf(M2 + kb, M2 + f) = bk[f]; Rf(M2 + NS + kb, M2 + NS + f) = bk[f]; }
        Rf(M2 + kb, M2 + NS + f) = 0.0;                   // x-foot row, y-foot column: cross (its mirror lies below the diagonal)
      }
    }
    double rki = 0.0;
#pragma unroll
    for (int f = 0; f < NS; ++f) rki = (fi == f) ? bk[f] : rki;
    const bool upd = foot && i > k;
#pragma unroll
    for (int f = 0; f < NS; ++f) {
      if (f > kb) {
        const double nv = acc[f] - bk[f] * rki;
        acc[f] = (upd && f >= fi) ? nv : acc[f];
      }
    }
  }
  return ok;
}

// ------------------------------------------------------------------ border columns of Z = R^-1 (:937-975), same structure
// Z(i,c) = -(sum_{k=i}^{c-1} Z(i,k) R(k,c)) / R(c,c) for the border columns c.  The sums start from +0.0, so terms that are
// exact zeros -- everything that pairs the x half with the y half, and the zeros below the diagonal of the constant block --
// change nothing wherever they stand; what is left of a cross entry is -(+0.0) / R(c,c) = -0.0 above the diagonal (and the
// +0.0 the reference stores below it).  The x half is computed, the y half is the same numbers.
template <int NH, int NS>
__device__ __forceinline__ void herdt_border_z(const QlView &q, int lane) {
  constexpr int M2 = 2 * NH;
  const int i = lane;
  const bool jerk = i < NH, foot = i >= NH && i < NH + NS, row = jerk || foot;
  const int fi = i - NH;
  const int gi = jerk ? i : M2 + (foot ? fi : 0);         // the row's index in the QP, and its twin in the y half
  const int gy = jerk ? NH + i : M2 + NS + (foot ? fi : 0);
  const int zi = jerk ? i : 0;
  double sum[NS];
#pragma unroll
  for (int f = 0; f < NS; ++f) sum[f] = 0.0;
#pragma unroll 4
  for (int k = 0; k < NH; ++k) {                          // k < i: Z(i,k) = +0.0, the products are exact zeros
    const double zk = Zm(zi, k);
#pragma unroll
    for (int f = 0; f < NS; ++f) sum[f] += zk * Rf(k, M2 + f);
  }
  double zl[NS];
#pragma unroll
  for (int f = 0; f < NS; ++f) {
    const int c = M2 + f;
    const double rcc = Rf(c, c);
    double sj = sum[f], sf = 0.0;
#pragma unroll
    for (int g = 0; g < NS; ++g) {
      if (g < f) {
        const double t = zl[g] * Rf(M2 + g, c);
        sj += t;                                          // jerk rows: every earlier foot row takes part
        sf = (g >= fi) ? sf + t : sf;                     // foot row fi: rows fi .. f-1
      }
    }
    double z = -sj / rcc;
    if (!jerk) z = (fi < f) ? -sf / rcc : ((fi == f) ? 1.0 / rcc : 0.0);
    zl[f] = z;
    if (row) {
      Zm(gi, c) = z; Zm(gy, M2 + NS + f) = z;             // the entry and its y twin
      Zm(gi, M2 + NS + f) = wg_kconst(-0.0);                         // x row, y-foot column: cross, above the diagonal
      Zm(gy, c) = jerk ? wg_kconst(-0.0) : 0.0;                      // y row, x-foot column: above (jerk-y rows) / below (y-foot rows) it
    }
  }
}


// Constant factor blocks of C = blockdiag(Qb, Qb) + border into the solver's R and Z: blockdiag(Rb, Rb) and blockdiag(Zb, Zb)
// with exact zeros in between (the cross-block products of the recurrences are x * 0), so only the first diagonal block is
// fetched from global memory; the zeros of Z's upper-right block are SIGNED (-(x * 0) / r), their signs come from a bit
// mask.  n = 2 NH + border columns; also clears the border rows of Z below the constant block.
template <int NH>
__device__ __forceinline__ void herdt_constant_blocks(const QlView &q, const double *R2, const double *Z2, const unsigned long long *z2sign,
                                                      int lane) {
  const int n = q.n;
  constexpr int M2 = 2 * NH;
  constexpr int NR = NH * (NH + 1) / 2, TR = (NR + 63) / 64, TZ = (NH * NH + 63) / 64;
  if constexpr (NH > 16) {
    // large horizon: one element at a time (the staged form below would keep 25 doubles per lane in flight; at this size the
    // copy is a per-mille of the tick)
    for (int e = lane; e < NR; e += 64) {
      int j = 0;
      while ((j + 1) * (j + 2) / 2 <= e) ++j;
      const int i = e - j * (j + 1) / 2;
      const double v = R2[e];
      q.Rf[e] = v;
      Rf(NH + i, NH + j) = v;
    }
    for (int e = lane; e < NH * NH; e += 64) { const int i = e % NH, j = NH + e / NH; Rf(i, j) = 0.0; }   // cross block of R
    for (int e = lane; e < NH * NH; e += 64) {
      const int i = e % NH, j = e / NH;
      const double v = Z2[i + j * M2];
      Zm(i, j) = v; Zm(NH + i, NH + j) = v;
      Zm(NH + i, j) = 0.0;
      Zm(i, NH + j) = ((z2sign[e >> 6] >> (e & 63)) & 1ull) ? wg_kconst(-0.0) : 0.0;
    }
    for (int e = lane; e < (n - M2) * M2; e += 64) { const int i = M2 + e % (n - M2), j = e / (n - M2); Zm(i, j) = 0.0; }
    return;
  }
  double rv[TR], zv[TZ];
#pragma unroll
  for (int t = 0; t < TR; ++t) { const int e = lane + 64 * t; rv[t] = e < NR ? R2[e] : 0.0; }
#pragma unroll
  for (int t = 0; t < TZ; ++t) { const int e = lane + 64 * t; zv[t] = e < NH * NH ? Z2[(e % NH) + (e / NH) * M2] : 0.0; }
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    const int e = lane + 64 * t;
    if (e < NR) {
      // e = j(j+1)/2 + i, i <= j < NH: column j is the largest with j(j+1)/2 <= e
      int j = 0;
      while ((j + 1) * (j + 2) / 2 <= e) ++j;
      const int i = e - j * (j + 1) / 2;
      q.Rf[e] = rv[t];
      Rf(NH + i, NH + j) = rv[t];
    }
  }
  for (int e = lane; e < NH * NH; e += 64) { const int i = e % NH, j = NH + e / NH; Rf(i, j) = 0.0; }   // cross block of R
#pragma unroll
  for (int t = 0; t < TZ; ++t) {
    const int e = lane + 64 * t;
    if (e < NH * NH) {
      const int i = e % NH, j = e / NH;
      Zm(i, j) = zv[t]; Zm(NH + i, NH + j) = zv[t];
      Zm(NH + i, j) = 0.0;
      Zm(i, NH + j) = ((z2sign[e >> 6] >> (e & 63)) & 1ull) ? wg_kconst(-0.0) : 0.0;
    }
  }
  for (int e = lane; e < (n - M2) * M2; e += 64) { const int i = M2 + e % (n - M2), j = e / (n - M2); Zm(i, j) = 0.0; }
}

template <int NH>
struct HerdtProb {
  static constexpr bool kCompact = true;
  static constexpr bool kNanExact = WG_TICK_NAN_EXACT != 0;   // NaN iterates end the way the reference ends them (scan_nan_exact)
  static constexpr bool kHasFactor = true;
  static constexpr bool kRowOps = false;       // the compact view has its own register-row paths
  static constexpr int kNM = 2 * NH + 2 * 2;   // n <= 2N + 2*2: at most two previewed steps (checked by wg_mpc_configure)
  static constexpr int kCopRows = 4 * NH;      // rows 1 .. 4N (lane k - 1 keeps row k's coefficients), then the foot-placement rows
  static constexpr bool kWideN = false;
  static constexpr int kFixedLdz = kNM | 1;    // Z in LDS, leading dimension of carve_fixed<kNM, ...>
  static_assert(4 * NH == 64, "one CoP row per lane needs 4N == 64");
  // ---- LDS / global tables (wave-uniform pointers) ----
  const double *Qb;       // global (L1/L2-resident constant of the model), NH x kQbLd
  const double *u;        // LDS, NH
  double *Gv;             // LDS, n x kGvLd: G(i, 2N + c)
  double *gd;             // LDS, n: current Hessian diagonal (shifted when needed)
  const double *rowA, *rowB;
  const int *rowK, *stepidx;
  const double *V_f;
  const double *R2, *Z2;  // global: constant 2N x 2N factor blocks (packed R, dense Z, ld 2N)
  const unsigned long long *z2sign;   // global: signs of the (zero) upper-right block of Z2
  double diag_b;          // the constant block's contribution to ql0002's diagonal test
  int blocks_ok;
  int ns;
  // ---- per-lane registers: the lane's own constraint rows ----
  double ax[NH], ay[NH];  // CoP row (lane+1): x- and y-jerk parts
  int fj;                 // its two foot-variable entries sit in columns 2N+fj, 2N+ns+fj (fj < 0: none); their values fa(), fb()
  // (the entries of the lane's CoP row toward the foot variables and of its foot-placement row are NOT kept: they are one
  //  addition away from the edge coefficients ra / rb / ga / gb below -- 17 registers of per-lane state that lived across the whole
  //  solve and were what the 256-register kernel spilled inside the active-set loop)
  // raw edge coefficients of the lane's two rows (rowA / rowB / rowK live in LDS only until the solver starts: their
  // storage is part of the pre-solve overlay): any row's (a, b, k) is one v_readlane away
  double ra, rb;          // CoP row lane+1
  double bcop;            // b of the CoP row lane+1 (constant during the solve; b itself lives in global memory, and the scan's
                          // first operation on a row is sum = -b: from a register it does not wait for an L2 round trip)
  double ga, gb; int gk;  // foot-placement row 1+4N+lane (gk < 0: unused row)
  // the assembly's 0.0 + (0.0 + a * 1.0) * 1.0 is a with a zero of either sign made +0.0: a + 0.0, exactly
  __device__ __forceinline__ double fa() const { return ra + 0.0; }
  __device__ __forceinline__ double fb() const { return rb + 0.0; }
  // The lane's foot-placement row (1 + 4N + lane), entries in column order [2N+kk-1] 2N+kk [2N+ns+kk-1] 2N+ns+kk (without a
  // predecessor step the two "-1" entries are zeros); an unused row is zeros on a valid column.  The assembly's expressions
  // 0.0 + (0.0 + a * -1.0) * -1.0 and 0.0 + (0.0 + a * 1.0) * -1.0 are a + 0.0 and 0.0 - a for every a (signed zeros included).
  struct FootRow { double v[4]; int c[4]; };
  __device__ __forceinline__ FootRow foot_row() const {
    const bool used = gk >= 0, pred = gk > 0;
    const int kk = used ? gk : 0;
    FootRow r;
    const int c1 = 2 * NH + kk, c3 = 2 * NH + ns + kk;
    r.c[0] = used ? (pred ? c1 - 1 : c1) : 2 * NH; r.c[1] = used ? c1 : 2 * NH;
    r.c[2] = used ? (pred ? c3 - 1 : c3) : 2 * NH; r.c[3] = used ? c3 : 2 * NH;
    r.v[0] = pred ? ga + 0.0 : 0.0; r.v[1] = used ? 0.0 - ga : 0.0;
    r.v[2] = pred ? gb + 0.0 : 0.0; r.v[3] = used ? 0.0 - gb : 0.0;
    return r;
  }

  // ------------------------------------------------------------------ element access (rare paths)
  __device__ __forceinline__ double G(const QlView &q, int i, int j) const {
    if (i == j) return gd[i];
    if (j < i) { const int t = i; i = j; j = t; }
    if (j < 2 * NH) {
      if (i < NH && j < NH) return Qb[i * kQbLd + j];
      if (i >= NH && j >= NH) return Qb[(i - NH) * kQbLd + (j - NH)];
      return 0.0;
    }
    return Gv[i * kGvLd + (j - 2 * NH)];
  }
  __device__ __forceinline__ double Gd(const QlView &, int i) const { return gd[i]; }
  __device__ __forceinline__ void setGd(const QlView &, int i, double v) const { gd[i] = v; }
  // QPProblem's bounds are the constants of qp-problem.cpp:118-121: no LDS copy
  __device__ __forceinline__ double xl(const QlView &, int) const { return -1e8; }
  __device__ __forceinline__ double xu(const QlView &, int) const { return 1e8; }

  // row k must be wave-uniform (it is at every call site: knext, or an entry of the active set)
  __device__ __forceinline__ double A(const QlView &, int k, int i) const {
    k = uni(k);
    if (k == 0) return 0.0;
    const bool cop = k <= 4 * NH;
    const int src = cop ? k - 1 : k - 1 - 4 * NH;
    const double a = cop ? rl(ra, src) : rl(ga, src), b = cop ? rl(rb, src) : rl(gb, src);
    const int kk = cop ? (k - 1) >> 2 : __builtin_amdgcn_readlane(gk, src);
    return elem(a, b, kk, k, i);
  }
  // the same element of the lane's OWN row -- CoP row k = lane + 1, or foot-placement row k = 1 + 4 NH + lane -- from the lane's
  // own registers: k differs from lane to lane (scan_nan_exact)
  __device__ __forceinline__ double A_own(int k, int i) const {
    const bool cop = k <= 4 * NH;
    return elem(cop ? ra : ga, cop ? rb : gb, cop ? (k - 1) >> 2 : gk, k, i);
  }
  __device__ __forceinline__ double elem(double a, double b, int kk, int k, int i) const {
    if (k <= 4 * NH) {
      const int r = kk;
      if (i < NH) return (i <= r) ? 0.0 + (0.0 + a * u[r - i]) * -1.0 : 0.0;
      if (i < 2 * NH) { const int c = i - NH; return (c <= r) ? 0.0 + (0.0 + b * u[r - c]) * -1.0 : 0.0; }
      int j = i - 2 * NH;
      if (j < ns) { const double v = (stepidx[r] == j + 1) ? 1.0 : 0.0; return 0.0 + (0.0 + a * v) * 1.0; }
      j -= ns;
      { const double v = (stepidx[r] == j + 1) ? 1.0 : 0.0; return 0.0 + (0.0 + b * v) * 1.0; }
    }
    if (kk < 0 || i < 2 * NH) return 0.0;
    int j = i - 2 * NH;
    if (j < ns) return 0.0 + (0.0 + a * V_f[kk * kSMaxQ + j]) * -1.0;
    j -= ns;
    return 0.0 + (0.0 + b * V_f[kk * kSMaxQ + j]) * -1.0;
  }

  // dst[i] = A(k, i) for i < n, lane i its own entry (k wave-uniform): the element access above without a lane-dependent branch
  // (every lane-dependent `if` in a loop of the solver costs an exec-mask save / branch / restore, ~27 cycles on the spot).  The
  // same expressions on the same operands, chosen by selects; lanes past n shadow lane n - 1 (same address, same value).
  __device__ __forceinline__ void fill_row(const QlView &q, int k, double *dst, int lane) const {
    const int n = q.n;
    const int i = lane < n ? lane : n - 1;
    k = uni(k);
    if (k == 0) { dst[i] = 0.0; return; }
    const bool cop = k <= 4 * NH;
    const int src = cop ? k - 1 : k - 1 - 4 * NH;
    const double a = cop ? rl(ra, src) : rl(ga, src), b = cop ? rl(rb, src) : rl(gb, src);
    const int kk = cop ? (k - 1) >> 2 : __builtin_amdgcn_readlane(gk, src);
    const bool jerk = i < 2 * NH, xpart = i < NH;
    const int jf = jerk ? 0 : i - 2 * NH;                    // foot-variable index 0 .. 2 ns - 1
    const bool second = jf >= ns;                            // its y half
    const int jj = second ? jf - ns : jf;
    double val;
    if (cop) {
      const int r = kk;
      const int c = xpart ? i : (jerk ? i - NH : 0);
      const bool low = c <= r;
      const double uu = u[low ? r - c : 0];
      const double jv = low ? 0.0 + (0.0 + (xpart ? a : b) * uu) * -1.0 : 0.0;
      const double v = (stepidx[r] == jj + 1) ? 1.0 : 0.0;
      const double fv = 0.0 + (0.0 + (second ? b : a) * v) * 1.0;
      val = jerk ? jv : fv;
    } else {
      const int kc = kk < 0 ? 0 : kk;
      const double vf = V_f[kc * kSMaxQ + jj];
      const double fv = 0.0 + (0.0 + (second ? b : a) * vf) * -1.0;
      val = (kk < 0 || jerk) ? 0.0 : fv;
    }
    dst[i] = val;
  }

  // ------------------------------------------------------------------ per-lane rows into registers
  __device__ __forceinline__ void load_rows(int lane, const double *bvec) {
    bcop = bvec[lane + 1];
    {
      const int k = lane + 1;
      const double a = rowA[k], b = rowB[k];
      const int r = rowK[k];
      ra = a; rb = b;
#pragma unroll
      for (int c = 0; c < NH; ++c) {
        const double uu = (c <= r) ? u[r - c] : 0.0;
        ax[c] = (c <= r) ? 0.0 + (0.0 + a * uu) * -1.0 : 0.0;
        ay[c] = (c <= r) ? 0.0 + (0.0 + b * uu) * -1.0 : 0.0;
      }
      fj = stepidx[r] - 1;
      if (fj >= ns) fj = -1;
    }
    ga = 0.0; gb = 0.0; gk = -1;
    if (lane < 5 * ns) {
      const int k = 1 + 4 * NH + lane;
      ga = rowA[k]; gb = rowB[k]; gk = rowK[k];
    }
  }

  // ------------------------------------------------------------------ qld.cpp:769-800
  __device__ __forceinline__ int norms(const QlView &q, int lane) const {
    int fatal = 0x7fffffff;
    {
      double sum = 0.0;
#pragma unroll
      for (int c = 0; c < NH; ++c) sum += ax[c] * ax[c];
#pragma unroll
      for (int c = 0; c < NH; ++c) sum += ay[c] * ay[c];
      if (fj >= 0) { sum += fa() * fa(); sum += fb() * fb(); }
      const int k = lane + 1;
      if (sum > 0.0) sum = 1.0 / sqrt(sum);
      else if (q.b[k] == 0.0) {}
      else if (!(q.b[k] <= 0.0)) fatal = k + 1;
      q.wa[k] = sum;
    }
    if (lane < 5 * ns) {
      double sum = 0.0;
      const FootRow fr = foot_row();
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += fr.v[e] * fr.v[e];
      const int k = 1 + 4 * NH + lane;
      if (sum > 0.0) sum = 1.0 / sqrt(sum);
      else if (q.b[k] == 0.0) {}
      else if (!(q.b[k] <= 0.0)) fatal = (k + 1 < fatal) ? k + 1 : fatal;
      q.wa[k] = sum;
    }
    if (lane == 0) q.wa[0] = 0.0;     // the dummy row: zero normal, zero rhs (qp-problem.cpp:428-440)
    return fatal;
  }

  // ------------------------------------------------------------------ row . x for the lane's rows
  // sum = init + sum_c x[c]*A(row,c),  asum = ainit + sum_c |x[c]*A(row,c)|   (column order)
  __device__ __forceinline__ void cop_row_dot(const double *xs, double &sum, double &asum) const {
#pragma unroll
    for (int c = 0; c < NH; ++c) { const double t = xs[c] * ax[c]; sum += t; asum += fabs(t); }
#pragma unroll
    for (int c = 0; c < NH; ++c) { const double t = xs[NH + c] * ay[c]; sum += t; asum += fabs(t); }
  }

  // ------------------------------------------------------------------ sum_k G(row,k)*v[k] on top of acc
  __device__ __forceinline__ double gdot_acc(const QlView &q, int row, const double *v, double acc) const {
    const int n = q.n;
    if (row < 2 * NH) {
      const int blk = (row < NH) ? 0 : NH;
      const double *qr = Qb + (row - blk) * kQbLd;
      // the row of Q_b comes from global memory (L2): all NH loads are issued before the first use -- left to itself
      // the register-limited build waits for each one in turn
      double qv[NH], vv[NH];
#pragma unroll
      for (int k = 0; k < NH; ++k) qv[k] = qr[k];
#pragma unroll
      for (int k = 0; k < NH; ++k) vv[k] = v[blk + k];
      const double gdr = gd[row];                    // hoisted: a select per term, not a branch around an LDS read
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const double g = (blk + k == row) ? gdr : qv[k];
        acc += g * vv[k];
      }
      for (int c = 0; c < n - 2 * NH; ++c) acc += Gv[row * kGvLd + c] * v[2 * NH + c];
      return acc;
    }
    const int cc = row - 2 * NH;
#pragma unroll
    for (int k0 = 0; k0 < 2 * NH; k0 += 8) {        // loads in groups of 8 ahead of the add chain
      double gg[8], vv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { gg[c] = Gv[(k0 + c) * kGvLd + cc]; vv[c] = v[k0 + c]; }
#pragma unroll
      for (int c = 0; c < 8; ++c) acc += gg[c] * vv[c];
    }
    for (int c = 0; c < n - 2 * NH; ++c) {
      const double g = (c == cc) ? gd[row] : Gv[row * kGvLd + c];
      acc += g * v[2 * NH + c];
    }
    return acc;
  }

  // ------------------------------------------------------------------ residual refresh (qld.cpp:1031-1099), compact forms
  // acc_i -= sum_k lam_k * A(active row k, i), k ascending -- the parameters of active constraint k are collected in
  // lane k (active_params, every lane takes part) and broadcast (scalar) once per k, so an element costs one LDS read (u).
  struct ActiveParams { double pa, pb, plam; int pr, ptype, pidx; };
  __device__ __forceinline__ ActiveParams active_params(const QlView &q, int nact, int lane) const {
    const int m = q.m, mn = q.mn;
    ActiveParams P;
    P.pa = 0.0; P.pb = 0.0; P.plam = 0.0; P.pr = -1; P.ptype = 4; P.pidx = -1;   // 0 CoP row, 1 foot row, 2 lower, 3 upper bound, 4 nothing
    int rk = 0;
    if (lane < nact) {
      const int kk = q.iact[lane];
      P.plam = q.lam[lane];
      if (kk <= m) {
        rk = kk - 1;
        P.ptype = (rk >= 1 && rk <= 4 * NH) ? 0 : 1;
        if (rk == 0) P.ptype = 4;                  // the dummy row: all zeros
      } else if (kk <= mn) { P.ptype = 2; P.pidx = kk - m - 1; }
      else { P.ptype = 3; P.pidx = kk - mn - 1; }
    }
    // row rk's coefficients sit in the registers of the lane that owns the row: gather through one scratch vector
    double *t = q.sc0;
    t[lane + 1] = ra;
    if (lane < 5 * ns) t[1 + 4 * NH + lane] = ga;
    WG_WSYNC();
    if (P.ptype <= 1) P.pa = t[rk];
    WG_WSYNC();
    t[lane + 1] = rb;
    if (lane < 5 * ns) t[1 + 4 * NH + lane] = gb;
    WG_WSYNC();
    if (P.ptype <= 1) P.pb = t[rk];
    WG_WSYNC();
    int *tk = reinterpret_cast<int *>(q.sc0);
    if (lane < 5 * ns) tk[lane] = gk;
    WG_WSYNC();
    if (P.ptype == 0) P.pr = (rk - 1) >> 2;
    else if (P.ptype == 1) P.pr = tk[rk - 1 - 4 * NH];
    WG_WSYNC();
    return P;
  }
  __device__ __forceinline__ double grad_minus_active(const QlView &q, const ActiveParams &P, int nact, int i, double acc) const {
    const bool xb = i < NH, jerk = i < 2 * NH;
    const int c = xb ? i : i - NH;                 // column inside the jerk block
    int fjx = -1, fjy = -1;                        // foot column index inside its group
    if (!jerk) { const int j = i - 2 * NH; if (j < ns) fjx = j; else fjy = j - ns; }
    for (int k = 0; k < nact; ++k) {
      const int type = __builtin_amdgcn_readlane(P.ptype, k);
      const double lam = rl(P.plam, k);
      if (type <= 1) {
        const double a = rl(P.pa, k), b = rl(P.pb, k);
        const int r = __builtin_amdgcn_readlane(P.pr, k);
        double e = 0.0;
        if (type == 0) {
          if (jerk) {
            const double ab = xb ? a : b;
            const int rc = (c <= r) ? r - c : 0;
            const double uu = u[rc];
            e = (c <= r) ? 0.0 + (0.0 + ab * uu) * -1.0 : 0.0;
          } else {
            const int sidx = stepidx[r];
            if (fjx >= 0) { const double v = (sidx == fjx + 1) ? 1.0 : 0.0; e = 0.0 + (0.0 + a * v) * 1.0; }
            else { const double v = (sidx == fjy + 1) ? 1.0 : 0.0; e = 0.0 + (0.0 + b * v) * 1.0; }
          }
        } else if (!jerk && r >= 0) {
          if (fjx >= 0) e = 0.0 + (0.0 + a * V_f[r * kSMaxQ + fjx]) * -1.0;
          else e = 0.0 + (0.0 + b * V_f[r * kSMaxQ + fjy]) * -1.0;
        }
        acc -= lam * e;
      } else if (type == 2) {
        if (__builtin_amdgcn_readlane(P.pidx, k) == i) acc -= lam;
      } else if (type == 3) {
        if (__builtin_amdgcn_readlane(P.pidx, k) == i) acc += lam;
      }
    }
    return acc;
  }
  // out[rk] = b[rk] - sum_i x_i A(rk, i) (i ascending) for every general row, each lane its own rows (register copies)
  __device__ __forceinline__ void row_residuals(const QlView &q, double *out, int lane) const {
    double xs[2 * NH];
#pragma unroll
    for (int c = 0; c < 2 * NH; ++c) xs[c] = q.x[c];
    {
      const int rk = lane + 1;
      double sk = q.b[rk];
#pragma unroll
      for (int c = 0; c < NH; ++c) sk -= xs[c] * ax[c];
#pragma unroll
      for (int c = 0; c < NH; ++c) sk -= xs[NH + c] * ay[c];
      if (fj >= 0) { sk -= q.x[2 * NH + fj] * fa(); sk -= q.x[2 * NH + ns + fj] * fb(); }
      out[rk] = sk;
    }
    if (lane < 5 * ns) {
      const int rk = 1 + 4 * NH + lane;
      double sk = q.b[rk];
      const FootRow fr = foot_row();
#pragma unroll
      for (int e = 0; e < 4; ++e) sk -= q.x[fr.c[e]] * fr.v[e];
      out[rk] = sk;
    }
    if (lane == 0) out[0] = q.b[0];
  }

  // ------------------------------------------------------------------ ql0002's diagonal test, :814-843
  __device__ __forceinline__ double diag_check(const QlView &q, double vsmall, int lane) const {
    const int n = q.n;
    double dl = diag_b;
    for (int i = lane; i < n; i += 64) {
      const double wdi = q.wd[i];
      if (i >= 2 * NH) dl = maxd(dl, vsmall - wdi);
      const int j0 = (i + 1 > 2 * NH) ? i + 1 : 2 * NH;
      for (int j = j0; j < n; ++j) {
        const double gjj = q.wd[j], gij = G(q, i, j);
        double ga = -mind(wdi, gjj);
        const double gb = fabs(wdi - gjj) + fabs(gij);
        if (gb > 0.0) ga += gij * gij / gb;
        dl = maxd(dl, ga);
      }
    }
    return wave_max(dl);
  }

  // ------------------------------------------------------------------ R and Z = R^-1 (:859-975)
  // Leading 2N columns: copied.  Remaining columns: the reference recurrences, restricted to them.
  // Returns false (nothing usable written) if a pivot fails -> caller falls back to the generic path.
  __device__ __forceinline__ bool factor(const QlView &q, double vsmall, int lane) const {
    const int n = q.n;
    constexpr int M2 = 2 * NH;
#ifdef WG_PROFILE
    const unsigned long long fp0 = clock64();
#endif
    herdt_constant_blocks<NH>(q, R2, Z2, z2sign, lane);
    WG_WSYNC();
    const int nb = n - M2;                                  // border columns: 2 ns <= 4
#ifdef WG_PROFILE
    const unsigned long long fp1 = clock64();
    if (lane == 0) atomicAdd(&g_prof[2], fp1 - fp0);       // slot "chol" (unused by the compact view): constant blocks into LDS
#endif
    // rows of R, columns >= 2N only; straight-line per step count (a wave-uniform switch)
    bool ok = true;
    if (nb == 2) ok = herdt_border_rows<NH, 1>(q, gd, Gv, vsmall, lane);
    else if (nb == 4) ok = herdt_border_rows<NH, 2>(q, gd, Gv, vsmall, lane);
    if (!WG_UBOOL(ok)) return false;
    WG_WSYNC();
#ifdef WG_PROFILE
    if (lane == 0) atomicAdd(&g_prof[31], clock64() - fp1);   // border rows of R
#endif
    if (nb == 2) herdt_border_z<NH, 1>(q, lane);
    else if (nb == 4) herdt_border_z<NH, 2>(q, lane);
    WG_WSYNC();
    return true;
  }

  // ------------------------------------------------------------------ s = Z' * (row rk of A), :2071-2085
  __device__ __forceinline__ void zt_row(const QlView &q, double *s, int rk, int lane) const {
    const int n = q.n;
    if (lane < n) {
      const int i = lane;
      const double *zc = q.Z + i * q.ldz;            // column i of Z is contiguous
      double acc = 0.0;
      if (rk >= 1 && rk <= 4 * NH) {
        // entries of the row beyond its instant are exact zeros: their products leave acc unchanged, so the
        // loops can run the full static length.  Loads are issued in groups of kZtChunk ahead of the add chain (left to
        // itself the register-limited build waits for every pair of loads: 16 LDS round trips instead of 4).
        // A CoP row of instant r has its entries in columns 0 .. r and NH .. NH + r: a chunk that lies wholly beyond r holds
        // nothing but exact zeros (their products are +-0, and acc -- started from +0.0 -- is never -0.0: adding them changes
        // nothing), so it is skipped: one wave-uniform test per chunk, a quarter of the chunks on average
        constexpr int kZtChunk = 8;
        const int r = (rk - 1) >> 2;
#pragma unroll
        for (int j0 = 0; j0 < 2 * NH; j0 += kZtChunk) {
          if ((j0 < NH ? j0 : j0 - NH) > r) continue;
          double zz[kZtChunk], wv[kZtChunk];
#pragma unroll
          for (int c = 0; c < kZtChunk; ++c) { zz[c] = zc[j0 + c]; wv[c] = q.ww[j0 + c]; }
#pragma unroll
          for (int c = 0; c < kZtChunk; ++c) acc += zz[c] * wv[c];
        }
      }
      for (int j = 2 * NH; j < n; ++j) acc += zc[j] * q.ww[j];
      s[i] = acc;
    }
    WG_WSYNC();
  }
};

// Element view of the same QP for any horizon (N <= 32, up to kSMaxQ previewed steps): G and A are never stored, every
// element is regenerated from the compact tables on access.  It plugs into the solver's generic (lane-strided) paths
// like DenseProb does, trading speed for LDS: at N = 32 (n <= 72, m <= 149) the dense matrices alone would need
// 42 KB + 87 KB, more than a CU has next to Z and R.  Same values as the dense assembly, element by element.
// border block: x-group rows (jerk-x rows and x-foot rows) only meet x-foot columns, y-group rows only y-foot columns, so
// one half-width table suffices: Gv[i][c'] with c' the column index inside the row's own group
constexpr int kGvLdElem = kSMaxQ;
// NHC > 0: the horizon is that compile-time constant (BASELINE config 5: 32) -- N folds into every loop bound and table stride,
// n = 2N + 2ns >= 64 is known (the solver compiles only its two-rows-per-lane forms: kWideN), the factor() dispatch is static;
// NHC == -1: any horizon, read from the member.
template <int NHC>
struct HerdtElemProbT {
  static constexpr bool kCompact = false;
  static constexpr bool kNanExact = WG_TICK_NAN_EXACT != 0;   // as in the compact view
  static constexpr bool kHasFactor = true;     // constant factor blocks + structured border (N == 32 only, see factor())
  static constexpr bool kRowOps = true;        // row products walk the row's structure instead of calling A() per element
  static constexpr int kNM = 0;
  static constexpr bool kWideN = NHC >= 32;    // 64 <= n <= 128 for every problem of the model
  static constexpr int kFixedLdz = 0;          // Z lives in the global slot
  static constexpr int kHorizon = NHC;         // (> 0: the compile-time horizon; rows 1 + 4 i + e are the CoP rows of instant i)
  struct NConst { static constexpr int v = NHC; __device__ __forceinline__ NConst &operator=(int) { return *this; } __device__ __forceinline__ operator int() const { return v; } };
  typename std::conditional<(NHC > 0), NConst, int>::type N;   // assigning to the constant form is a no-op
  int ns;
  const double *Qb;       // global, N x kQbLd
  const double *u;        // LDS, N
  double *Gv;             // LDS, n x kGvLdElem: G(i, 2N + c) of the row's own column group
  double *gd;             // LDS, n
  const double *rowA, *rowB;
  const int *rowK, *stepidx;
  const double *V_f;
  const double *R2, *Z2;  // global: constant 2N x 2N factor blocks (TickTables)
  const unsigned long long *z2sign;
  int blocks_ok;
  // R and Z = R^-1 (:859-975) from the constant blocks + the structured border, exactly as the compact view does it
  // (herdt_constant_blocks, herdt_border_rows, herdt_border_z): instantiated for BASELINE config 5's horizon, N = 32, with
  // 1..4 previewed steps; other horizons of this view take the solver's generic factorisation (return false).
  __device__ __forceinline__ bool factor(const QlView &q, double vsmall, int lane) const {
    if ((int)N != 32 || ns < 1 || ns > 4) return false;
    constexpr int NHc = 32;
    herdt_constant_blocks<NHc>(q, R2, Z2, z2sign, lane);
    WG_WSYNC();
    bool ok = true;
    switch (ns) {
      case 1: ok = herdt_border_rows<NHc, 1>(q, gd, Gv, vsmall, lane); break;
      case 2: ok = herdt_border_rows<NHc, 2>(q, gd, Gv, vsmall, lane); break;
      case 3: ok = herdt_border_rows<NHc, 3>(q, gd, Gv, vsmall, lane); break;
      default: ok = herdt_border_rows<NHc, 4>(q, gd, Gv, vsmall, lane); break;
    }
    if (!WG_UBOOL(ok)) return false;
    WG_WSYNC();
    switch (ns) {
      case 1: herdt_border_z<NHc, 1>(q, lane); break;
      case 2: herdt_border_z<NHc, 2>(q, lane); break;
      case 3: herdt_border_z<NHc, 3>(q, lane); break;
      default: herdt_border_z<NHc, 4>(q, lane); break;
    }
    WG_WSYNC();
    return true;
  }
  __device__ __forceinline__ double G(const QlView &q, int i, int j) const {
    if (i == j) return gd[i];
    if (j < i) { const int t = i; i = j; j = t; }
    if (j < 2 * N) {
      if (i < N && j < N) return Qb[i * kQbLd + j];
      if (i >= N && j >= N) return Qb[(i - N) * kQbLd + (j - N)];
      return 0.0;
    }
    const int c = j - 2 * N;
    const bool xcol = c < ns, xrow = (i < N) || (i >= 2 * N && i < 2 * N + ns);
    if (xcol != xrow) return 0.0;                 // never written by the assembly: stays +0.0 in the dense matrix
    return Gv[i * kGvLdElem + (xcol ? c : c - ns)];
  }
  __device__ __forceinline__ double Gd(const QlView &, int i) const { return gd[i]; }
  __device__ __forceinline__ void setGd(const QlView &, int i, double v) const { gd[i] = v; }
  // QPProblem's bounds are the constants of qp-problem.cpp:118-121: no LDS copy
  __device__ __forceinline__ double xl(const QlView &, int) const { return -1e8; }
  __device__ __forceinline__ double xu(const QlView &, int) const { return 1e8; }
  __device__ __forceinline__ double A(const QlView &, int k, int i) const {
    if (k == 0) return 0.0;
    const double a = rowA[k], b = rowB[k];
    const int kk = rowK[k];
    if (k <= 4 * N) {
      const int r = kk;
      if (i < N) return (i <= r) ? 0.0 + (0.0 + a * u[r - i]) * -1.0 : 0.0;
      if (i < 2 * N) { const int c = i - N; return (c <= r) ? 0.0 + (0.0 + b * u[r - c]) * -1.0 : 0.0; }
      int j = i - 2 * N;
      if (j < ns) { const double v = (stepidx[r] == j + 1) ? 1.0 : 0.0; return 0.0 + (0.0 + a * v) * 1.0; }
      j -= ns;
      { const double v = (stepidx[r] == j + 1) ? 1.0 : 0.0; return 0.0 + (0.0 + b * v) * 1.0; }
    }
    if (kk < 0 || i < 2 * N) return 0.0;
    int j = i - 2 * N;
    if (j < ns) return 0.0 + (0.0 + a * V_f[kk * kSMaxQ + j]) * -1.0;
    j -= ns;
    return 0.0 + (0.0 + b * V_f[kk * kSMaxQ + j]) * -1.0;
  }
  // acc (+)= sum_i v[i] * A(k, i)  (ABS: |v[i] * A(k, i)|), i ascending, over the row's structural non-zeros only: the
  // skipped entries are exact zeros, whose products leave a running sum unchanged.  Same element expressions as A().
  template <bool ABS>
  __device__ __forceinline__ double row_dot(const QlView &, int k, const double *v, double acc) const {
    if (k == 0) return acc;
    const double a = rowA[k], b = rowB[k];
    const int kk = rowK[k];
    if (k <= 4 * N) {
      const int r = kk;
      for (int c = 0; c <= r; ++c) { const double t = v[c] * (0.0 + (0.0 + a * u[r - c]) * -1.0); acc += ABS ? fabs(t) : t; }
      for (int c = 0; c <= r; ++c) { const double t = v[N + c] * (0.0 + (0.0 + b * u[r - c]) * -1.0); acc += ABS ? fabs(t) : t; }
      const int j = stepidx[r] - 1;
      if (j >= 0 && j < ns) {
        { const double t = v[2 * N + j] * (0.0 + (0.0 + a * 1.0) * 1.0); acc += ABS ? fabs(t) : t; }
        { const double t = v[2 * N + ns + j] * (0.0 + (0.0 + b * 1.0) * 1.0); acc += ABS ? fabs(t) : t; }
      }
      return acc;
    }
    if (kk < 0) return acc;
    for (int j = 0; j < ns; ++j) { const double t = v[2 * N + j] * (0.0 + (0.0 + a * V_f[kk * kSMaxQ + j]) * -1.0); acc += ABS ? fabs(t) : t; }
    for (int j = 0; j < ns; ++j) { const double t = v[2 * N + ns + j] * (0.0 + (0.0 + b * V_f[kk * kSMaxQ + j]) * -1.0); acc += ABS ? fabs(t) : t; }
    return acc;
  }
  // Both sums of the violation scan -- sum += v[i] A(k,i) and asum += |v[i] A(k,i)|, i ascending -- in ONE walk of the row, the
  // same for every lane of the pass (k = k0 + lane): trip counts depend on k0 only, the operands of four terms are fetched
  // ahead of the two add chains, and a lane whose row ends earlier (or is no CoP / foot row at all) walks on over exact-zero
  // coefficients (u[-1] = 0.0 in LDS, or a zeroed edge coefficient): its extra terms are +-0.0, which change neither a
  // non-zero sum nor any decision taken on a zero one (sumx = -(+-0) wak is not > 0 either way).
  __device__ __forceinline__ void row_dot_both(const QlView &q, int k, int k0, const double *v, double &sum, double &asum) const {
    row_dot_both(q, k, k0, rowA[k], rowB[k], rowK[k], v, sum, asum);
  }
  // the same with the row's (a, b, instant) already in registers: the row tables live in global memory (L2), and the scan
  // fetches the parameters of all its passes before the first one
  __device__ __forceinline__ void row_dot_both(const QlView &, int k, int k0, double a, double b, int kk, const double *v, double &sum,
                                               double &asum) const {
    // The coefficients are the assembly's expressions 0.0 + (0.0 + p) * -1.0 and 0.0 + (0.0 + p) * 1.0 of a product p, written as
    // 0.0 - p and p + 0.0: the same value for every p (a zero product of either sign gives +0.0 both ways), two operations
    // instead of four per element.
    if (k0 <= 4 * N) {                                      // the pass holds CoP rows (wave-uniform)
      const bool cop = k >= 1 && k <= 4 * N;
      const int r = cop ? kk : -1;
      const int klast = (k0 + 63 < 4 * N) ? k0 + 63 : 4 * N;
      const int rmax = (klast - 1) >> 2;                    // instant of the pass's last CoP row
      const double ac = cop ? a : 0.0, bc = cop ? b : 0.0;
#ifndef WG_SCAN_CH
#define WG_SCAN_CH 4
#endif
      constexpr int CH = WG_SCAN_CH;                         // terms whose operands are requested together, ahead of the two add chains
#pragma unroll 1
      for (int c0 = 0; c0 <= rmax; c0 += CH) {
        double vv[CH], uu[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) { const int d = r - c0 - e, ci = c0 + e < N ? c0 + e : N - 1; vv[e] = v[ci]; uu[e] = u[d > -1 ? d : -1]; }
#pragma unroll
        for (int e = 0; e < CH; ++e) { const double t = vv[e] * (0.0 - ac * uu[e]); sum += t; asum += fabs(t); }
      }
#pragma unroll 1
      for (int c0 = 0; c0 <= rmax; c0 += CH) {
        double vv[CH], uu[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) { const int d = r - c0 - e, ci = c0 + e < N ? c0 + e : N - 1; vv[e] = v[N + ci]; uu[e] = u[d > -1 ? d : -1]; }
#pragma unroll
        for (int e = 0; e < CH; ++e) { const double t = vv[e] * (0.0 - bc * uu[e]); sum += t; asum += fabs(t); }
      }
      const int j = (r >= 0) ? stepidx[r] - 1 : -1;
      const bool st = j >= 0 && j < ns;
      const int jc = st ? j : 0;
      { const double t = v[2 * N + jc] * ((st ? a : 0.0) * 1.0 + 0.0); sum += t; asum += fabs(t); }
      { const double t = v[2 * N + ns + jc] * ((st ? b : 0.0) * 1.0 + 0.0); sum += t; asum += fabs(t); }
    }
    if (k0 + 63 > 4 * N) {                                  // the pass holds foot-placement rows (wave-uniform)
      const bool ft = k > 4 * N && kk >= 0;
      const int kf = ft ? kk : 0;
      const double af = ft ? a : 0.0, bf = ft ? b : 0.0;
      for (int j = 0; j < ns; ++j) { const double t = v[2 * N + j] * (0.0 - af * V_f[kf * kSMaxQ + j]); sum += t; asum += fabs(t); }
      for (int j = 0; j < ns; ++j) { const double t = v[2 * N + ns + j] * (0.0 - bf * V_f[kf * kSMaxQ + j]); sum += t; asum += fabs(t); }
    }
  }
  __device__ __forceinline__ double row_sqnorm(const QlView &, int k) const {
    double sum = 0.0;
    if (k == 0) return sum;
    const double a = rowA[k], b = rowB[k];
    const int kk = rowK[k];
    if (k <= 4 * N) {
      const int r = kk;
      for (int c = 0; c <= r; ++c) { const double e = 0.0 + (0.0 + a * u[r - c]) * -1.0; sum += e * e; }
      for (int c = 0; c <= r; ++c) { const double e = 0.0 + (0.0 + b * u[r - c]) * -1.0; sum += e * e; }
      const int j = stepidx[r] - 1;
      if (j >= 0 && j < ns) {
        { const double e = 0.0 + (0.0 + a * 1.0) * 1.0; sum += e * e; }
        { const double e = 0.0 + (0.0 + b * 1.0) * 1.0; sum += e * e; }
      }
      return sum;
    }
    if (kk < 0) return sum;
    for (int j = 0; j < ns; ++j) { const double e = 0.0 + (0.0 + a * V_f[kk * kSMaxQ + j]) * -1.0; sum += e * e; }
    for (int j = 0; j < ns; ++j) { const double e = 0.0 + (0.0 + b * V_f[kk * kSMaxQ + j]) * -1.0; sum += e * e; }
    return sum;
  }
};
typedef HerdtElemProbT<-1> HerdtElemProb;

}  // namespace wg
