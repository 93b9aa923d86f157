// wg_pldp_device.hpp -- PLDP primal active-set solve with OptCholesky row updates, one problem per wavefront (gfx950).
//
// Restates, for a 64-lane wave,
//   PLDPSolver::SolveProblem and helpers      src/Mathematics/PLDPSolver.cpp:287-1036
//   OptCholesky::UpdateCholeskyMatrixFortran  src/Mathematics/OptCholesky.cpp:171-223
// with the bit-exactness rule of the QL kernel: lanes run over independent outputs only; every inner sum runs inside
// one lane in the reference's order (ascending index), no FMA contraction; '/' and sqrt are IEEE on gfx950.
//
//   stage                                   lanes
//   initial solution (:287-340)             lane i < 2N owns v[i]
//   v1 = E c (:455-470)                     lane li < S owns active row li
//   forward substitution L y = v1 (:342-365)  lane i owns y[i]; step k: lane k divides, broadcasts y[k] (v_readlane),
//                                           lanes i > k subtract L[i,k] y[k]  -> per-lane order k ascending, as the reference
//   backward substitution L' v2 = y (:367-400)  the reference sums k = i+1 .. S-1 ascending while v2[k] become known
//                                           descending: a true serial chain; lane i walks it from LDS when its turn comes
//   d = c - E' v2 (:509-519)                lane li < 2N
//   step length (:534-653)                  one row per lane (two when m > 64): A_i d, then A_i v for rows with
//                                           A_i d < 0; SimilarConstraint reuse resolved through LDS; arg-min with
//                                           first-index-wins by wave shuffle
//   row append of the Cholesky factor       lane lj owns L[new, lj]: Gram entry from A (k ascending), then the same
//   (OptCholesky.cpp:171-223)               broadcast recurrence as the forward substitution
//
// LDS per problem: A staged with an odd leading dimension (conflict-free by rows and by columns), packed lower L
// (WG_PLDP_ACTIVE_CAP rows), vectors.  See PldpLds::bytes().
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/wg_mpc.h"
#include "wg_ql_device.hpp"

namespace wg {

struct PldpModel {               // device-resident constants (wg_pldp_configure)
  int N, pad_;
  double iPu[WG_PLDP_N * WG_PLDP_N], Px[WG_PLDP_N * 3], Pu[WG_PLDP_N * WG_PLDP_N], iPuPx[2 * WG_PLDP_N * 6];
};

struct PldpLds {
  double *A, *L, *b, *c, *d, *Vk, *v2, *tmp1;
  double *c0, *c1, *PuL;         // structured view: A(row, k) = c0[row] * Pu[k][slot[row]], A(row, k+N) = c1[row] * Pu[k][slot[row]]
  // per-row marks in their narrowest types (similar: offset to an earlier row, -mcap < s <= 0; state: 1 active, 0 not, 2 pending):
  // with ints the dense view took 10.9 KB at mcap = 128 -- nine LDS granules, 14 problems per CU; 10 240 B are eight, 16 per CU
  short *similar;
  int *act;
  signed char *state;
  int *slot;
  int lda, cap, N;
  __host__ __device__ static int lda_for(int mcap) { return (mcap + 1) | 1; }
  // dense view: A staged with an odd leading dimension; structured view: 2 coefficients + the instant per row
  // a_lds (dense view only): the constraint matrix staged in LDS; false: read in place from global memory (L2)
  __host__ __device__ static size_t bytes(int mcap, int cap = WG_PLDP_ACTIVE_CAP, bool structured = false, bool a_lds = true) {
    const size_t n = 2 * WG_PLDP_N;
    size_t dbl = (structured ? (size_t)2 * mcap + WG_PLDP_N * WG_PLDP_N : (a_lds ? (size_t)lda_for(mcap) * n : 0)) +
                 (size_t)cap * (cap + 1) / 2 + mcap /*b*/ + 3 * n /*c d Vk*/ + cap /*v2*/ + mcap /*tmp1*/;
    size_t marks = (size_t)4 * cap /*act*/ + (structured ? (size_t)4 * mcap : 0) /*slot*/ + (size_t)2 * mcap /*similar*/ + mcap /*state*/;
    return dbl * 8 + ((marks + 7) & ~size_t(7));
  }
  __device__ void carve(unsigned char *base, int mcap, int cap_ = WG_PLDP_ACTIVE_CAP, bool structured = false, int N_ = WG_PLDP_N,
                        bool a_lds = true) {
    const int n = 2 * WG_PLDP_N;
    double *p = reinterpret_cast<double *>(base);
    lda = lda_for(mcap); cap = cap_; N = N_;
    A = nullptr; c0 = c1 = PuL = nullptr; slot = nullptr;
    if (structured) { c0 = p; p += mcap; c1 = p; p += mcap; PuL = p; p += WG_PLDP_N * WG_PLDP_N; }
    else if (a_lds) { A = p; p += lda * n; }
    L = p; p += cap * (cap + 1) / 2;
    b = p; p += mcap;
    c = p; p += n;
    d = p; p += n;
    Vk = p; p += n;
    v2 = p; p += cap;
    tmp1 = p; p += mcap;
    int *q = reinterpret_cast<int *>(p);
    act = q; q += cap;
    if (structured) { slot = q; q += mcap; }
    similar = reinterpret_cast<short *>(q);
    state = reinterpret_cast<signed char *>(similar + mcap);
  }
  // element (row, col) of the constraint matrix; ST selects the structured view at compile time.  The structured
  // product c * Pu is the very multiplication BuildConstraintMatrices performs (:893-905), so the value is identical.
  template <bool ST>
  __device__ __forceinline__ double a(int row, int col) const {
    if constexpr (ST) {
      const bool y = col >= N;
      return (y ? c1[row] : c0[row]) * PuL[(y ? col - N : col) * N + slot[row]];
    } else {
      return A[row + col * lda];
    }
  }
};

__device__ __forceinline__ int ltri(int i, int j) { return i * (i + 1) / 2 + j; }   // packed lower, j <= i

// arg-min over the wave: smaller v wins, equal v -> smaller idx.  idx < 0 = no candidate (then idx stays < 0).
// On the DPP path like wave_argmax_first (wg_ql_device.hpp); candidates must be finite.
__device__ __forceinline__ void wave_argmin_first(double &v, int &idx) {
  double nv = -v;                                   // exact, order-reversing; -(+-0) compares equal either way
  wave_argmax_first(nv, idx);
  v = -nv;
}

// OptCholesky::AddActiveConstraint + UpdateCholeskyMatrixFortran: append `row` as active row S (S < cap checked by caller)
template <bool ST>
__device__ __forceinline__ void pldp_add_row(const PldpLds &W, int n, int S, int row, int lane) {
  if (lane == 0) { W.act[S] = row; W.state[row] = 1; W.v2[S] = 0.0; }
  WG_WSYNC();
  double r = 0.0;
  if (lane <= S) {
    const int rj = W.act[lane];
    double mij = 0.0;
    for (int k = 0; k < n; k++) mij += W.a<ST>(row, k) * W.a<ST>(rj, k);
    r = mij;
  }
  for (int lk = 0; lk <= S; lk++) {
    double val = 0.0;
    if (lane == lk) {
      val = (lk != S) ? r / W.L[ltri(lk, lk)] : sqrt(r);
      W.L[ltri(S, lk)] = val;
    }
    val = rl(val, lk);
    if (lane > lk && lane <= S) r = r - val * (lane == S ? val : W.L[ltri(lane, lk)]);
  }
  WG_WSYNC();
}

// one SolveProblem; returns ret (uniform).  Outputs written by the caller from LDS/registers.
template <bool ST>
__device__ int pldp_solve(const PldpModel &M, const PldpLds &W, int m, const double *__restrict__ D_g,
                          const double *__restrict__ zmpref, const double *__restrict__ xkyk, int n_removed,
                          int starting, int max_iter, wg_pldp_state_t *st, int &S_out, int &it_out) {
  const int lane = threadIdx.x;
  const int N = M.N, n = 2 * N;
  const double tol = 1e-8;                                     // m_tol, PLDPSolver.cpp:49

  // ComputeInitialSolution :287-340
  double vk = 0.0, Dl = 0.0;
  if (lane < n) {
    Dl = D_g[lane];
    const int i = lane < N ? lane : lane - N;
    const int off = lane < N ? 0 : N;
    const int j0 = lane < N ? 0 : 3;
    double acc = 0.0;
    for (int j = j0; j < j0 + 3; j++) acc -= M.iPuPx[lane * 6 + j] * xkyk[j];
    if (!starting) {
      for (int j = 0; j < N - 1; j++) acc += M.iPu[j * N + i] * st->prev_zmp[j + off + 1];
      acc += M.iPu[(N - 1) * N + i] * zmpref[N - 1 + off];
    } else {
      for (int j = 0; j < N; j++) acc += M.iPu[j * N + i] * zmpref[j + off];
    }
    vk = acc;
  }
  // hot start :777-791
  int S = 0;
  const int n_prev = st->n_prev;
  int rc = 0;
  for (int i = 0; i < n_prev; i++) {
    const int lindex = st->prev_active[i] - n_removed;
    if (lindex >= 0) {
      if (S >= W.cap || lindex >= m) { rc = WG_PLDP_CAPACITY; break; }
      pldp_add_row<ST>(W, n, S, lindex, lane);
      S++;
    }
  }
  if (lane == 0) {
    st->n_prev = 0;                                            // m_PreviouslyActivatedConstraints.clear(), :792
    if (starting) st->internal_time = 0.0;                     // :667-668
  }
  int it = 0;
  bool go = (rc == 0);
  double dl = 0.0;
  while (go) {
    const double cl = -Dl - vk;                                // :806-808
    if (lane < n) { W.c[lane] = cl; W.Vk[lane] = vk; }
    WG_WSYNC();
    // v1 :455-470 and forward substitution :342-365
    double acc = 0.0;
    if (lane < S) {
      const int row = W.act[lane];
      for (int lj = 0; lj < n; lj++) acc += W.a<ST>(row, lj) * W.c[lj];
    }
    for (int k = 0; k < S; k++) {
      double yk = 0.0;
      if (lane == k) {
        const double lkk = W.L[ltri(k, k)];
        if (lkk != 0.0) acc /= lkk;
        yk = acc;
      }
      yk = rl(yk, k);
      if (lane > k && lane < S) acc += -W.L[ltri(lane, k)] * yk;
    }
    // backward substitution :367-400 (acc = y[lane]); lane i runs when v2[i+1..S-1] are in LDS
    for (int i = S - 1; i >= 0; i--) {
      if (lane == i) {
        double v = acc;
        for (int k = i + 1; k < S; k++) v -= W.L[ltri(k, i)] * W.v2[k];
        v = v / W.L[ltri(i, i)];
        W.v2[i] = v;
      }
      WG_WSYNC();
    }
    // d = c - E' v2 :509-519
    if (lane < n) {
      double dd = cl;
      for (int lj = 0; lj < S; lj++) dd -= W.a<ST>(W.act[lj], lane) * W.v2[lj];
      dl = dd;
      W.d[lane] = dd;
    }
    WG_WSYNC();
    // ComputeAlpha :534-653
    double best = 0.0; int bidx = -1;
    {
      // pass 1: A_i d for every row that does not reuse a similar row's value
      bool pending[2] = {false, false};
      double t1[2] = {0.0, 0.0};
      bool cand[2] = {false, false};
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int li = lane + 64 * h;
        if (li < m && !W.state[li]) {
          cand[h] = true;
          const int sim = W.similar[li];
          // m_ConstraintsValueComputed[li+sim] is true exactly when that (earlier) row is not active
          if (sim != 0 && !W.state[li + sim]) pending[h] = true;
          else {
            double s = 0.0;
            for (int lj = 0; lj < n; lj++) s += W.a<ST>(li, lj) * W.d[lj];
            t1[h] = s;
            W.tmp1[li] = s;
          }
        }
      }
      // resolve reuse chains (depth 1 with the reference's SimilarConstraints); state 2 marks "value still pending"
#pragma unroll
      for (int h = 0; h < 2; h++) if (pending[h]) W.state[lane + 64 * h] = 2;
      WG_WSYNC();
      while (__ballot(pending[0] || pending[1])) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int li = lane + 64 * h;
          if (pending[h]) {
            const int src = li + W.similar[li];
            if (W.state[src] == 0) { t1[h] = -W.tmp1[src]; W.tmp1[li] = t1[h]; pending[h] = false; }
          }
        }
        WG_WSYNC();
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int li = lane + 64 * h;
          if (cand[h] && !pending[h] && W.state[li] == 2) W.state[li] = 0;
        }
        WG_WSYNC();
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int li = lane + 64 * h;
        if (cand[h] && t1[h] < 0.0) {
          double t2 = -W.b[li];
          for (int lj = 0; lj < n; lj++) t2 -= W.a<ST>(li, lj) * W.Vk[lj];
          if (t2 > tol) { /* reference prints "PB ON constraint" */ }
          else if (t2 > 0.0) t2 = -tol;
          const double la = t2 / t1[h];
          if (bidx < 0 ? true : (best > la)) { if (!(la != la)) { best = la; bidx = li; } }
        }
      }
    }
    wave_argmin_first(best, bidx);
    double alpha = 10000000.0;
    int add = -1;
    if (bidx >= 0 && alpha > best) { alpha = best; if (alpha < 1) add = bidx; }
    if (alpha >= 1.0) { alpha = 1.0; go = false; }
    if (alpha < 0.0) { rc = WG_PLDP_NEG_ALPHA; break; }
    vk = vk + alpha * dl;                                      // :841-844
    if (go && add >= 0) {
      if (S >= W.cap) { rc = WG_PLDP_CAPACITY; break; }
      pldp_add_row<ST>(W, n, S, add, lane);
      S++;
    }
    if (max_iter > 0 && it + 1 >= max_iter) go = false;
    it++;
  }
  if (lane < n) W.Vk[lane] = vk;
  WG_WSYNC();
  if (rc == 0) {
    // rows kept for the next hot start :959-968, in activation order
    const bool keep = lane < S && W.v2[lane] < 0.0;
    const unsigned long long mask = __ballot(keep);
    if (keep) st->prev_active[__popcll(mask & ((1ull << lane) - 1ull))] = W.act[lane];
    if (lane == 0) st->n_prev = __popcll(mask);
    // StoreCurrentZMPSolution :1010-1036
    if (lane < n) {
      const int i = lane < N ? lane : lane - N;
      const int off = lane < N ? 0 : N;
      double z = 0.0;
      for (int j = 0; j < N; j++) z += M.Pu[j * N + i] * W.Vk[j + off];
      for (int j = 0; j < 3; j++) z += M.Px[i * 3 + j] * xkyk[j + (lane < N ? 0 : 3)];
      st->prev_zmp[lane] = z;
    }
    const double x0 = W.Vk[0], xN = W.Vk[N];
    const bool bad = (x0 != x0) || (xN != xN) || fabs(x0) == INFINITY || fabs(xN) == INFINITY;
    if (bad) rc = WG_PLDP_NAN;
    else if (lane == 0) st->internal_time += 0.02;
  }
  S_out = S;
  it_out = it;
  return rc;
}


// kernel body: stage problem `b` into LDS, solve, write the outputs
template <bool kALds = true>                               // A staged in LDS, or read in place from global memory (L2)
__device__ __forceinline__ void pldp_problem(const PldpModel &M, unsigned char *lds, int mcap, int m, const double *__restrict__ D,
                             const double *__restrict__ A, const double *__restrict__ bvec,
                             const double *__restrict__ zmpref, const double *__restrict__ xkyk,
                             const int *__restrict__ similar, int n_removed, int starting, int max_iter,
                             wg_pldp_state_t *st, double *X, int *ret, int *n_iter, int *active, int *n_active) {
  const int lane = threadIdx.x;
  const int n = 2 * M.N;
  PldpLds W;
  W.carve(lds, mcap, WG_PLDP_ACTIVE_CAP, false, WG_PLDP_N, kALds);
  bool bad = false;
  for (int li = lane; li < m; li += 64) {
    const int sim = similar[li];
    W.similar[li] = (short)sim;
    W.state[li] = 0;
    W.b[li] = bvec[li];
    if (sim > 0 || li + sim < 0) bad = true;
  }
  const int ldg = m + 1;
  if constexpr (kALds) {
    for (int col = 0; col < n; col++)
      for (int row = lane; row < m; row += 64) W.A[row + col * W.lda] = A[row + col * ldg];
  } else {                                         // the solver only reads A: in place, the reference's own layout
    W.A = const_cast<double *>(A);
    W.lda = ldg;
  }
  if (lane < W.cap) W.v2[lane] = 0.0;
  WG_WSYNC();
  int S = 0, it = 0, rc;
  if (__ballot(bad)) rc = WG_PLDP_BAD_INPUT;
  else rc = pldp_solve<false>(M, W, m, D, zmpref, xkyk, n_removed, starting, max_iter, st, S, it);
  if (lane < n) X[lane] = W.Vk[lane];
  if (lane == 0) {
    *ret = rc;
    if (n_iter) *n_iter = it;
    if (n_active) *n_active = S;
  }
  if (active && lane < S) active[lane] = W.act[lane];
  WG_WSYNC();
}

}  // namespace wg
