// wg_dimitrov_device.hpp -- one pass of the Dimitrov-2008 receding-horizon loop per wavefront (gfx950):
// constraint matrices from ZMP polytopes, cost vector, PLDP solve (wg_pldp_device.hpp), un-preconditioning, LIPM.
// Restates src/ZMPRefTrajectoryGeneration/ZMPConstrainedQPFastFormulation.cpp:759-1022, 1254-1262, 1322-1392 and
// src/PreviewControl/LinearizedInvertedPendulum2D.cpp:157-264 with the same operation order; lanes run over
// independent outputs only.
#pragma once
#include "wg_pldp_device.hpp"

namespace wg {

struct DimitrovConst {           // device-resident (wg_dimitrov_configure)
  int N, solver;
  double T, Tctrl, h;
  double OptB[2 * WG_PLDP_N * 6], OptC[2 * WG_PLDP_N * 2 * WG_PLDP_N], iLQ[2 * WG_PLDP_N * 2 * WG_PLDP_N];
  PldpModel pldp;
  // mode QLD (m_FastFormulationMode == QLD): the problem as InitConstants builds it without the LQ preconditioning --
  // Qq = m_Q = OptA in ql0001_'s column-major layout (:382-389), OptBq / OptCq the cost terms before iLQ is applied (:545-556),
  // PuTq = Pu' as :629-637 fills it (PuTq[k N + i], k <= i)
  double Qq[2 * WG_PLDP_N * 2 * WG_PLDP_N], OptBq[2 * WG_PLDP_N * 6], OptCq[2 * WG_PLDP_N * 2 * WG_PLDP_N], PuTq[WG_PLDP_N * WG_PLDP_N];
};

// host: the constants of InitConstants (ZMPConstrainedQPFastFormulation.cpp:158-246, 384-560, 597-690), row-major.
// Matrix products are plain i-j-k loops with k ascending (what ublas prod does on dense matrices).
struct DimitrovHost {
  static void matmul(const double *A, const double *B, double *C, int r, int inner, int c) {
    for (int i = 0; i < r; i++)
      for (int j = 0; j < c; j++) {
        double s = 0.0;
        for (int k = 0; k < inner; k++) s += A[i * inner + k] * B[k * c + j];
        C[i * c + j] = s;
      }
  }
  // false if the LQ factor or the inverse of Pu does not exist
  static bool build(const wg_dimitrov_model_t &m, DimitrovConst &K) {
    const unsigned N = (unsigned)m.N, n = 2 * N;
    const double T = m.T;
    std::vector<double> PPu(n * n, 0.0), VPu(n * n, 0.0), PPx(n * 6, 0.0), VPx(n * 6, 0.0), Px(N * 3);
    for (unsigned i = 0; i < N; i++) {                                 // :165-211
      VPx[i * 6 + 1] = 1.0; VPx[i * 6 + 2] = (i + 1) * T;
      VPx[(i + N) * 6 + 4] = 1.0; VPx[(i + N) * 6 + 5] = (i + 1) * T;
      PPx[i * 6 + 0] = 1.0; PPx[i * 6 + 1] = (i + 1) * T; PPx[i * 6 + 2] = (i + 1) * (i + 1) * T * T * 0.5;
      PPx[(i + N) * 6 + 3] = 1.0; PPx[(i + N) * 6 + 4] = (i + 1) * T; PPx[(i + N) * 6 + 5] = (i + 1) * (i + 1) * T * T * 0.5;
      for (unsigned j = 0; j <= i; j++) {
        const double v = (2 * (i - j) + 1) * T * T * 0.5;
        const double p = (1 + 3 * (i - j) + 3 * (i - j) * (i - j)) * T * T * T / 6.0;
        VPu[i * n + j] = v; VPu[(i + N) * n + j + N] = v;
        PPu[i * n + j] = p; PPu[(i + N) * n + j + N] = p;
      }
    }
    for (unsigned li = 0; li < N; li++) {                              // :214-220
      Px[li * 3 + 0] = 1.0;
      Px[li * 3 + 1] = (double)(1.0 + li) * T;
      Px[li * 3 + 2] = (li + 1.0) * (li + 1.0) * T * T * 0.5 - m.com_height / 9.81;
    }
    auto transpose = [](const std::vector<double> &A, int r, int c) {
      std::vector<double> t((size_t)r * c);
      for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) t[(size_t)j * r + i] = A[(size_t)i * c + j];
      return t;
    };
    const std::vector<double> PPuT = transpose(PPu, n, n), VPuT = transpose(VPu, n, n);
    // OptA = Id + beta PPu'PPu + alpha VPu'   (the reference scales VPu', not VPu'VPu: :524-527)
    std::vector<double> tmp(n * n), OptA(n * n);
    matmul(PPuT.data(), PPu.data(), tmp.data(), n, n, n);
    for (unsigned i = 0; i < n; i++)
      for (unsigned j = 0; j < n; j++)
        OptA[i * n + j] = ((i == j ? 1.0 : 0.0) + m.beta * tmp[i * n + j]) + m.alpha * VPuT[i * n + j];
    // linear part: OptB = alpha VPu'VPx + beta PPu'PPx ; OptC = beta PPu'   (:545-556)
    std::vector<double> l1(n * 6), OptB(n * 6), OptC(n * n);
    matmul(PPuT.data(), PPx.data(), l1.data(), n, n, 6);
    matmul(VPuT.data(), VPx.data(), OptB.data(), n, n, 6);
    for (unsigned i = 0; i < n * 6; i++) OptB[i] = m.alpha * OptB[i] + m.beta * l1[i];
    for (unsigned i = 0; i < n * n; i++) OptC[i] = m.beta * PPuT[i];
    // LQ factor of the leading N x N block and its inverse, OptCholesky::ComputeNormalCholeskyOnANormal /
    // ComputeInverseCholeskyNormal (OptCholesky.cpp:225-302): only the lower triangle of OptA is read  (:384-420)
    std::vector<double> LQ(N * N, 0.0), iLQ1(N * N, 0.0);
    for (unsigned li = 0; li < N; li++)
      for (unsigned lj = 0; lj <= li; lj++) {
        double r = OptA[li * n + lj];
        for (unsigned lk = 0; lk < lj; lk++) r = r - LQ[li * N + lk] * LQ[lj * N + lk];
        if (lj != li) LQ[li * N + lj] = r / LQ[lj * N + lj];
        else { if (!(r > 0.0)) return false; LQ[li * N + lj] = sqrt(r); }
      }
    for (int lj = (int)N - 1; lj >= 0; lj--) {
      const double d = 1 / LQ[lj * N + lj];
      iLQ1[lj * N + lj] = d;
      for (int li = lj + 1; li < (int)N; li++) {
        double r = 0.0;
        for (int lk = lj + 1; lk < (int)N; lk++) r = r + iLQ1[li * N + lk] * LQ[lk * N + lj];
        iLQ1[li * N + lj] = -d * r;
      }
    }
    std::vector<double> iLQ(n * n, 0.0);                               // :448-461
    for (unsigned i = 0; i < N; i++)
      for (unsigned j = 0; j < N; j++) { iLQ[i * n + j] = iLQ1[i * N + j]; iLQ[(i + N) * n + j + N] = iLQ1[i * N + j]; }
    std::vector<double> OptB2(n * 6), OptC2(n * n);                    // :464-467
    matmul(iLQ.data(), OptB.data(), OptB2.data(), n, n, 6);
    matmul(iLQ.data(), OptC.data(), OptC2.data(), n, n, n);
    // constraint side: Pu' (transposed storage), m_Pu = iLQ Pu', iPu = inverse  (:597-690)
    std::vector<double> PuT(N * N, 0.0), Pu(N * N), iPu(N * N);
    for (unsigned i = 0; i < N; i++)
      for (unsigned k = 0; k <= i; k++)
        PuT[k * N + i] = ((1 + 3 * (i - k) + 3 * (i - k) * (i - k)) * T * T * T / 6.0 - T * m.com_height / 9.81);
    for (unsigned i = 0; i < N; i++)
      for (unsigned j = 0; j < N; j++) {
        double s = 0;
        for (unsigned k = 0; k < N; k++) s += iLQ[i * n + k] * PuT[k * N + j];
        Pu[i * N + j] = s;
      }
    {  // MAL_INVERSE (jrl-mal -> LAPACK, unpinned): Gauss-Jordan with partial pivoting
      std::vector<double> a(Pu), inv(N * N, 0.0);
      for (unsigned i = 0; i < N; i++) inv[i * N + i] = 1.0;
      for (unsigned c = 0; c < N; c++) {
        unsigned p = c; double best = fabs(a[c * N + c]);
        for (unsigned r = c + 1; r < N; r++) if (fabs(a[r * N + c]) > best) { best = fabs(a[r * N + c]); p = r; }
        if (!(best > 0.0)) return false;
        if (p != c) for (unsigned j = 0; j < N; j++) { std::swap(a[c * N + j], a[p * N + j]); std::swap(inv[c * N + j], inv[p * N + j]); }
        const double d = a[c * N + c];
        for (unsigned j = 0; j < N; j++) { a[c * N + j] /= d; inv[c * N + j] /= d; }
        for (unsigned r = 0; r < N; r++) {
          if (r == c) continue;
          const double f = a[r * N + c];
          if (f == 0.0) continue;
          for (unsigned j = 0; j < N; j++) { a[r * N + j] -= f * a[c * N + j]; inv[r * N + j] -= f * inv[c * N + j]; }
        }
      }
      iPu = inv;
    }
    memset(&K, 0, sizeof K);
    K.N = (int)N; K.T = T; K.Tctrl = m.Tctrl; K.h = m.com_height; K.solver = m.solver;
    // mode QLD: BuildingConstantPartOfTheObjectiveFunctionQLD (:382-389) stores m_Q[i 2N + j] = OptA(j, i), which ql0001_ reads as
    // the column-major matrix C(j, i) = OptA(j, i); OptB / OptC stay as built; Pu' is the raw table
    for (unsigned i = 0; i < n; i++)
      for (unsigned j = 0; j < n; j++) K.Qq[i * n + j] = OptA[j * n + i];
    memcpy(K.OptBq, OptB.data(), sizeof(double) * n * 6);
    memcpy(K.OptCq, OptC.data(), sizeof(double) * n * n);
    memcpy(K.PuTq, PuT.data(), sizeof(double) * N * N);
    memcpy(K.OptB, OptB2.data(), sizeof(double) * n * 6);
    memcpy(K.OptC, OptC2.data(), sizeof(double) * n * n);
    memcpy(K.iLQ, iLQ.data(), sizeof(double) * n * n);
    K.pldp.N = (int)N;
    memcpy(K.pldp.iPu, iPu.data(), sizeof(double) * N * N);
    memcpy(K.pldp.Pu, Pu.data(), sizeof(double) * N * N);
    memcpy(K.pldp.Px, Px.data(), sizeof(double) * N * 3);
    for (unsigned i = 0; i < N; i++)                                   // PLDPSolver::PrecomputeiPuPx
      for (unsigned j = 0; j < 3; j++) {
        double s = 0.0;
        for (unsigned k = 0; k < N; k++) s += iPu[k * N + i] * Px[k * 3 + j];
        K.pldp.iPuPx[i * 6 + j] = s;
        K.pldp.iPuPx[(i + N) * 6 + j + 3] = s;
      }
    return true;
  }
};

constexpr int kDimitrovActiveCap = 40;   // E E' is singular beyond 2N = 32 active rows

// one gait, one tick.  LDS: the structured PLDP work area (mcap = 8N) + 4 * 2N doubles (D, zmpref, NewX, X) + 8 (xk).
__device__ int dimitrov_tick(const DimitrovConst &K, unsigned char *lds, const wg_zmp_polytope_t *__restrict__ polys,
                             wg_dimitrov_state_t *st, wg_dimitrov_out_t *out, int max_iter) {   // returns PLDP's iteration count
  const int lane = threadIdx.x;
  const int N = K.N, n = 2 * N, mcap = WG_POLY_MAX_ROWS * N;
  PldpLds W;
  W.carve(lds, mcap, kDimitrovActiveCap, true, N);
  double *D = reinterpret_cast<double *>(lds + PldpLds::bytes(mcap, kDimitrovActiveCap, true));
  double *zr = D + 2 * WG_PLDP_N, *NewX = zr + 2 * WG_PLDP_N, *X = NewX + 2 * WG_PLDP_N, *xk = X + 2 * WG_PLDP_N;
  int *rowbase = reinterpret_cast<int *>(xk + 8);                 // [N+1] first row of each instant
  if (lane < 6) xk[lane] = st->xk[lane];
  // ---- BuildConstraintMatrices :759-1022 ----
  if (lane == 0) {
    int idx = 0;
    for (int i = 0; i < N; i++) { rowbase[i] = idx; int r = polys[i].nrows; r = r < 0 ? 0 : (r > WG_POLY_MAX_ROWS ? WG_POLY_MAX_ROWS : r); idx += r; }
    rowbase[N] = idx;
  }
  WG_WSYNC();
  const int m = rowbase[N];
  bool bad = false;
  for (int e = lane; e < N * WG_POLY_MAX_ROWS; e += 64) {
    const int i = e / WG_POLY_MAX_ROWS, j = e % WG_POLY_MAX_ROWS;
    const int nr = rowbase[i + 1] - rowbase[i];
    if (j < nr) {
      const int idx = rowbase[i] + j;
      const double a0 = polys[i].A[j][0], a1 = polys[i].A[j][1];
      const double *px = K.pldp.Px + i * 3;
      W.b[idx] = (xk[0] * px[0] + xk[1] * px[1] + xk[2] * px[2]) * a0 + (xk[3] * px[0] + xk[4] * px[1] + xk[5] * px[2]) * a1 +
                 polys[i].B[j];
      const int sim = polys[i].similar[j];
      W.similar[idx] = (short)sim;
      W.state[idx] = 0;
      if (sim > 0 || idx + sim < 0) bad = true;
      W.c0[idx] = a0; W.c1[idx] = a1; W.slot[idx] = i;          // DPu(idx, k) = a0 Pu[k][i], DPu(idx, k+N) = a1 Pu[k][i]
    }
  }
  if (lane < N) { zr[lane] = polys[lane].centre[0]; zr[lane + N] = polys[lane].centre[1]; }
  for (int e = lane; e < N * N; e += 64) W.PuL[e] = K.pldp.Pu[e];
  if (lane < W.cap) W.v2[lane] = 0.0;
  WG_WSYNC();
  // ---- D = OptB xk - OptC ZMPRef :1254-1262 ----
  if (lane < n) {
    double l1 = 0.0, od = 0.0;
    for (int j = 0; j < n; j++) l1 += K.OptC[lane * n + j] * zr[j];
    for (int j = 0; j < 6; j++) od += K.OptB[lane * 6 + j] * xk[j];
    D[lane] = od - l1;
  }
  WG_WSYNC();
  int S = 0, it = 0, rc;
  const int first_rows = rowbase[1] - rowbase[0];                  // NextNumberOfRemovedConstraints :823
  if (__ballot(bad)) rc = WG_PLDP_BAD_INPUT;
  else rc = pldp_solve<true>(K.pldp, W, m, D, zr, xk, st->n_removed, st->starting, max_iter, &st->pldp, S, it);
  if (lane < n) X[lane] = W.Vk[lane];
  WG_WSYNC();
  // ---- X <- iLQ' X :1355-1381 ----
  if (lane < n) {
    double s = 0.0;
    for (int j = lane; j < n; j++) s += K.iLQ[j * n + lane] * X[j];
    NewX[lane] = s;
  }
  WG_WSYNC();
  const double jx = NewX[0], jy = NewX[N];
  if (out && lane < 2 * WG_PLDP_N) out->X[lane] = lane < n ? NewX[lane] : 0.0;
  if (rc == 0) {
    // ---- LinearizedInvertedPendulum2D::Interpolation :157-227 (lk = 0..interval) ----
    if (out && lane <= WG_SAMPLES_PER_TICK) {
      const double t = (lane + 1) * K.Tctrl;
      const double c02 = -K.h / 9.81;
      const double cx0 = xk[0] + t * xk[1] + 0.5 * t * t * xk[2] + t * t * t * jx / 6.0;
      const double cx1 = xk[1] + t * xk[2] + 0.5 * t * t * jx;
      const double cx2 = xk[2] + t * jx;
      const double cy0 = xk[3] + t * xk[4] + 0.5 * t * t * xk[5] + t * t * t * jy / 6.0;
      const double cy1 = xk[4] + t * xk[5] + 0.5 * t * t * jy;
      const double cy2 = xk[5] + t * jy;
      out->com_x[lane][0] = cx0; out->com_x[lane][1] = cx1; out->com_x[lane][2] = cx2;
      out->com_y[lane][0] = cy0; out->com_y[lane][1] = cy1; out->com_y[lane][2] = cy2;
      out->zmp_x[lane] = 1.0 * cx0 + 0.0 * cx1 + c02 * cx2;
      out->zmp_y[lane] = 1.0 * cy0 + 0.0 * cy1 + c02 * cy2;
    }
    // ---- OneIteration :230-264 ----
    if (lane < 2) {
      const double T = K.T;
      const double A01 = T, A02 = T * T / 2.0, A12 = T, B0 = T * T * T / 6.0, B1 = T * T / 2.0, B2 = T;
      const double u = lane == 0 ? jx : jy;
      const double *c = xk + 3 * lane;
      const double n0 = 0.0 + 1.0 * c[0] + A01 * c[1] + A02 * c[2];
      const double n1 = 0.0 + 0.0 * c[0] + 1.0 * c[1] + A12 * c[2];
      const double n2 = 0.0 + 0.0 * c[0] + 0.0 * c[1] + 1.0 * c[2];
      st->xk[3 * lane + 0] = n0 + u * B0; st->xk[3 * lane + 1] = n1 + u * B1; st->xk[3 * lane + 2] = n2 + u * B2;
    }
  }
  if (lane == 0) {
    st->starting = 0;                                              // :1339
    st->n_removed = first_rows;                                    // :1340
    if (out) { out->jerk_x = jx; out->jerk_y = jy; out->ret = rc; out->n_iter = it; out->n_active = S; out->m = m; }
  }
  WG_WSYNC();
  return it;
}

// ---- the same tick with ql0001_ as the back-end (m_FastFormulationMode == QLD / QLDANDLQ, :1297-1320) --------------------------
// QLD: the problem as it stands -- Q = OptA (full matrix, iwar[0] = 1), D = OptB xk - OptC ZMPRef (constants as built: no LQ
//   preconditioning), DPu(idx, k) = A_j0 Pu'[k N + i] and DPu(idx, k + N) = A_j1 Pu'[k N + i] for k <= i, zero beyond (:891-904:
//   "In this case, Pu is triangular"), X[0], X[N] are the jerks (:1383).
// QLDANDLQ: the preconditioned problem PLDP gets -- Q = I (handed to ql0001_ as its own Cholesky factor, iwar[0] = 0), D from the
//   iLQ-premultiplied OptB / OptC, DPu(idx, k) = A_j0 (iLQ Pu')[k N + i] for every k (:905-918), X <- iLQ' X (:1355-1381).
// DPx, the bounds -+1e8 and eps = 1e-8 are the same in every mode.
// The in-wave ql0002 (wg_ql_device.hpp) solves it through a problem view that never materialises DPu: a row is its two polytope
// coefficients and its instant, an element their product with one entry of the Pu table -- the value the dense matrix holds.
// ql0002 is instantiated for a full Hessian (lql): on the identity the reference's factor-given branch (iwar[0] = 0) returns the
// same bits -- held against the compiled qld.cpp called the driver's way by tests/test_dimitrov_gpu.py.
template <bool kLQ>                                       // kLQ: mode QLDANDLQ (identity Hessian, full rows)
struct DimitrovQldProb {
  static constexpr bool kCompact = false;
  static constexpr bool kNanExact = WG_TICK_NAN_EXACT != 0;   // NaN iterates end the way the reference ends them (scan_nan_exact)
  static constexpr bool kHasFactor = false;
  static constexpr bool kRowOps = false;
  static constexpr bool kWideN = false;
  static constexpr int kFixedLdz = 0;                     // run-time carve (n = 2N varies with the model)
  static constexpr int kNM = 2 * WG_PLDP_N;               // n = 2N <= 32: the compile-time-bounded forms of the solver
  // Read-only operands stay where they are (global memory, L1-resident: the polytopes of the gait, the model's Pu table): the
  // rows the scan walks are in registers (below), so these are read once per solve (row norms, load_rows, residual refresh) --
  // and the 4 KB of LDS they took as copies are what separates six from eight gaits per CU.
  const wg_zmp_polytope_t *polys;                         // global: the gait's N polytopes (row k = face k - rowbase[i] of instant i)
  const int *slot;                                        // LDS: the row's instant
  const int *rowbase;                                     // LDS: first row of every instant
  const double *PuT;                                      // global: the Pu table, N x N (entry [k N + i]; QLD: Pu', zero for k > i)
  int N;
  __device__ __forceinline__ double coef(int k, int inst, int axis) const { return polys[inst].A[k - rowbase[inst]][axis]; }
  __device__ __forceinline__ double G(const QlView &q, int i, int j) const {
    if constexpr (kLQ) return i == j ? q.Gdiag[i] : 0.0;    // m_Q = identity (:537-540)
    else {
      const double g = q.G[i + j * q.ldg], dg = q.Gdiag[i];   // Q read in place (a constant of the model), its diagonal in LDS
      return i == j ? dg : g;
    }
  }
  __device__ __forceinline__ double Gd(const QlView &q, int i) const { return q.Gdiag[i]; }
  __device__ __forceinline__ void setGd(const QlView &q, int i, double v) const { q.Gdiag[i] = v; }
  __device__ __forceinline__ double xl(const QlView &, int) const { return -1e8; }
  __device__ __forceinline__ double xu(const QlView &, int) const { return 1e8; }
  __device__ __forceinline__ double A(const QlView &, int k, int i) const {
    const bool second = i >= N;
    const int kk = second ? i - N : i;
    const int inst = slot[k];
    const double a = coef(k, inst, second ? 1 : 0);
    if constexpr (kLQ) return a * PuT[kk * N + inst];
    else {
      const double pu = PuT[kk * N + (kk <= inst ? inst : kk)];      // clamped address, selected value
      return kk <= inst ? a * pu : 0.0;                              // memset zero beyond the triangle (:782)
    }
  }
  // m <= WG_PLDP_MMAX = 128: the two rows a lane owns (lane, 64 + lane) as their n <= 32 entries in registers, formed ONCE per solve
  // from the same expression as A() -- the violation scan then reads nothing but x, the new normal is written out by its owner
  // (the register-row interface of the solver, like DenseRegProb)
  static constexpr bool kRegRows = true;
  double ar0[kNM], ar1[kNM];
  __device__ __forceinline__ void load_rows(const QlView &q, int lane) {
    const int m = q.m, n = q.n;
    if (m <= 0) {                              // every polytope empty: no row to form (slot[] holds nothing: not even slot[0])
#pragma unroll
      for (int i = 0; i < kNM; ++i) { ar0[i] = 0.0; ar1[i] = 0.0; }
      return;
    }
    const int k0 = lane < m ? lane : m - 1, k1 = lane + 64 < m ? lane + 64 : m - 1;
    const int i0 = slot[k0], i1 = slot[k1];
    const double a00 = coef(k0, i0, 0), a01 = coef(k0, i0, 1), a10 = coef(k1, i1, 0), a11 = coef(k1, i1, 1);
    // the element expression of A(), with the row's (instant, coefficients) fetched once
#pragma unroll
    for (int i = 0; i < kNM; ++i) {
      const int ic = i < n ? i : n - 1;
      const bool second = ic >= N;
      const int kk = second ? ic - N : ic;
      if constexpr (kLQ) {
        ar0[i] = (second ? a01 : a00) * PuT[kk * N + i0];
        ar1[i] = (second ? a11 : a10) * PuT[kk * N + i1];
      } else {
        const double p0 = PuT[kk * N + (kk <= i0 ? i0 : kk)], p1 = PuT[kk * N + (kk <= i1 ? i1 : kk)];
        ar0[i] = kk <= i0 ? (second ? a01 : a00) * p0 : 0.0;
        ar1[i] = kk <= i1 ? (second ? a11 : a10) * p1 : 0.0;
      }
    }
  }
  __device__ __forceinline__ void row_to(const QlView &q, int k, double *dst, int lane) const {
    const int n = q.n;
    if (lane == (k & 63)) {
#pragma unroll
      for (int i = 0; i < kNM; ++i)
        if (i < n) dst[i] = k < 64 ? ar0[i] : ar1[i];
    }
  }
};

constexpr int kDimQldNsc = 2 * WG_PLDP_N;
__host__ __device__ inline size_t dimitrov_qld_ql_bytes() {
  return (QlDims(2 * WG_PLDP_N, WG_PLDP_MMAX, WG_PLDP_MMAX, true, false, kDimQldNsc, false, true, true, true, 0, false).bytes() + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t dimitrov_qld_lds_bytes() {
  // solver area | zr (2N) | NewX (2N) | xk (8) | slot (mcap ints) | rowbase (N + 1 ints)
  return dimitrov_qld_ql_bytes() + 8 * (size_t)(4 * WG_PLDP_N + 8) +
         4 * (size_t)(WG_PLDP_MMAX + ((WG_PLDP_N + 2) & ~1)) + 16;
}

template <bool kLQ>                                          // returns ql0002's iteration count (the launcher's start order, nothing else)
__device__ int dimitrov_qld_tick(const DimitrovConst &K, double *lds, const wg_zmp_polytope_t *__restrict__ polys,
                                 wg_dimitrov_state_t *st, wg_dimitrov_out_t *out) {
  const int lane = wg_lane();
  const int N = K.N, n = 2 * N, mcap = WG_PLDP_MMAX;
  double *t = reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + dimitrov_qld_ql_bytes());
  double *zr = t; t += 2 * WG_PLDP_N;
  double *NewX = t; t += 2 * WG_PLDP_N;
  double *xk = t; t += 8;
  int *slot = reinterpret_cast<int *>(t);
  int *rowbase = slot + mcap;
  if (lane < 6) xk[lane] = st->xk[lane];
  if (lane == 0) {
    int idx = 0;
    for (int i = 0; i < N; i++) { rowbase[i] = idx; int r = polys[i].nrows; r = r < 0 ? 0 : (r > WG_POLY_MAX_ROWS ? WG_POLY_MAX_ROWS : r); idx += r; }
    rowbase[N] = idx;
  }
  WG_WSYNC();
  const int m = uni(rowbase[N]);
  // the solver's view: n = 2N variables, m rows; Q read in place (or the identity), wa | b in LDS
  QlDims D(n, m, m, true, false, kDimQldNsc, false, true, true, true, 0, false);
  QlView q;
  q.template carve<false, true, true>(lds, D, 0);
  q.G = const_cast<double *>(K.Qq);
  q.ldg = n;
  const double *OptB = kLQ ? K.OptB : K.OptBq, *OptC = kLQ ? K.OptC : K.OptCq, *PuSrc = kLQ ? K.pldp.Pu : K.PuTq;
  // ---- BuildConstraintMatrices :759-1022 ----
  for (int e = lane; e < N * WG_POLY_MAX_ROWS; e += 64) {
    const int i = e / WG_POLY_MAX_ROWS, j = e % WG_POLY_MAX_ROWS;
    const int nr = rowbase[i + 1] - rowbase[i];
    if (j < nr) {
      const int idx = rowbase[i] + j;
      const double a0 = polys[i].A[j][0], a1 = polys[i].A[j][1];
      const double *px = K.pldp.Px + i * 3;
      const double dpx = (xk[0] * px[0] + xk[1] * px[1] + xk[2] * px[2]) * a0 + (xk[3] * px[0] + xk[4] * px[1] + xk[5] * px[2]) * a1 +
                         polys[i].B[j];
      q.b[idx] = -dpx;                                         // inner sign, qld.cpp:469-475
      slot[idx] = i;
    }
  }
  if (lane < N) { zr[lane] = polys[lane].centre[0]; zr[lane + N] = polys[lane].centre[1]; }
  if (lane < n) q.Gdiag[lane] = kLQ ? 1.0 : K.Qq[lane + lane * n];
  WG_WSYNC();
  // ---- D = OptB xk - OptC ZMPRef :1254-1262 ----
  if (lane < n) {
    double l1 = 0.0, od = 0.0;
    for (int j = 0; j < n; j++) l1 += OptC[lane * n + j] * zr[j];
    for (int j = 0; j < 6; j++) od += OptB[lane * 6 + j] * xk[j];
    q.d[lane] = od - l1;
  }
  DimitrovQldProb<kLQ> prob;
  prob.polys = polys; prob.slot = slot; prob.rowbase = rowbase; prob.PuT = PuSrc; prob.N = N;
  if (lane == 0 && fabs(prob.Gd(q, n - 1)) == 0.0) prob.setGd(q, n - 1, 1e-8);    // qld.cpp:442-444 (nmax == n)
  WG_WSYNC();
  const QlResult r = ql_solve(q, prob, 1e-8, nullptr, 0);
  WG_WSYNC();
  if constexpr (kLQ) {
    // ---- X <- iLQ' X :1355-1381 ----
    if (lane < n) {
      double s = 0.0;
      for (int j = lane; j < n; j++) s += K.iLQ[j * n + lane] * q.x[j];
      NewX[lane] = s;
    }
  } else if (lane < n) NewX[lane] = q.x[lane];
  WG_WSYNC();
  const double jx = NewX[0], jy = NewX[N];
  if (out && lane < 2 * WG_PLDP_N) out->X[lane] = lane < n ? NewX[lane] : 0.0;
  const int first_rows = rowbase[1] - rowbase[0];
  WG_WSYNC();
  if (r.ifail == 0) {
    // ---- LinearizedInvertedPendulum2D::Interpolation :157-227 (lk = 0..interval) ----
    if (out && lane <= WG_SAMPLES_PER_TICK) {
      const double tt = (lane + 1) * K.Tctrl;
      const double c02 = -K.h / 9.81;
      const double cx0 = xk[0] + tt * xk[1] + 0.5 * tt * tt * xk[2] + tt * tt * tt * jx / 6.0;
      const double cx1 = xk[1] + tt * xk[2] + 0.5 * tt * tt * jx;
      const double cx2 = xk[2] + tt * jx;
      const double cy0 = xk[3] + tt * xk[4] + 0.5 * tt * tt * xk[5] + tt * tt * tt * jy / 6.0;
      const double cy1 = xk[4] + tt * xk[5] + 0.5 * tt * tt * jy;
      const double cy2 = xk[5] + tt * jy;
      out->com_x[lane][0] = cx0; out->com_x[lane][1] = cx1; out->com_x[lane][2] = cx2;
      out->com_y[lane][0] = cy0; out->com_y[lane][1] = cy1; out->com_y[lane][2] = cy2;
      out->zmp_x[lane] = 1.0 * cx0 + 0.0 * cx1 + c02 * cx2;
      out->zmp_y[lane] = 1.0 * cy0 + 0.0 * cy1 + c02 * cy2;
    }
    // ---- OneIteration :230-264 ----
    if (lane < 2) {
      const double T = K.T;
      const double A01 = T, A02 = T * T / 2.0, A12 = T, B0 = T * T * T / 6.0, B1 = T * T / 2.0, B2 = T;
      const double u = lane == 0 ? jx : jy;
      const double *c = xk + 3 * lane;
      const double n0 = 0.0 + 1.0 * c[0] + A01 * c[1] + A02 * c[2];
      const double n1 = 0.0 + 0.0 * c[0] + 1.0 * c[1] + A12 * c[2];
      const double n2 = 0.0 + 0.0 * c[0] + 0.0 * c[1] + 1.0 * c[2];
      st->xk[3 * lane + 0] = n0 + u * B0; st->xk[3 * lane + 1] = n1 + u * B1; st->xk[3 * lane + 2] = n2 + u * B2;
    }
  }
  if (lane == 0) {
    st->starting = 0;
    st->n_removed = first_rows;
    if (out) { out->jerk_x = jx; out->jerk_y = jy; out->ret = r.ifail; out->n_iter = r.n_iter; out->n_active = r.nact; out->m = m; }
  }
  WG_WSYNC();
  return r.n_iter;
}

}  // namespace wg
