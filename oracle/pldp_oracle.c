/*
 * pldp_oracle.c -- TEST INFRASTRUCTURE ONLY (see wg_oracle.h).
 *
 * CPU restatement of the reference's Dimitrov-2008 back-end:
 *   OptCholesky   /root/reference/src/Mathematics/OptCholesky.cpp
 *   PLDPSolver    /root/reference/src/Mathematics/PLDPSolver.cpp
 * Same operations in the same order (every inner sum ascending like the reference's loops), so results are meant to
 * be bit-identical to a build of those files.
 *
 * PARITY UNPINNED against a compiled reference: both translation units include <jrl/mal/matrixabstractlayer.hh>
 * (jrl-mal >= 1.9.0, CMakeLists.txt:45), which is not in this image, so they are unbuildable here.  The pins that exist
 * are the reference's own self-checking test tests/TestOptCholesky.cpp (srand(0), 12 x 15, |A A' - L L'|_F <= 1e-6),
 * replayed in tests/test_pldp_oracle.py, and the optimality (KKT) conditions of the QP the solver claims to solve.
 *
 * Deliberate deviations, all at places where the reference is not reproducible:
 *   - the 1.3 ms gettimeofday budget (PLDPSolver.cpp:51-52, 889-900) is an iteration cap (max_iter);
 *   - alpha < 0 makes the reference call exit(0) (:833-838); here the solve returns WG_PLDP_NEG_ALPHA;
 *   - m_ConstraintsValueComputed is uninitialised heap memory in the reference (:139); the flags are only ever read
 *     for rows below the current one when SimilarConstraint offsets are negative (what FindSimilarConstraints emits),
 *     so positive offsets are rejected instead of reading stale data;
 *   - diagnostic printing and file dumps are dropped.
 */
#include <math.h>
#include <string.h>

#include "wg_oracle.h"

/* ---- OptCholesky ---------------------------------------------------------------------------------------------- */

/* OptCholesky::UpdateCholeskyMatrixNormal, OptCholesky.cpp:123-169: the newest entry of set[] is row i of L.
 * A row-major, card_u doubles per row; L row-major with leading dimension ldl. */
void wgo_optchol_update_normal(const double *A, int card_u, const int *set, int nset, double *L, int ldl) {
  const int i = nset > 0 ? nset - 1 : 0;
  const double *arow_i = A + (size_t)card_u * set[i];
  for (int lj = 0; lj < nset; lj++) {
    const double *arow_j = A + (size_t)card_u * set[lj];
    double mij = 0.0;
    for (int lk = 0; lk < card_u; lk++) mij += arow_i[lk] * arow_j[lk];
    double r = mij;
    for (int lk = 0; lk < lj; lk++) r = r - L[i * ldl + lk] * L[lj * ldl + lk];
    if (lj != nset - 1) L[i * ldl + lj] = r / L[lj * ldl + lj];
    else L[i * ldl + lj] = sqrt(r);
  }
}

/* OptCholesky::UpdateCholeskyMatrixFortran, OptCholesky.cpp:171-223: A column-major, leading dimension m+1. */
void wgo_optchol_update_fortran(const double *A, int m, int card_u, const int *set, int nset, double *L, int ldl) {
  const int i = nset > 0 ? nset - 1 : 0;
  const int lda = m + 1;
  for (int lj = 0; lj < nset; lj++) {
    const double *pi = A + set[i], *pj = A + set[lj];
    double mij = 0.0;
    for (int lk = 0; lk < card_u; lk++) { mij += (*pi) * (*pj); pi += lda; pj += lda; }
    double r = mij;
    for (int lk = 0; lk < lj; lk++) r = r - L[i * ldl + lk] * L[lj * ldl + lk];
    if (lj != nset - 1) L[i * ldl + lj] = r / L[lj * ldl + lj];
    else L[i * ldl + lj] = sqrt(r);
  }
}

/* OptCholesky::ComputeNormalCholeskyOnANormal, OptCholesky.cpp:225-259 (A is n x n, row-major; only the lower
 * triangle of L is written) */
void wgo_chol_normal(const double *A, int n, double *L) {
  for (int li = 0; li < n; li++)
    for (int lj = 0; lj <= li; lj++) {
      double r = A[li * n + lj];
      for (int lk = 0; lk < lj; lk++) r = r - L[li * n + lk] * L[lj * n + lk];
      if (lj != li) L[li * n + lj] = r / L[lj * n + lj];
      else L[li * n + lj] = sqrt(r);
    }
}

/* OptCholesky::ComputeInverseCholeskyNormal, OptCholesky.cpp:261-302, over the leading `size` rows */
void wgo_chol_inverse(const double *L, int n, int size, double *iL) {
  for (int lj = size - 1; lj >= 0; lj--) {
    const double d = 1 / L[lj * n + lj];
    iL[lj * n + lj] = d;
    for (int li = lj + 1; li < size; li++) {
      double r = 0.0;
      for (int lk = lj + 1; lk < size; lk++) r = r + iL[li * n + lk] * L[lk * n + lj];
      iL[li * n + lj] = -d * r;
    }
  }
}

/* ---- PLDPSolver ----------------------------------------------------------------------------------------------- */

/* constructor + PrecomputeiPuPx, PLDPSolver.cpp:40-96, 208-285 */
int wgo_pldp_setup(wgo_pldp_model_t *M, int N, const double *iPu, const double *Px, const double *Pu) {
  if (N < 1 || N > WG_PLDP_N) return -1;
  memset(M, 0, sizeof(*M));
  M->N = N;
  memcpy(M->iPu, iPu, sizeof(double) * N * N);
  memcpy(M->Pu, Pu, sizeof(double) * N * N);
  memcpy(M->Px, Px, sizeof(double) * N * 3);
  for (int i = 0; i < N; i++)
    for (int j = 0; j < 3; j++) {
      M->iPuPx[i * 6 + j] = 0.0;
      M->iPuPx[i * 6 + j + 3] = 0.0;
      M->iPuPx[(i + N) * 6 + j] = 0.0;
      M->iPuPx[(i + N) * 6 + j + 3] = 0.0;
      for (int k = 0; k < N; k++) {
        const double tmp = iPu[k * N + i] * Px[k * 3 + j];
        M->iPuPx[i * 6 + j] += tmp;
        M->iPuPx[(i + N) * 6 + j + 3] += tmp;
      }
    }
  return 0;
}

typedef struct {
  const wgo_pldp_model_t *M;
  int m, lda, nact;
  const double *A, *b, *D;
  int act[WG_PLDP_MMAX];
  double Vk[2 * WG_PLDP_N], c[2 * WG_PLDP_N], d[2 * WG_PLDP_N];
  double v1[WG_PLDP_MMAX], v2[WG_PLDP_MMAX], y[WG_PLDP_MMAX], tmp1[WG_PLDP_MMAX], tmp2[WG_PLDP_MMAX];
  unsigned char computed[2 * WG_PLDP_MMAX];
  double L[WG_PLDP_MMAX * WG_PLDP_MMAX];
} pldp_work_t;

/* ComputeInitialSolution, PLDPSolver.cpp:287-340 (m_HotStart is always true, :46) */
static void initial_solution(pldp_work_t *w, const wg_pldp_state_t *st, const double *zmpref, const double *xkyk,
                             int starting) {
  const wgo_pldp_model_t *M = w->M;
  const int N = M->N;
  for (int i = 0; i < N; i++) {
    w->Vk[i] = 0.0;
    w->Vk[i + N] = 0.0;
    for (int j = 0; j < 3; j++) w->Vk[i] -= M->iPuPx[i * 6 + j] * xkyk[j];
    for (int j = 3; j < 6; j++) w->Vk[i + N] -= M->iPuPx[(i + N) * 6 + j] * xkyk[j];
    if (!starting) {
      for (int j = 0; j < N - 1; j++) w->Vk[i] += M->iPu[j * N + i] * st->prev_zmp[j + 1];
      w->Vk[i] += M->iPu[(N - 1) * N + i] * zmpref[N - 1];
      for (int j = 0; j < N - 1; j++) w->Vk[i + N] += M->iPu[j * N + i] * st->prev_zmp[j + N + 1];
      w->Vk[i + N] += M->iPu[(N - 1) * N + i] * zmpref[N - 1 + N];
    } else {
      for (int j = 0; j < N; j++) w->Vk[i] += M->iPu[j * N + i] * zmpref[j];
      for (int j = 0; j < N; j++) w->Vk[i + N] += M->iPu[j * N + i] * zmpref[j + N];
    }
  }
}

/* ComputeProjectedDescentDirection with Forward/BackwardSubstitution, PLDPSolver.cpp:342-532 */
static void projected_direction(pldp_work_t *w) {
  const int n = 2 * w->M->N, S = w->nact, ld = WG_PLDP_MMAX;
  for (int li = 0; li < S; li++) {
    w->v1[li] = 0.0;
    const int row = w->act[li];
    for (int lj = 0; lj < n; lj++) w->v1[li] += w->A[row + lj * w->lda] * w->c[lj];
  }
  for (int i = 0; i < S; i++) {
    w->y[i] = w->v1[i];
    for (int k = 0; k < i; k++) w->y[i] += -w->L[i * ld + k] * w->y[k];
    if (w->L[i * ld + i] != 0.0) w->y[i] /= w->L[i * ld + i];
  }
  for (int i = S - 1; i >= 0; i--) {
    w->v2[i] = w->y[i];
    for (int k = i + 1; k < S; k++) w->v2[i] -= w->L[k * ld + i] * w->v2[k];
    w->v2[i] = w->v2[i] / w->L[i * ld + i];
  }
  for (int li = 0; li < n; li++) {
    w->d[li] = w->c[li];
    for (int lj = 0; lj < S; lj++) w->d[li] -= w->A[w->act[lj] + li * w->lda] * w->v2[lj];
  }
}

/* ComputeAlpha, PLDPSolver.cpp:534-653.  *add receives the row to activate or -1. */
static double compute_alpha(pldp_work_t *w, const int *similar, int *add) {
  const int n = 2 * w->M->N, m = w->m;
  const double tol = 1e-8;                                  /* m_tol, :49 */
  double alpha = 10000000.0;
  int to_add = 0, which = 0;
  for (int li = 0; li < m; li++) {
    int found = 0;
    w->computed[li] = 0;
    w->computed[li + m] = 0;
    for (int ci = 0; ci < w->nact; ci++)
      if (w->act[ci] == li) { found = 1; break; }
    if (found) continue;
    const double *pa = w->A + li;
    w->tmp1[li] = 0.0;
    {
      int compute = 1;
      if (similar[li] != 0) {
        const int lindex = li + similar[li];
        if (w->computed[lindex]) { w->tmp1[li] = -w->tmp1[lindex]; compute = 0; }
      }
      if (compute)
        for (int lj = 0; lj < n; lj++) { w->tmp1[li] += *pa * w->d[lj]; pa += w->lda; }
    }
    w->computed[li] = 1;
    if (w->tmp1[li] < 0.0) {
      const double *pa2 = w->A + li;
      w->tmp2[li] = -w->b[li];
      {
        int compute = 1;
        if (similar[li] != 0) {
          const int lindex = li + similar[li];
          if (w->computed[lindex + m]) { w->tmp2[li] += -w->tmp2[lindex] - w->b[lindex]; compute = 0; }
        }
        if (compute)
          for (int lj = 0; lj < n; lj++) { w->tmp2[li] -= *pa2 * w->Vk[lj]; pa2 += w->lda; }
      }
      if (w->tmp2[li] > tol) { /* the reference only prints "PB ON constraint" here */ }
      else if (w->tmp2[li] > 0.0) w->tmp2[li] = -tol;
      const double lalpha = w->tmp2[li] / w->tmp1[li];
      if (alpha > lalpha) {
        alpha = lalpha;
        if (alpha < 1) { to_add = 1; which = li; }
      }
    }
  }
  *add = to_add ? which : -1;
  return alpha;
}

static void add_active(pldp_work_t *w, int row) {
  w->v2[w->nact] = 0.0;       /* reference: whatever m_v2 held; only read if max_iter ends the loop right after an add */
  w->act[w->nact++] = row;
  wgo_optchol_update_fortran(w->A, w->m, 2 * w->M->N, w->act, w->nact, w->L, WG_PLDP_MMAX);
}

/* SolveProblem, PLDPSolver.cpp:654-1007 */
int wgo_pldp_solve(const wgo_pldp_model_t *M, wg_pldp_state_t *st, const double *D, int m, const double *A,
                   const double *b, const double *zmpref, const double *xkyk, const int *similar, int n_removed,
                   int starting, int max_iter, double *X, int *n_iter, int *active, int *n_active) {
  static pldp_work_t work;                                   /* test infrastructure: single-threaded */
  pldp_work_t *w = &work;
  const int N = M->N, n = 2 * N;
  if (m < 0 || m > WG_PLDP_MMAX) return -100;
  for (int i = 0; i < m; i++)
    if (similar[i] > 0 || i + similar[i] < 0) return -100;
  w->M = M; w->m = m; w->lda = m + 1; w->A = A; w->b = b; w->D = D; w->nact = 0;
  if (starting) st->internal_time = 0.0;
  initial_solution(w, st, zmpref, xkyk, starting);
  for (int i = 0; i < st->n_prev; i++) {                     /* hot start :777-791 */
    const int lindex = st->prev_active[i] - n_removed;
    if (lindex >= 0) w->act[w->nact++] = lindex;
  }
  {
    const int total = w->nact;
    w->nact = 0;
    for (int i = 0; i < total; i++) add_active(w, w->act[i]);
  }
  st->n_prev = 0;
  int it = 0, go = 1, rc = 0;
  while (go) {
    for (int i = 0; i < n; i++) w->c[i] = -D[i] - w->Vk[i];
    projected_direction(w);
    int add = -1;
    double alpha = compute_alpha(w, similar, &add);
    if (alpha >= 1.0) { alpha = 1.0; go = 0; }
    if (alpha < 0.0) { rc = WG_PLDP_NEG_ALPHA; break; }
    for (int i = 0; i < n; i++) w->Vk[i] = w->Vk[i] + alpha * w->d[i];
    if (go && add >= 0) add_active(w, add);
    if (max_iter > 0 && it + 1 >= max_iter) go = 0;
    it++;
  }
  for (int i = 0; i < n; i++) X[i] = w->Vk[i];
  if (rc == 0) {
    for (int i = 0; i < w->nact; i++)                        /* :959-968; v2 of the last projection */
      if (w->v2[i] < 0.0) st->prev_active[st->n_prev++] = w->act[i];
    for (int i = 0; i < N; i++) {                            /* StoreCurrentZMPSolution :1010-1036 */
      st->prev_zmp[i] = 0.0;
      st->prev_zmp[i + N] = 0.0;
      for (int j = 0; j < N; j++) {
        st->prev_zmp[i] += M->Pu[j * N + i] * w->Vk[j];
        st->prev_zmp[i + N] += M->Pu[j * N + i] * w->Vk[j + N];
      }
      for (int j = 0; j < 3; j++) {
        st->prev_zmp[i] += M->Px[i * 3 + j] * xkyk[j];
        st->prev_zmp[i + N] += M->Px[i * 3 + j] * xkyk[j + 3];
      }
    }
    if (isnan(X[0]) || isnan(X[N]) || isinf(X[0]) || isinf(X[N])) rc = WG_PLDP_NAN;
    else st->internal_time += 0.02;
  }
  if (n_iter) *n_iter = it;
  if (n_active) *n_active = w->nact;
  if (active) for (int i = 0; i < w->nact; i++) active[i] = w->act[i];
  return rc;
}

/* ---- Dimitrov-2008 tick around PLDP ---------------------------------------------------------------------------------
 * One pass of the loop body of ZMPConstrainedQPFastFormulation::BuildZMPTrajectoryFromFootTrajectory
 * (/root/reference/src/ZMPRefTrajectoryGeneration/ZMPConstrainedQPFastFormulation.cpp:1180-1400):
 * BuildConstraintMatrices :759-1022, D :1254-1262, SolveProblem :1322-1339, X <- iLQ' X :1355-1381,
 * LinearizedInvertedPendulum2D::Interpolation / OneIteration (LinearizedInvertedPendulum2D.cpp:157-264).
 * The constants (OptB, OptC, iLQ: row-major 2N x 6 / 2N x 2N / 2N x 2N) are inputs; PARITY UNPINNED like the rest of
 * this file (the translation unit needs jrl-mal). */
int wgo_dimitrov_tick(const wgo_pldp_model_t *M, const double *OptB, const double *OptC, const double *iLQ, double T,
                      double Tctrl, double com_height, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *st,
                      wg_dimitrov_out_t *out, int max_iter) {
  const int N = M->N, n = 2 * N;
  static double A[(WG_PLDP_MMAX + 1) * 2 * WG_PLDP_N];
  double b[WG_PLDP_MMAX], zr[2 * WG_PLDP_N], D[2 * WG_PLDP_N] = {0}, X[2 * WG_PLDP_N], NewX[2 * WG_PLDP_N] = {0};
  int sim[WG_PLDP_MMAX], act[WG_PLDP_MMAX];
  const double *xk = st->xk;
  int m = 0;
  for (int i = 0; i < N; i++) m += polys[i].nrows;
  if (m > WG_PLDP_MMAX) return -100;
  memset(A, 0, sizeof(double) * (size_t)(m + 1) * n);
  int idx = 0;
  for (int i = 0; i < N; i++) {
    zr[i] = polys[i].centre[0];
    zr[i + N] = polys[i].centre[1];
    for (int j = 0; j < polys[i].nrows; j++) {
      b[idx] = (xk[0] * M->Px[i * 3 + 0] + xk[1] * M->Px[i * 3 + 1] + xk[2] * M->Px[i * 3 + 2]) * polys[i].A[j][0] +
               (xk[3] * M->Px[i * 3 + 0] + xk[4] * M->Px[i * 3 + 1] + xk[5] * M->Px[i * 3 + 2]) * polys[i].A[j][1] +
               polys[i].B[j];
      sim[idx] = polys[i].similar[j];
      for (int k = 0; k < N; k++) {
        A[idx + k * (m + 1)] = polys[i].A[j][0] * M->Pu[k * N + i];
        A[idx + (k + N) * (m + 1)] = polys[i].A[j][1] * M->Pu[k * N + i];
      }
      idx++;
    }
  }
  for (int i = 0; i < n; i++) {
    double l1 = 0.0, od = 0.0;
    for (int j = 0; j < n; j++) l1 += OptC[i * n + j] * zr[j];
    for (int j = 0; j < 6; j++) od += OptB[i * 6 + j] * xk[j];
    D[i] = od - l1;
  }
  int nit = 0, nact = 0;
  int rc = wgo_pldp_solve(M, &st->pldp, D, m, A, b, zr, xk, sim, st->n_removed, st->starting, max_iter, X, &nit, act, &nact);
  if (rc == -100) rc = WG_PLDP_BAD_INPUT;
  for (int i = 0; i < n; i++) {
    double s = 0.0;
    for (int j = i; j < n; j++) s += iLQ[j * n + i] * X[j];
    NewX[i] = s;
  }
  const double jx = NewX[0], jy = NewX[N];
  if (rc == 0) {
    const double c02 = -com_height / 9.81;
    if (out)
      for (int lk = 0; lk <= WG_SAMPLES_PER_TICK; lk++) {
        const double t = (lk + 1) * Tctrl;
        const double cx0 = xk[0] + t * xk[1] + 0.5 * t * t * xk[2] + t * t * t * jx / 6.0;
        const double cx1 = xk[1] + t * xk[2] + 0.5 * t * t * jx;
        const double cx2 = xk[2] + t * jx;
        const double cy0 = xk[3] + t * xk[4] + 0.5 * t * t * xk[5] + t * t * t * jy / 6.0;
        const double cy1 = xk[4] + t * xk[5] + 0.5 * t * t * jy;
        const double cy2 = xk[5] + t * jy;
        out->com_x[lk][0] = cx0; out->com_x[lk][1] = cx1; out->com_x[lk][2] = cx2;
        out->com_y[lk][0] = cy0; out->com_y[lk][1] = cy1; out->com_y[lk][2] = cy2;
        out->zmp_x[lk] = 1.0 * cx0 + 0.0 * cx1 + c02 * cx2;
        out->zmp_y[lk] = 1.0 * cy0 + 0.0 * cy1 + c02 * cy2;
      }
    const double A01 = T, A02 = T * T / 2.0, A12 = T, B0 = T * T * T / 6.0, B1 = T * T / 2.0, B2 = T;
    for (int a = 0; a < 2; a++) {
      const double u = a == 0 ? jx : jy;
      double *c = st->xk + 3 * a;
      const double n0 = 0.0 + 1.0 * c[0] + A01 * c[1] + A02 * c[2];
      const double n1 = 0.0 + 0.0 * c[0] + 1.0 * c[1] + A12 * c[2];
      const double n2 = 0.0 + 0.0 * c[0] + 0.0 * c[1] + 1.0 * c[2];
      c[0] = n0 + u * B0; c[1] = n1 + u * B1; c[2] = n2 + u * B2;
    }
  }
  st->starting = 0;
  st->n_removed = polys[0].nrows;
  if (out) {
    out->jerk_x = jx; out->jerk_y = jy; out->ret = rc; out->n_iter = nit; out->n_active = nact; out->m = m;
    for (int i = 0; i < 2 * WG_PLDP_N; i++) out->X[i] = i < n ? NewX[i] : 0.0;
  }
  return rc;
}

/* ---- the Dimitrov-2008 tick in modes QLD and QLDANDLQ ------------------------------------------------------------------
 * ZMPConstrainedQPFastFormulation::BuildZMPTrajectoryFromFootTrajectory with m_FastFormulationMode == QLD / QLDANDLQ
 * (/root/reference/src/ZMPRefTrajectoryGeneration/ZMPConstrainedQPFastFormulation.cpp): BuildConstraintMatrices :759-1022 with the
 * triangular fill of :891-904 (QLD) or the full one of :905-918 (QLDANDLQ), D :1254-1262, XL / XU = -+1e8 and X = 0 (:1280-1286),
 * iwar[0] = 1 / 0 (:1288-1291), ql0001_ (:1297-1307) on m_Q = OptA (:382-389) / the identity (:537-540), X used as it is (QLD,
 * :1383) or un-preconditioned (QLDANDLQ, :1355-1381), Interpolation + OneIteration.
 * The arithmetic core -- the solve -- is the reference's own compiled qld.cpp when the harness sets it (wgo_ql_call); the dense
 * arrays are built here exactly as the driver builds them (DPu column-major with leading dimension m + 1, zero-filled). */
int wgo_dimitrov_qld_tick(int mode, int N, const double *Q, const double *OptB, const double *OptC, const double *PuT,
                          const double *Px, const double *iLQ, double T, double Tctrl, double com_height,
                          const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *st, wg_dimitrov_out_t *out) {
  const int n = 2 * N;
  const int lq = mode == 2;
  static double A[(WG_PLDP_MMAX + 1) * 2 * WG_PLDP_N], Cq[4 * WG_PLDP_N * WG_PLDP_N], U[WG_PLDP_MMAX + 1 + 4 * WG_PLDP_N];
  double b[WG_PLDP_MMAX + 1], zr[2 * WG_PLDP_N], D[2 * WG_PLDP_N], X[2 * WG_PLDP_N], XL[2 * WG_PLDP_N], XU[2 * WG_PLDP_N];
  double NewX[2 * WG_PLDP_N];
  const double *xk = st->xk;
  int m = 0;
  for (int i = 0; i < N; i++) m += polys[i].nrows;
  if (m > WG_PLDP_MMAX) return -100;
  const int mmax = m + 1;
  memset(A, 0, sizeof(double) * (size_t)mmax * n);
  memset(b, 0, sizeof b);
  int idx = 0;
  for (int i = 0; i < N; i++) {
    zr[i] = polys[i].centre[0];
    zr[i + N] = polys[i].centre[1];
    for (int j = 0; j < polys[i].nrows; j++) {
      b[idx] = (xk[0] * Px[i * 3 + 0] + xk[1] * Px[i * 3 + 1] + xk[2] * Px[i * 3 + 2]) * polys[i].A[j][0] +
               (xk[3] * Px[i * 3 + 0] + xk[4] * Px[i * 3 + 1] + xk[5] * Px[i * 3 + 2]) * polys[i].A[j][1] +
               polys[i].B[j];
      const int kend = lq ? N - 1 : i;                    /* QLD: "Pu is triangular" (:891-904); QLDANDLQ: it is not (:905-918) */
      for (int k = 0; k <= kend; k++) {
        A[idx + k * mmax] = polys[i].A[j][0] * PuT[k * N + i];
        A[idx + (k + N) * mmax] = polys[i].A[j][1] * PuT[k * N + i];
      }
      idx++;
    }
  }
  for (int i = 0; i < n; i++) {
    double l1 = 0.0, od = 0.0;
    for (int j = 0; j < n; j++) l1 += OptC[i * n + j] * zr[j];
    for (int j = 0; j < 6; j++) od += OptB[i * 6 + j] * xk[j];
    D[i] = od - l1;
    XL[i] = -1e8; XU[i] = 1e8; X[i] = 0.0;
  }
  if (lq) { memset(Cq, 0, sizeof(double) * (size_t)n * n); for (int i = 0; i < n; i++) Cq[i * n + i] = 1.0; }
  else memcpy(Cq, Q, sizeof(double) * (size_t)n * n);
  int ifail = 0, nit = 0, nact = 0;
  wgo_ql_call(lq ? 0 : 1, m, 0, mmax, n, n, Cq, D, A, b, XL, XU, X, U, &ifail, &nit, &nact);
  for (int i = 0; i < n; i++) {
    if (!lq) { NewX[i] = X[i]; continue; }
    double s = 0.0;
    for (int j = i; j < n; j++) s += iLQ[j * n + i] * X[j];
    NewX[i] = s;
  }
  const double jx = NewX[0], jy = NewX[N];
  if (ifail == 0) {
    const double c02 = -com_height / 9.81;
    if (out)
      for (int lk = 0; lk <= WG_SAMPLES_PER_TICK; lk++) {
        const double t = (lk + 1) * Tctrl;
        const double cx0 = xk[0] + t * xk[1] + 0.5 * t * t * xk[2] + t * t * t * jx / 6.0;
        const double cx1 = xk[1] + t * xk[2] + 0.5 * t * t * jx;
        const double cx2 = xk[2] + t * jx;
        const double cy0 = xk[3] + t * xk[4] + 0.5 * t * t * xk[5] + t * t * t * jy / 6.0;
        const double cy1 = xk[4] + t * xk[5] + 0.5 * t * t * jy;
        const double cy2 = xk[5] + t * jy;
        out->com_x[lk][0] = cx0; out->com_x[lk][1] = cx1; out->com_x[lk][2] = cx2;
        out->com_y[lk][0] = cy0; out->com_y[lk][1] = cy1; out->com_y[lk][2] = cy2;
        out->zmp_x[lk] = 1.0 * cx0 + 0.0 * cx1 + c02 * cx2;
        out->zmp_y[lk] = 1.0 * cy0 + 0.0 * cy1 + c02 * cy2;
      }
    const double A01 = T, A02 = T * T / 2.0, A12 = T, B0 = T * T * T / 6.0, B1 = T * T / 2.0, B2 = T;
    for (int a = 0; a < 2; a++) {
      const double u = a == 0 ? jx : jy;
      double *c = st->xk + 3 * a;
      const double n0 = 0.0 + 1.0 * c[0] + A01 * c[1] + A02 * c[2];
      const double n1 = 0.0 + 0.0 * c[0] + 1.0 * c[1] + A12 * c[2];
      const double n2 = 0.0 + 0.0 * c[0] + 0.0 * c[1] + 1.0 * c[2];
      c[0] = n0 + u * B0; c[1] = n1 + u * B1; c[2] = n2 + u * B2;
    }
  }
  st->starting = 0;
  st->n_removed = polys[0].nrows;
  if (out) {
    out->jerk_x = jx; out->jerk_y = jy; out->ret = ifail; out->n_iter = nit; out->n_active = nact; out->m = m;
    for (int i = 0; i < 2 * WG_PLDP_N; i++) out->X[i] = i < n ? NewX[i] : 0.0;
  }
  return ifail;
}
