/*
 * ql_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the dense convex QP solver that sits on the reference's
 * Herdt-2010 hot path:
 *     ql0001_  /root/reference/src/Mathematics/qld.cpp:378-612   (driver)
 *     ql0002_  /root/reference/src/Mathematics/qld.cpp:621-2091  (Powell /
 *              Schittkowski dual active-set method, Givens-updated Z and R)
 *
 * The reference is f2c output: 1-based arrays, one flat work array, every
 * local `static`, control flow by computed goto.  This file restates the same
 * arithmetic -- every sum in the same order, every tolerance test the same --
 * as structured, re-entrant C99 with separate arrays, so that
 *   (a) tests can run it from several threads / processes, and
 *   (b) the HIP kernel (jrl-walkgen_amd/csrc/wg_ql_device.hpp) can be checked
 *       against something that is readable next to it.
 * Parity pin: tests/test_ql_oracle.py drives this file and the *compiled
 * reference* (oracle/_ref/libqld_ref.so, built from the reference source by
 * oracle/Makefile) on the same QPs and requires bit-identical x, u, ifail and
 * final active set.
 *
 * Only the `lql == TRUE` mode (iwar[0] == 1: full symmetric Hessian given,
 * solver does its own Cholesky) is restated; it is the only mode the
 * reference's QPProblem::solve uses (qp-problem.cpp:264).
 *
 * Build: -O2 -ffp-contract=off (no FMA contraction; the reference x86-64
 * build has none either).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wg_oracle.h"

typedef struct {
  int n, m, meq, mn;
  int lda;              /* leading dimension of a (mmax)                      */
  double *g;            /* n x n Hessian copy, column-major, ld n             */
  const double *a;      /* m x n, column-major, ld lda                        */
  const double *b;      /* inner right-hand side (= -b_user), qld.cpp:469-475 */
  const double *grad, *xl, *xu;
  double *x;
  double *lam;          /* w[1..n]      multipliers, aligned with iact        */
  double *z;            /* w[iwz+..]    n x n, column-major                   */
  double *r;            /* w[iwr+..]    packed upper triangle + n tail        */
  double *ww;           /* w[iww+..]                                          */
  double *wd;           /* w[iwd+..]    saved Hessian diagonal                */
  double *wx;           /* w[iwx+..]    x snapshot                            */
  double *wa;           /* w[iwa+..]    reciprocal norms, m + n               */
  int *iact;
  int nact;
  double vsmall, diag;
  /* add / drop log */
  int *hist; int hist_cap, hist_len;
} ql_t;

/* f2c.h max/min semantics (qld.cpp:269-270) */
#define MAXD(a, b) ((a) >= (b) ? (a) : (b))
#define MIND(a, b) ((a) <= (b) ? (a) : (b))
#define G(i, j) (q->g[(i) + (size_t)(j) * q->n])
#define A(k, i) (q->a[(k) + (size_t)(i) * q->lda])
#define Z(i, j) (q->z[(i) + (size_t)(j) * q->n])
#define RP(i, j) (q->r[(size_t)(j) * ((j) + 1) / 2 + (i)]) /* i <= j */

static void log_event(ql_t *q, int code) {
  if (q->hist && q->hist_len < q->hist_cap) q->hist[q->hist_len] = code;
  q->hist_len++;
}

/* index into wa[] of an active-set code (1-based code), qld.cpp:1767-1771 */
static int wa_slot(const ql_t *q, int code) {
  int ia = code - 1;
  if (code > q->mn) ia -= q->n;
  return ia;
}

/* qld.cpp:859-890: column-wise Cholesky into packed R.  Returns -1 on success
 * or the failing column j (0-based) with *tfail = the offending pivot. */
static int chol_packed(ql_t *q, double *tfail) {
  const int n = q->n;
  for (int j = 0; j < n; ++j) {
    double temp = 0.0;
    for (int i = 0; i <= j; ++i) {
      temp = G(i, j);
      for (int k = 0; k < i; ++k) temp -= RP(k, j) * RP(k, i);
      if (i < j) RP(i, j) = temp / RP(i, i);
    }
    if (temp < q->vsmall) { *tfail = temp; return j; }
    RP(j, j) = sqrt(temp);
  }
  return -1;
}

/* qld.cpp:895-918: estimate of the further diagonal shift after a failed
 * pivot in column j.  Uses lam[] as scratch exactly like the reference. */
static void chol_bump(ql_t *q, int j, double temp) {
  double *v = q->lam;
  double sumx = 1.0;
  v[j] = 1.0;
  for (int k = j; k >= 1; --k) {           /* computes v[k-1] */
    double sum = 0.0;
    for (int i = k; i <= j; ++i) sum -= RP(k - 1, i) * v[i];
    v[k - 1] = sum / RP(k - 1, k - 1);
    sumx += v[k - 1] * v[k - 1];
  }
  q->diag = q->diag + q->vsmall - temp / sumx;
}

/* qld.cpp:937-975: Z = R^{-1} (upper triangular), strictly-lower part zeroed */
static void invert_r(ql_t *q) {
  const int n = q->n;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < i; ++j) Z(i, j) = 0.0;
    Z(i, i) = 1.0 / RP(i, i);
    for (int j = i; j < n - 1; ++j) {
      double sum = 0.0;
      for (int k = i; k <= j; ++k) sum += Z(i, k) * RP(k, j + 1);
      Z(i, j + 1) = -sum / RP(j + 1, j + 1);
    }
  }
}

/* qld.cpp:2071-2085: s[i] = sum_j Z(j,i) * ww[j] */
static void zt_times_ww(ql_t *q, double *s) {
  const int n = q->n;
  for (int i = 0; i < n; ++i) {
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc += Z(j, i) * q->ww[j];
    s[i] = acc;
  }
}

/* qld.cpp:1824-1851: ww[0..nact) = R^{-1} s[0..nact) (back substitution,
 * inner sums ascending). */
static void backsub(ql_t *q, const double *s) {
  for (int i = q->nact - 1; i >= 0; --i) {
    double sum = 0.0;
    for (int j = i + 1; j < q->nact; ++j) sum += RP(i, j) * q->ww[j];
    q->ww[i] = (s[i] - sum) / RP(i, i);
  }
}

/* qld.cpp:1861-1889: choose the active constraint whose multiplier reaches
 * zero first along the dual step; first index wins ties. */
static int pick_drop(const ql_t *q, double res, double *ratio) {
  int kdrop = -1;
  for (int k = 0; k < q->nact; ++k) {
    if (q->iact[k] <= q->meq) continue;
    if (res * q->ww[k] >= 0.0) continue;
    double temp = q->lam[k] / q->ww[k];
    if (kdrop >= 0 && fabs(temp) >= fabs(*ratio)) continue;
    kdrop = k;
    *ratio = temp;
  }
  return kdrop;
}

/* one plane rotation built from (p, q), qld.cpp:1921-1930 / 2005-2014 */
static void givens(double p, double qq, double *ga, double *gb, double *nrm) {
  double t = MAXD(fabs(p), fabs(qq));
  double d1 = p / t, d2 = qq / t;
  double sum = t * sqrt(d1 * d1 + d2 * d2);
  *ga = p / sum;
  *gb = qq / sum;
  *nrm = sum;
}

/* qld.cpp:1903-1982: delete the constraint at position kdrop (0-based) of the
 * active set; R columns kdrop+1..nu-1 (0-based, nu = number of columns that
 * take part, i.e. nact or nact+1 when the S column rides along) are
 * re-triangularised and the matching columns of Z rotated. */
static void drop_constraint(ql_t *q, int kdrop, int nu) {
  const int n = q->n;
  log_event(q, -q->iact[kdrop]);
  int ia = wa_slot(q, q->iact[kdrop]);
  q->wa[ia] = -q->wa[ia];
  for (int k = kdrop; k < q->nact - 1; ++k) {
    /* rotation from R(k,k+1), R(k+1,k+1) */
    double ga, gb, nrm;
    givens(RP(k, k + 1), RP(k + 1, k + 1), &ga, &gb, &nrm);
    /* exchange the leading k+1 entries of columns k and k+1 */
    for (int i = 0; i <= k; ++i) {
      double t = RP(i, k + 1);
      RP(i, k + 1) = RP(i, k);
      RP(i, k) = t;
    }
    RP(k + 1, k + 1) = 0.0;
    RP(k, k) = nrm;
    /* rotate rows k, k+1 of columns k+1 .. nu-1 */
    for (int c = k + 1; c < nu; ++c) {
      double t = ga * RP(k, c) + gb * RP(k + 1, c);
      RP(k + 1, c) = ga * RP(k + 1, c) - gb * RP(k, c);
      RP(k, c) = t;
    }
    /* rotate columns k, k+1 of Z */
    for (int i = 0; i < n; ++i) {
      double t = ga * Z(i, k) + gb * Z(i, k + 1);
      Z(i, k + 1) = ga * Z(i, k + 1) - gb * Z(i, k);
      Z(i, k) = t;
    }
    q->iact[k] = q->iact[k + 1];
    q->lam[k] = q->lam[k + 1];
  }
  q->nact--;
}

/* qld.cpp:1992-2030: rotate s[nu-1], s[nu-2], ... down to s[nact+1] into
 * s[nact] (0-based), applying each rotation to the matching Z columns. */
static void sweep(ql_t *q, double *s, int nu) {
  const int n = q->n;
  for (int c = nu - 1; c > q->nact; --c) {
    if (s[c] == 0.0) continue;
    double ga, gb, nrm;
    givens(s[c - 1], s[c], &ga, &gb, &nrm);
    s[c - 1] = nrm;
    for (int i = n - 1; i >= 0; --i) {
      double t = ga * Z(i, c - 1) + gb * Z(i, c);
      Z(i, c) = ga * Z(i, c) - gb * Z(i, c - 1);
      Z(i, c - 1) = t;
    }
  }
}

/* qld.cpp:2039-2058 (lql branch): magnitude estimate of x */
static double xmag_sum(const ql_t *q, double vfact) {
  double sum = 0.0;
  for (int i = 0; i < q->n; ++i)
    sum += fabs(q->x[i]) * vfact * (fabs(q->grad[i]) + fabs(G(i, i) * q->x[i]));
  return sum;
}

/* two-sided "is it more than rounding" guard used all over ql0002 */
static int significant(double base, double delta_abs) {
  double temp = base + delta_abs * .1;
  double tempa = base + delta_abs * .2;
  if (temp <= base) return 0;
  if (tempa <= temp) return 0;
  return 1;
}

/* qld.cpp:1547-1658: second half of the linear-dependence test.  Returns 1 if
 * the new normal is independent of the active ones in some coordinate. */
static int independent_coordinate(const ql_t *q, int knext) {
  const int n = q->n, m = q->m;
  int k1 = 0;
  if (knext > m) { k1 = knext - m; if (k1 > n) k1 -= n; }
  for (int i = 1; i <= n; ++i) {           /* 1-based coordinate */
    double suma;
    if (knext <= m) suma = A(knext - 1, i - 1);
    else { suma = 0.0; if (i == k1) suma = (knext > q->mn) ? -1.0 : 1.0; }
    double sumb = fabs(suma);
    for (int k = 0; k < q->nact; ++k) {
      int kk = q->iact[k];
      double temp;
      if (kk <= m) temp = q->ww[k] * A(kk - 1, i - 1);
      else {
        /* the reference indexes ww by the *variable* number here
         * (qld.cpp:1564-1572, 1626-1634); kept as is. */
        kk -= m; temp = 0.0;
        if (kk == i) temp = q->ww[kk - 1];
        kk -= n;
        if (kk == i) temp = -q->ww[kk - 1];
      }
      suma -= temp;
      sumb += fabs(temp);
    }
    if (knext <= m && suma <= q->vsmall) continue;
    if (significant(sumb, fabs(suma))) return 1;
  }
  return 0;
}

int wgo_ql_solve(int m, int me, int mmax, int n, int nmax,
                 const double *c, const double *d, const double *a,
                 const double *b, const double *xl, const double *xu,
                 double eps, double *x, double *u, int *ifail, int *iact_out,
                 int *nact_out, int *n_iter, int *hist, int hist_cap,
                 int *hist_len) {
  ql_t Q, *q = &Q;
  memset(q, 0, sizeof Q);
  const size_t rlen = (size_t)n * (n + 1) / 2 + n;
  size_t nd = (size_t)n * n * 2 + rlen + 5 * (size_t)n + 2 * (size_t)m + n + 8;
  double *mem = (double *)calloc(nd, sizeof(double));
  int *iact = (int *)calloc((size_t)n + 1, sizeof(int));
  if (!mem || !iact) { free(mem); free(iact); *ifail = 5; return -1; }
  double *p = mem;
  q->g = p;   p += (size_t)n * n;
  q->z = p;   p += (size_t)n * n;
  q->r = p;   p += rlen;
  q->lam = p; p += n;
  q->ww = p;  p += n;
  q->wd = p;  p += n;
  q->wx = p;  p += n;
  q->wa = p;  p += (size_t)m + n;
  double *binner = p; p += m;
  q->n = n; q->m = m; q->meq = me; q->mn = m + n; q->lda = mmax;
  q->a = a; q->b = binner; q->grad = d; q->xl = xl; q->xu = xu; q->x = x;
  q->iact = iact; q->nact = 0; q->vsmall = eps; q->diag = 0.0;
  q->hist = hist; q->hist_cap = hist_cap; q->hist_len = 0;

  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) G(i, j) = c[i + (size_t)j * nmax];
  /* qld.cpp:442-444 (the reference patches the caller's array; we patch the copy) */
  /* the element patched is c(nmax,nmax): inside the n x n block only if nmax == n */
  if (nmax == n && fabs(G(n - 1, n - 1)) == 0.0) G(n - 1, n - 1) = eps;
  for (int j = 0; j < m; ++j) binner[j] = -b[j];            /* :469-475 */
  const int maxit = (m + n) * 40;                           /* :459 */

  int info = 0, iterc = 1, itref = 0, iflag = 0;
  double *s_tail = q->r + (size_t)n * (n + 1) / 2;
  double *s = s_tail;
  const double onha = 1.5, xmagr = .01, diagr = 2.0;
  const int ifinc = 3, kfinc = (n > 10) ? n : 10;
  int jfinc = -kfinc;
  double xmag = 0.0, vfact = 1.0;
  double res = 0.0, ratio = 0.0, cvmax;
  int knext = 0;
  int done_info_set = 0;

  /* --- reciprocal lengths of the constraint normals, :769-807 --- */
  for (int k = 0; k < m; ++k) {
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += A(k, i) * A(k, i);
    if (sum > 0.0) sum = 1.0 / sqrt(sum);
    else if (binner[k] == 0.0) { /* keep 0 */ }
    else {
      info = -(k + 1);
      if (k + 1 <= me || !(binner[k] <= 0.0)) { done_info_set = 1; break; }   /* :789 `if (b <= 0) ok else fail`: a NaN fails */
    }
    q->wa[k] = sum;
  }
  if (done_info_set) goto finish_noresore;
  for (int k = 0; k < n; ++k) q->wa[m + k] = 1.0;

  /* --- make the Hessian numerically positive definite, :814-854 --- */
  for (int i = 0; i < n; ++i) {
    q->wd[i] = G(i, i);
    q->diag = MAXD(q->diag, q->vsmall - q->wd[i]);
    for (int j = i + 1; j < n; ++j) {
      double ga = -MIND(q->wd[i], G(j, j));
      double gb = fabs(q->wd[i] - G(j, j)) + fabs(G(i, j));
      if (gb > 0.0) ga += G(i, j) * G(i, j) / gb;
      q->diag = MAXD(q->diag, ga);
    }
  }
  {
    int need_shift = !(q->diag <= 0.0);                       /* :844 `if (diag <= 0) goto L90`: a NaN shifts */
    for (;;) {
      if (need_shift) {
        q->diag = diagr * q->diag;
        for (int i = 0; i < n; ++i) G(i, i) = q->diag + q->wd[i];
      }
      double tfail;
      int jf = chol_packed(q, &tfail);
      if (jf < 0) break;
      if (jf == 0) { q->diag = q->diag + q->vsmall - tfail; } /* unreachable in the reference */
      else chol_bump(q, jf, tfail);
      need_shift = 1;
    }
  }
  invert_r(q);

  enum { ST_RESET, ST_RESID, ST_SCAN, ST_CONVERGED, ST_FINISH } st = ST_RESET;
  for (;;) {
    if (st == ST_RESET || st == ST_RESID) {
      s = s_tail;
      if (st == ST_RESET) {                                   /* :989-1027 */
        iflag = 1;
        for (int i = 0; i < n; ++i) {
          x[i] = 0.0;
          q->ww[i] = d[i];
          if (i >= q->nact) continue;
          q->lam[i] = 0.0;
          int k = iact[i];
          if (k <= m) s[i] = binner[k - 1];
          else if (k > q->mn) s[i] = -xu[k - q->mn - 1];
          else s[i] = xl[k - m - 1];
        }
        xmag = 0.0;
        vfact = 1.0;
      } else {                                                /* :1031-1099 */
        iflag = 2;
        for (int i = 0; i < n; ++i) {
          double acc = d[i];
          for (int j = 0; j < n; ++j) acc += G(i, j) * x[j];
          q->ww[i] = acc;
        }
        for (int k = 0; k < q->nact; ++k) {
          int kk = iact[k];
          if (kk <= m) {
            double sk = binner[kk - 1];
            for (int i = 0; i < n; ++i) {
              q->ww[i] -= q->lam[k] * A(kk - 1, i);
              sk -= x[i] * A(kk - 1, i);
            }
            s[k] = sk;
          } else if (kk <= q->mn) {
            int k1 = kk - m - 1;
            q->ww[k1] -= q->lam[k];
            s[k] = xl[k1] - x[k1];
          } else {
            int k1 = kk - q->mn - 1;
            q->ww[k1] += q->lam[k];
            s[k] = -xu[k1] + x[k1];
          }
        }
      }
      if (q->nact > 0) {                                      /* :1104-1170 */
        for (int i = 0; i < q->nact; ++i) {
          double sum = 0.0;
          for (int j = 0; j < i; ++j) sum += RP(j, i) * s[j];
          s[i] = (s[i] - sum) / RP(i, i);
        }
        for (int i = 0; i < n; ++i) {
          double sum = 0.0;
          for (int j = 0; j < q->nact; ++j) sum += s[j] * Z(i, j);
          x[i] += sum;
          for (int j = 0; j < n; ++j) q->ww[j] += sum * G(i, j);
        }
      }
      zt_times_ww(q, s);                                      /* :1175-1177 */
      if (q->nact != n) {                                     /* :1186-1201 */
        for (int i = 0; i < n; ++i) {
          double sum = 0.0;
          for (int j = q->nact; j < n; ++j) sum += Z(i, j) * s[j];
          x[i] -= sum;
        }
        info = 0;
      }
      if (q->nact != 0) {                                     /* :1208-1217 */
        backsub(q, s);
        for (int k = 0; k < q->nact; ++k) q->lam[k] += q->ww[k];
      }
      { double sm = xmag_sum(q, vfact); xmag = MAXD(xmag, sm); }  /* :1222-1224 */
      if (iflag == itref) { st = ST_RESID; continue; }        /* :1226 */
      /* delete the first inequality with a negative multiplier, :1233-1249 */
      int kd = -1;
      for (int k = 0; k < q->nact; ++k)
        /* :1237 `if (w[kdrop] >= zero) goto next`: a NaN multiplier IS dropped */
        if (!(q->lam[k] >= 0.0) && iact[k] > me) { kd = k; break; }
      if (kd >= 0) { drop_constraint(q, kd, q->nact); st = ST_RESID; continue; }
      st = ST_SCAN;
    }

    if (st == ST_SCAN) {
      /* most violated normalised constraint, :1255-1331 */
      cvmax = 0.0;
      for (int k = 0; k < m; ++k) {
        if (q->wa[k] <= 0.0) continue;
        double sum = -binner[k];
        for (int i = 0; i < n; ++i) sum += x[i] * A(k, i);
        double sumx = -sum * q->wa[k];
        if (k + 1 <= me) sumx = fabs(sumx);
        if (sumx <= cvmax) continue;
        double temp = fabs(binner[k]);
        for (int i = 0; i < n; ++i) temp += fabs(x[i] * A(k, i));
        double tempa = temp + fabs(sum);
        if (tempa <= temp) continue;
        temp += onha * fabs(sum);
        if (temp <= tempa) continue;
        cvmax = sumx; res = sum; knext = k + 1;
      }
      for (int k = 0; k < n; ++k) {
        if (q->wa[m + k] <= 0.0) continue;
        int lower = 1;
        double sum = xl[k] - x[k];
        if (sum == 0.0) continue;
        if (sum < 0.0) { sum = x[k] - xu[k]; lower = 0; }
        if (sum <= cvmax) continue;
        cvmax = sum; res = -sum;
        knext = lower ? k + 1 + m : k + 1 + q->mn;
      }
      info = 0;
      if (cvmax <= q->vsmall) { st = ST_CONVERGED; continue; }   /* :1336 */

      /* has the objective stopped increasing?  :1343-1408 */
      ++jfinc;
      if (jfinc == 0 || jfinc == ifinc) {
        if (jfinc == ifinc) {
          double fdiff = 0.0, fdiffa = 0.0;
          for (int i = 0; i < n; ++i) {
            double sum = 2.0 * d[i];
            double sumx = fabs(sum);
            for (int j = 0; j < n; ++j) {
              double temp = G(i, j) * (q->wx[j] + x[j]);
              sum += temp;
              sumx += fabs(temp);
            }
            fdiff += sum * (x[i] - q->wx[i]);
            fdiffa += sumx * fabs(x[i] - q->wx[i]);
          }
          info = 2;
          double sum = fdiffa + fdiff;
          if (sum <= fdiffa) { st = ST_CONVERGED; continue; }
          double temp = fdiffa + onha * fdiff;
          if (temp <= sum) { st = ST_CONVERGED; continue; }
          jfinc = 0;
          info = 0;
        }
        for (int i = 0; i < n; ++i) q->wx[i] = x[i];
      }

      ++iterc;                                                /* :1415-1420 */
      if (iterc > maxit) { info = 1; st = ST_FINISH; continue; }

      /* new normal and its products with the columns of Z, :1422-1470 */
      s = q->r + (size_t)q->nact * (q->nact + 1) / 2;
      if (knext <= m) {
        for (int i = 0; i < n; ++i) q->ww[i] = A(knext - 1, i);
        zt_times_ww(q, s);
      } else {
        for (int i = 0; i < n; ++i) q->ww[i] = 0.0;
        int k1 = knext - m;
        if (k1 <= n) {
          q->ww[k1 - 1] = 1.0;
          for (int i = 0; i < n; ++i) s[i] = Z(k1 - 1, i);
        } else {
          k1 = knext - q->mn;
          q->ww[k1 - 1] = -1.0;
          for (int i = 0; i < n; ++i) s[i] = -Z(k1 - 1, i);
        }
      }
      double parnew = 0.0, parinc = 0.0, step = 0.0, sumy;
      int kdrop = -1;
      /* route: 0 = take a step (:1717), 1 = dependent, multipliers needed
       * (:1541), 2 = dependent, multipliers already in ww (:1598) */
      int route;
      if (q->nact == n) route = 1;                            /* :1477 */
      else {
        sweep(q, s, n);                                       /* :1480-1482 */
        if (q->nact == 0) route = 0;                          /* :1488 */
        else {                                                /* :1491-1532 */
          double suma = 0.0, sumb = 0.0, sumc = 0.0;
          for (int i = 0; i < n; ++i) {
            double zi = Z(i, q->nact);
            suma += q->ww[i] * zi;
            sumb += fabs(q->ww[i] * zi);
            sumc += zi * zi;
          }
#ifdef WGO_DEBUG_ROUTE
          if (iterc == 3) for (int i = 0; i < n; i++) fprintf(stderr, "ORCW %d %.17g %.17g\n", i, q->ww[i], Z(i, q->nact));
          fprintf(stderr, "ORC it %d knext %d nact %d suma %.17g sumb %.17g sumc %.17g wa %.17g\n", iterc, knext, q->nact, suma, sumb, sumc, knext <= m ? q->wa[knext - 1] : 0.0);
#endif
          if (!significant(sumb, fabs(suma)) || !(sumb > q->vsmall)) route = 1;
          else {
            sumc = sqrt(sumc);
            if (knext <= m) sumc /= q->wa[knext - 1];
            if (significant(sumc, fabs(suma))) route = 0;
            else {                                            /* :1538-1540 */
              backsub(q, s);
              route = independent_coordinate(q, knext) ? 0 : 2;
            }
          }
        }
      }
      int leave = 0;       /* 1: go to ST_CONVERGED with info < 0 */
      if (route != 0) {
        if (route == 1) backsub(q, s);
        kdrop = pick_drop(q, res, &ratio);
        info = -knext;                                        /* :1663 */
        /* (the reference printf()s a diagnostic here, :1664) */
        if (kdrop < 0) leave = 1;
        else { parinc = ratio; parnew = parinc; }
      }
      if (leave) { st = ST_CONVERGED; continue; }

      /* partial steps, each ending in a deletion, until a full step fits,
       * :1673-1759 */
      int have_dual_only_update = (route != 0);
      for (;;) {
        if (!have_dual_only_update) {
          sumy = s[q->nact];                                  /* :1718-1720 */
          step = -res / sumy;
          parinc = step / sumy;
          kdrop = -1;
          if (q->nact > 0) {
            backsub(q, s);
            kdrop = pick_drop(q, res, &ratio);
            if (kdrop >= 0) {                                 /* :1734-1743 */
              double temp = 1.0 - ratio / parinc;
              if (temp <= 0.0) kdrop = -1;
              else { step = ratio * sumy; parinc = ratio; res = temp * res; }
            }
          }
          for (int i = 0; i < n; ++i) x[i] += step * Z(i, q->nact);  /* :1749-1755 */
          parnew += parinc;
          if (q->nact < 1) break;
        }
        have_dual_only_update = 0;
        for (int k = 0; k < q->nact; ++k) {                   /* :1677-1687 */
          q->lam[k] -= parinc * q->ww[k];
          if (iact[k] > me) q->lam[k] = MAXD(0.0, q->lam[k]);
        }
        if (kdrop < 0) break;
        {                                                     /* :1697-1711 */
          int nu = q->nact + 1;
          drop_constraint(q, kdrop, nu);
          double *snew = s - (q->nact + 1);
          if (nu > n) nu = n;
          for (int i = 0; i < nu; ++i) snew[i] = s[i];  /* same as w[is]=w[is+nact+1] */
          s = snew;
          sweep(q, s, nu);
        }
      }

      /* add the new constraint, :1764-1771 */
      q->lam[q->nact] = parnew;
      iact[q->nact] = knext;
      q->nact++;
      log_event(q, knext);
      { int ia = wa_slot(q, knext); q->wa[ia] = -q->wa[ia]; }
      double sm = xmag_sum(q, vfact);                         /* :1776-1786 */
      xmag = MAXD(xmag, sm);
      if (sm < xmagr * xmag) st = ST_RESET;
      else if (itref <= 0) st = ST_SCAN;
      else st = ST_RESID;
      continue;
    }

    if (st == ST_CONVERGED) {                                 /* :1791-1799 */
      ++itref;
      jfinc = -1;
      if (itref == 1) { st = ST_RESID; continue; }
      st = ST_FINISH;
    }
    if (st == ST_FINISH) break;
  }

  /* ql0002 exit: the caller's Hessian diagonal is restored there (:1804-1809);
   * we worked on a copy, nothing to do. */
finish_noresore:
  /* --- ql0001 epilogue, :497-608 --- */
  *ifail = 0;
  if (info == 1) *ifail = 1;
  else if (info == 2) *ifail = 2;
  else if (info < 0) *ifail = -info + 10;
  else {
    for (int j = 0; j < m + 2 * n; ++j) u[j] = 0.0;
    for (int i = 0; i < q->nact; ++i) u[iact[i] - 1] = q->lam[i];
  }
  if (iact_out) for (int i = 0; i < n; ++i) iact_out[i] = (i < q->nact) ? iact[i] : 0;
  if (nact_out) *nact_out = q->nact;
  if (n_iter) *n_iter = iterc;
  if (hist_len) *hist_len = q->hist_len;
  free(mem);
  free(iact);
  return 0;
}
