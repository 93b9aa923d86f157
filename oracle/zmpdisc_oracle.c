/*
 * zmpdisc_oracle.c -- TEST INFRASTRUCTURE ONLY (see wg_oracle.h).
 *
 * CPU restatement of the data producers either side of the Kajita stage-1 / Dimitrov paths:
 *
 *   ZMPDiscretization::GetZMPDiscretization      src/ZMPRefTrajectoryGeneration/ZMPDiscretization.cpp:143-173
 *     InitializeFilter :240-262, InitOnLine :319-513, UpdateCurrentSupportFootPosition :515-558,
 *     OnLineAddFoot :573-1020, FilterOutValues :1045-1109, EndPhaseOfTheWalking :1129-1300
 *   FootTrajectoryGenerationStandard::SetParameters / UpdateFootPosition
 *                                                src/FootTrajectoryGeneration/FootTrajectoryGenerationStandard.cpp:150-186, 411-566
 *   Polynome::Compute, Polynome3/4/5::SetParameters  src/Mathematics/Polynome.cpp:44-53, PolynomeFoot.cpp:41-57, 100-120, 174-195
 *   FootConstraintsAsLinearSystem                src/Mathematics/FootConstraintsAsLinearSystem.cpp:55-92, 97-256, 258-539
 *   ComputeConvexHull::DoComputeConvexHull       src/Mathematics/ConvexHull.cpp:37-203
 *
 * Pinned (tests/test_zmpdisc_oracle.py) to the reference's golden files TestKajita2003{StraightWalking, PbFlorentSeq1,
 * Circle}TestFGPI.datref: their columns 11-13, 20-22, 23-25, 32-34 are the feet deques and 35-36 the ZMP
 * reference deque of this code, popped one sample per control step (DoubleStagePreviewControlStrategy.cpp:128-150,
 * tests/TestObject.cpp:354-382).
 *
 * The boost::ublas products of the reference (MAL_RET_A_by_B, MAL_C_eq_A_by_B) accumulate k ascending from 0.
 */
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>

#include "../include/wg_mpc.h"
#include "wg_oracle.h"

#ifdef WGO_PORTABLE_TRIG
#include "../include/wg_trig.h"
#define WSIN(x) wg_sin(x)
#define WCOS(x) wg_cos(x)
#else
#define WSIN(x) sin(x)
#define WCOS(x) cos(x)
#endif

#define WIN_MAX 64

typedef struct {
  double px, py, theta, time;
  int type;
} zmp_t;

typedef struct {
  double x, y, z, theta, omega, omega2, time;
  int type;
} foot_t;

typedef struct {
  double c[6];
  int n;
} poly_t;

typedef struct {
  const wg_zmpdisc_model_t *M;
  double win[WIN_MAX];
  int nwin;
  double sfp[3][3], prev_sfp[3][3];          /* m_CurrentSupportFootPosition, m_PrevCurrentSupportFootPosition */
  double vdiffpre[2];                        /* m_vdiffsupppre */
  double ang_support, ang_zmp;               /* m_AngleDiffToSupportFootTheta, m_AngleDiffFromZMPThetaToSupportFootTheta */
  double now;                                /* m_CurrentTime */
  wg_rel_step_t rel[2];                      /* m_RelativeFootPositions (never more than two entries) */
  int nrel;
  zmp_t *fz;                                 /* FinalZMPPositions */
  foot_t *fl, *fr;                           /* FinalLeft/RightFootAbsolutePositions */
  int nz, nf, cap;
  poly_t px, py, pz, ptheta, pomega, pomega2, pzmptheta;
} zd_t;

/* Polynome::Compute, Polynome.cpp:44-53 */
static double poly_eval(const poly_t *p, double t) {
  double r = 0.0, pt = 1.0;
  for (int i = 0; i < p->n; i++) {
    r += p->c[i] * pt;
    pt *= t;
  }
  return r;
}
/* Polynome3::SetParameters, PolynomeFoot.cpp:41-57 */
static void poly3_set(poly_t *p, double FT, double FP) {
  p->n = 4;
  p->c[0] = 0.0;
  p->c[1] = 0.0;
  double tmp = FT * FT;
  if (FP == 0.0 || FT == 0.0) {
    p->c[2] = 0.0;
    p->c[3] = 0.0;
  } else {
    p->c[2] = 3.0 * FP / tmp;
    p->c[3] = -2.0 * FP / (tmp * FT);
  }
}
/* Polynome4::SetParameters, PolynomeFoot.cpp:100-120 */
static void poly4_set(poly_t *p, double FT, double MP) {
  p->n = 5;
  p->c[0] = 0.0;
  p->c[1] = 0.0;
  double tmp = FT * FT;
  if (MP == 0.0 || tmp == 0.0) {
    p->c[2] = p->c[3] = p->c[4] = 0.0;
  } else {
    p->c[2] = 16.0 * MP / tmp;
    tmp = tmp * FT;
    p->c[3] = -32.0 * MP / tmp;
    tmp = tmp * FT;
    p->c[4] = 16.0 * MP / tmp;
  }
}
/* Polynome5::SetParameters, PolynomeFoot.cpp:174-195 */
static void poly5_set(poly_t *p, double FT, double FP) {
  p->n = 6;
  p->c[0] = p->c[1] = p->c[2] = 0.0;
  double tmp = FT * FT * FT;
  if (FP == 0.0 || tmp == 0.0) {
    p->c[3] = p->c[4] = p->c[5] = 0.0;
  } else {
    p->c[3] = 10 * FP / tmp;
    tmp *= FT;
    p->c[4] = -15 * FP / tmp;
    tmp *= FT;
    p->c[5] = 6 * FP / tmp;
  }
}

/* InitializeFilter, ZMPDiscretization.cpp:240-262 */
static int init_filter(zd_t *z) {
  const double T = 0.05;
  int n = (int)floor(T / z->M->T);
  if (n < 1 || n + 1 > WIN_MAX) return -1;
  double sum = 0;
  for (int i = 0; i < n + 1; i++) {
    double tmp = WSIN((M_PI * i) / n);
    z->win[i] = tmp * tmp;
  }
  for (int i = 0; i < n + 1; i++) sum += z->win[i];
  for (int i = 0; i < n + 1; i++) z->win[i] /= sum;
  z->nwin = n + 1;
  return 0;
}

/* FilterOutValues, :1045-1109.  `o` is re-read from the growing output deque on every sample, like the reference. */
static int filter_out(zd_t *z, const zmp_t *Z, int nZ, int init_step) {
  const unsigned lshift = 2;
  if (z->nz + nZ > z->cap) return WG_ZMPDISC_CAPACITY;
  for (unsigned i = 0; i < (unsigned)nZ; i++) {
    double l0 = 0, l1 = 0;
    int o = (int)((unsigned)z->nz - 1u - lshift);
    for (unsigned j = 0; j < (unsigned)z->nwin; j++) {
      int r = (int)(i - j + lshift);
      if (r < 0) {
        if (init_step) {
          l0 += z->win[j] * Z[lshift].px;
          l1 += z->win[j] * Z[lshift].py;
        } else if (-r < o) {
          l0 += z->win[j] * z->fz[o + r].px;
          l1 += z->win[j] * z->fz[o + r].py;
        } else {
          l0 += z->win[j] * z->fz[0].px;
          l1 += z->win[j] * z->fz[0].py;
        }
      } else {
        if (r >= nZ) r = nZ - 1;
        l0 += z->win[j] * Z[r].px;
        l1 += z->win[j] * Z[r].py;
      }
    }
    zmp_t a = Z[i];
    a.px = l0;
    a.py = l1;
    z->fz[z->nz++] = a;
  }
  return 0;
}

/* UpdateCurrentSupportFootPosition, :515-558 */
static void update_support(zd_t *z, const wg_rel_step_t *s) {
  memcpy(z->prev_sfp, z->sfp, sizeof z->sfp);
  double c = WCOS(s->theta * M_PI / 180.0);
  double sn = WSIN(s->theta * M_PI / 180.0);
  double MM[2][2] = {{c, -sn}, {sn, c}}, O[2][2], v2[2];
  for (int k = 0; k < 2; k++)
    for (int l = 0; l < 2; l++) {
      double t = 0.0;
      for (int q = 0; q < 2; q++) t += MM[k][q] * z->sfp[q][l];
      O[k][l] = t;
    }
  for (int k = 0; k < 2; k++) {
    double t = 0.0;
    t += O[k][0] * s->sx;
    t += O[k][1] * s->sy;
    v2[k] = t;
  }
  for (int k = 0; k < 2; k++)
    for (int l = 0; l < 2; l++) z->sfp[k][l] = O[k][l];
  for (int k = 0; k < 2; k++) z->sfp[k][2] += v2[k];
}

static void zmp_in_world(const zd_t *z, double w[2]) {
  const double f[3] = {z->M->zmp_neutral[0], z->M->zmp_neutral[1], 1.0};
  for (int k = 0; k < 2; k++) {
    double t = 0.0;
    for (int q = 0; q < 3; q++) t += z->sfp[k][q] * f[q];
    w[k] = t;
  }
}

/* FootTrajectoryGenerationStandard::UpdateFootPosition, FootTrajectoryGenerationStandard.cpp:411-566 */
static void update_foot(const zd_t *z, foot_t *sup, foot_t *non, int cur, int init, double mod_sst, int step_type) {
  const wg_zmpdisc_model_t *M = z->M;
  unsigned k = (unsigned)(cur - init);
  double local = k * M->T;
  double end_lift = (M->t_single - mod_sst) * 0.5;
  double start_land = end_lift + mod_sst;
  sup[cur] = sup[cur - 1];
  sup[cur].type = (-1) * step_type;
  foot_t *c = &non[cur];
  const foot_t *i0 = &non[init];
  c->type = step_type;
  if (local < end_lift) {
    c->x = i0->x;
    c->y = i0->y;
    c->theta = i0->theta;
  } else if (local < start_land) {
    c->x = i0->x + poly_eval(&z->px, local - end_lift);
    c->y = i0->y + poly_eval(&z->py, local - end_lift);
    c->theta = i0->theta + poly_eval(&z->ptheta, local - end_lift);
  } else {
    c->x = i0->x + poly_eval(&z->px, mod_sst);
    c->y = i0->y + poly_eval(&z->py, mod_sst);
    c->theta = i0->theta + poly_eval(&z->ptheta, mod_sst);
  }
  c->z = i0->z + poly_eval(&z->pz, local);
  if (local < end_lift)
    c->omega = poly_eval(&z->pomega, local);
  else if (local < start_land)
    c->omega = M->omega - poly_eval(&z->pomega2, local - end_lift);
  else
    c->omega = poly_eval(&z->pomega, local - start_land) - M->omega;
  double dFX, dFY, dFZ;
  double lo = c->omega * M_PI / 180.0;
  double lt = c->theta * M_PI / 180.0;
  double ct = WCOS(lt), st = WSIN(lt);
  {
    double dX, Z1, Z2, X1, X2;
    double B = M->foot_b, H = M->foot_h, F = M->foot_f;
    if (lo < 0) {
      X1 = B * WCOS(-lo);
      X2 = H * WSIN(-lo);
      Z1 = H * WCOS(-lo);
      Z2 = B * WSIN(-lo);
      dX = -(B - X1 + X2);
      dFZ = Z1 + Z2 - H;
    } else {
      X1 = F * WCOS(lo);
      X2 = H * WSIN(lo);
      Z1 = H * WCOS(lo);
      Z2 = F * WSIN(lo);
      dX = (F - X1 + X2);
      dFZ = Z1 + Z2 - H;
    }
    dFX = ct * dX;
    dFY = st * dX;
  }
  c->x += dFX;
  c->y += dFY;
  c->z += dFZ;
}

static void support_bookkeeping(zd_t *z, const foot_t *L, const foot_t *R, double zmp_theta, int *who) {
  if (z->rel[0].sy < 0) {
    *who = -1;
    z->vdiffpre[0] = R->x - L->x;
    z->vdiffpre[1] = R->y - L->y;
    z->ang_support = R->theta - L->theta;
    z->ang_zmp = R->theta - zmp_theta;
  } else {
    *who = 1;
    z->vdiffpre[0] = -R->x + L->x;
    z->vdiffpre[1] = -R->y + L->y;
    z->ang_support = L->theta - R->theta;
    z->ang_zmp = L->theta - zmp_theta;
  }
}

/* OnLineAddFoot, :573-1020 (EndSequence = false) */
static int add_foot(zd_t *z, const wg_rel_step_t *new_step) {
  const wg_zmpdisc_model_t *M = z->M;
  if (z->nrel != 1) return WG_ZMPDISC_BAD_INPUT;
  foot_t curL = z->fl[z->nf - 1], curR = z->fr[z->nf - 1];
  double cur_zmp_theta = z->fz[z->nz - 1].theta;
  z->rel[1] = *new_step;
  z->nrel = 2;
  int who = 1;
  double lTdble = M->t_double, lTsingle = M->t_single;
  if (z->rel[1].ds_time != 0.0) {
    lTdble = z->rel[1].ds_time;
    lTsingle = z->rel[1].ss_time;
  }
  double t_first = lTdble;
  support_bookkeeping(z, &curL, &curR, cur_zmp_theta, &who);
  double t_this = t_first + lTsingle;
  int add = (int)(unsigned)round(t_this / M->T);
  if (add <= 0 || add > (1 << 20)) return WG_ZMPDISC_BAD_INPUT;
  if (z->nf + add > z->cap || z->nz + add > z->cap) return WG_ZMPDISC_CAPACITY;
  zmp_t *Z = (zmp_t *)calloc((size_t)add, sizeof(zmp_t));
  foot_t *L = (foot_t *)calloc((size_t)add, sizeof(foot_t));
  foot_t *R = (foot_t *)calloc((size_t)add, sizeof(foot_t));
  int rc = 0, idx = 0;
  update_support(z, &z->rel[0]);
  unsigned n1 = (unsigned)round(t_first / M->T);
  unsigned n2 = (unsigned)round(lTsingle / M->T);
  if (n1 == 0 || n1 + n2 > (unsigned)add) {           /* the reference would index ZMPPositions out of bounds */
    rc = WG_ZMPDISC_BAD_INPUT;
    goto done;
  }
  double px0 = z->fz[z->nz - 1].px, py0 = z->fz[z->nz - 1].py, theta0 = z->fz[z->nz - 1].theta;
  double w[2];
  zmp_in_world(z, w);
  double dx = (w[0] - px0) / n1, dy = (w[1] - py0) / n1;
  const int t1 = z->rel[1].step_type;
  if (t1 == 3) {
    dx = (z->sfp[0][2] + M->zmp_shift[0] - px0) / n1;
    dy = (z->sfp[1][2] - py0) / n1;
  }
  if (t1 == 4) {
    dx = (z->sfp[0][2] + M->zmp_shift[2] - px0) / n1;
    dy = (z->sfp[1][2] - py0) / n1;
  }
  if (t1 == 5) {
    dx = (z->sfp[0][2] - (M->zmp_shift[0] + M->zmp_shift[2] + M->zmp_shift[1] + M->zmp_shift[3]) - px0) / n1;
    dy = (z->sfp[1][2] - py0) / n1;
  }
  for (unsigned k = 0; k < n1; k++) {                  /* double support */
    Z[idx].px = px0 + k * dx;
    Z[idx].py = py0 + k * dy;
    Z[idx].theta = theta0;
    Z[idx].time = z->now;
    Z[idx].type = t1 + 10;
    L[idx] = z->fl[z->nf - 1];
    L[idx].z = 0.0;
    R[idx] = z->fr[z->nf - 1];
    R[idx].z = 0.0;
    L[idx].time = R[idx].time = z->now;
    L[idx].type = R[idx].type = t1 + 10;
    z->now += M->T;
    idx++;
  }
  double step_h, next_theta, rel_theta, rel_zmp_theta, vdiff[2] = {0, 0}, vrel[2];
  {                                                    /* m_RelativeFootPositions.size() > 1 always holds here */
    next_theta = z->rel[1].theta;
    rel_theta = next_theta + z->ang_support;
    rel_zmp_theta = next_theta + z->ang_zmp;
    step_h = M->step_height;
    double c = WCOS(next_theta * M_PI / 180.0), s = WSIN(next_theta * M_PI / 180.0);
    double Rm[2][2] = {{c, -s}, {s, c}}, O[2][2];
    for (int k = 0; k < 2; k++)
      for (int l = 0; l < 2; l++) {
        double t = 0.0;
        for (int q = 0; q < 2; q++) t += Rm[k][q] * z->sfp[q][l];
        O[k][l] = t;
      }
    for (int k = 0; k < 2; k++) {
      double t = 0.0;
      t += O[k][0] * z->rel[1].sx;
      t += O[k][1] * z->rel[1].sy;
      vdiff[k] = t;
    }
    vrel[0] = vdiff[0] + z->vdiffpre[0];
    vrel[1] = vdiff[1] + z->vdiffpre[1];
  }
  z->vdiffpre[0] = vdiff[0];
  z->vdiffpre[1] = vdiff[1];
  double mod_sst = lTsingle * M->modulation;
  double end_lift = (lTsingle - mod_sst) * 0.5;
  poly5_set(&z->px, mod_sst, vrel[0]);
  poly5_set(&z->py, mod_sst, vrel[1]);
  poly4_set(&z->pz, M->t_single, step_h);
  poly3_set(&z->ptheta, mod_sst, rel_theta);
  poly3_set(&z->pomega, end_lift, M->omega);
  poly3_set(&z->pomega2, mod_sst, 2 * M->omega);
  poly3_set(&z->pzmptheta, lTsingle, rel_zmp_theta);
  int init = idx - 1;
  double px02 = Z[idx - 1].px, py02 = Z[idx - 1].py;
  for (unsigned k = 0; k < n2; k++) {                  /* single support */
    zmp_in_world(z, w);
    Z[idx].px = w[0];
    Z[idx].py = w[1];
    Z[idx].time = z->now;
    if (t1 == 3 || t1 == 4) {
      if (t1 == 3) {
        dx = (z->sfp[0][2] + M->zmp_shift[1] - px02) / n2;
        dy = (z->sfp[1][2] - py02) / n2;
      } else {
        dx = (z->sfp[0][2] + M->zmp_shift[3] - px02) / n2;
        dy = (z->sfp[1][2] - py02) / n2;
      }
      Z[idx].px = Z[idx - 1].px + dx;
      Z[idx].py = Z[idx - 1].py + dy;
    }
    Z[idx].theta = poly_eval(&z->pzmptheta, k * M->T) + Z[init].theta;
    Z[idx].type = who * z->rel[0].step_type;
    if (who == 1)
      update_foot(z, L, R, idx, init, mod_sst, t1);
    else
      update_foot(z, R, L, idx, init, mod_sst, t1);
    L[idx].time = R[idx].time = z->now;
    z->now += M->T;
    idx++;
  }
  z->rel[0] = z->rel[1];                               /* pop_front */
  z->nrel = 1;
  for (int i = 0; i < add; i++) {
    z->fl[z->nf] = L[i];
    z->fr[z->nf] = R[i];
    z->nf++;
  }
  rc = filter_out(z, Z, add, 0);
done:
  free(Z);
  free(L);
  free(R);
  return rc;
}

/* EndPhaseOfTheWalking, :1129-1300 */
static int end_phase(zd_t *z) {
  const wg_zmpdisc_model_t *M = z->M;
  if (z->nrel > 0) update_support(z, &z->rel[0]);
  unsigned n_end = (unsigned)round(M->t_double / (2 * M->T));
  int n_rest = (int)(3.0 * M->preview_time / M->T);
  if (n_end == 0 || n_rest < 0) return WG_ZMPDISC_BAD_INPUT;
  int total = (int)n_end + n_rest;
  if (z->nf + total > z->cap || z->nz + total > z->cap) return WG_ZMPDISC_CAPACITY;
  zmp_t *Z = (zmp_t *)calloc((size_t)total, sizeof(zmp_t));
  int idx = 0;
  double px0 = z->fz[z->nz - 1].px, py0 = z->fz[z->nz - 1].py;
  double pxf = 0.5 * (z->sfp[0][2] + z->prev_sfp[0][2]);
  double pyf = 0.5 * (z->sfp[1][2] + z->prev_sfp[1][2]);
  double dx = (pxf - px0) / (double)n_end, dy = (pyf - py0) / (double)n_end;
  Z[0].px = px0 + dx;
  Z[0].py = py0 + dy;
  Z[0].time = z->now;
  Z[0].theta = z->fz[z->nz - 1].theta;
  Z[0].type = 0;
  idx++;
  for (int i = 0; i < total; i++) {
    if (i >= 1) {
      if ((unsigned)i < n_end) {
        Z[idx].px = Z[idx - 1].px + dx;
        Z[idx].py = Z[idx - 1].py + dy;
      } else {
        Z[idx].px = Z[idx - 1].px;
        Z[idx].py = Z[idx - 1].py;
      }
      Z[idx].time = z->now;
      Z[idx].theta = Z[idx - 1].theta;
      Z[idx].type = 0;
      idx++;
    }
    foot_t l = z->fl[z->nf - 1], r = z->fr[z->nf - 1];
    l.time = r.time = z->now;
    l.type = r.type = 0;
    z->fl[z->nf] = l;
    z->fr[z->nf] = r;
    z->nf++;
    z->now += M->T;
  }
  int rc = filter_out(z, Z, total, 0);
  free(Z);
  return rc;
}

int wgo_zmpdisc_length(const wg_zmpdisc_model_t *M, const wg_rel_step_t *steps, int n_steps) {
  if (n_steps < 2 || n_steps > WG_ZMPDISC_MAX_STEPS || !(M->T > 0)) return WG_ZMPDISC_BAD_INPUT;
  long n = (int)(2 * M->preview_time / M->T);
  for (int i = 1; i < n_steps; i++) {
    double d = M->t_double, s = M->t_single;
    if (steps[i].ds_time != 0.0) {
      d = steps[i].ds_time;
      s = steps[i].ss_time;
    }
    n += (int)(unsigned)round((d + s) / M->T);
  }
  n += (int)(unsigned)round(M->t_double / (2 * M->T)) + (int)(3.0 * M->preview_time / M->T);
  return n > (1 << 24) ? WG_ZMPDISC_BAD_INPUT : (int)n;
}

/* GetZMPDiscretization for one gait.  init_feet: left x, y, theta, right x, y, theta.  Arrays as wg_zmpdisc_batch with
 * B = 1.  Returns L or a negative code. */
int wgo_zmpdisc(const wg_zmpdisc_model_t *M, const wg_rel_step_t *steps, int n_steps, const double *init_feet, int lcap,
                double *zmp, double *zmp_theta, int *zmp_type, double *left, int *left_type, double *right,
                int *right_type, double *time) {
  int Ltot = wgo_zmpdisc_length(M, steps, n_steps);
  if (Ltot < 0) return Ltot;
  if (Ltot > lcap) return WG_ZMPDISC_CAPACITY;
  zd_t z;
  memset(&z, 0, sizeof z);
  z.M = M;
  if (init_filter(&z)) return WG_ZMPDISC_BAD_INPUT;
  z.cap = Ltot;
  z.fz = (zmp_t *)calloc((size_t)Ltot, sizeof(zmp_t));
  z.fl = (foot_t *)calloc((size_t)Ltot, sizeof(foot_t));
  z.fr = (foot_t *)calloc((size_t)Ltot, sizeof(foot_t));
  int rc = 0;
  /* InitOnLine, :319-513 (m_CurrentTime starts at the internal clock, 0) */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) z.sfp[i][j] = i == j ? 1.0 : 0.0;
  memcpy(z.prev_sfp, z.sfp, sizeof z.sfp);
  foot_t curL, curR;
  memset(&curL, 0, sizeof curL);
  memset(&curR, 0, sizeof curR);
  curL.x = init_feet[0];
  curL.y = init_feet[1];
  curL.theta = init_feet[2];
  curR.x = init_feet[3];
  curR.y = init_feet[4];
  curR.theta = init_feet[5];
  {
    double zt = (curR.theta + curL.theta) / 2.0;
    int who;
    z.rel[0] = steps[0];
    support_bookkeeping(&z, &curL, &curR, zt, &who);
  }
  int add = (int)(2 * M->preview_time / M->T);
  if (add < 3) {                                       /* FilterOutValues reads ZMPPositions[2] */
    rc = WG_ZMPDISC_BAD_INPUT;
    goto done;
  }
  {
    zmp_t *Z = (zmp_t *)calloc((size_t)add, sizeof(zmp_t));
    const double start[2] = {0.0, 0.0};
    const double fin[2] = {M->zmp_neutral[0], M->zmp_neutral[1]};
    for (int i = 0; i < add; i++) {
      double coef = (double)i / (double)add;
      Z[i].px = start[0] + (fin[0] - start[0]) * coef;
      Z[i].py = start[1] + (fin[1] - start[1]) * coef;
      Z[i].theta = 0.0;
      Z[i].time = z.now;
      Z[i].type = 0;
      z.fl[z.nf] = curL;
      z.fr[z.nf] = curR;
      z.fl[z.nf].time = z.fr[z.nf].time = z.now;
      z.fl[z.nf].type = z.fr[z.nf].type = 10;
      z.nf++;
      z.now += M->T;
    }
    z.nrel = 1;                                        /* m_RelativeFootPositions.push_back(RelativeFootPositions[0]) */
    rc = filter_out(&z, Z, add, 1);
    free(Z);
    if (rc) goto done;
  }
  for (int i = 1; i < n_steps; i++) {
    rc = add_foot(&z, &steps[i]);
    if (rc) goto done;
  }
  rc = end_phase(&z);
  if (rc) goto done;
  if (z.nz != Ltot || z.nf != Ltot) {
    rc = WG_ZMPDISC_BAD_INPUT;
    goto done;
  }
  for (int i = 0; i < Ltot; i++) {
    if (zmp) {
      zmp[2 * i] = z.fz[i].px;
      zmp[2 * i + 1] = z.fz[i].py;
    }
    if (zmp_theta) zmp_theta[i] = z.fz[i].theta;
    if (zmp_type) zmp_type[i] = z.fz[i].type;
    if (time) time[i] = z.fz[i].time;
    const foot_t *f[2] = {&z.fl[i], &z.fr[i]};
    double *o[2] = {left, right};
    int *ot[2] = {left_type, right_type};
    for (int s = 0; s < 2; s++) {
      if (o[s]) {
        double *p = o[s] + 6 * (size_t)i;
        p[0] = f[s]->x;
        p[1] = f[s]->y;
        p[2] = f[s]->z;
        p[3] = f[s]->theta;
        p[4] = f[s]->omega;
        p[5] = f[s]->omega2;
      }
      if (ot[s]) ot[s][i] = f[s]->type;
    }
  }
  rc = Ltot;
done:
  free(z.fz);
  free(z.fl);
  free(z.fr);
  return rc;
}

/* ---- FootConstraintsAsLinearSystem ---------------------------------------------------------------------------------- */

typedef struct {
  double col, row;
} chpt_t;

/* DoComputeConvexHull, ConvexHull.cpp:88-203: Graham scan around the lowest point; the reference keeps the candidates
 * in a std::set ordered by the sign of the cross product about p0 (same direction: the farther point survives). */
static int convex_hull(const chpt_t *pts, int n, chpt_t *hull) {
  if (n == 0) return 0;
  chpt_t p0 = pts[0];
  for (int i = 0; i < n; i++)
    if (pts[i].row < p0.row) p0 = pts[i];
  chpt_t set[16];
  int ns = 0;
  for (int i = 0; i < n; i++) {
    int insert = 1;
    double x2 = pts[i].col - p0.col, y2 = pts[i].row - p0.row;
    for (int k = 0; k < ns;) {
      double x1 = set[k].col - p0.col, y1 = set[k].row - p0.row;
      int del = 0;
      if ((x1 * y2 - x2 * y1) == 0.0) {
        double d1 = sqrt(x1 * x1 + y1 * y1), d2 = sqrt(x2 * x2 + y2 * y2);
        if (d1 <= d2)
          del = 1;
        else
          insert = 0;
      }
      if (del) {
        for (int q = k; q < ns - 1; q++) set[q] = set[q + 1];
        ns--;
      } else
        k++;
    }
    if (insert) {
      int pos = 0, dup = 0;
      for (; pos < ns; pos++) {                        /* first element the new point orders before */
        double x1 = set[pos].col - p0.col, y1 = set[pos].row - p0.row;
        if ((x2 * y1 - x1 * y2) > 0.0) break;          /* comp(new, elem) */
        if (!((x1 * y2 - x2 * y1) > 0.0)) dup = 1;     /* neither orders before the other: std::set refuses it */
      }
      if (!dup) {
        for (int q = ns; q > pos; q--) set[q] = set[q - 1];
        set[pos] = pts[i];
        ns++;
      }
    }
  }
  if (ns < 2) return -1;
  int nh = 0, it = 0;
  hull[nh++] = p0;
  hull[nh++] = set[it++];
  hull[nh++] = set[it++];
  while (it < ns) {
    chpt_t pi = set[it];
    int ok;
    do {
      if (nh >= 2) {
        chpt_t s1 = hull[nh - 1], s2 = hull[nh - 2];
        double x1 = s1.col - s2.col, x2 = pi.col - s2.col, y1 = s1.row - s2.row, y2 = pi.row - s2.row;
        ok = (x1 * y2 - x2 * y1) > 0.0;
      } else
        ok = 1;
      if (!ok) nh--;
    } while (!ok);
    hull[nh++] = set[it];
    it++;
  }
  return nh;
}

/* one edge of ComputeLinearSystem, :124-179 / :192-236: p -> q */
static void edge_row(chpt_t p, chpt_t q, double *a_out, double *c_out, double *b_out) {
  double a, b, c;
  if (fabs(q.col - p.col) > 1e-7) {
    double y1, x1, y2, x2, lmul = -1.0;
    if (q.col < p.col) {
      lmul = 1.0;
      y2 = p.row;
      y1 = q.row;
      x2 = p.col;
      x1 = q.col;
    } else {
      y2 = q.row;
      y1 = p.row;
      x2 = q.col;
      x1 = p.col;
    }
    a = (y2 - y1) / (x2 - x1);
    b = (p.row - a * p.col);
    a = lmul * a;
    b = lmul * b;
    c = -lmul;
  } else {
    c = 0.0;
    a = -1.0;
    b = q.col;
    if (q.row < p.row) {
      a = -a;
      b = -b;
    }
  }
  *a_out = a;
  *c_out = c;
  *b_out = b;
}

/* ComputeLinearSystem :97-256 + FindSimilarConstraints :55-92 */
static int linear_system(const chpt_t *h, int n, wg_zmp_polytope_t *P) {
  if (n < 2 || n > WG_POLY_MAX_ROWS) return -1;
  memset(P, 0, sizeof *P);
  P->nrows = n;
  double C0 = 0.0, C1 = 0.0;
  for (int i = 0; i < n - 1; i++) {
    C0 += h[i].col;
    C1 += h[i].row;
    edge_row(h[i], h[i + 1], &P->A[i][0], &P->A[i][1], &P->B[i]);
  }
  C0 += h[n - 1].col;
  C1 += h[n - 1].row;
  C0 /= (double)n;
  C1 /= (double)n;
  /* the closing edge: the reference orders the end points differently (:192-233) -- same line, written out */
  {
    chpt_t p = h[n - 1], q = h[0];
    double a, b, c;
    if (fabs(q.col - p.col) > 1e-7) {
      double y1, x1, y2, x2, lmul = -1.0;
      if (q.col < p.col) {
        lmul = 1.0;
        y2 = p.row;
        y1 = q.row;
        x2 = p.col;
        x1 = q.col;
      } else {
        y2 = q.row;
        y1 = p.row;
        x2 = q.col;
        x1 = p.col;
      }
      a = (y2 - y1) / (x2 - x1);
      b = (q.row - a * q.col);
      a = lmul * a;
      b = lmul * b;
      c = -lmul;
    } else {
      c = 0.0;
      a = -1.0;
      b = q.col;
      if (q.row < p.row) {
        a = -a;
        b = -b;
      }
    }
    P->A[n - 1][0] = a;
    P->A[n - 1][1] = c;
    P->B[n - 1] = b;
  }
  P->centre[0] = C0;
  P->centre[1] = C1;
  if (n == 4) {
    if (P->A[0][0] == -P->A[2][0] && P->A[0][1] == -P->A[2][1]) P->similar[2] = -2;
    if (P->A[1][0] == -P->A[3][0] && P->A[1][1] == -P->A[3][1]) P->similar[3] = -2;
  } else if (n == 6) {
    for (int k = 0; k < 3; k++)
      if (P->A[k][0] == -P->A[k + 3][0] && P->A[k][1] == -P->A[k + 3][1]) P->similar[k + 3] = -3;
  }
  return 0;
}

static void foot_corners(double lx, double ly, double theta_deg, double hw, double hh, chpt_t *out) {
  static const double lxc[4] = {1.0, 1.0, -1.0, -1.0}, lyc[4] = {-1.0, 1.0, 1.0, -1.0};
  double s = WSIN(theta_deg * M_PI / 180.0), c = WCOS(theta_deg * M_PI / 180.0);
  for (int j = 0; j < 4; j++) {
    out[j].col = lx + (lxc[j] * hw * c - lyc[j] * hh * s);
    out[j].row = ly + (lxc[j] * hw * s + lyc[j] * hh * c);
  }
}

/* BuildLinearConstraintInequalities, :258-539; arguments as wg_foot_constraints */
int wgo_foot_constraints(int n, const double *time, const double *left, const int *left_type, const double *right,
                         double sole_w, double sole_h, double cx, double cy, int cap, wg_zmp_polytope_t *polys,
                         double *t_start, double *t_end) {
  double lhw = sole_w * 0.5, lhh = sole_h * 0.5, rhw = sole_w * 0.5, rhh = sole_h * 0.5;
  lhh -= cy;
  rhh -= cy;
  lhw -= cx;
  rhw -= cx;
  int state = 0, count = 0;
  for (int i = 0; i < n; i++) {
    const double *L = left + 6 * (size_t)i, *R = right + 6 * (size_t)i;
    int compute = 0;
    if (i == 0) {
      compute = 1;
      state = 3;
    }
    if (left_type[i] >= 10) {
      if (state != 3) compute = 1;
      state = 3;
    } else {
      const double thr = 0.00001;
      if (L[2] > thr) {
        if (state != 2) compute = 1;
        state = 2;
      } else if (R[2] > thr) {
        if (state != 1) compute = 1;
        state = 1;
      } else if (R[2] < thr && L[2] < thr) {
        if (state != 3) compute = 1;
        state = 3;
      }
    }
    if (compute) {
      chpt_t hull[16];
      int nh;
      if (state == 3) {
        chpt_t pts[8];
        foot_corners(L[0], L[1], L[3], lhw, lhh, pts);
        foot_corners(R[0], R[1], R[3], rhw, rhh, pts + 4);
        nh = convex_hull(pts, 8, hull);
      } else {
        if (L[2] < R[2])
          foot_corners(L[0], L[1], L[3], lhw, lhh, hull);
        else
          foot_corners(R[0], R[1], R[3], rhw, rhh, hull);
        nh = 4;
      }
      wg_zmp_polytope_t P;
      if (linear_system(hull, nh, &P)) return -1;
      if (count > 0 && count - 1 < cap) t_end[count - 1] = time[i];
      if (count < cap) {
        polys[count] = P;
        t_start[count] = time[i];
        t_end[count] = time[i];
      }
      count++;
    }
    if (i == n - 1 && count > 0 && count - 1 < cap) t_end[count - 1] = time[i];
  }
  return count;
}

/* ---- StepStackHandler: the step generators behind ":supportfoot", ":arc", ":lastsupport" ------------------------------
 * src/StepStackHandler.cpp:754-764 (PrepareForSupportFoot), :299-457 (CreateArcInStepStack), :872-882
 * (FinishOnTheLastCorrectSupportFoot) -- the commands of TestKajita2003's TurningOnTheCircle (tests/TestKajita2003.cpp:68-93).
 * The reference leaves RelativeFootPosition::stepType uninitialised in the first two; 0 here.
 * Appends to out[*n]; *keep is m_KeepLastCorrectSupportFoot.  Returns 0, or -1 when cap is too small. */
static int push_step(wg_rel_step_t *out, int *n, int cap, double sx, double sy, double theta, double ss, double ds) {
  if (*n >= cap) return -1;
  wg_rel_step_t s;
  memset(&s, 0, sizeof s);
  s.sx = sx;
  s.sy = sy;
  s.theta = theta;
  s.ss_time = ss;
  s.ds_time = ds;
  s.step_type = 0;
  out[(*n)++] = s;
  return 0;
}
int wgo_steps_support_foot(int support_foot, double ss, double ds, wg_rel_step_t *out, int *n, int cap) {
  return push_step(out, n, cap, 0, support_foot * 0.095, 0, ss, ds);
}
int wgo_steps_last_support(int keep, double ss, double ds, wg_rel_step_t *out, int *n, int cap) {
  return push_step(out, n, cap, 0, keep * 0.19, 0, ss, ds);
}
int wgo_steps_arc(double x, double y, double arc_deg, int SupportFoot, double ss, double ds, wg_rel_step_t *out, int *n,
                  int cap, int *keep) {
  double StepMax = 0.15, LastStep, NumberOfStepFloat, OmegaStep, OmegaTotal = arc_deg * M_PI / 180.0, LastOmegaStep;
  int NumberOfStep, DirectionRay = -1;
  double R = sqrt(x * x + y * y);
  NumberOfStepFloat = OmegaTotal * R / StepMax;
  NumberOfStep = (int)floor(NumberOfStepFloat);
  LastStep = OmegaTotal * R - NumberOfStep * StepMax;
  OmegaStep = StepMax / R;
  LastOmegaStep = OmegaTotal - OmegaStep * NumberOfStep;
  OmegaStep = OmegaStep * 180.0 / M_PI;
  LastOmegaStep = LastOmegaStep * 180.0 / M_PI;
  if (x < 0) {
    StepMax = -StepMax;
    LastStep = -LastStep;
    DirectionRay = 1;
  }
  if (y < 0) {
    OmegaStep = -OmegaStep;
    LastOmegaStep = -LastOmegaStep;
  }
  double Omegakp, Omegak = 0.0;
  for (int i = 0; i < NumberOfStep + 1; i++) {
    const int last = i == NumberOfStep;
    if (last && LastStep == 0.0) break;
    const double dOmega = last ? LastOmegaStep : OmegaStep;
    Omegakp = Omegak;
    Omegak = Omegak + dOmega;
    double c = WCOS(Omegak * M_PI / 180.0), s = WSIN(Omegak * M_PI / 180.0);
    double cp = WCOS(Omegakp * M_PI / 180.0), sp = WSIN(Omegakp * M_PI / 180.0);
    double lv0 = (R + DirectionRay * SupportFoot * 0.095) * s - (R - DirectionRay * SupportFoot * 0.095) * sp;
    double lv1 = -((R + DirectionRay * SupportFoot * 0.095) * c - (R - DirectionRay * SupportFoot * 0.095) * cp);
    double a = 0.0, b = 0.0;                             /* lv2 = A lv, A = [c s; -s c] */
    a += c * lv0;
    a += s * lv1;
    b += -s * lv0;
    b += c * lv1;
    if (push_step(out, n, cap, a, b, dOmega, ss, ds)) return -1;
    SupportFoot = -SupportFoot;
  }
  *keep = SupportFoot;
  return 0;
}
