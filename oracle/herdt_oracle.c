/*
 * herdt_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of one Herdt-2010 MPC tick, i.e. of the body of
 *   ZMPVelocityReferencedQP::OnLine   src/ZMPRefTrajectoryGeneration/ZMPVelocityReferencedQP.cpp:346-452
 * and everything it calls (file:line given at each function), operating on the
 * flat per-gait state of include/wg_mpc.h instead of the reference's objects
 * and deques.  The reference sources on this path need boost::ublas, jrl-mal
 * and abstract-robot-dynamics, none of which is in this image, so they cannot
 * be compiled here ("unbuildable"); parity of this file is pinned end to end
 * by the reference's own golden file
 *   tests/TestHerdt2010EmergencyStopTestFGPI.datref.cmake
 * (tests/test_herdt_oracle.py, tolerance 1e-6 like tests/TestObject.cpp:477),
 * and the QP it assembles is solved by ql_oracle.c, itself bit-identical to
 * the compiled reference qld.cpp.
 *
 * Arithmetic follows the reference's evaluation order: ublas prod() sums k
 * ascending from 0.0; compute_term() scales AFTER the product
 * (generator-vel-ref.cpp:751-787); add_term_to() accumulates with += in call
 * order (qp-problem.cpp:464-480).
 */
#define _USE_MATH_DEFINES
#define _DEFAULT_SOURCE
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/wg_mpc.h"
#include "wg_oracle.h"

/* Trigonometry: libm by default (what the reference calls).  -DWGO_PORTABLE_TRIG builds the
 * variant that shares include/wg_trig.h with the HIP kernels so GPU parity can be bit-exact. */
#ifdef WGO_PORTABLE_TRIG
#include "../include/wg_trig.h"
#define WSIN(x) wg_sin(x)
#define WCOS(x) wg_cos(x)
#else
#define WSIN(x) sin(x)
#define WCOS(x) cos(x)
#endif

#define NMAXH 32                     /* largest horizon supported            */
#define SMAX 6                       /* largest number of previewed steps    */
#define NV (2 * NMAXH + 2 * SMAX)
#define MC (1 + 4 * NMAXH + 5 * SMAX)

typedef struct {
  int phase, foot, nb_steps_left, step_number, state_changed;
  double time_limit, start_time, x, y, yaw;
} sup_t;

typedef struct {
  double Sv[NMAXH][3], Uv[NMAXH][NMAXH];
  double Sz[NMAXH][3], Uz[NMAXH][NMAXH];
  double Qb[NMAXH][NMAXH];
} tables_t;

/* ---------------------------------------------------------------------- */
/* constants                                                               */
/* ---------------------------------------------------------------------- */
void wgo_model_defaults(wg_model_t *m) {
  memset(m, 0, sizeof *m);
  m->N = 16; m->T = 0.1; m->Tctrl = 0.005; m->com_height_qp = 0.814;
  m->alpha = 1.0; m->beta = 0.00001; m->gamma = 0.000001;
  /* robot dependent; defaults = jrl-dynamics' sample robot as fitted on the reference's golden
   * file (ZMP sticks to +-0.085 / +-0.03 around the support foot => sole 0.25 x 0.14) */
  m->sole_w = 0.25; m->sole_h = 0.14;
  m->margin_x = 0.04; m->margin_y = 0.04; m->ds_feet_distance = 0.2;
  /* the sample robot declares no hip-yaw limits: OrientationsPreview.cpp:46-66 then falls back to
   * -30/+45 deg for BOTH legs, and |upperVelocityBound| = 0 (:68) */
  m->hip_l_lo = -30.0 / 180.0 * M_PI; m->hip_l_hi = 45.0 / 180.0 * M_PI;
  m->hip_r_lo = -30.0 / 180.0 * M_PI; m->hip_r_hi = 45.0 / 180.0 * M_PI;
  m->hip_vmax = 0.0;
  m->hip_amax = 0.1; m->feet_cross_max = 5.0 / 180.0 * M_PI;
  m->step_period = 0.8; m->ds_period = 1e9; m->dsss_period = 0.8;
  m->t_single = 0.7; m->t_double = 0.1; m->step_height = 0.05; m->feet_distance = 0.2;
}

/* RigidBodySystem::compute_dyn_cjerk, rigid-body-system.cpp:377-452, and
 * GeneratorVelRef::build_invariant_part, generator-vel-ref.cpp:587-614 */
static void build_tables(const wg_model_t *m, tables_t *t) {
  const int N = m->N;
  const double T = m->T, h = m->com_height_qp;
  memset(t, 0, sizeof *t);
  for (unsigned i = 0; i < (unsigned)N; i++) {
    t->Sv[i][0] = 0.0; t->Sv[i][1] = 1.0; t->Sv[i][2] = (i + 1) * T;
    t->Sz[i][0] = 1.0; t->Sz[i][1] = (i + 1) * T;
    t->Sz[i][2] = (i + 1) * (i + 1) * T * T * 0.5 - h / 9.81;
    for (unsigned j = 0; j < (unsigned)N; j++) {
      if (j <= i) {
        t->Uv[i][j] = (2 * (i - j) + 1) * T * T * 0.5;
        t->Uz[i][j] = (1 + 3 * (i - j) + 3 * (i - j) * (i - j)) * T * T * T / 6.0 - T * h / 9.81;
      } else { t->Uv[i][j] = 0.0; t->Uz[i][j] = 0.0; }
    }
  }
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      double pj = 0.0, pv = 0.0, pz = 0.0;
      for (int k = 0; k < N; k++) {
        pj += ((k == i) ? 1.0 : 0.0) * ((k == j) ? 1.0 : 0.0);
        pv += t->Uv[k][i] * t->Uv[k][j];
        pz += t->Uz[k][i] * t->Uz[k][j];
      }
      double q = 0.0;
      q += pj * m->beta;
      q += pv * m->alpha;
      q += pz * m->gamma;
      t->Qb[i][j] = q;
    }
}

/* the invariant Hessian block alone (N x N, row-major): checker of wg_gramian_batch */
int wgo_invariant_hessian(const wg_model_t *m, double *Qb) {
  if (!m || !Qb || m->N < 1 || m->N > NMAXH) return -1;
  static tables_t t;
  build_tables(m, &t);
  for (int i = 0; i < m->N; i++)
    for (int j = 0; j < m->N; j++) Qb[i * m->N + j] = t.Qb[i][j];
  return 0;
}

/* ZMPVelocityReferencedQP::InitOnLine, ZMPVelocityReferencedQP.cpp:212-319 */
void wgo_gait_init(const wg_model_t *m, wg_gait_state_t *s, const double com0[3],
                   const double left_xyt[3], const double right_xyt[3]) {
  (void)m;
  memset(s, 0, sizeof *s);
  s->upper_time_limit = 0.0;
  s->online = 1; s->ending_phase = 0; s->time_to_stop = -1.0; s->running = 0;
  s->clock = 0.0;
  for (int k = 0; k < 3; k++) {
    s->lf[k].x = left_xyt[0]; s->lf[k].y = left_xyt[1]; s->lf[k].theta = left_xyt[2];
    s->rf[k].x = right_xyt[0]; s->rf[k].y = right_xyt[1]; s->rf[k].theta = right_xyt[2];
  }
  s->phase = WG_DS; s->foot = WG_LEFT; s->time_limit = 1000000000; s->nb_steps_left = 1;
  s->state_changed = 0; s->sup_x = left_xyt[0]; s->sup_y = left_xyt[1];
  s->sup_yaw = left_xyt[2] * M_PI / 180; s->start_time = 0.0; s->step_number = 0;
  s->com_x[0] = com0[0]; s->com_y[0] = com0[1]; s->com_z = com0[2];
  s->front_com_x[0] = com0[0]; s->front_com_y[0] = com0[1];
  s->nb_steps_ssds = 2;                 /* ZMPVelocityReferencedQP.cpp:79 */
  s->rot_support_foot = WG_LEFT;        /* SupportFSM.cpp:38 */
}

/* ---------------------------------------------------------------------- */
/* SupportFSM, src/PreviewControl/SupportFSM.cpp                           */
/* ---------------------------------------------------------------------- */
#define FSM_EPS 1e-6

/* SupportFSM::update_vel_reference :57-90 */
static void fsm_update_vel_reference(wg_gait_state_t *s, double ref[3], int cur_foot) {
  s->in_translation = (fabs(ref[0]) > 2 * FSM_EPS || fabs(ref[1]) > 2 * FSM_EPS) ? 1 : 0;
  if (fabs(ref[2]) > FSM_EPS) s->in_rotation = 1;
  else {
    if (s->in_rotation && !s->in_translation) {
      ref[0] = 2 * FSM_EPS; ref[1] = 2 * FSM_EPS;
      if (!s->post_rotation_phase) {
        s->rot_support_foot = cur_foot; s->nb_steps_after_rotation = 0; s->post_rotation_phase = 1;
      } else {
        if (s->rot_support_foot != cur_foot) { s->rot_support_foot = cur_foot; ++s->nb_steps_after_rotation; }
        if (s->nb_steps_after_rotation > 2) { s->in_rotation = 0; s->post_rotation_phase = 0; }
      }
    } else s->in_rotation = 0;
  }
}

/* SupportFSM::set_support_state :93-153 */
static void fsm_set_support_state(const wg_model_t *m, int nb_steps_ssds, double time, unsigned pi,
                                  sup_t *S, const double ref[3]) {
  const double T = m->T;
  S->state_changed = 0;
  int given = (fabs(ref[0]) > FSM_EPS || fabs(ref[1]) > FSM_EPS || fabs(ref[2]) > FSM_EPS);
  if (given && S->phase == WG_DS && (S->time_limit - time - FSM_EPS) > m->dsss_period) {
    S->time_limit = time + m->dsss_period - T / 10.0;
    S->nb_steps_left = nb_steps_ssds;
  }
  if (time + FSM_EPS + pi * T >= S->time_limit) {
    if (S->phase == WG_SS && !given && S->nb_steps_left == 0) {
      S->phase = WG_DS;
      S->time_limit = time + pi * T + m->ds_period - T / 10.0;
      S->state_changed = 1;
    } else if ((S->phase == WG_DS && given) || (S->phase == WG_DS && S->nb_steps_left > 0)) {
      S->phase = WG_SS;
      S->time_limit = time + pi * T + m->step_period - T / 10.0;
      S->nb_steps_left = nb_steps_ssds;
      S->state_changed = 1;
    } else if ((S->phase == WG_SS && S->nb_steps_left > 0) || (S->nb_steps_left == 0 && given)) {
      S->foot = (S->foot == WG_LEFT) ? WG_RIGHT : WG_LEFT;
      S->state_changed = 1;
      S->time_limit = time + pi * T + m->step_period - T / 10.0;
      if (pi != 1) ++S->step_number;
      if (!given) S->nb_steps_left = S->nb_steps_left - 1;
      if (given) S->nb_steps_left = nb_steps_ssds;
    }
  }
}

/* ---------------------------------------------------------------------- */
/* RelativeFeetInequalities, src/Mathematics/relative-feet-inequalities.cpp */
/* ---------------------------------------------------------------------- */
typedef struct { int nv; double X[5], Y[5], A[5], B[5], D[5]; } hull_t;

/* init_convex_hulls :88-149 + FootHalfSize.cpp:62-84, then set_vertices :185-234 */
static void hull_set_vertices(const wg_model_t *m, hull_t *H, const sup_t *S, int feet) {
  if (!feet) {
    static const double lxr[4] = {1.0, 1.0, -1.0, -1.0}, lyr[4] = {-1.0, 1.0, 1.0, -1.0};
    static const double lxl[4] = {1.0, 1.0, -1.0, -1.0}, lyl[4] = {1.0, -1.0, -1.0, 1.0};
    double hw = 0.5 * m->sole_w; hw -= m->margin_x;
    double hh = 0.5 * m->sole_h; hh -= m->margin_y;
    double hhds = hh + m->ds_feet_distance / 2.0;
    H->nv = 4;
    for (int j = 0; j < 4; j++) {
      if (S->foot == WG_LEFT) {
        H->X[j] = lxl[j] * hw;
        H->Y[j] = (S->phase == WG_DS) ? lyl[j] * hhds - m->ds_feet_distance / 2.0 : lyl[j] * hh;
      } else {
        H->X[j] = lxr[j] * hw;
        H->Y[j] = (S->phase == WG_DS) ? lyr[j] * hhds + m->ds_feet_distance / 2.0 : lyr[j] * hh;
      }
    }
  } else {
    static const double px[5] = {-0.28, -0.2, 0.0, 0.2, 0.28};
    static const double py[5] = {-0.2, -0.3, -0.4, -0.3, -0.2};
    H->nv = 5;
    for (int j = 0; j < 5; j++) { H->X[j] = px[j]; H->Y[j] = (S->foot == WG_LEFT) ? py[j] : -py[j]; }
  }
  /* convex_hull_t::rotate, privatepgtypes.cpp:157-185 */
  for (int j = 0; j < H->nv; j++) {
    double xo = H->X[j], yo = H->Y[j];
    H->X[j] = (xo * WCOS(S->yaw) - yo * WSIN(S->yaw));
    H->Y[j] = (xo * WSIN(S->yaw) + yo * WCOS(S->yaw));
  }
}

/* compute_linear_system :264-319 */
static void hull_linear_system(hull_t *H, int foot) {
  double sign = (foot == WG_LEFT) ? 1.0 : -1.0;
  for (int i = 0; i < H->nv; i++) {
    int i2 = (i + 1 == H->nv) ? 0 : i + 1;
    double y1 = H->Y[i], y2 = H->Y[i2], x1 = H->X[i], x2 = H->X[i2];
    double dx = y1 - y2, dy = x2 - x1;
    double dc = dx * x1 + dy * y1;
    H->A[i] = sign * dx; H->B[i] = sign * dy; H->D[i] = sign * dc;
  }
}

/* ---------------------------------------------------------------------- */
/* OrientationsPreview, src/ZMPRefTrajectoryGeneration/OrientationsPreview.cpp */
/* ---------------------------------------------------------------------- */
#define OP_EPS 0.00000001

typedef struct { double sup_time_passed, sign_rot_vel; } op_tmp_t;

/* verify_angle_hip_joint :273-303 */
static int op_verify_angle(const wg_model_t *m, wg_gait_state_t *s, op_tmp_t *o, const sup_t *cur,
                           double trunk_end, double cur_sup_angle, unsigned step_number) {
  double ul, ll;
  if (cur->foot == WG_LEFT) { ul = m->hip_l_hi; ll = m->hip_l_lo; } else { ul = m->hip_r_hi; ll = m->hip_r_lo; }
  double lim = (s->trunkT_yaw[1] < 0.0) ? ll : ul;
  if (fabs(trunk_end - cur_sup_angle) > fabs(lim)) {
    s->trunkT_yaw[1] = (cur_sup_angle + 0.9 * lim - s->trunk_yaw[0] - s->trunk_yaw[1] * m->T / 2.0) /
                       (o->sup_time_passed + step_number * m->step_period - m->T / 2.0);
    return 0;
  }
  return 1;
}

/* preview_orientations :79-251.  sup[0..N] are the previewed support states
 * (their Yaw is rewritten), sup_angles[] / trunk[] the two output deques. */
static void op_preview(const wg_model_t *m, wg_gait_state_t *s, double time, const double ref[3],
                       sup_t *sup, double *sup_angles, int *n_sup_angles, double *trunk) {
  const int N = m->N;
  const double T = m->T, SSP = m->step_period;
  op_tmp_t o;
  sup_t cur = sup[0];
  /* verify_acceleration_hip_joint :254-270 */
  if (cur.phase != WG_DS) {
    if (fabs(ref[2] - s->trunk_yaw[1]) > 2.0 / 3.0 * T * m->hip_amax) {
      double sg = (ref[2] - s->trunk_yaw[1] < 0.0) ? -1.0 : 1.0;
      s->trunkT_yaw[1] = s->trunk_yaw[1] + sg * 2.0 / 3.0 * T * m->hip_amax;
    } else s->trunkT_yaw[1] = ref[2];
  } else s->trunkT_yaw[1] = 0.0;

  const wg_foot_sample_t *LB = &s->lf[2], *RB = &s->rf[2];
  int vel_ok = 0, angle_ok = 0;
  double first_prw = 0.0;
  o.sign_rot_vel = (s->trunkT_yaw[1] < 0.0) ? -1.0 : 1.0;
  o.sup_time_passed = 0.0;
  unsigned step_number = 0;
  double trunk_end = 0.0;
  int na = 0;
  /* The reference loops `while(!TrunkVelOK)` / `while(!TrunkAngleOK)` without bound (OrientationsPreview.cpp:118, 126);
   * for some references the conditions are never met and it hangs.  Both this restatement and the kernel stop after 64
   * passes (tools/probe_run.py found such a gait: seed 20100, gait 2012, tick 161). */
  int guard = 0;
  while (!vel_ok && guard++ < 64) {
    double cur_sup_angle = (cur.foot == WG_LEFT) ? s->lf[0].theta * M_PI / 180.0 : s->rf[0].theta * M_PI / 180.0;
    if (cur.phase != WG_DS) {
      angle_ok = 0;
      int g2 = 0;
      while (!angle_ok && g2++ < 64) {
        if (fabs(s->trunkT_yaw[1] - s->trunk_yaw[1]) > OP_EPS) {
          double a = s->trunk_yaw[0], b = s->trunk_yaw[1], c = 0.0;
          double d = 3.0 * (s->trunkT_yaw[1] - s->trunk_yaw[1]) / (T * T);
          double e = -2.0 * d / (3.0 * T);
          s->trunkT_yaw[0] = a + b * T + 1.0 / 2.0 * c * T * T + 1.0 / 3.0 * d * T * T * T + 1.0 / 4.0 * e * T * T * T * T;
        } else s->trunkT_yaw[0] = s->trunk_yaw[0] + s->trunk_yaw[1] * T;
        o.sup_time_passed = cur.time_limit - time;
        trunk_end = s->trunkT_yaw[0] + s->trunkT_yaw[1] * (o.sup_time_passed - T);
        angle_ok = op_verify_angle(m, s, &o, &cur, trunk_end, cur_sup_angle, step_number);
      }
    } else {
      o.sup_time_passed = cur.time_limit + SSP - time;
      first_prw = 1;
      sup_angles[na++] = cur_sup_angle;
      s->trunkT_yaw[0] = trunk_end = s->trunk_yaw[0];
    }
    double prev_sup_angle = cur_sup_angle;
    double prw_foot = (cur.foot == WG_LEFT) ? 1.0 : -1.0;
    double cur_l_angle = LB->theta * M_PI / 180.0, cur_r_angle = RB->theta * M_PI / 180.0;
    unsigned last = (unsigned)((int)ceil((N + 1) * T / SSP));
    for (step_number = (unsigned)first_prw; step_number <= last; step_number++) {
      prw_foot = -prw_foot;
      double prw_angle = trunk_end + s->trunkT_yaw[1] * SSP / 2.0;
      /* verify_velocity_hip_joint (:306-365) takes its angle by value: no effect */
      if ((double)prw_foot * (prev_sup_angle - prw_angle) - OP_EPS > m->feet_cross_max)
        prw_angle = prev_sup_angle + (double)o.sign_rot_vel * m->feet_cross_max;
      else if (fabs(prw_angle - prev_sup_angle) > m->hip_vmax * SSP)
        prw_angle = prev_sup_angle + (double)prw_foot * m->hip_vmax * (SSP - T);
      angle_ok = op_verify_angle(m, s, &o, &cur, trunk_end, cur_sup_angle, step_number);
      if (!angle_ok) { na = 0; vel_ok = 0; break; }
      else if (na < 8) sup_angles[na++] = prw_angle;
      trunk_end = trunk_end + SSP * s->trunkT_yaw[1];
      prev_sup_angle = prw_angle;
      if (prw_foot == 1) cur_l_angle = prw_angle; else cur_r_angle = prw_angle;
      vel_ok = 1;
    }
    (void)cur_l_angle; (void)cur_r_angle;
  }
  *n_sup_angles = na;
  trunk[0] = s->trunk_yaw[0];
  trunk[1] = s->trunkT_yaw[0];
  for (int i = 1; i < N; i++) trunk[i + 1] = s->trunkT_yaw[0] + s->trunkT_yaw[1] * T;
  double sup_angle = sup[0].yaw;
  int j = 0;
  for (int i = 1; i <= N; i++) {
    if (sup[i].state_changed) { sup_angle = sup_angles[j]; j++; }
    sup[i].yaw = sup_angle;
  }
}

/* interpolate_trunk_orientation :368-418 */
static void op_interpolate_trunk(const wg_model_t *m, wg_gait_state_t *s, double time, const sup_t *cur,
                                 wg_tick_out_t *out) {
  const double T = m->T, dt = m->Tctrl;
  const int K = (int)(T / dt);
  if (cur->phase == WG_SS && time + 3.0 / 2.0 * T < cur->time_limit) {
    double a = s->trunk_yaw[1];
    double c = 3.0 * (s->trunkT_yaw[1] - s->trunk_yaw[1]) / (T * T);
    double d = -2.0 * c / (3.0 * T);
    double theta = s->trunk_yaw[0];
    if (out) { out->com_yaw[0][0] = s->trunk_yaw[0]; out->com_yaw[0][1] = s->trunk_yaw[1]; }
    for (int k = 0; k < K; k++) {
      double tT = (double)(k + 1) * dt;
      if (fabs(s->trunkT_yaw[1] - s->trunk_yaw[1]) - 0.000001 > 0) {
        s->trunk_yaw[0] = (((1.0 / 4.0 * d * tT + 1.0 / 3.0 * c) * tT) * tT + a) * tT + theta;
        s->trunk_yaw[1] = ((d * tT + c) * tT) * tT + a;
        s->trunk_yaw[2] = (3.0 * d * tT + 2.0 * c) * tT;
      } else s->trunk_yaw[0] += dt * s->trunkT_yaw[1];
      if (out) { out->com_yaw[k][0] = s->trunk_yaw[0]; out->com_yaw[k][1] = s->trunk_yaw[1]; }
    }
  } else if (cur->phase == WG_DS || time + 3.0 / 2.0 * T > cur->time_limit) {
    for (int k = 0; k < K; k++)
      if (out) { out->com_yaw[k][0] = s->trunk_yaw[0]; out->com_yaw[k][1] = s->trunk_yaw[1]; }
  }
}

/* ---------------------------------------------------------------------- */
/* foot polynomials, src/Mathematics/PolynomeFoot.cpp / Polynome.cpp       */
/* ---------------------------------------------------------------------- */
static double poly_eval(const double *c, int deg, double t) {      /* Polynome.cpp:44-54 */
  double r = 0.0, pt = 1.0;
  for (int i = 0; i <= deg; i++) { r += c[i] * pt; pt *= t; }
  return r;
}
static double poly_d1(const double *c, int deg, double t) {        /* :56-65 */
  double r = 0, pt = 1;
  for (int i = 1; i <= deg; i++) { r += i * c[i] * pt; pt *= t; }
  return r;
}
static double poly_d2(const double *c, int deg, double t) {        /* :67-76 */
  double r = 0, pt = 1;
  for (int i = 2; i <= deg; i++) { r += i * (i - 1) * c[i] * pt; pt *= t; }
  return r;
}
/* Polynome5::SetParameters(FT, FP, InitPos, InitSpeed, InitAcc)  PolynomeFoot.cpp:226-240 */
static void poly5_set(double *c, double FT, double FP, double p0, double v0, double a0) {
  double tmp;
  c[0] = p0; c[1] = v0; c[2] = a0 / 2.0;
  tmp = FT * FT * FT;
  c[3] = (-3.0 / 2.0 * a0 * FT * FT - 6.0 * v0 * FT - 10.0 * p0 + 10.0 * FP) / tmp;
  tmp = tmp * FT;
  c[4] = (3.0 / 2.0 * a0 * FT * FT + 8.0 * v0 * FT + 15.0 * p0 - 15.0 * FP) / tmp;
  tmp = tmp * FT;
  c[5] = (-1.0 / 2.0 * a0 * FT * FT - 3.0 * v0 * FT - 6.0 * p0 + 6.0 * FP) / tmp;
}
/* Polynome4::SetParameters(FT, MP)  :100-120 */
static void poly4_set(double *c, double FT, double MP) {
  double tmp;
  c[0] = 0.0; c[1] = 0.0;
  tmp = FT * FT;
  if (MP == 0.0 || tmp == 0.0) { c[2] = 0.0; c[3] = 0.0; c[4] = 0.0; }
  else {
    c[2] = 16.0 * MP / tmp;
    tmp = tmp * FT;
    c[3] = -32.0 * MP / tmp;
    tmp = tmp * FT;
    c[4] = 16.0 * MP / tmp;
  }
}
/* Polynome3::SetParametersWithInitPosInitSpeed  :59-79 */
static void poly3_set(double *c, double FT, double FP, double p0, double v0) {
  double tmp;
  c[0] = p0; c[1] = v0;
  tmp = FT * FT;
  if (FT == 0.0) { c[2] = 0.0; c[3] = 0.0; }
  else {
    c[2] = (3 * FP - 3 * p0 - 2 * v0 * FT) / tmp;
    c[3] = (v0 * FT + 2 * p0 - 2 * FP) / (tmp * FT);
  }
}

/* ---------------------------------------------------------------------- */
/* the tick                                                                */
/* ---------------------------------------------------------------------- */
typedef struct {
  int n, m, mmax;
  double *C, *d, *A, *b, *xl, *xu;     /* column-major, ld n / mmax */
} qp_t;

typedef int (*wgo_ql0001_fn)(int *m, int *me, int *mmax, int *n, int *nmax, int *mnn, double *c, double *d, double *a,
                             double *b, double *xl, double *xu, double *x, double *u, int *iout, int *ifail, int *iprint,
                             double *war, int *lwar, int *iwar, int *liwar, double *eps1);
static wgo_ql0001_fn g_ref_ql = 0;
/* bench.py's "solve-only" CPU figure: wall time spent inside the reference's ql0001_ alone, next to the
 * assemble + solve time of the whole tick */
static int g_solve_timer = 0;
static double g_solve_seconds = 0.0;
static long g_solve_count = 0;
void wgo_solve_timer(int enable, double *seconds, long *count) {
  if (seconds) *seconds = g_solve_seconds;
  if (count) *count = g_solve_count;
  g_solve_timer = enable; g_solve_seconds = 0.0; g_solve_count = 0;
}
/* route the tick's QP through the reference's own compiled solver (NULL = this directory's restatement) */
void wgo_set_reference_ql(void *ql0001_entry) { g_ref_ql = (wgo_ql0001_fn)ql0001_entry; }

/* One ql0001_ call the way the reference's drivers make it (QPProblem::solve, qp-problem.cpp:256-279, and
 * ZMPConstrainedQPFastFormulation.cpp:1280-1307): iwar[0] = lql (1: C is the Hessian; 0: C is its upper-triangular Cholesky
 * factor, what mode QLDANDLQ says of its identity), eps = 1e-8, lwar = 3 nmax^2/2 + 10 nmax + 2 mmax + 20000 -- by the reference's
 * COMPILED ql0001_ when wgo_set_reference_ql gave one (then *n_iter = -1: it does not report iterations), by the restatement
 * otherwise.  The restatement has the lql branch only (ql_oracle.c); it is handed C as a Hessian either way -- for the identity
 * the same matrix, and the compiled reference returns the same bits on it with iwar[0] = 0 and 1 (tests/test_dimitrov_gpu.py
 * runs the compiled one the driver's way).  *nact = number of non-zero multipliers (compiled) / active constraints (restated). */
int wgo_ql_call(int lql, int m, int me, int mmax, int n, int nmax, double *C, double *d, double *A, double *b, double *xl, double *xu,
                double *x, double *u, int *ifail, int *n_iter, int *nact) {
  int *iact = (int *)calloc((size_t)n + 1, sizeof(int));
  *nact = 0;
  if (g_ref_ql) {
    int m_ = m, me_ = me, mmax_ = mmax, n_ = n, nmax_ = nmax, mnn_ = m + 2 * n, iout_ = 0, iprint_ = 1, liwar_ = n;
    int lwar_ = 3 * nmax * nmax / 2 + 10 * nmax + 2 * mmax + 20000;
    double eps_ = 1e-8;
    double *war = (double *)calloc((size_t)lwar_, sizeof(double));
    iact[0] = lql ? 1 : 0;
    g_ref_ql(&m_, &me_, &mmax_, &n_, &nmax_, &mnn_, C, d, A, b, xl, xu, x, u, &iout_, ifail, &iprint_, war, &lwar_, iact, &liwar_, &eps_);
    free(war);
    *n_iter = -1;
    for (int i = 0; i < m + 2 * n; i++) *nact += (u[i] != 0.0);
  } else {
    int hlen = 0;
    wgo_ql_solve(m, me, mmax, n, nmax, C, d, A, b, xl, xu, 1e-8, x, u, ifail, iact, nact, n_iter, NULL, 0, &hlen);
  }
  free(iact);
  return *ifail;
}

int wgo_mpc_tick(const wg_model_t *m, wg_gait_state_t *s, wg_tick_out_t *out, wgo_qp_dump_t *dump) {
  const int N = m->N;
  const double T = m->T;
  const int K = (int)(T / m->Tctrl);
  if (N > NMAXH || K != WG_SAMPLES_PER_TICK) return -1;
  tables_t *tb = (tables_t *)malloc(sizeof(tables_t));
  if (!tb) return -2;
  build_tables(m, tb);
  const double time = s->clock;
  int rc = 0;

  /* --- :353-357 reference + current support ---------------------------------- */
  double ref[3] = {s->vref[0], s->vref[1], s->vref[2]};
  fsm_update_vel_reference(s, ref, s->foot);
  sup_t sup[NMAXH + 1];
  sup_t cur;
  cur.phase = s->phase; cur.foot = s->foot; cur.nb_steps_left = s->nb_steps_left;
  cur.step_number = s->step_number; cur.state_changed = s->state_changed;
  cur.time_limit = s->time_limit; cur.start_time = s->start_time;
  cur.x = s->sup_x; cur.y = s->sup_y; cur.yaw = s->sup_yaw;

  /* --- preview_support_states, generator-vel-ref.cpp:70-134 ------------------ */
  fsm_set_support_state(m, s->nb_steps_ssds, time, 0, &cur, ref);
  if (cur.state_changed) {
    const wg_foot_sample_t *f = (cur.foot == WG_LEFT) ? &s->lf[0] : &s->rf[0];
    cur.x = f->x; cur.y = f->y; cur.yaw = f->theta * M_PI / 180.0; cur.start_time = time;
  }
  sup[0] = cur;
  s->phase = cur.phase; s->foot = cur.foot; s->nb_steps_left = cur.nb_steps_left;
  s->step_number = cur.step_number; s->state_changed = cur.state_changed;
  s->time_limit = cur.time_limit; s->start_time = cur.start_time;
  s->sup_x = cur.x; s->sup_y = cur.y; s->sup_yaw = cur.yaw;
  {
    sup_t prw = cur;
    prw.step_number = 0;
    for (unsigned pi = 1; pi <= (unsigned)N; pi++) {
      fsm_set_support_state(m, s->nb_steps_ssds, time, pi, &prw, ref);
      if (prw.state_changed) {
        if (pi == 1) {
          const wg_foot_sample_t *f = (prw.foot == WG_LEFT) ? &s->lf[2] : &s->rf[2];
          prw.x = f->x; prw.y = f->y; prw.yaw = f->theta * M_PI / 180.0; prw.start_time = time + pi * T;
        }
        if (prw.step_number > 0) { prw.x = 0.0; prw.y = 0.0; }
      }
      sup[pi] = prw;
    }
  }
  const int ns = sup[N].step_number;            /* NbPrwSteps */
  if (ns > SMAX) { free(tb); return -3; }

  /* --- generate_selection_matrices :137-208 ----------------------------------- */
  int stepidx[NMAXH];                            /* V(i, stepidx-1) = 1 */
  double VcX[NMAXH], VcY[NMAXH], Vc_fX[SMAX], Vc_fY[SMAX], V_f[SMAX][SMAX];
  memset(V_f, 0, sizeof V_f);
  for (int k = 0; k < SMAX; k++) { Vc_fX[k] = 0.0; Vc_fY[k] = 0.0; }
  for (int i = 0; i < N; i++) {
    const sup_t *S = &sup[i + 1];
    VcX[i] = 0.0; VcY[i] = 0.0; stepidx[i] = 0;
    if (S->step_number > 0) {
      stepidx[i] = S->step_number;
      if (S->step_number == 1 && S->state_changed && S->phase == WG_SS) {
        Vc_fX[0] = sup[i].x; Vc_fY[0] = sup[i].y;
        V_f[0][0] = 1.0;
      } else if (S->step_number > 1) {
        V_f[S->step_number - 1][S->step_number - 2] = -1.0;
        V_f[S->step_number - 1][S->step_number - 1] = 1.0;
      }
    } else { VcX[i] = S->x; VcY[i] = S->y; }
  }

  /* --- preview_orientations ---------------------------------------------------- */
  double sup_angles[8] = {0, 0, 0, 0, 0, 0, 0, 0}, trunk[NMAXH + 1];
  int n_sup_angles = 0;
  op_preview(m, s, time, ref, sup, sup_angles, &n_sup_angles, trunk);

  /* --- compute_global_reference :211-229 --------------------------------------- */
  double refx[NMAXH], refy[NMAXH];
  for (int i = 0; i < N; i++) {
    double yt = trunk[i];
    refx[i] = ref[0] * WCOS(yt) - ref[1] * WSIN(yt);
    refy[i] = ref[1] * WCOS(yt) + ref[0] * WSIN(yt);
  }

  /* --- QP storage ------------------------------------------------------------- */
  const int n = 2 * N + 2 * ns;
  const int mreal = 4 * N + 5 * ns;
  const int mq = mreal + 1, mmax = mq + 1;       /* qp-problem.cpp:248-253 */
  double *C = (double *)calloc((size_t)n * n, sizeof(double));
  double *d = (double *)calloc((size_t)n, sizeof(double));
  double *A = (double *)calloc((size_t)mmax * n, sizeof(double));
  double *b = (double *)calloc((size_t)mmax, sizeof(double));
  double *xl = (double *)malloc(sizeof(double) * (size_t)n), *xu = (double *)malloc(sizeof(double) * (size_t)n);
  double *x = (double *)calloc((size_t)n, sizeof(double));
  double *u = (double *)calloc((size_t)mq + 2 * (size_t)n, sizeof(double));
#define Cq(i, j) C[(i) + (size_t)(j) * n]
#define Aq(r, c) A[(r) + (size_t)(c) * mmax]
  for (int i = 0; i < n; i++) { xl[i] = -1e8; xu[i] = 1e8; }
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) { Cq(i, j) = tb->Qb[i][j]; Cq(N + i, N + j) = tb->Qb[i][j]; }

  /* --- update_problem :617-674 -------------------------------------------------- */
  {
    double svx[NMAXH], svy[NMAXH], szx[NMAXH], szy[NMAXH];
    for (int i = 0; i < N; i++) {                 /* MV2_ = prod(S, CoM) */
      double ax = 0.0, ay = 0.0, bx = 0.0, by = 0.0;
      for (int k = 0; k < 3; k++) {
        ax += tb->Sv[i][k] * s->com_x[k]; ay += tb->Sv[i][k] * s->com_y[k];
        bx += tb->Sz[i][k] * s->com_x[k]; by += tb->Sz[i][k] * s->com_y[k];
      }
      svx[i] = ax; svy[i] = ay; szx[i] = bx; szy[i] = by;
    }
    for (int i = 0; i < N; i++) {
      double t1x = 0.0, t1y = 0.0, t2x = 0.0, t2y = 0.0;
      for (int k = 0; k < N; k++) {               /* UT(i,k) = Uv[k][i] */
        t1x += tb->Uv[k][i] * svx[k]; t1y += tb->Uv[k][i] * svy[k];
        t2x += tb->Uv[k][i] * refx[k]; t2y += tb->Uv[k][i] * refy[k];
      }
      d[i] += t1x * m->alpha; d[N + i] += t1y * m->alpha;
      d[i] += t2x * (-m->alpha); d[N + i] += t2y * (-m->alpha);
    }
    /* Hessian: -g*UzT*V, -g*VT*Uz, g*VT*V */
    for (int i = 0; i < N; i++)
      for (int j = 0; j < ns; j++) {
        double p = 0.0, pt = 0.0;
        for (int k = 0; k < N; k++) {
          double v = (stepidx[k] == j + 1) ? 1.0 : 0.0;
          p += tb->Uz[k][i] * v;                  /* (UzT V)(i,j) */
          pt += v * tb->Uz[k][i];                 /* (VT Uz)(j,i) */
        }
        p *= -m->gamma; pt *= -m->gamma;
        Cq(i, 2 * N + j) += p; Cq(N + i, 2 * N + ns + j) += p;
        Cq(2 * N + j, i) += pt; Cq(2 * N + ns + j, N + i) += pt;
      }
    for (int i = 0; i < ns; i++)
      for (int j = 0; j < ns; j++) {
        double p = 0.0;
        for (int k = 0; k < N; k++) {
          double vi = (stepidx[k] == i + 1) ? 1.0 : 0.0, vj = (stepidx[k] == j + 1) ? 1.0 : 0.0;
          p += vi * vj;
        }
        p *= m->gamma;
        Cq(2 * N + i, 2 * N + j) += p; Cq(2 * N + ns + i, 2 * N + ns + j) += p;
      }
    for (int j = 0; j < ns; j++) {
      double px = 0.0, py = 0.0, qx = 0.0, qy = 0.0;
      for (int k = 0; k < N; k++) {
        double v = (stepidx[k] == j + 1) ? 1.0 : 0.0;
        px += v * szx[k]; py += v * szy[k];
        qx += v * VcX[k]; qy += v * VcY[k];
      }
      d[2 * N + j] += px * (-m->gamma); d[2 * N + ns + j] += py * (-m->gamma);
      d[2 * N + j] += qx * m->gamma; d[2 * N + ns + j] += qy * m->gamma;
    }

    /* --- build_constraints :554-584 ------------------------------------------- */
    /* CoP rows, build_inequalities_cop :284-314 + build_constraints_cop :393-448 */
    hull_t H;
    hull_set_vertices(m, &H, &sup[0], 0);
    for (int i = 0; i < N; i++) {
      const sup_t *S = &sup[i + 1];
      if (S->state_changed) hull_set_vertices(m, &H, S, 0);
      hull_linear_system(&H, S->foot);
      for (int e = 0; e < 4; e++) {
        const int r = 1 + 4 * i + e;              /* row 0 of DU/DS is the dummy */
        const double a = H.A[e], bb = H.B[e];
        for (int c = 0; c < N; c++) {
          double px = 0.0 + a * tb->Uz[i][c], py = 0.0 + bb * tb->Uz[i][c];
          Aq(r, c) += px * -1.0; Aq(r, N + c) += py * -1.0;
        }
        for (int j = 0; j < ns; j++) {
          double v = (stepidx[i] == j + 1) ? 1.0 : 0.0;
          double px = 0.0 + a * v, py = 0.0 + bb * v;
          Aq(r, 2 * N + j) += px * 1.0; Aq(r, 2 * N + ns + j) += py * 1.0;
        }
        b[r] += H.D[e];
        b[r] += (0.0 + a * szx[i]) * -1.0;
        b[r] += (0.0 + bb * szy[i]) * -1.0;
        b[r] += (0.0 + a * VcX[i]) * 1.0;
        b[r] += (0.0 + bb * VcY[i]) * 1.0;
      }
    }
    /* foot rows, build_inequalities_feet :317-354 + build_constraints_feet :451-474 */
    for (int i = 0; i < N; i++) {
      const sup_t *S = &sup[i + 1];
      if (S->state_changed && S->step_number > 0 && S->phase != WG_DS) {
        hull_set_vertices(m, &H, &sup[i], 1);
        hull_linear_system(&H, S->foot);
        const int k = S->step_number - 1;
        for (int e = 0; e < 5; e++) {
          const int r = 1 + 4 * N + 5 * k + e;
          const double a = H.A[e], bb = H.B[e];
          for (int j = 0; j < ns; j++) {
            double px = 0.0 + a * V_f[k][j], py = 0.0 + bb * V_f[k][j];
            Aq(r, 2 * N + j) += px * -1.0; Aq(r, 2 * N + ns + j) += py * -1.0;
          }
          b[r] += H.D[e];
          b[r] += (0.0 + a * Vc_fX[k]) * 1.0;
          b[r] += (0.0 + bb * Vc_fY[k]) * 1.0;
        }
      }
    }
  }

  /* --- QPProblem::solve, qp-problem.cpp:245-294 ------------------------------- */
  int ifail = 0, nact = 0, nit = 0, hlen = 0;
  int *iact = (int *)calloc((size_t)n + 1, sizeof(int));
  if (g_ref_ql) {
    /* the REFERENCE's compiled ql0001_ (oracle/_ref/libqld_ref.so) does the solve, called exactly like
     * QPProblem::solve does (qp-problem.cpp:256-279): iwar[0] = 1, lwar = 3 nmax^2/2 + 10 nmax + 2 mmax + 20000 */
    int m_ = mq, me_ = 0, mmax_ = mmax, n_ = n, nmax_ = n, mnn_ = mq + 2 * n, iout_ = 0, iprint_ = 1, liwar_ = n;
    int lwar_ = 3 * n * n / 2 + 10 * n + 2 * mmax + 20000;
    double eps_ = 1e-8;
    double *war = (double *)calloc((size_t)lwar_, sizeof(double));
    iact[0] = 1;
    struct timespec t0_, t1_;
    if (g_solve_timer) clock_gettime(CLOCK_MONOTONIC, &t0_);
    g_ref_ql(&m_, &me_, &mmax_, &n_, &nmax_, &mnn_, C, d, A, b, xl, xu, x, u, &iout_, &ifail, &iprint_, war, &lwar_, iact,
             &liwar_, &eps_);
    if (g_solve_timer) {
      clock_gettime(CLOCK_MONOTONIC, &t1_);
      g_solve_seconds += (double)(t1_.tv_sec - t0_.tv_sec) + 1e-9 * (double)(t1_.tv_nsec - t0_.tv_nsec);
      g_solve_count++;
    }
    free(war);
    for (int i = 0; i < mq + 2 * n; i++) nact += (u[i] != 0.0);
  } else
  wgo_ql_solve(mq, 0, mmax, n, n, C, d, A, b, xl, xu, 1e-8, x, u, &ifail, iact, &nact, &nit,
               dump ? dump->hist : NULL, dump ? WGO_HIST_CAP : 0, &hlen);
  if (dump) {
    dump->n = n; dump->m = mq; dump->mmax = mmax; dump->ifail = ifail; dump->nact = nact; dump->n_iter = nit;
    dump->hist_len = hlen;
    if ((size_t)n * n <= WGO_DUMP_C && (size_t)mmax * n <= WGO_DUMP_A) {
      memcpy(dump->C, C, sizeof(double) * (size_t)n * n);
      memcpy(dump->A, A, sizeof(double) * (size_t)mmax * n);
      memcpy(dump->d, d, sizeof(double) * (size_t)n);
      memcpy(dump->b, b, sizeof(double) * (size_t)mmax);
      memcpy(dump->x, x, sizeof(double) * (size_t)n);
      for (int i = 0; i < n && i < 128; i++) dump->iact[i] = iact[i];
    }
  }

  /* --- :405-428 CoM interpolation + state step ------------------------------- */
  double jx, jy;
  if (sup[0].nb_steps_left == 0 && !(m->flags & WG_FLAG_NO_STOP_CENTERING)) {
    jx = (s->lf[0].x + s->rf[0].x) / 2 - s->front_com_x[0];
    jy = (s->lf[0].y + s->rf[0].y) / 2 - s->front_com_y[0];
    if (fabs(jx) < 1e-3 && fabs(jy) < 1e-3) s->running = 0;
    const double tf = 0.75;
    jx = 6 / (tf * tf * tf) * (jx - tf * s->front_com_x[1] - (tf * tf / 2) * s->front_com_x[2]);
    jy = 6 / (tf * tf * tf) * (jy - tf * s->front_com_y[1] - (tf * tf / 2) * s->front_com_y[2]);
  } else {
    s->running = 1;
    jx = x[0]; jy = x[N];
  }
  {
    /* LinearizedInvertedPendulum2D::Interpolation :157-227 */
    const double c02 = -s->com_z / 9.81;
    double fcx[3] = {0, 0, 0}, fcy[3] = {0, 0, 0};
    for (int lk = 0; lk < K; lk++) {
      double t = (lk + 1) * m->Tctrl;
      double cx0 = s->com_x[0] + t * s->com_x[1] + 0.5 * t * t * s->com_x[2] + t * t * t * jx / 6.0;
      double cx1 = s->com_x[1] + t * s->com_x[2] + 0.5 * t * t * jx;
      double cx2 = s->com_x[2] + t * jx;
      double cy0 = s->com_y[0] + t * s->com_y[1] + 0.5 * t * t * s->com_y[2] + t * t * t * jy / 6.0;
      double cy1 = s->com_y[1] + t * s->com_y[2] + 0.5 * t * t * jy;
      double cy2 = s->com_y[2] + t * jy;
      double zx = 1.0 * cx0 + 0.0 * cx1 + c02 * cx2;
      double zy = 1.0 * cy0 + 0.0 * cy1 + c02 * cy2;
      if (out) {
        out->com_x[lk][0] = cx0; out->com_x[lk][1] = cx1; out->com_x[lk][2] = cx2;
        out->com_y[lk][0] = cy0; out->com_y[lk][1] = cy1; out->com_y[lk][2] = cy2;
        out->com_yaw[lk][0] = 0.0; out->com_yaw[lk][1] = 0.0;
        out->zmp_x[lk] = zx; out->zmp_y[lk] = zy;
      }
      if (lk == 11) { fcx[0] = cx0; fcx[1] = cx1; fcx[2] = cx2; fcy[0] = cy0; fcy[1] = cy1; fcy[2] = cy2; }
    }
    for (int k = 0; k < 3; k++) { s->front_com_x[k] = fcx[k]; s->front_com_y[k] = fcy[k]; }
    /* OneIteration :230-264:  x <- A x + B u  (ublas prod, then vector add) */
    const double A01 = T, A02 = T * T / 2.0, A12 = T;
    const double B0 = T * T * T / 6.0, B1 = T * T / 2.0, B2 = T;
    double nx[3], ny[3];
    nx[0] = 0.0 + 1.0 * s->com_x[0] + A01 * s->com_x[1] + A02 * s->com_x[2];
    nx[1] = 0.0 + 0.0 * s->com_x[0] + 1.0 * s->com_x[1] + A12 * s->com_x[2];
    nx[2] = 0.0 + 0.0 * s->com_x[0] + 0.0 * s->com_x[1] + 1.0 * s->com_x[2];
    ny[0] = 0.0 + 1.0 * s->com_y[0] + A01 * s->com_y[1] + A02 * s->com_y[2];
    ny[1] = 0.0 + 0.0 * s->com_y[0] + 1.0 * s->com_y[1] + A12 * s->com_y[2];
    ny[2] = 0.0 + 0.0 * s->com_y[0] + 0.0 * s->com_y[1] + 1.0 * s->com_y[2];
    s->com_x[0] = nx[0] + jx * B0; s->com_x[1] = nx[1] + jx * B1; s->com_x[2] = nx[2] + jx * B2;
    s->com_y[0] = ny[0] + jy * B0; s->com_y[1] = ny[1] + jy * B1; s->com_y[2] = ny[2] + jy * B2;
  }
  if (out) {
    out->jerk_x = jx; out->jerk_y = jy; out->ifail = ifail; out->n_iter = nit; out->nact = nact;
    out->n = n; out->m = mq; out->nb_prw_steps = ns;
  }

  /* --- interpolate_trunk_orientation ------------------------------------------ */
  op_interpolate_trunk(m, s, time, &sup[0], out);

  /* --- interpolate_feet_positions, OnLineFootTrajectoryGeneration.cpp:235-346 -- */
  {
    const sup_t *cs = &sup[0];
    double FPx = 0.0, FPy = 0.0;
    if (cs->phase != WG_DS) {                      /* interpret_solution :202-232 */
      double sign = (cs->foot == WG_LEFT) ? 1.0 : -1.0;
      if (cs->nb_steps_left > 0 && ns > 0) { FPx = x[2 * N]; FPy = x[2 * N + ns]; }
      else {
        FPx = cs->x + sign * WSIN(cs->yaw) * m->feet_distance;
        FPy = cs->y - sign * WCOS(cs->yaw) * m->feet_distance;
      }
    }
    const double dt = m->Tctrl;
    const double local_t = time - (cs->time_limit - (m->t_double + m->t_single));
    wg_foot_sample_t L[WG_SAMPLES_PER_TICK + 1], R[WG_SAMPLES_PER_TICK + 1];   /* [0] = old back */
    L[0] = s->lf[2]; R[0] = s->rf[2];
    if (cs->phase == WG_SS && time + 3.0 / 2.0 * T < cs->time_limit) {
      const double unlocked = m->t_single * 0.9;
      const double end_lift = (m->t_single - unlocked) * 0.5;
      double swing_passed = 0.0;
      if (local_t > end_lift) swing_passed = local_t - end_lift;
      wg_foot_sample_t *SW = (cs->foot == WG_LEFT) ? R : L;     /* swing */
      wg_foot_sample_t *ST = (cs->foot == WG_LEFT) ? L : R;     /* stance */
      const wg_foot_sample_t *st_prev = (cs->foot == WG_LEFT) ? &s->lf[1] : &s->rf[1];
      const wg_foot_sample_t *last = &SW[0];
      const double ti = unlocked - swing_passed;
      double px5[6], py5[6], pth[4], pom[4], pom2[4];
      poly5_set(px5, ti, FPx, last->x, last->dx, last->ddx);
      poly5_set(py5, ti, FPy, last->y, last->dy, last->ddy);
      if (cs->state_changed) poly4_set(s->poly_z, m->t_single, m->step_height);
      poly3_set(pth, ti, sup_angles[0] * 180.0 / M_PI, last->theta, last->dtheta);
      poly3_set(pom, ti, 0.0 * 180.0 / M_PI, last->omega, last->domega);
      poly3_set(pom2, ti, 2 * 0.0 * 180.0 / M_PI, last->omega2, last->domega2);
      const double start_landing = end_lift + unlocked;
      const double omega_cmd = 0.0;                /* m_Omega (":omega 0.0") */
      const wg_foot_sample_t *sw_prev = (cs->foot == WG_LEFT) ? &s->rf[1] : &s->lf[1];  /* [StartIndex-1] */
      for (int k = 1; k <= K; k++) {               /* UpdateFootPosition :50-199 */
        const double it = (double)k * dt;
        wg_foot_sample_t *c = &SW[k];
        const wg_foot_sample_t *p = &SW[k - 1];
        *c = (wg_foot_sample_t){0};
        ST[k] = *st_prev;
        if (local_t + it <= end_lift || local_t + it >= start_landing) {
          c->x = p->x; c->y = p->y; c->theta = p->theta;
        } else if (local_t < end_lift && local_t + it > end_lift) {
          double rt = local_t + it - end_lift;
          c->x = poly_eval(px5, 5, rt); c->dx = poly_d1(px5, 5, rt); c->ddx = poly_d2(px5, 5, rt);
          c->y = poly_eval(py5, 5, rt); c->dy = poly_d1(py5, 5, rt); c->ddy = poly_d2(py5, 5, rt);
          c->theta = poly_eval(pth, 3, rt); c->dtheta = poly_d1(pth, 3, rt);
        } else {
          c->x = poly_eval(px5, 5, it); c->dx = poly_d1(px5, 5, it); c->ddx = poly_d2(px5, 5, it);
          c->y = poly_eval(py5, 5, it); c->dy = poly_d1(py5, 5, it); c->ddy = poly_d2(py5, 5, it);
          c->theta = poly_eval(pth, 3, it); c->dtheta = poly_d1(pth, 3, it);
        }
        c->z = poly_eval(s->poly_z, 4, local_t + it);
        c->dz = poly_d1(s->poly_z, 4, local_t + it);
        if (local_t + it < end_lift) {
          c->omega = poly_eval(pom, 3, it); c->domega = poly_d1(pom, 3, it);
        } else if (local_t + it < start_landing) {
          c->omega = omega_cmd - poly_eval(pom2, 3, local_t + it - end_lift) - sw_prev->omega2;
        } else {
          c->omega = poly_eval(pom, 3, local_t + it - start_landing) + sw_prev->omega - omega_cmd;
        }
        /* :150-198 floor-penetration shift: with omega == 0 every term is exactly 0,
         * but keep the arithmetic (dX = F - F*WCOS(0) + H*WSIN(0)) shape-free: */
        {
          double lOmega = c->omega * M_PI / 180.0, lTheta = c->theta * M_PI / 180.0;
          double cth = WCOS(lTheta), sth = WSIN(lTheta);
          /* B, H, F are ankle-geometry constants; they only multiply (1-cos) and sin of
           * lOmega, which is 0 on this path (":omega 0.0"), so any finite value gives 0 */
          double Bf = 0.0, Hf = 0.105, Ff = 0.105, dX, dFZ;
          if (lOmega < 0) { dX = -(Bf - Bf * WCOS(-lOmega) + Hf * WSIN(-lOmega)); dFZ = Hf * WCOS(-lOmega) + Bf * WSIN(-lOmega) - Hf; }
          else { dX = (Ff - Ff * WCOS(lOmega) + Hf * WSIN(lOmega)); dFZ = Hf * WCOS(lOmega) + Ff * WSIN(lOmega) - Hf; }
          c->x += cth * dX; c->y += sth * dX; c->z += dFZ;
        }
      }
    } else if (cs->phase == WG_DS || time + 3.0 / 2.0 * T > cs->time_limit) {
      L[0] = s->lf[1]; R[0] = s->rf[1];           /* k = 0: back <- back-1 (:333-336) */
      for (int k = 1; k <= K; k++) { R[k] = R[k - 1]; L[k] = L[k - 1]; }
    } else {
      for (int k = 1; k <= K; k++) { memset(&L[k], 0, sizeof L[k]); memset(&R[k], 0, sizeof R[k]); }
    }
    if (out) { for (int k = 0; k < K; k++) { out->lf[k] = L[k + 1]; out->rf[k] = R[k + 1]; } out->lf_back = L[0]; out->rf_back = R[0];
               for (size_t k = 0; k < sizeof(out->pad_) / sizeof(double); k++) out->pad_[k] = 0.0; }
    /* the back sample itself may have been rewritten (DS branch): it is still in
     * the queue and will be consumed later, so callers replaying the queue need it */
    if (dump) { dump->lf_back_rewritten = L[0]; dump->rf_back_rewritten = R[0]; }
    s->lf[0] = L[12]; s->lf[1] = L[19]; s->lf[2] = L[20];
    s->rf[0] = R[12]; s->rf[1] = R[19]; s->rf[2] = R[20];
  }

  /* --- :446-450 ------------------------------------------------------------------ */
  if (!s->ending_phase) s->time_to_stop = s->upper_time_limit + T * N;
  s->upper_time_limit = s->upper_time_limit + T;
  s->tick_count++;

  free(iact); free(C); free(d); free(A); free(b); free(xl); free(xu); free(x); free(u); free(tb);
  return rc;
}

/* bench.py's cpu_baseline leg: the benchmark workload (velocity table vel[seg][gait][3], redrawn every `redraw`
 * ticks; the 5 ms clock advanced like RunOneStepOfTheControlLoop does) for gaits [0, n_gaits), entirely in C. */
int wgo_mpc_run(const wg_model_t *model, wg_gait_state_t *states, int n_gaits, int n_ticks, const double *vel, int redraw) {
  for (int tick = 0; tick < n_ticks; tick++) {
    const int adv = tick == 0 ? 1 : (tick == 1 ? 19 : 20);
    const int seg = tick / redraw;
    for (int g = 0; g < n_gaits; g++) {
      wg_gait_state_t *st = &states[g];
      if (tick % redraw == 0) {
        const double *v = vel + ((size_t)seg * n_gaits + g) * 3;
        st->vref[0] = v[0]; st->vref[1] = v[1]; st->vref[2] = v[2];
      }
      double c = st->clock;
      for (int k = 0; k < adv; k++) c += model->Tctrl;
      st->clock = c;
      int rc = wgo_mpc_tick(model, st, 0, 0);
      if (rc) return rc;
    }
  }
  return 0;
}
