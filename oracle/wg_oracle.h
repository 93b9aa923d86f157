/*
 * wg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference's ZMP-MPC hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this directory; the product (jrl-walkgen_amd/) never
 * does.  Each function cites the reference file:line it follows in its .c.
 */
#ifndef WG_ORACLE_H
#define WG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ql0001_/ql0002_ restatement (ql_oracle.c).  Same argument meaning as
 * qld.hh:27-31 with lql = TRUE; c is not modified.  Extra outputs:
 *   iact_out[n]  final active set in activation order (QL codes: 1..m general
 *                row, m+1..m+n lower bound, m+n+1..m+2n upper bound), 0-padded
 *   nact_out     its length
 *   n_iter       ql0002's iteration counter (iterc)
 *   hist         add (+code) / drop (-code) log, up to hist_cap entries;
 *                hist_len receives the number of events (may exceed cap). */
int wgo_ql_solve(int m, int me, int mmax, int n, int nmax,
                 const double *c, const double *d, const double *a,
                 const double *b, const double *xl, const double *xu,
                 double eps, double *x, double *u, int *ifail, int *iact_out,
                 int *nact_out, int *n_iter, int *hist, int hist_cap,
                 int *hist_len);

/* ---- Herdt-2010 tick restatement (herdt_oracle.c) ------------------------- */
#include "../include/wg_mpc.h"   /* POD layouts only (wg_model_t, wg_gait_state_t, wg_tick_out_t) */

#define WGO_HIST_CAP 512
#define WGO_DUMP_C (80 * 80)
#define WGO_DUMP_A (200 * 80)
/* optional per-tick introspection: the assembled QP exactly as it is handed to QL */
typedef struct wgo_qp_dump {
  int n, m, mmax, ifail, nact, n_iter, hist_len;
  int iact[128];
  int hist[WGO_HIST_CAP];
  double C[WGO_DUMP_C], A[WGO_DUMP_A], d[80], b[200], x[80];
  wg_foot_sample_t lf_back_rewritten, rf_back_rewritten;
} wgo_qp_dump_t;

void wgo_model_defaults(wg_model_t *model);
void wgo_gait_init(const wg_model_t *model, wg_gait_state_t *state, const double com0[3],
                   const double left_xyt[3], const double right_xyt[3]);
/* one tick at time state->clock (the caller advances the clock, see tests/herdt_replay.py);
 * out and dump may be NULL.  Returns 0, or <0 if the sizes are unsupported. */
int wgo_mpc_tick(const wg_model_t *model, wg_gait_state_t *state, wg_tick_out_t *out, wgo_qp_dump_t *dump);
/* Q_b = beta I + alpha Uv'Uv + gamma Uz'Uz of the model (N x N, row-major), summed in the reference's order */
int wgo_invariant_hessian(const wg_model_t *model, double *Qb);
/* NULL (default): the tick solves with wgo_ql_solve; otherwise with the given ql0001_ entry point (the reference's own
 * compiled qld.cpp from oracle/_ref/libqld_ref.so) */
void wgo_set_reference_ql(void *ql0001_entry);
/* one ql0001_ call as the reference's drivers make it (iwar[0] = lql): by the compiled reference when one was set, else by the
 * restatement */
int wgo_ql_call(int lql, int m, int me, int mmax, int n, int nmax, double *C, double *d, double *A, double *b, double *xl, double *xu,
                double *x, double *u, int *ifail, int *n_iter, int *nact);
/* enable / disable and read-and-reset the wall time spent inside the reference's ql0001_ (solve only) */
void wgo_solve_timer(int enable, double *seconds, long *count);
/* the benchmark workload on the CPU, loop in C (see herdt_oracle.c) */
int wgo_mpc_run(const wg_model_t *model, wg_gait_state_t *states, int n_gaits, int n_ticks, const double *vel, int redraw);

/* ---- Dimitrov back-end: OptCholesky + PLDPSolver restatement (pldp_oracle.c) -------------------------------------- */
typedef struct wgo_pldp_model {
  int N, pad_;
  double iPu[WG_PLDP_N * WG_PLDP_N], Px[WG_PLDP_N * 3], Pu[WG_PLDP_N * WG_PLDP_N], iPuPx[2 * WG_PLDP_N * 6];
} wgo_pldp_model_t;

void wgo_optchol_update_normal(const double *A, int card_u, const int *set, int nset, double *L, int ldl);
void wgo_optchol_update_fortran(const double *A, int m, int card_u, const int *set, int nset, double *L, int ldl);
void wgo_chol_normal(const double *A, int n, double *L);
void wgo_chol_inverse(const double *L, int n, int size, double *iL);
int wgo_pldp_setup(wgo_pldp_model_t *M, int N, const double *iPu, const double *Px, const double *Pu);
/* arguments as wg_pldp_solve_batch (include/wg_mpc.h) for one problem; returns its `ret` (or -100 on bad input) */
int wgo_pldp_solve(const wgo_pldp_model_t *M, wg_pldp_state_t *st, const double *D, int m, const double *A,
                   const double *b, const double *zmpref, const double *xkyk, const int *similar, int n_removed,
                   int starting, int max_iter, double *X, int *n_iter, int *active, int *n_active);
/* one Dimitrov-2008 tick (see pldp_oracle.c); constants row-major */
int wgo_dimitrov_tick(const wgo_pldp_model_t *M, const double *OptB, const double *OptC, const double *iLQ, double T,
                      double Tctrl, double com_height, const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *st,
                      wg_dimitrov_out_t *out, int max_iter);
/* the same tick in the reference's modes QLD (mode = 1) and QLDANDLQ (mode = 2) (ZMPConstrainedQPFastFormulation.cpp:1297-1320):
 *   QLD       ql0001_ on (Q = OptA, D, DPu, DPx) without the LQ preconditioning, iwar[0] = 1: Q = ql0001_'s column-major 2N x 2N,
 *             OptB (2N x 6) / OptC (2N x 2N) row-major AS BUILT, PuT = Pu' (N x N, :629-637); iLQ unused
 *   QLDANDLQ  ql0001_ on the preconditioned problem, Q = identity handed over as its own factor (iwar[0] = 0): OptB / OptC
 *             premultiplied by iLQ, PuT = iLQ Pu' (full), X <- iLQ' X; Q unused
 * Px N x 3.  The solve goes through wgo_ql_call: PINNED to the compiled reference when wgo_set_reference_ql was given oracle/_ref's
 * entry point. */
int wgo_dimitrov_qld_tick(int mode, int N, const double *Q, const double *OptB, const double *OptC, const double *PuT,
                          const double *Px, const double *iLQ, double T, double Tctrl, double com_height,
                          const wg_zmp_polytope_t *polys, wg_dimitrov_state_t *st, wg_dimitrov_out_t *out);

/* ---- Kajita stage-1 preview-control iteration (preview_oracle.c); arguments as wg_preview_run_batch ---------------- */
int wgo_preview_run(const wg_preview_gains_t *g, const double *F, int B, int L, const double *zmp_x, const double *zmp_y,
                    double *state, double *com, double *zmp2, int simulation);

/* ---- ZMPDiscretization / FootConstraintsAsLinearSystem restatement (zmpdisc_oracle.c) ------------------------------ */
int wgo_zmpdisc_length(const wg_zmpdisc_model_t *model, const wg_rel_step_t *steps, int n_steps);
/* arguments as wg_zmpdisc_batch (include/wg_mpc.h) for one gait, plus the sample times (may be NULL); returns L */
int wgo_zmpdisc(const wg_zmpdisc_model_t *model, const wg_rel_step_t *steps, int n_steps, const double *init_feet,
                int lcap, double *zmp, double *zmp_theta, int *zmp_type, double *left, int *left_type, double *right,
                int *right_type, double *time);
/* arguments as wg_foot_constraints */
int wgo_foot_constraints(int n, const double *time, const double *left, const int *left_type, const double *right,
                         double sole_w, double sole_h, double constraint_x, double constraint_y, int cap,
                         wg_zmp_polytope_t *polys, double *t_start, double *t_end);

/* StepStackHandler's generators behind ":supportfoot", ":arc", ":lastsupport" (see zmpdisc_oracle.c); append to out[*n] */
int wgo_steps_support_foot(int support_foot, double ss, double ds, wg_rel_step_t *out, int *n, int cap);
int wgo_steps_last_support(int keep, double ss, double ds, wg_rel_step_t *out, int *n, int cap);
int wgo_steps_arc(double x, double y, double arc_deg, int support_foot, double ss, double ds, wg_rel_step_t *out, int *n,
                  int cap, int *keep);

#ifdef __cplusplus
}
#endif
#endif
