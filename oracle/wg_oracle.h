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

#ifdef __cplusplus
}
#endif
#endif
