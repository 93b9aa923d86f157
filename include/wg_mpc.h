/*
 * wg_mpc.h -- C ABI of the MI355X-native ZMP-MPC hot path (libwg_mpc.so).
 *
 * Drop-in boundary for the per-tick inner loop of jrl-walkgen's Herdt-2010
 * pattern generator.  Plain pointers and sizes only; no C++/torch types.
 * Every entry point names the reference interface it replaces.
 *
 * Conventions
 *   - return value: 0 on success, negative on error (wg_last_error() has text);
 *   - per-QP `ifail` keeps the QL codes of the reference
 *     (src/privatepgtypes.hh:349-356): 0 ok, 1 too many iterations,
 *     2 accuracy insufficient, 5 workspace too short, >10 inconsistent
 *     constraints (10 + constraint code);
 *   - all matrices column-major (Fortran), one QP after the other, strides
 *     given by (nmax, mmax) exactly as ql0001_ takes them;
 *   - `_dev` variants take DEVICE pointers and a hipStream_t (passed as void*)
 *     and are asynchronous; the others take HOST pointers and are synchronous;
 *   - the library never falls back to a CPU path: without a usable HIP device
 *     every compute entry point fails with WG_ERR_NO_DEVICE.
 */
#ifndef WG_MPC_H
#define WG_MPC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WG_OK 0
#define WG_ERR_NO_DEVICE (-1)
#define WG_ERR_BAD_ARG (-2)
#define WG_ERR_HIP (-3)
#define WG_ERR_TOO_LARGE (-4)

/* Library / device management ------------------------------------------- */

/* Select the HIP device this process drives (one process per GPU). */
int wg_init(int device_ordinal);
void wg_shutdown(void);
const char *wg_last_error(void);
/* ABI version, bumped on any signature change. */
int wg_abi_version(void);
/* LDS bytes one QP of size (n, m) occupies in the solver kernel; lets callers
 * reason about occupancy (160 KiB per CU on gfx950). */
size_t wg_qp_lds_bytes(int n, int m);

/* Batched dense QP solve -------------------------------------------------
 *
 * Replaces, batched over B independent problems,
 *     ql0001_(m, me, mmax, n, nmax, mnn, c, d, a, b, xl, xu, x, u, iout,
 *             ifail, iprint, war, lwar, iwar, liwar, eps1)
 *     src/Mathematics/qld.hh:27-31, called from QPProblem::solve
 *     src/ZMPRefTrajectoryGeneration/qp-problem.cpp:275-279
 * with iwar[0] = 1 (solver factorises the Hessian itself):
 *
 *     minimise  1/2 x'Cx + d'x   s.t.  A_j x + b_j  = 0  (j <  me)
 *                                      A_j x + b_j >= 0  (me <= j < m)
 *                                      xl <= x <= xu
 *
 *   B        number of QPs
 *   nmax     leading dimension of every C, stride of d/xl/xu/x/iact (>= max n)
 *   mmax     leading dimension of every A, stride of b             (>= max m)
 *   n,m,me   per-QP sizes, B ints each (n may be NULL => all nmax;
 *            m NULL => all mmax-1, the reference's convention m = mmax-1;
 *            me NULL => all 0)
 *   C        B * nmax*nmax     d, xl, xu   B * nmax
 *   A        B * mmax*nmax     b           B * mmax
 *   eps      the reference passes 1e-8 (qp-problem.cpp:260)
 * outputs
 *   x        B * nmax
 *   u        B * (mmax + 2*nmax): multipliers [general m | lower n | upper n]
 *            packed with the QP's own m and n like ql0001_'s u (may be NULL)
 *   ifail    B
 *   n_iter   B   ql0002's iteration counter                 (may be NULL)
 *   iact     B * nmax final active set in activation order, QL codes
 *            (1..m row, m+1..m+n lower bound, m+n+1.. upper), 0 padded
 *            -- what ql0001_ leaves in iwar[0..nact)          (may be NULL)
 *   nact     B                                               (may be NULL)
 *   hist     B * hist_cap add(+code)/drop(-code) log          (may be NULL)
 *   hist_len B   number of events (can exceed hist_cap)       (NULL iff hist NULL)
 */
int wg_qp_solve_batch(int B, int nmax, int mmax, const int *n, const int *m,
                      const int *me, const double *C, const double *d,
                      const double *A, const double *b, const double *xl,
                      const double *xu, double eps, double *x, double *u,
                      int *ifail, int *n_iter, int *iact, int *nact, int *hist,
                      int hist_cap, int *hist_len);

int wg_qp_solve_batch_dev(int B, int nmax, int mmax, const int *n, const int *m,
                          const int *me, const double *C, const double *d,
                          const double *A, const double *b, const double *xl,
                          const double *xu, double eps, double *x, double *u,
                          int *ifail, int *n_iter, int *iact, int *nact,
                          int *hist, int hist_cap, int *hist_len,
                          void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* WG_MPC_H */
