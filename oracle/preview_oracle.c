/*
 * preview_oracle.c -- TEST INFRASTRUCTURE ONLY (see wg_oracle.h).
 *
 * CPU restatement of the Kajita stage-1 preview-control iteration:
 *   PreviewControl::OneIterationOfPreview     /root/reference/src/PreviewControl/PreviewControl.cpp:324-374
 *   (OneIterationOfPreview1D :376-420 is the same arithmetic on one axis)
 * with m_A, m_B, m_C as ComputeOptimalWeights / ReadPrecomputedFile fill them (:169-181, :203-214).
 * jrl-mal's MAL_RET_A_by_B is boost::ublas prod: each element is a sum over k ascending that starts from 0.
 *
 * Pin: PARITY UNPINNED for the iteration itself -- the reference's tests hold no golden for stage 1 alone
 * (TestKajita2003*.datref contain the CoM after the multi-body second stage, which needs the HRP-2 model).  The gains
 * it is driven with ARE pinned (src/data/PreviewControlParameters.ini, tests/test_riccati.py), and the closed-loop
 * property the reference relies on (output ZMP tracks the reference) is asserted in tests/test_preview_oracle.py.
 */
#include "wg_oracle.h"

int wgo_preview_run(const wg_preview_gains_t *g, const double *F, int B, int L, const double *zmp_x, const double *zmp_y,
                    double *state, double *com, double *zmp2, int simulation) {
  if (!g || !F || B < 0 || L < 0 || g->nl < 1) return -1;
  const double T = g->T;
  const double A[3][3] = {{1.0, T, T * T / 2.0}, {0.0, 1.0, T}, {0.0, 0.0, 1.0}};
  const double Bm[3] = {T * T * T / 6.0, T * T / 2.0, T};
  const double C[3] = {1.0, 0.0, -g->zc / 9.81};
  const int nl = g->nl;
  const long Lz = (long)L + nl - 1;
  for (int b = 0; b < B; b++)
    for (int axis = 0; axis < 2; axis++) {
      const double *z = (axis ? zmp_y : zmp_x) + (long)b * Lz;
      double *st = state + (long)b * 8;
      double x[3] = {st[3 * axis], st[3 * axis + 1], st[3 * axis + 2]};
      double s = st[6 + axis];
      for (int l = 0; l < L; l++) {
        double r = 0.0;
        for (int k = 0; k < 3; k++) r += g->Kx[k] * x[k];
        double u = -r + g->Ks * s;
        for (int i = 0; i < nl; i++) u += F[i] * z[l + i];
        double ax[3];
        for (int i = 0; i < 3; i++) {
          double t = 0.0;
          for (int k = 0; k < 3; k++) t += A[i][k] * x[k];
          ax[i] = t;
        }
        for (int i = 0; i < 3; i++) x[i] = ax[i] + u * Bm[i];
        double p = 0.0;
        for (int i = 0; i < 3; i++) p += C[i] * x[i];
        if (simulation) s += (z[l] - p);
        if (com) for (int i = 0; i < 3; i++) com[((long)b * L + l) * 6 + 3 * axis + i] = x[i];
        if (zmp2) zmp2[((long)b * L + l) * 2 + axis] = p;
      }
      st[3 * axis] = x[0]; st[3 * axis + 1] = x[1]; st[3 * axis + 2] = x[2]; st[6 + axis] = s;
    }
  return 0;
}
