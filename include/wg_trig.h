/* wg_trig.h -- sin/cos as a fixed sequence of IEEE-754 double +,-,* (no FMA, no
 * table, no libm), so that the HIP kernels and a CPU build produce the SAME bits.
 *
 * Why: the reference calls libm's sin/cos when it rotates the ZMP / foot-placement
 * polygons and the velocity reference (privatepgtypes.cpp:157-185,
 * generator-vel-ref.cpp:211-229).  glibc's and the GPU's math libraries are both
 * < 1 ulp but not bit-identical, and the parity bar for this path is "same
 * active-set sequence".  A self-contained kernel removes the library from the
 * comparison: tests build the CPU oracle once against libm (faithful to the
 * reference, pinned to its golden file) and once against this header (bit-exact
 * partner of the GPU), and check that the two oracles agree to ~1e-16.
 *
 * Algorithm: Cody-Waite reduction by pi/2 with a two-word tail, then the classic
 * odd/even minimax polynomials on [-pi/4, pi/4] (the coefficient values are the
 * well-known Sun/fdlibm ones).  Error < 1 ulp for |x| < 1e5, far beyond any yaw
 * angle this path sees.  Include from C or HIP; define WG_TRIG_FN for qualifiers.
 */
#ifndef WG_TRIG_H
#define WG_TRIG_H

#ifndef WG_TRIG_FN
#define WG_TRIG_FN static inline
#endif

/* r + rt ~= x - n*pi/2, returns n mod 4 */
WG_TRIG_FN int wg_trig_reduce(double x, double *r, double *rt) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
  const double pio2_1t = 6.07710050650619224932e-11;  /* pi/2 - pio2_1 */
  const double pio2_2 = 6.07710050630396597660e-11;   /* second 33 bits */
  const double pio2_2t = 2.02226624879595063154e-21;  /* pi/2 - (pio2_1 + pio2_2) */
  double t = x * invpio2;
  int n = (int)(t + (t >= 0.0 ? 0.5 : -0.5));
  double fn = (double)n;
  /* two-stage subtraction keeps ~100 bits of pi/2 in play */
  double a = x - fn * pio2_1;         /* exact: pio2_1 has 33 bits, |fn| small */
  double w = fn * pio2_2;
  double b = a - w;
  double c = (a - b) - w;              /* rounding error of b */
  double w2 = fn * pio2_2t - c;
  double y0 = b - w2;
  double y1 = (b - y0) - w2;
  (void)pio2_1t;
  *r = y0;
  *rt = y1;
  return n & 3;
}

WG_TRIG_FN double wg_trig_ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

WG_TRIG_FN double wg_trig_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

WG_TRIG_FN double wg_sin(double x) {
  double r, rt;
  if (x == 0.0) return x;
  switch (wg_trig_reduce(x, &r, &rt)) {
    case 0: return wg_trig_ksin(r, rt);
    case 1: return wg_trig_kcos(r, rt);
    case 2: return -wg_trig_ksin(r, rt);
    default: return -wg_trig_kcos(r, rt);
  }
}

WG_TRIG_FN double wg_cos(double x) {
  double r, rt;
  if (x == 0.0) return 1.0;
  switch (wg_trig_reduce(x, &r, &rt)) {
    case 0: return wg_trig_kcos(r, rt);
    case 1: return -wg_trig_ksin(r, rt);
    case 2: return -wg_trig_kcos(r, rt);
    default: return wg_trig_ksin(r, rt);
  }
}

#endif /* WG_TRIG_H */
