// Kajita preview-control gains (host side of the C ABI, include/wg_mpc.h: wg_riccati_solve / wg_riccati_gains).
//
// Replaces OptimalControllerSolver::ComputeWeights (src/PreviewControl/OptimalControllerSolver.cpp:200-352) and the
// system set-up of PreviewControl::ComputeOptimalWeights (src/PreviewControl/PreviewControl.cpp:198-322).
//
// The reference obtains the stabilising solution P of
//     P = A'PA - A'Pb (R + b'Pb)^-1 b'PA + c'Qc
// from an ordered generalised Schur form of the symplectic pencil (LAPACK dgges, OptimalControllerSolver.cpp:133-198).
// LAPACK is not part of this image and the system is 3x3 / 4x4, so P is computed here with the structure-preserving
// doubling algorithm (quadratically convergent, same fixed point: the unique stabilising solution) followed by two
// Newton-free fixed-point polishing steps of the Riccati map.  Everything downstream of P (K, the F recursion) follows
// the reference's operation order.  Microseconds on the host: no GPU kernel.
#include <cmath>
#include <cstring>

#include "../../include/wg_mpc.h"

namespace {

constexpr int kMaxN = 8;

struct Mat {
  int r = 0, c = 0;
  double v[kMaxN * kMaxN];
  double &operator()(int i, int j) { return v[i * c + j]; }
  double operator()(int i, int j) const { return v[i * c + j]; }
};

Mat zeros(int r, int c) { Mat m; m.r = r; m.c = c; std::memset(m.v, 0, sizeof(m.v)); return m; }
Mat eye(int n) { Mat m = zeros(n, n); for (int i = 0; i < n; i++) m(i, i) = 1.0; return m; }

// plain triple loop, k ascending (the order of ublas prod on dense matrices)
Mat mul(const Mat &a, const Mat &b) {
  Mat o = zeros(a.r, b.c);
  for (int i = 0; i < a.r; i++)
    for (int j = 0; j < b.c; j++) {
      double s = 0.0;
      for (int k = 0; k < a.c; k++) s += a(i, k) * b(k, j);
      o(i, j) = s;
    }
  return o;
}
Mat tr(const Mat &a) { Mat o = zeros(a.c, a.r); for (int i = 0; i < a.r; i++) for (int j = 0; j < a.c; j++) o(j, i) = a(i, j); return o; }
Mat add(const Mat &a, const Mat &b) { Mat o = a; for (int i = 0; i < a.r * a.c; i++) o.v[i] += b.v[i]; return o; }
Mat sub(const Mat &a, const Mat &b) { Mat o = a; for (int i = 0; i < a.r * a.c; i++) o.v[i] -= b.v[i]; return o; }
Mat scale(const Mat &a, double s) { Mat o = a; for (int i = 0; i < a.r * a.c; i++) o.v[i] *= s; return o; }

// X = W^-1 * B by Gaussian elimination with partial pivoting; false when W is numerically singular
bool solve(Mat W, Mat B, Mat &X) {
  const int n = W.r;
  for (int k = 0; k < n; k++) {
    int p = k; double best = std::fabs(W(k, k));
    for (int i = k + 1; i < n; i++) if (std::fabs(W(i, k)) > best) { best = std::fabs(W(i, k)); p = i; }
    if (!(best > 0.0)) return false;
    if (p != k) {
      for (int j = 0; j < n; j++) { double t = W(k, j); W(k, j) = W(p, j); W(p, j) = t; }
      for (int j = 0; j < B.c; j++) { double t = B(k, j); B(k, j) = B(p, j); B(p, j) = t; }
    }
    for (int i = k + 1; i < n; i++) {
      const double f = W(i, k) / W(k, k);
      if (f == 0.0) continue;
      for (int j = k; j < n; j++) W(i, j) -= f * W(k, j);
      for (int j = 0; j < B.c; j++) B(i, j) -= f * B(k, j);
    }
  }
  X = zeros(n, B.c);
  for (int j = 0; j < B.c; j++)
    for (int i = n - 1; i >= 0; i--) {
      double s = B(i, j);
      for (int k = i + 1; k < n; k++) s -= W(i, k) * X(k, j);
      X(i, j) = s / W(i, i);
    }
  return true;
}

double maxabs(const Mat &a) { double m = 0.0; for (int i = 0; i < a.r * a.c; i++) m = std::fmax(m, std::fabs(a.v[i])); return m; }

// one application of the Riccati map  P -> A'PA - A'Pb (R + b'Pb)^-1 b'PA + H
Mat riccati_map(const Mat &A, const Mat &b, const Mat &H, double R, const Mat &P) {
  const Mat At = tr(A), bt = tr(b);
  const Mat PA = mul(P, A);
  const Mat btPA = mul(bt, PA);                 // 1 x n
  const double den = R + mul(mul(bt, P), b)(0, 0);
  Mat out = add(mul(At, PA), H);
  const Mat corr = scale(mul(tr(btPA), btPA), 1.0 / den);
  return sub(out, corr);
}

// structure-preserving doubling:  A_{k+1} = A_k W^-1 A_k,  G_{k+1} = G_k + A_k W^-1 G_k A_k',
//                                 H_{k+1} = H_k + A_k' H_k W^-1 A_k,   W = I + G_k H_k;   H_k -> P
bool dare(const Mat &A0, const Mat &b, const Mat &c, double Q, double R, Mat &P) {
  const int n = A0.r;
  Mat A = A0;
  Mat G = scale(mul(b, tr(b)), 1.0 / R);
  const Mat H0 = mul(scale(tr(c), Q), c);
  Mat H = H0;
  bool converged = false;
  for (int it = 0; it < 60; it++) {
    const Mat W = add(eye(n), mul(G, H));
    Mat WiA, WiG;
    if (!solve(W, A, WiA) || !solve(W, G, WiG)) return false;
    const Mat An = mul(A, WiA);
    const Mat Gn = add(G, mul(mul(A, WiG), tr(A)));
    const Mat Hn = add(H, mul(mul(tr(A), H), WiA));
    const double dh = maxabs(sub(Hn, H)), hs = maxabs(Hn);
    A = An; G = Gn; H = Hn;
    for (int i = 0; i < n * n; i++) if (!std::isfinite(H.v[i])) return false;
    if (dh <= 1e-15 * hs) { converged = true; break; }
  }
  if (!converged) return false;
  // symmetrise, then polish on the original equation
  P = scale(add(H, tr(H)), 0.5);
  for (int k = 0; k < 2; k++) {
    const Mat Pn = riccati_map(A0, b, H0, R, P);
    P = scale(add(Pn, tr(Pn)), 0.5);
  }
  return true;
}

}  // namespace

extern "C" int wg_riccati_solve(int n, const double *A_rm, const double *b, const double *c, double Q, double R,
                                int Nl, int mode, double *K, double *F) {
  if (n < 1 || n > kMaxN || !A_rm || !b || !c || !K || (Nl > 0 && !F) || Nl < 0 || !(R > 0.0)) return WG_ERR_BAD_ARG;
  Mat A = zeros(n, n), B = zeros(n, 1), C = zeros(1, n);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) A(i, j) = A_rm[i * n + j];
    B(i, 0) = b[i];
    C(0, i) = c[i];
  }
  Mat P;
  if (!dare(A, B, C, Q, R, P)) return WG_ERR_BAD_ARG;

  // OptimalControllerSolver.cpp:300-352
  const Mat tb = tr(B);
  double la = R + mul(mul(tb, P), B)(0, 0);
  la = 1 / la;
  Mat Km = scale(mul(tb, mul(P, A)), la);                 // K = la * b' (P A)
  for (int i = 0; i < n; i++) K[i] = Km(0, i);

  const Mat Pre = scale(tb, la);
  const Mat Base = tr(sub(A, mul(B, Km)));
  Mat Post = scale(tr(C), Q);
  if (mode == WG_RICCATI_WITHOUT_INITIALPOS) Post = mul(P, Post);
  Mat Rec = Post;
  for (int k = 0; k < Nl; k++) {
    F[k] = mul(Pre, Rec)(0, 0);
    Rec = mul(Base, Rec);
  }
  return WG_OK;
}

extern "C" int wg_riccati_gains(double T, double zc, double Q, double R, int Nl, int mode, double *K, double *F) {
  if (!(T > 0.0) || !K) return WG_ERR_BAD_ARG;
  // PreviewControl.cpp:203-214
  const double A3[9] = {1.0, T, T * T / 2.0, 0.0, 1.0, T, 0.0, 0.0, 1.0};
  const double B3[3] = {T * T * T / 6.0, T * T / 2.0, T};
  const double C3[3] = {1.0, 0.0, -zc / 9.81};
  if (mode == WG_RICCATI_WITH_INITIALPOS) {               // :297-315  (K[0] doubles as Ks there; K[3] unused)
    double K3[3];
    const int rc = wg_riccati_solve(3, A3, B3, C3, Q, R, Nl, mode, K3, F);
    if (rc != WG_OK) return rc;
    K[0] = K3[0]; K[1] = K3[1]; K[2] = K3[2]; K[3] = 0.0;
    return WG_OK;
  }
  if (mode != WG_RICCATI_WITHOUT_INITIALPOS) return WG_ERR_BAD_ARG;
  // :232-273  augmented ("derivated") system: state (accumulated ZMP error, dx)
  double Ax[16] = {0}, bx[4], cx[4] = {1.0, 0.0, 0.0, 0.0};
  double tmpA[3];
  for (int j = 0; j < 3; j++) {
    double s = 0.0;
    for (int k = 0; k < 3; k++) s += C3[k] * A3[k * 3 + j];
    tmpA[j] = s;
  }
  Ax[0] = 1.0;
  for (int i = 0; i < 3; i++) {
    Ax[0 * 4 + i + 1] = tmpA[i];
    for (int j = 0; j < 3; j++) Ax[(i + 1) * 4 + j + 1] = A3[i * 3 + j];
  }
  double tb = 0.0;
  for (int k = 0; k < 3; k++) tb += C3[k] * B3[k];
  bx[0] = tb;
  for (int i = 0; i < 3; i++) bx[i + 1] = B3[i];
  return wg_riccati_solve(4, Ax, bx, cx, Q, R, Nl, mode, K, F);   // K = [Ks, Kx0, Kx1, Kx2]
}
