// wg_footcons.cpp -- ZMP polytopes of a feet trajectory (host part of libwg_mpc.so).
//
//   FootConstraintsAsLinearSystem::BuildLinearConstraintInequalities  src/Mathematics/FootConstraintsAsLinearSystem.cpp:258-539
//   FootConstraintsAsLinearSystem::ComputeLinearSystem                :97-256
//   FootConstraintsAsLinearSystem::FindSimilarConstraints             :55-92
//   ComputeConvexHull::DoComputeConvexHull                            src/Mathematics/ConvexHull.cpp:88-203
//
// Runs once per step sequence (one polytope per support phase), far off the per-tick path: plain host C++.  Its output
// feeds wg_dimitrov_tick_batch (one wg_zmp_polytope_t per previewed instant, picked by time like
// ZMPConstrainedQPFastFormulation::BuildConstraintMatrices :783-795, 822-840 does).  sin / cos from include/wg_trig.h,
// like every other trigonometric value in this library.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/wg_mpc.h"
#define WG_TRIG_FN static inline
#include "../../include/wg_trig.h"

namespace {

const double kPi = 3.14159265358979323846;

struct Pt {
  double col, row;   // CH_Point: col = x, row = y
};

inline double cross_about(const Pt &o, const Pt &a, const Pt &b) {
  const double x1 = a.col - o.col, x2 = b.col - o.col, y1 = a.row - o.row, y2 = b.row - o.row;
  return x1 * y2 - x2 * y1;
}

// Graham scan about the lowest point.  The reference keeps the candidates in a std::set ordered by the sign of the cross
// product about p0; two candidates in the same direction are reduced to the farther one before the insertion.
std::vector<Pt> graham_hull(const std::vector<Pt> &pts) {
  std::vector<Pt> hull;
  if (pts.empty()) return hull;
  Pt p0 = pts[0];
  for (const Pt &p : pts)
    if (p.row < p0.row) p0 = p;
  std::vector<Pt> order;                                  // ascending polar angle about p0
  for (const Pt &p : pts) {
    bool insert = true;
    for (size_t k = 0; k < order.size();) {
      if (cross_about(p0, order[k], p) == 0.0) {
        const double dk = std::sqrt((order[k].col - p0.col) * (order[k].col - p0.col) + (order[k].row - p0.row) * (order[k].row - p0.row));
        const double dp = std::sqrt((p.col - p0.col) * (p.col - p0.col) + (p.row - p0.row) * (p.row - p0.row));
        if (dk <= dp) {
          order.erase(order.begin() + (long)k);
          continue;
        }
        insert = false;
      }
      k++;
    }
    if (!insert) continue;
    size_t pos = 0;
    bool equivalent = false;
    for (; pos < order.size(); pos++) {
      if (cross_about(p0, p, order[pos]) > 0.0) break;    // p orders before order[pos]
      if (!(cross_about(p0, order[pos], p) > 0.0)) equivalent = true;
    }
    if (!equivalent) order.insert(order.begin() + (long)pos, p);
  }
  if (order.size() < 2) return hull;
  hull.push_back(p0);
  hull.push_back(order[0]);
  hull.push_back(order[1]);
  for (size_t it = 2; it < order.size(); it++) {
    const Pt &pi = order[it];
    for (;;) {
      bool ok = true;
      if (hull.size() >= 2) {
        const Pt &s1 = hull[hull.size() - 1], &s2 = hull[hull.size() - 2];
        ok = cross_about(s2, s1, pi) > 0.0;
      }
      if (ok) break;
      hull.pop_back();
    }
    hull.push_back(pi);
  }
  return hull;
}

// the half plane left of the edge p -> q as a x + c y + b >= 0; `anchor` is the point the reference takes the offset at
void half_plane(const Pt &p, const Pt &q, const Pt &anchor, double &a, double &c, double &b) {
  if (std::fabs(q.col - p.col) > 1e-7) {
    double x1, y1, x2, y2, lmul = -1.0;
    if (q.col < p.col) {
      lmul = 1.0;
      x1 = q.col; y1 = q.row; x2 = p.col; y2 = p.row;
    } else {
      x1 = p.col; y1 = p.row; x2 = q.col; y2 = q.row;
    }
    a = (y2 - y1) / (x2 - x1);
    b = (anchor.row - a * anchor.col);
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
}

bool polytope_of(const std::vector<Pt> &h, wg_zmp_polytope_t *P) {
  const int n = (int)h.size();
  if (n < 2 || n > WG_POLY_MAX_ROWS) return false;
  std::memset(P, 0, sizeof *P);
  P->nrows = n;
  double cx = 0.0, cy = 0.0;
  for (int i = 0; i < n - 1; i++) {
    cx += h[i].col;
    cy += h[i].row;
    half_plane(h[i], h[i + 1], h[i], P->A[i][0], P->A[i][1], P->B[i]);          // offset at the edge's first point
  }
  cx += h[n - 1].col;
  cy += h[n - 1].row;
  P->centre[0] = cx / (double)n;
  P->centre[1] = cy / (double)n;
  half_plane(h[n - 1], h[0], h[0], P->A[n - 1][0], P->A[n - 1][1], P->B[n - 1]);  // closing edge: offset at its last point
  const int half = n == 4 ? 2 : (n == 6 ? 3 : 0);       // FindSimilarConstraints knows rectangles and hexagons
  for (int k = 0; k < half; k++)
    if (P->A[k][0] == -P->A[k + half][0] && P->A[k][1] == -P->A[k + half][1]) P->similar[k + half] = -half;
  return true;
}

void sole_corners(const double *f, double hw, double hh, Pt *out) {   // f: x, y, z, theta (degrees), ...
  static const double sx[4] = {1.0, 1.0, -1.0, -1.0}, sy[4] = {-1.0, 1.0, 1.0, -1.0};   // counter-clockwise
  const double s = wg_sin(f[3] * kPi / 180.0), c = wg_cos(f[3] * kPi / 180.0);
  for (int j = 0; j < 4; j++) {
    out[j].col = f[0] + (sx[j] * hw * c - sy[j] * hh * s);
    out[j].row = f[1] + (sx[j] * hw * s + sy[j] * hh * c);
  }
}

}  // namespace

extern "C" int wg_foot_constraints(int n, const double *time, const double *left, const int *left_type, const double *right,
                                   double sole_w, double sole_h, double constraint_x, double constraint_y, int cap,
                                   wg_zmp_polytope_t *polys, double *t_start, double *t_end) {
  if (n < 0 || cap < 0 || (n > 0 && (!time || !left || !left_type || !right)) || (cap > 0 && (!polys || !t_start || !t_end)))
    return WG_ERR_BAD_ARG;
  double hw = sole_w * 0.5, hh = sole_h * 0.5;
  hh -= constraint_y;
  hw -= constraint_x;
  enum { START = 0, RIGHT_SUPPORT = 1, LEFT_SUPPORT = 2, DOUBLE_SUPPORT = 3 };
  int state = START, count = 0;
  for (int i = 0; i < n; i++) {
    const double *L = left + 6 * (size_t)i, *R = right + 6 * (size_t)i;
    int next = state;
    bool fresh = false;
    if (i == 0) {
      fresh = true;
      next = DOUBLE_SUPPORT;
      state = DOUBLE_SUPPORT;
    }
    if (left_type[i] >= 10)
      next = DOUBLE_SUPPORT;
    else {
      const double lifting = 0.00001;
      if (L[2] > lifting)
        next = LEFT_SUPPORT;                              // the reference's state 2: the LEFT foot is in the air
      else if (R[2] > lifting)
        next = RIGHT_SUPPORT;
      else if (R[2] < lifting && L[2] < lifting)
        next = DOUBLE_SUPPORT;
    }
    if (next != state) fresh = true;
    state = next;
    if (fresh) {
      std::vector<Pt> hull;
      if (state == DOUBLE_SUPPORT) {
        std::vector<Pt> pts(8);
        sole_corners(L, hw, hh, pts.data());
        sole_corners(R, hw, hh, pts.data() + 4);
        hull = graham_hull(pts);
      } else {
        hull.resize(4);
        sole_corners(L[2] < R[2] ? L : R, hw, hh, hull.data());
      }
      wg_zmp_polytope_t P;
      if (!polytope_of(hull, &P)) return WG_ERR_BAD_ARG;
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
