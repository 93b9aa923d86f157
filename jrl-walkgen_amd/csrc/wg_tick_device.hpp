// wg_tick_device.hpp -- one Herdt-2010 MPC tick per wavefront.
//
// Device-side replacement for the body of
//   ZMPVelocityReferencedQP::OnLine   src/ZMPRefTrajectoryGeneration/ZMPVelocityReferencedQP.cpp:346-452
// and what it calls on the reference's CPU path:
//   SupportFSM                        src/PreviewControl/SupportFSM.cpp:57-153
//   GeneratorVelRef                   src/ZMPRefTrajectoryGeneration/generator-vel-ref.cpp:70-229, 284-474, 554-674
//   OrientationsPreview               src/ZMPRefTrajectoryGeneration/OrientationsPreview.cpp:79-418
//   RelativeFeetInequalities          src/Mathematics/relative-feet-inequalities.cpp:185-319
//   QPProblem::solve -> ql0001_       (wg_ql_device.hpp)
//   LinearizedInvertedPendulum2D      src/PreviewControl/LinearizedInvertedPendulum2D.cpp:157-264
//   OnLineFootTrajectoryGeneration    src/FootTrajectoryGeneration/OnLineFootTrajectoryGeneration.cpp:50-346
//
// One wave owns one gait.  The branchy scalar bookkeeping (support FSM, orientation
// preview, polygon edges) runs on lane 0 and leaves its results in LDS; QP assembly,
// the solve and the sample interpolation are lane-parallel.  Every floating-point
// expression keeps the reference's evaluation order (ublas prod sums k ascending from
// 0.0, compute_term scales after the product, add_term_to accumulates with +=).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/wg_mpc.h"
#define WG_TRIG_FN __host__ __device__ static inline
#include "../../include/wg_trig.h"
#include "wg_ql_device.hpp"
#include "wg_ql_herdt.hpp"

namespace wg {

#define GmL(i, j) q.G[(i) + (j) * q.ldg]
#define AmL(k, i) q.A[(k) + (i) * q.lda]

constexpr int kNMaxH = 32;   // largest horizon
constexpr int kSMax = 4;     // largest number of previewed steps
constexpr double kPi = 3.14159265358979323846;

// Condensed cart-table maps + invariant Hessian block, built once per model on the host
// (RigidBodySystem::compute_dyn_cjerk, rigid-body-system.cpp:377-452;
//  GeneratorVelRef::build_invariant_part, generator-vel-ref.cpp:587-614).
struct TickTables {
  double Sv[kNMaxH][3], Sz[kNMaxH][3];
  double Uv[kNMaxH][kNMaxH], Uz[kNMaxH][kNMaxH];
  double Qb[kNMaxH][kNMaxH];
  // gait-independent part of ql0002's set-up for C = blockdiag(Qb, Qb) + border (wg_ql_herdt.hpp):
  double R2[2 * kNMaxH * (2 * kNMaxH + 1) / 2];   // packed Cholesky factor of blockdiag(Qb, Qb), qld.cpp:859-890
  double Z2[4 * kNMaxH * kNMaxH];                 // its inverse, column-major ld 2N, qld.cpp:937-975
  double diag_b;                                  // diagonal test over the constant block, qld.cpp:814-843
  int blocks_ok, pad_;
  // Z2 is blockdiag(Zb, Zb) with zeros in between, but the recurrence leaves SIGNED zeros in the upper-right block
  // (-(x * 0) / r); bit i + j*N = sign of Z2(i, N + j).  The kernel fetches only Zb and re-creates the rest from this.
  unsigned long long z2_cross_sign[(kNMaxH * kNMaxH + 63) / 64];
};

// qb_override (N x N, row-major) replaces the host loop's Q_b: the matrix-core Gramian of wg_gramian_batch
// (WG_FLAG_GRAMIAN_MFMA_*); everything derived from Q_b (factor blocks, diagonal test) then follows the override.
inline void build_tables(const wg_model_t &m, TickTables &t, const double *qb_override = nullptr) {
  const int N = m.N;
  const double T = m.T, h = m.com_height_qp;
  for (unsigned i = 0; i < (unsigned)kNMaxH; i++)
    for (unsigned j = 0; j < (unsigned)kNMaxH; j++) { t.Uv[i][j] = 0.0; t.Uz[i][j] = 0.0; t.Qb[i][j] = 0.0; }
  for (unsigned i = 0; i < (unsigned)N; i++) {
    t.Sv[i][0] = 0.0; t.Sv[i][1] = 1.0; t.Sv[i][2] = (i + 1) * T;
    t.Sz[i][0] = 1.0; t.Sz[i][1] = (i + 1) * T;
    t.Sz[i][2] = (i + 1) * (i + 1) * T * T * 0.5 - h / 9.81;
    for (unsigned j = 0; j <= i; j++) {
      t.Uv[i][j] = (2 * (i - j) + 1) * T * T * 0.5;
      t.Uz[i][j] = (1 + 3 * (i - j) + 3 * (i - j) * (i - j)) * T * T * T / 6.0 - T * h / 9.81;
    }
  }
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      double pj = 0.0, pv = 0.0, pz = 0.0;
      for (int k = 0; k < N; k++) {
        pj += ((k == i) ? 1.0 : 0.0) * ((k == j) ? 1.0 : 0.0);
        pv += t.Uv[k][i] * t.Uv[k][j];
        pz += t.Uz[k][i] * t.Uz[k][j];
      }
      double q = 0.0;
      q += pj * m.beta;
      q += pv * m.alpha;
      q += pz * m.gamma;
      t.Qb[i][j] = qb_override ? qb_override[i * N + j] : q;
    }
  // ---- constant factor blocks: exactly ql0002's recurrences on blockdiag(Qb, Qb), eps = 1e-8 ----
  {
    const int M = 2 * N;
    const double vsmall = 1e-8;
    auto g2 = [&](int i, int j) -> double {
      if (i < N && j < N) return t.Qb[i][j];
      if (i >= N && j >= N) return t.Qb[i - N][j - N];
      return 0.0;
    };
    double diag = 0.0;
    for (int i = 0; i < M; ++i) {
      const double wdi = g2(i, i);
      diag = (diag >= vsmall - wdi) ? diag : vsmall - wdi;
      for (int j = i + 1; j < M; ++j) {
        const double gjj = g2(j, j), gij = g2(i, j);
        double ga = -((wdi <= gjj) ? wdi : gjj);
        const double gb = fabs(wdi - gjj) + fabs(gij);
        if (gb > 0.0) ga += gij * gij / gb;
        diag = (diag >= ga) ? diag : ga;
      }
    }
    t.diag_b = diag;
    t.blocks_ok = (diag <= 0.0) ? 1 : 0;
    t.pad_ = 0;
    auto R = [&](int i, int j) -> double & { return t.R2[(size_t)j * (j + 1) / 2 + i]; };
    for (int j = 0; j < M && t.blocks_ok; ++j) {
      double temp = 0.0;
      for (int i = 0; i <= j; ++i) {
        temp = g2(i, j);
        for (int k = 0; k < i; ++k) temp -= R(k, j) * R(k, i);
        if (i < j) R(i, j) = temp / R(i, i);
      }
      if (temp < vsmall) { t.blocks_ok = 0; break; }
      R(j, j) = sqrt(temp);
    }
    auto Z = [&](int i, int j) -> double & { return t.Z2[(size_t)i + (size_t)j * M]; };
    if (t.blocks_ok)
      for (int i = 0; i < M; ++i) {
        for (int j = 0; j < i; ++j) Z(i, j) = 0.0;
        Z(i, i) = 1.0 / R(i, i);
        for (int j = i; j < M - 1; ++j) {
          double sum = 0.0;
          for (int k = i; k <= j; ++k) sum += Z(i, k) * R(k, j + 1);
          Z(i, j + 1) = -sum / R(j + 1, j + 1);
        }
      }
    for (auto &w : t.z2_cross_sign) w = 0ull;
    if (t.blocks_ok)
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i)
          if (std::signbit(Z(i, N + j))) t.z2_cross_sign[(i + j * N) / 64] |= 1ull << ((i + j * N) % 64);
  }
}

struct Sup {
  int phase, foot, nb_steps_left, step_number, state_changed, pad;
  double time_limit, start_time, x, y, yaw;
};

// LDS scratch of the tick (besides the solver's QlView).  Two groups:
//   * persistent: read by the solver's problem view or by the post-solve interpolation;
//   * pre-solve only (support preview, selection vectors, reference, S*c products, edge offsets): dead once the QP is
//     assembled.  In the compact build they are overlaid on the solver's Z matrix, which the solver writes first thing
//     (factor()) and nobody reads before -- 2.8 KB less LDS per gait.
struct TickLds {
  // ---- pre-solve only ----
  Sup *sup;                       // [N+1]
  double *VcX, *VcY;              // [N]
  double *Vc_fX, *Vc_fY;          // [kSMax]
  double *trunk;                  // [N+1]
  double *refx, *refy;            // [N]
  double *svx, *svy, *szx, *szy;  // [N]
  double *rowD;                   // [m]
  // ---- persistent ----
  Sup *sup0;                      // [1] copy of sup[0] for the post-solve phase
  int *stepidx;                   // [N]
  double *V_f;                    // [kSMax*kSMax]
  double *sup_angles;             // [8]
  double *rowA, *rowB;            // [m]  polygon edge per constraint row
  int *rowK;                      // [m]  instant (CoP rows) or step (foot rows)
  double *misc;                   // [16] scalars handed from lane 0 to the wave
  double *uvec, *Gv, *gd;         // compact problem view (wg_ql_herdt.hpp): N, nmax x kGvLd, nmax
  wg_gait_state_t *st;            // working copy of the state
  __host__ __device__ static size_t pre_doubles(int N, int m) { return (size_t)(2 * N + 2 * kSMax + (N + 1) + 6 * N + m); }
  __host__ __device__ static size_t pre_bytes(int N, int smax) {
    const int m = 1 + 4 * N + 5 * smax;
    return sizeof(Sup) * (N + 1) + 8 * pre_doubles(N, m);
  }
  // gvld > 0: a table view (compact: kGvLd, element: kGvLdElem) -- its tables are present and the pre-solve group lives
  // elsewhere (overlay); gvld == 0: dense view, G and A are LDS matrices of the solver area
  // rows_presolve: rowA / rowB / rowK are only read while the QP is assembled and the register rows are loaded (the compact
  // view keeps every row's coefficients in registers afterwards) -- they join the pre-solve group
  // gv_lds = false: the border block Gv lives in a per-block slot of global memory (see mpc_tick)
  // rows_lds = false (with rows_presolve = false): rowA / rowB / rowK live in the per-block slot of global memory too
  // lean = true (element view): the Hessian diagonal gd joins Gv in the global slot and the state copy is parked on the overlay
  __host__ __device__ static size_t bytes(int N, int smax = kSMax, int gvld = 0, bool rows_presolve = false, bool gv_lds = true,
                                          bool rows_lds = true, bool lean = false) {
    const int m = 1 + 4 * N + 5 * smax;
    const int nmax = 2 * N + 2 * smax;
    const bool compact = gvld > 0;
    const bool rows_here = !rows_presolve && rows_lds;
    size_t b = sizeof(Sup) + 8 * (size_t)(kSMax * kSMax + 8 + (rows_here ? 2 * m : 0) + 16 +
                                          (compact ? (N + 2) + (gv_lds ? nmax * gvld : 0) + (lean ? 0 : nmax) : 0)) +
               4 * (size_t)(((N + 1) & ~1) + (rows_here ? ((m + 1) & ~1) : 0)) +
               ((rows_presolve || lean) ? 0 : ((sizeof(wg_gait_state_t) + 15) & ~(size_t)15));
    if (!compact) b += pre_bytes(N, smax);
    return (b + 15) & ~(size_t)15;
  }
  // kExtGv: Gv is ext_gv (global memory); kExtRows: rowA | rowB | rowK are ext_rows (global memory, m doubles each, then m
  // ints) -- decided at compile time so that their accesses are global_ instructions, not flat_
  // kLean (element view): gd is ext_gd (global memory) and the state copy sits on the overlay (parked during the solve)
  template <bool kExtGv = false, bool kExtRows = false, bool kLean = false>
  __device__ __forceinline__ void carve(char *base, int N, int smax, int gvld, char *overlay, bool rows_presolve, double *ext_gv = nullptr,
                                        double *ext_rows = nullptr, double *ext_gd = nullptr) {
    const bool compact = gvld > 0;
    const int m = 1 + 4 * N + 5 * smax;
    char *p = base;
    if constexpr (!kLean) { if (!rows_presolve) { st = reinterpret_cast<wg_gait_state_t *>(p); p += (sizeof(wg_gait_state_t) + 15) & ~(size_t)15; } }
    sup0 = reinterpret_cast<Sup *>(p); p += sizeof(Sup);
    double *d = reinterpret_cast<double *>(p);
    V_f = d; d += kSMax * kSMax; sup_angles = d; d += 8;
    if constexpr (kExtRows) { rowA = ext_rows; rowB = ext_rows + m; rowK = reinterpret_cast<int *>(ext_rows + 2 * m); }
    else if (!rows_presolve) { rowA = d; d += m; rowB = d; d += m; }
    misc = d; d += 16;
    uvec = Gv = gd = nullptr;
    if (compact) {
      const int nmax = 2 * N + 2 * smax;
      uvec = d + 2; d += N + 2;          // uvec[-1] = uvec[-2] = 0.0: a row walked past its instant reads an exact-zero coefficient
      if constexpr (kExtGv) Gv = ext_gv; else { Gv = d; d += nmax * gvld; }
      if constexpr (kLean) gd = ext_gd; else { gd = d; d += nmax; }
    }
    int *ip = reinterpret_cast<int *>(d);
    stepidx = ip; ip += (N + 1) & ~1;
    if constexpr (!kExtRows) { if (!rows_presolve) { rowK = ip; ip += (m + 1) & ~1; } }
    char *o = compact ? overlay : reinterpret_cast<char *>(ip);
    sup = reinterpret_cast<Sup *>(o); o += sizeof(Sup) * (N + 1);
    double *e = reinterpret_cast<double *>(o);
    VcX = e; e += N; VcY = e; e += N; Vc_fX = e; e += kSMax; Vc_fY = e; e += kSMax;
    trunk = e; e += N + 1; refx = e; e += N; refy = e; e += N;
    svx = e; e += N; svy = e; e += N; szx = e; e += N; szy = e; e += N;
    rowD = e; e += m;
    if constexpr (kLean) {
      char *q = reinterpret_cast<char *>(e);
      q += (0 - reinterpret_cast<size_t>(q)) & 15;
      st = reinterpret_cast<wg_gait_state_t *>(q);           // parked in its HBM slot during the solve (mpc_tick)
    }
    if (rows_presolve) {
      rowA = e; e += m; rowB = e; e += m; rowK = reinterpret_cast<int *>(e);
      // the working copy of the state also lives on Z: it is written back before the solver starts and fetched again
      // after it (mpc_tick), so that it costs no LDS while the solve is resident
      char *q = reinterpret_cast<char *>(rowK + ((m + 1) & ~1));
      q += (0 - reinterpret_cast<size_t>(q)) & 15;           // up to 16 bytes, as pointer arithmetic (an integer round trip would
                                                              // make `st` a generic pointer: flat_ instead of ds_ accesses)
      st = reinterpret_cast<wg_gait_state_t *>(q);
    }
  }
  // Element view: its pre-solve group and the state copy lie over R and the four scratch vectors of the solver area (dead outside
  // the solve; the state copy is fetched back from its HBM slot right after the solve, while x still holds the solution).  Short
  // horizons (N <= 13) have a smaller R than that: their overlay gets its own bytes behind the tick's arrays instead -- the
  // solver area itself stays at the start of the wave's LDS for every model (a run-time offset there turns every address of R,
  // x, ... from an immediate into a register: measured at N = 32, 70 more spilled registers and -4.7 %)
  __host__ __device__ static size_t elem_overlay_need(int N, size_t state_bytes) { return (pre_bytes(N, kSMax) + state_bytes + 32 + 15) & ~(size_t)15; }
  __host__ __device__ static bool elem_overlay_apart(int N, size_t state_bytes) {
    return elem_overlay_need(N, state_bytes) > (size_t)8 * ((size_t)(2 * N) * (2 * N + 1) / 2 + 2 * N + 4 * 2 * N);
  }
  __host__ __device__ static size_t overlay_bytes(int N, int smax) {   // what the compact view parks on Z before the solve
    const int m = 1 + 4 * N + 5 * smax;
    return pre_bytes(N, smax) + 16 * (size_t)m + 4 * (size_t)((m + 1) & ~1) + 16 + ((sizeof(wg_gait_state_t) + 15) & ~(size_t)15);
  }
};

// ---------------------------------------------------------------------------
// lane-0 scalar code
// ---------------------------------------------------------------------------
#define WG_FSM_EPS 1e-6
#define WG_OP_EPS 0.00000001

// SupportFSM::update_vel_reference, SupportFSM.cpp:57-90
__device__ inline void fsm_update_vel_reference(wg_gait_state_t *s, double *ref, int cur_foot) {
  s->in_translation = (fabs(ref[0]) > 2 * WG_FSM_EPS || fabs(ref[1]) > 2 * WG_FSM_EPS) ? 1 : 0;
  if (fabs(ref[2]) > WG_FSM_EPS) s->in_rotation = 1;
  else if (s->in_rotation && !s->in_translation) {
    ref[0] = 2 * WG_FSM_EPS; ref[1] = 2 * WG_FSM_EPS;
    if (!s->post_rotation_phase) {
      s->rot_support_foot = cur_foot; s->nb_steps_after_rotation = 0; s->post_rotation_phase = 1;
    } else {
      if (s->rot_support_foot != cur_foot) { s->rot_support_foot = cur_foot; ++s->nb_steps_after_rotation; }
      if (s->nb_steps_after_rotation > 2) { s->in_rotation = 0; s->post_rotation_phase = 0; }
    }
  } else s->in_rotation = 0;
}

// SupportFSM::set_support_state, SupportFSM.cpp:93-153
__device__ inline void fsm_set_support_state(const wg_model_t &m, int nb_steps_ssds, double time, unsigned pi, Sup &S,
                                             const double *ref) {
  const double T = m.T;
  S.state_changed = 0;
  const bool given = fabs(ref[0]) > WG_FSM_EPS || fabs(ref[1]) > WG_FSM_EPS || fabs(ref[2]) > WG_FSM_EPS;
  if (given && S.phase == WG_DS && (S.time_limit - time - WG_FSM_EPS) > m.dsss_period) {
    S.time_limit = time + m.dsss_period - T / 10.0;
    S.nb_steps_left = nb_steps_ssds;
  }
  if (time + WG_FSM_EPS + pi * T >= S.time_limit) {
    if (S.phase == WG_SS && !given && S.nb_steps_left == 0) {
      S.phase = WG_DS;
      S.time_limit = time + pi * T + m.ds_period - T / 10.0;
      S.state_changed = 1;
    } else if ((S.phase == WG_DS && given) || (S.phase == WG_DS && S.nb_steps_left > 0)) {
      S.phase = WG_SS;
      S.time_limit = time + pi * T + m.step_period - T / 10.0;
      S.nb_steps_left = nb_steps_ssds;
      S.state_changed = 1;
    } else if ((S.phase == WG_SS && S.nb_steps_left > 0) || (S.nb_steps_left == 0 && given)) {
      S.foot = (S.foot == WG_LEFT) ? WG_RIGHT : WG_LEFT;
      S.state_changed = 1;
      S.time_limit = time + pi * T + m.step_period - T / 10.0;
      if (pi != 1) ++S.step_number;
      if (!given) S.nb_steps_left = S.nb_steps_left - 1;
      if (given) S.nb_steps_left = nb_steps_ssds;
    }
  }
}

struct Hull { int nv; double X[5], Y[5], A[5], B[5], D[5]; };

// RelativeFeetInequalities::set_vertices (:185-234) on the hulls of init_convex_hulls (:88-149) followed by
// compute_linear_system (:264-319).  NV is a compile-time constant so the vertex arrays stay in registers.
template <int NV>
__device__ __forceinline__ void hull_edges(const wg_model_t &m, int hull_foot, int hull_phase, double yaw, int sign_foot,
                                           double *A, double *B, double *D) {
  double X[NV], Y[NV];
  if (NV == 4) {
    const double lx[4] = {1.0, 1.0, -1.0, -1.0};
    const double lyr[4] = {-1.0, 1.0, 1.0, -1.0}, lyl[4] = {1.0, -1.0, -1.0, 1.0};
    double hw = 0.5 * m.sole_w; hw -= m.margin_x;      // FootHalfSize.cpp:62-70
    double hh = 0.5 * m.sole_h; hh -= m.margin_y;
    const double hhds = hh + m.ds_feet_distance / 2.0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      X[j] = lx[j] * hw;
      if (hull_foot == WG_LEFT) Y[j] = (hull_phase == WG_DS) ? lyl[j] * hhds - m.ds_feet_distance / 2.0 : lyl[j] * hh;
      else Y[j] = (hull_phase == WG_DS) ? lyr[j] * hhds + m.ds_feet_distance / 2.0 : lyr[j] * hh;
    }
  } else {
    const double px[5] = {-0.28, -0.2, 0.0, 0.2, 0.28};
    const double py[5] = {-0.2, -0.3, -0.4, -0.3, -0.2};
#pragma unroll
    for (int j = 0; j < NV; j++) { X[j] = px[j]; Y[j] = (hull_foot == WG_LEFT) ? py[j] : -py[j]; }
  }
  const double c = wg_cos(yaw), s = wg_sin(yaw);        // convex_hull_t::rotate, privatepgtypes.cpp:157-185
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const double xo = X[j], yo = Y[j];
    X[j] = (xo * c - yo * s);
    Y[j] = (xo * s + yo * c);
  }
  const double sign = (sign_foot == WG_LEFT) ? 1.0 : -1.0;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int i2 = (i + 1 == NV) ? 0 : i + 1;
    const double y1 = Y[i], y2 = Y[i2], x1 = X[i], x2 = X[i2];
    const double dx = y1 - y2, dy = x2 - x1;
    const double dc = dx * x1 + dy * y1;
    A[i] = sign * dx; B[i] = sign * dy; D[i] = sign * dc;
  }
}

// OrientationsPreview::verify_angle_hip_joint :273-303
__device__ inline bool op_verify_angle(const wg_model_t &m, wg_gait_state_t *s, double sup_time_passed, const Sup &cur,
                                       double trunk_end, double cur_sup_angle, unsigned step_number) {
  double ul, ll;
  if (cur.foot == WG_LEFT) { ul = m.hip_l_hi; ll = m.hip_l_lo; } else { ul = m.hip_r_hi; ll = m.hip_r_lo; }
  const double lim = (s->trunkT_yaw[1] < 0.0) ? ll : ul;
  if (fabs(trunk_end - cur_sup_angle) > fabs(lim)) {
    s->trunkT_yaw[1] = (cur_sup_angle + 0.9 * lim - s->trunk_yaw[0] - s->trunk_yaw[1] * m.T / 2.0) /
                       (sup_time_passed + step_number * m.step_period - m.T / 2.0);
    return false;
  }
  return true;
}

// OrientationsPreview::preview_orientations :79-251
__device__ inline void op_preview(const wg_model_t &m, wg_gait_state_t *s, double time, const double *ref, Sup *sup,
                                  double *sup_angles, double *trunk) {
  const int N = m.N;
  const double T = m.T, SSP = m.step_period;
  const Sup cur = sup[0];
  if (cur.phase != WG_DS) {                                  // verify_acceleration_hip_joint :254-270
    if (fabs(ref[2] - s->trunk_yaw[1]) > 2.0 / 3.0 * T * m.hip_amax) {
      const double sg = (ref[2] - s->trunk_yaw[1] < 0.0) ? -1.0 : 1.0;
      s->trunkT_yaw[1] = s->trunk_yaw[1] + sg * 2.0 / 3.0 * T * m.hip_amax;
    } else s->trunkT_yaw[1] = ref[2];
  } else s->trunkT_yaw[1] = 0.0;

  for (int e = 0; e < 8; e++) sup_angles[e] = 0.0;             // defined contents even when a pass is abandoned (guards below)
  bool vel_ok = false, angle_ok = false;
  double first_prw = 0.0;
  const double sign_rot_vel = (s->trunkT_yaw[1] < 0.0) ? -1.0 : 1.0;
  double sup_time_passed = 0.0;
  unsigned step_number = 0;
  double trunk_end = 0.0;
  int na = 0;
  const unsigned last = (unsigned)((int)ceil((N + 1) * T / SSP));
  int guard = 0;
  while (!vel_ok && guard++ < 64) {
    const double cur_sup_angle = (cur.foot == WG_LEFT) ? s->lf[0].theta * kPi / 180.0 : s->rf[0].theta * kPi / 180.0;
    if (cur.phase != WG_DS) {
      angle_ok = false;
      int g2 = 0;
      while (!angle_ok && g2++ < 64) {
        if (fabs(s->trunkT_yaw[1] - s->trunk_yaw[1]) > WG_OP_EPS) {
          const double a = s->trunk_yaw[0], b = s->trunk_yaw[1], c = 0.0;
          const double d = 3.0 * (s->trunkT_yaw[1] - s->trunk_yaw[1]) / (T * T);
          const double e = -2.0 * d / (3.0 * T);
          s->trunkT_yaw[0] = a + b * T + 1.0 / 2.0 * c * T * T + 1.0 / 3.0 * d * T * T * T + 1.0 / 4.0 * e * T * T * T * T;
        } else s->trunkT_yaw[0] = s->trunk_yaw[0] + s->trunk_yaw[1] * T;
        sup_time_passed = cur.time_limit - time;
        trunk_end = s->trunkT_yaw[0] + s->trunkT_yaw[1] * (sup_time_passed - T);
        angle_ok = op_verify_angle(m, s, sup_time_passed, cur, trunk_end, cur_sup_angle, step_number);
      }
    } else {
      sup_time_passed = cur.time_limit + SSP - time;
      first_prw = 1;
      sup_angles[na++] = cur_sup_angle;
      s->trunkT_yaw[0] = trunk_end = s->trunk_yaw[0];
    }
    double prev_sup_angle = cur_sup_angle;
    double prw_foot = (cur.foot == WG_LEFT) ? 1.0 : -1.0;
    for (step_number = (unsigned)first_prw; step_number <= last; step_number++) {
      prw_foot = -prw_foot;
      double prw_angle = trunk_end + s->trunkT_yaw[1] * SSP / 2.0;
      // verify_velocity_hip_joint (:306-365) receives its angle by value: no effect
      if ((double)prw_foot * (prev_sup_angle - prw_angle) - WG_OP_EPS > m.feet_cross_max)
        prw_angle = prev_sup_angle + (double)sign_rot_vel * m.feet_cross_max;
      else if (fabs(prw_angle - prev_sup_angle) > m.hip_vmax * SSP)
        prw_angle = prev_sup_angle + (double)prw_foot * m.hip_vmax * (SSP - T);
      angle_ok = op_verify_angle(m, s, sup_time_passed, cur, trunk_end, cur_sup_angle, step_number);
      if (!angle_ok) { na = 0; vel_ok = false; break; }
      else if (na < 8) sup_angles[na++] = prw_angle;
      trunk_end = trunk_end + SSP * s->trunkT_yaw[1];
      prev_sup_angle = prw_angle;
      vel_ok = true;
    }
  }
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

// dst = cond ? a : b, field by field (keeps the samples in registers)
__device__ __forceinline__ void foot_select(wg_foot_sample_t &dst, bool cond, const wg_foot_sample_t &a,
                                            const wg_foot_sample_t &b) {
#define WG_FSEL(f) dst.f = cond ? a.f : b.f
  WG_FSEL(x); WG_FSEL(y); WG_FSEL(z); WG_FSEL(theta); WG_FSEL(omega); WG_FSEL(omega2);
  WG_FSEL(dx); WG_FSEL(dy); WG_FSEL(dz); WG_FSEL(dtheta); WG_FSEL(domega); WG_FSEL(domega2);
  WG_FSEL(ddx); WG_FSEL(ddy); WG_FSEL(ddz); WG_FSEL(ddtheta); WG_FSEL(ddomega); WG_FSEL(ddomega2);
#undef WG_FSEL
}

// polynomial helpers, Polynome.cpp:44-76, PolynomeFoot.cpp:59-79, 100-120, 226-240
__device__ inline double poly_eval(const double *c, int deg, double t) {
  double r = 0.0, pt = 1.0;
  for (int i = 0; i <= deg; i++) { r += c[i] * pt; pt *= t; }
  return r;
}
__device__ inline double poly_d1(const double *c, int deg, double t) {
  double r = 0, pt = 1;
  for (int i = 1; i <= deg; i++) { r += i * c[i] * pt; pt *= t; }
  return r;
}
__device__ inline double poly_d2(const double *c, int deg, double t) {
  double r = 0, pt = 1;
  for (int i = 2; i <= deg; i++) { r += i * (i - 1) * c[i] * pt; pt *= t; }
  return r;
}
__device__ inline void poly5_set(double *c, double FT, double FP, double p0, double v0, double a0) {
  double tmp;
  c[0] = p0; c[1] = v0; c[2] = a0 / 2.0;
  tmp = FT * FT * FT;
  c[3] = (-3.0 / 2.0 * a0 * FT * FT - 6.0 * v0 * FT - 10.0 * p0 + 10.0 * FP) / tmp;
  tmp = tmp * FT;
  c[4] = (3.0 / 2.0 * a0 * FT * FT + 8.0 * v0 * FT + 15.0 * p0 - 15.0 * FP) / tmp;
  tmp = tmp * FT;
  c[5] = (-1.0 / 2.0 * a0 * FT * FT - 3.0 * v0 * FT - 6.0 * p0 + 6.0 * FP) / tmp;
}
__device__ inline void poly4_set(double *c, double FT, double MP) {
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
__device__ inline void poly3_set(double *c, double FT, double FP, double p0, double v0) {
  double tmp;
  c[0] = p0; c[1] = v0;
  tmp = FT * FT;
  if (FT == 0.0) { c[2] = 0.0; c[3] = 0.0; }
  else {
    c[2] = (3 * FP - 3 * p0 - 2 * v0 * FT) / tmp;
    c[3] = (v0 * FT + 2 * p0 - 2 * FP) / (tmp * FT);
  }
}

// ---------------------------------------------------------------------------
// the tick: executed by one wave.  `lds` = QlView area followed by TickLds area.
// ---------------------------------------------------------------------------
struct TickDiag { int ifail, n_iter, nact, n, m, ns; };

// NH == 16: compact problem view (rows in registers, no G / A anywhere); NH == 0: generic dense view (G, A in LDS);
// NH == -1: element view (any N <= 32: G / A regenerated per element from the compact tables, wg_ql_herdt.hpp)
// NH == 32: the element view with BASELINE config 5's horizon as a compile-time constant: every offset of the per-block global
//           slot and of the wave's LDS is a constant behind one base, the solver keeps only its two-rows-per-lane forms
// Where the dense view writes the assembled QP instead of solving it (wg_mpc_assemble_batch): one gait's slice of the
// arrays QPProblem::solve hands to ql0001_ (qp-problem.cpp:245-279) -- what QPProblem::dump_problem prints (:639-653).
struct QpDumpOut {
  double *C, *d, *A, *b, *xl, *xu;     // nmax x nmax | nmax | mmax x nmax (column-major) | mmax | nmax | nmax
  int *n, *m;
  int nmax, mmax;
};

// NHP == 33: NH = 32 with Z in REGISTERS (wg_ql_device.hpp, ZRegs): one wave per SIMD (512 registers), four gaits per CU, the
//            wave's LDS also holds the Z^T a tile; the global slot keeps Z only between the factorisation and the load
template <int NHP>
__device__ __forceinline__ TickDiag mpc_tick(const wg_model_t &m, const TickTables *__restrict__ tb, wg_gait_state_t *gstate,
                                    wg_tick_out_t *out, double *lds_ql, char *lds_tick, int *hist, int hist_cap,
                                    int *hist_len, double *zglobal = nullptr, int elem_nact_cap = 0,
                                    const QpDumpOut *dump = nullptr) {
  const int lane = wg_lane();
  constexpr bool kRegZ = (NHP == 33);
  constexpr int NH = kRegZ ? 32 : NHP;
  constexpr bool kElem = (NH == -1 || NH == 32);  // element view: any horizon (-1), or the horizon as a constant (32)
  // compact view: the horizon is a compile-time constant everywhere (checked by the host).  Fixed element view: a constant where
  // it decides ADDRESSES (kNC below: slot and LDS offsets, table strides), but the trip count of the assembly's loops stays a
  // scalar the compiler cannot see through -- unrolled 32-fold they cost the kernel 856 spilled VGPRs and 1248 B of scratch
  constexpr int kNC = (NH > 0) ? NH : 0;
  int N = (NH == 16) ? 16 : m.N;
  if constexpr (NH == 32) { N = 32; asm volatile("" : "+s"(N)); }
  const double T = m.T;
  const int K = WG_SAMPLES_PER_TICK;
  TickLds L;
  // compact: the pre-solve group is overlaid on Z, the first array of the solver's area (QlView::carve without G)
  static_assert(sizeof(Sup) % 8 == 0, "Sup must keep doubles aligned");
  constexpr int kGvStride = (NH == 16) ? kGvLd : (kElem ? kGvLdElem : 0);
  constexpr int kGvOff = kElem ? 0 : 1;
  // compact view with a global slot: [wa (kMmax + kNmax) | b (kMmax) | Gv (kNmax x kGvLd)] leave the LDS -- 2.6 KB, the
  // difference between seven and eight gaits per CU.  All three are read lane-parallel, early in their phases.
  constexpr int kExtWab = (2 * 16 + 4) + 2 * (1 + 4 * 16 + 10);
  double *ext16 = (NH == 16) ? zglobal : nullptr;         // never null for the compact view (the host reserves the slot)
  // element view: the slot holds [Z (nmax x (nmax|1)) | wa (mmax + nmax) | b (mmax) | Gv (nmax x kGvLdElem) | rowA | rowB | rowK
  // (2 mmax + (mmax + 1) / 2 + 2 doubles) | gd | d | wd | wx (nmax each)] for the largest problem of the model -- with all of
  // them in LDS a CU holds four gaits at N = 32, without them six
  const int eNmax = 2 * (kNC ? kNC : N) + 2 * kSMax, eMmax = 1 + 4 * (kNC ? kNC : N) + 5 * kSMax;
  // fixed element view: Z's leading dimension is eNmax itself -- every column is 9 whole 64-byte lines and every 64-row load of the
  // sweep 8 whole lines (the slot is 64-byte aligned); the odd leading dimension of the other views (bank-conflict-free in LDS)
  // makes each column straddle a tenth line: measured 1.29 x the useful bytes in the Z^T a walk (profiles/round3_fetchcal_*)
  double *extE = kElem ? zglobal + (size_t)eNmax * (NH == 32 ? eNmax : (eNmax | 1)) : nullptr;
  const int eWab = (eMmax + eNmax) + eMmax;
  const int eRows = eWab + eNmax * kGvStride;               // offset of the row tables, then of gd | d | wd | wx
  const int eCold = eRows + 2 * eMmax + (eMmax + 1) / 2 + 2;
  const int eRfull = eCold + 4 * eNmax;                    // a full-size R (element view with a column cap on its LDS part)
  if constexpr (NH == 32) {
    // fixed layout: the tick's own arrays first (their size is a constant of the horizon), the solver area behind them with R
    // last (QlView::carve_fixed_elem) -- every LDS address of the tick is the wave's base plus a constant; the pre-solve overlay
    // lies over the solver's scratch vectors and R
    const size_t tb_bytes = (TickLds::bytes(32, kSMax, kGvStride, false, false, false, true) + 15) & ~(size_t)15;
    lds_tick = reinterpret_cast<char *>(lds_ql);
    lds_ql = reinterpret_cast<double *>(lds_tick + tb_bytes);
    char *ovl = reinterpret_cast<char *>(lds_ql + QlView::fixed_elem_head<2 * 32 + 2 * kSMax>());
    L.template carve<true, true, true>(lds_tick, kNC, kSMax, kGvStride, ovl, false, extE + eWab, extE + eRows, extE + eCold);
  }
  else if constexpr (NH == -1) {
    char *ovl = reinterpret_cast<char *>(lds_ql);
    if (TickLds::elem_overlay_apart(N, sizeof(wg_gait_state_t)))           // short horizons: behind the tick's own arrays
      ovl = lds_tick + ((TickLds::bytes(N, kSMax, kGvStride, false, false, false, true) + 15) & ~(size_t)15);
    L.template carve<true, true, true>(lds_tick, N, kSMax, kGvStride, ovl, false, extE + eWab, extE + eRows, extE + eCold);
  }
  else
    L.template carve<NH == 16>(lds_tick, N, (NH == 16) ? 2 : kSMax, kGvStride, reinterpret_cast<char *>(lds_ql), NH == 16,
                               (NH == 16) ? ext16 + kExtWab : nullptr);
  wg_gait_state_t *s = L.st;
#ifdef WG_PROFILE
  unsigned long long tk0 = clock64(), tk1 = 0, tk2 = 0, tk3 = 0, tka = 0, tkb = 0, tkc = 0, tkd = 0, tke = 0, tkf = 0, tkg = 0;
#endif

  // ---- state: HBM -> LDS (coalesced 8-byte lanes) ----
  {
    // agent-scope loads: in a multi-tick launch the previous tick of this gait may have run on another XCD
    const double *src = reinterpret_cast<const double *>(gstate);
    double *dst = reinterpret_cast<double *>(s);
    for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64)
      dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  WG_WSYNC();
  const double time = s->clock;
#ifdef WG_PROFILE
  tka = clock64();
#endif

  // ---- lane 0: support FSM, selection, orientations, polygon edges ----
  if (lane == 0) {
    double ref[3] = {s->vref[0], s->vref[1], s->vref[2]};
    fsm_update_vel_reference(s, ref, s->foot);                       // :353-357
    Sup cur;
    cur.phase = s->phase; cur.foot = s->foot; cur.nb_steps_left = s->nb_steps_left; cur.step_number = s->step_number;
    cur.state_changed = s->state_changed; cur.pad = 0; cur.time_limit = s->time_limit; cur.start_time = s->start_time;
    cur.x = s->sup_x; cur.y = s->sup_y; cur.yaw = s->sup_yaw;
    // preview_support_states, generator-vel-ref.cpp:70-134
    fsm_set_support_state(m, s->nb_steps_ssds, time, 0, cur, ref);
    if (cur.state_changed) {
      const wg_foot_sample_t *f = (cur.foot == WG_LEFT) ? &s->lf[0] : &s->rf[0];
      cur.x = f->x; cur.y = f->y; cur.yaw = f->theta * kPi / 180.0; cur.start_time = time;
    }
    L.sup[0] = cur;
    s->phase = cur.phase; s->foot = cur.foot; s->nb_steps_left = cur.nb_steps_left; s->step_number = cur.step_number;
    s->state_changed = cur.state_changed; s->time_limit = cur.time_limit; s->start_time = cur.start_time;
    s->sup_x = cur.x; s->sup_y = cur.y; s->sup_yaw = cur.yaw;
    {
      Sup prw = cur;
      prw.step_number = 0;
      int hull_src = 0;
      for (unsigned pi = 1; pi <= (unsigned)N; pi++) {
        fsm_set_support_state(m, s->nb_steps_ssds, time, pi, prw, ref);
        if (prw.state_changed) {
          if (pi == 1) {
            const wg_foot_sample_t *f = (prw.foot == WG_LEFT) ? &s->lf[2] : &s->rf[2];
            prw.x = f->x; prw.y = f->y; prw.yaw = f->theta * kPi / 180.0; prw.start_time = time + pi * T;
          }
          if (prw.step_number > 0) { prw.x = 0.0; prw.y = 0.0; }
        }
        L.sup[pi] = prw;
        // the CoP hull changes only where the support state changes: remember which previewed state carries it
        if (prw.state_changed) hull_src = (int)pi;
        L.stepidx[pi - 1] = hull_src << 8;                        // low byte (step index) is filled in below, per lane
      }
    }
    int ns = L.sup[N].step_number;
    constexpr int kSCap = (NH == 16) ? 2 : kSMax;   // compact kernel: N*T <= 2*step_period is checked at configure time
    if (ns > kSCap) ns = kSCap;                     // cannot happen; keeps every index in range
#ifdef WG_PROFILE
    tkb = clock64();
#endif
    op_preview(m, s, time, ref, L.sup, L.sup_angles, L.trunk);
    *L.sup0 = L.sup[0];
    L.misc[0] = (double)ns;
    L.misc[1] = ref[0]; L.misc[2] = ref[1]; L.misc[3] = ref[2];
  }
  WG_WSYNC();
#ifdef WG_PROFILE
  tkc = clock64();
#endif

  // ---- lane-parallel part of the bookkeeping: one previewed instant per lane ----
  {
    const int ns0 = uni((int)L.misc[0]);
    const double ref0 = L.misc[1], ref1 = L.misc[2];
    // generate_selection_matrices :137-208, one instant per lane.  The few scalar cells (Vc_f, V_f) are written by the
    // lanes whose instant defines them; lanes that hit the same cell write the same value.
    if (lane < kSMax) { L.Vc_fX[lane] = 0.0; L.Vc_fY[lane] = 0.0; }
    if (lane < kSMax * kSMax) L.V_f[lane] = 0.0;
    WG_WSYNC();
    if (lane < N) {
      const int i = lane;
      const Sup &S = L.sup[i + 1];
      double vx = 0.0, vy = 0.0;
      int sidx = 0;
      if (S.step_number > 0) {
        sidx = S.step_number;
        if (S.step_number == 1 && S.state_changed && S.phase == WG_SS) {
          L.Vc_fX[0] = L.sup[i].x; L.Vc_fY[0] = L.sup[i].y;
          L.V_f[0] = 1.0;
        } else if (S.step_number > 1 && S.step_number <= kSMax) {
          L.V_f[(S.step_number - 1) * kSMax + (S.step_number - 2)] = -1.0;
          L.V_f[(S.step_number - 1) * kSMax + (S.step_number - 1)] = 1.0;
        }
      } else { vx = S.x; vy = S.y; }
      L.VcX[i] = vx; L.VcY[i] = vy;
      L.stepidx[i] = (L.stepidx[i] & ~0xff) | sidx;
    }
    WG_WSYNC();
    // compute_global_reference :211-229 (trunk[2..N] all hold the same angle, OrientationsPreview.cpp:229-233)
    if (lane < N) {
      const int i = lane;
      const int e = i < 2 ? i : 2;
      const double yt = L.trunk[e < N ? e : N - 1];
      const double cs = wg_cos(yt), sn = wg_sin(yt);
      L.refx[i] = ref0 * cs - ref1 * sn;
      L.refy[i] = ref1 * cs + ref0 * sn;
    }
    // polygon edges per constraint row: build_inequalities_cop :284-314, build_inequalities_feet :317-354
    const int mq0 = 1 + 4 * N + 5 * ns0;
    if (lane == 0) { L.rowA[0] = 0.0; L.rowB[0] = 0.0; L.rowD[0] = 0.0; L.rowK[0] = -1; }
    for (int r = 1 + 4 * N + lane; r < mq0; r += 64) { L.rowA[r] = 0.0; L.rowB[r] = 0.0; L.rowD[r] = 0.0; L.rowK[r] = -1; }
    WG_WSYNC();
    if (lane < N) {
      const int i = lane;
      const int packed = L.stepidx[i];
      const int src = packed >> 8;
      const Sup &H = L.sup[src];                                 // the state whose hull is in force at instant i
      const Sup &S = L.sup[i + 1];
      double eA[4], eB[4], eD[4];
      hull_edges<4>(m, H.foot, H.phase, H.yaw, S.foot, eA, eB, eD);   // edges re-signed by the instant's own foot
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int r = 1 + 4 * i + e;
        L.rowA[r] = eA[e]; L.rowB[r] = eB[e]; L.rowD[r] = eD[e]; L.rowK[r] = i;
      }
      if (S.state_changed && S.step_number > 0 && S.step_number <= ns0 && S.phase != WG_DS) {
        double fA[5], fB[5], fD[5];
        hull_edges<5>(m, L.sup[i].foot, L.sup[i].phase, L.sup[i].yaw, S.foot, fA, fB, fD);
        const int k = S.step_number - 1;
#pragma unroll
        for (int e = 0; e < 5; e++) {
          const int r = 1 + 4 * N + 5 * k + e;
          L.rowA[r] = fA[e]; L.rowB[r] = fB[e]; L.rowD[r] = fD[e]; L.rowK[r] = k;
        }
      }
    }
    WG_WSYNC();
    if (lane < N) L.stepidx[lane] &= 0xff;                       // back to the plain step index
  }
  WG_WSYNC();

#ifdef WG_PROFILE
  tk1 = clock64();
#endif
  const int ns = uni((int)L.misc[0]);     // the same in every lane: keep it (and n, m, every solver address) scalar
  const int n = 2 * N + 2 * ns;
  const int mq = 1 + 4 * N + 5 * ns;     // rows incl. the dummy row 0 (qp-problem.cpp:248)
  constexpr bool kCompactView = (NH == 16);
  constexpr bool kTableView = (NH == 16) || kElem;        // Hessian / constraints kept as compact tables
  // element view with a Z slot in global memory: Z leaves the LDS (it is the operand that caps the residency at N = 32)
  constexpr bool z_in_lds = !kElem;
  constexpr bool kElemView = kElem;
  QlDims D(n, mq, mq, !kTableView, true, kCompactView ? 2 * NH + 4 : 0, !kTableView, z_in_lds, !kElemView, !kElemView,
           kElemView ? (elem_nact_cap & 0xffff) : 0);       // ordered sums run the static length
  QlView q;
  if constexpr (kCompactView) {
    constexpr int kNmax = 2 * NH + 4, kMmax = 1 + 4 * NH + 10;     // two previewed steps at most (wg_mpc_configure)
    // same footprint as QlDims(kNmax, kMmax, kMmax, dense = false, nsc = kNmax, bounds = false), which sized the LDS on the host
    q.template carve_fixed<kNmax, kMmax, kNmax, true>(lds_ql, n, mq, 0, ext16);
  } else {
    if constexpr (NH == 32) {
      q.template carve_fixed_elem<2 * 32 + 2 * kSMax, 1 + 4 * 32 + 5 * kSMax>(lds_ql, n, mq, 0, elem_nact_cap & 0xffff, zglobal, extE, extE + (eMmax + eNmax),
                                                                              extE + eCold + eNmax, extE + eCold + 2 * eNmax, extE + eCold + 3 * eNmax,
                                                                              extE + eRfull);
      if (q.nact_cap > 0 && (elem_nact_cap >> 16) > 0 && (elem_nact_cap >> 16) < q.nact_cap) q.nact_cap = elem_nact_cap >> 16;   // tests
      if constexpr (kRegZ) {                                // the Z^T a tile behind R's LDS part (the layout's cap, not the tests')
        const int rc = elem_nact_cap & 0xffff, nn = 2 * 32 + 2 * kSMax;
        q.ztile = q.R + ((rc > 0 && rc < nn) ? rc * (rc + 1) / 2 : nn * (nn + 1) / 2) + nn;
      }
    }
    else if constexpr (kElemView) {
      q.template carve<false, false, false>(lds_ql, D, 0, extE, eMmax + eNmax, extE + eCold + eNmax, eNmax);
      q.Z = zglobal;
      q.Rf = extE + eRfull;                                 // the LDS may hold only r_cols columns of R: the Cholesky factor needs all n
      if (q.nact_cap > 0 && (elem_nact_cap >> 16) > 0 && (elem_nact_cap >> 16) < q.nact_cap) q.nact_cap = elem_nact_cap >> 16;   // tests
    }
    else q.carve(lds_ql, D, 0);
  }

  WG_REP(9) {                                               // attribution builds: the QP assembly twice (it only reads the state)
  // ---- S*c products (MV2_ = prod(S, CoM), generator-vel-ref.cpp:780-787) ----
  for (int i = lane; i < N; i += 64) {
    double ax = 0.0, ay = 0.0, bx = 0.0, by = 0.0;
    for (int k = 0; k < 3; k++) {
      ax += tb->Sv[i][k] * s->com_x[k]; ay += tb->Sv[i][k] * s->com_y[k];
      bx += tb->Sz[i][k] * s->com_x[k]; by += tb->Sz[i][k] * s->com_y[k];
    }
    L.svx[i] = ax; L.svy[i] = ay; L.szx[i] = bx; L.szy[i] = by;
  }
  WG_WSYNC();

  // ---- gradient, update_problem :617-674 ----
  for (int i = lane; i < N; i += 64) {
    double t1x = 0.0, t1y = 0.0, t2x = 0.0, t2y = 0.0;
    for (int k = 0; k < N; k++) {
      const double u = tb->Uv[k][i];
      t1x += u * L.svx[k]; t1y += u * L.svy[k];
      t2x += u * L.refx[k]; t2y += u * L.refy[k];
    }
    double dx = 0.0, dy = 0.0;
    dx += t1x * m.alpha; dy += t1y * m.alpha;
    dx += t2x * (-m.alpha); dy += t2y * (-m.alpha);
    q.d[i] = dx; q.d[N + i] = dy;
  }
  for (int j = lane; j < ns; j += 64) {
    double px = 0.0, py = 0.0, qx = 0.0, qy = 0.0;
    for (int k = 0; k < N; k++) {
      const double v = (L.stepidx[k] == j + 1) ? 1.0 : 0.0;
      px += v * L.szx[k]; py += v * L.szy[k];
      qx += v * L.VcX[k]; qy += v * L.VcY[k];
    }
    double dx = 0.0, dy = 0.0;
    dx += px * (-m.gamma); dy += py * (-m.gamma);
    dx += qx * m.gamma; dy += qy * m.gamma;
    q.d[2 * N + j] = dx; q.d[2 * N + ns + j] = dy;
  }
  if constexpr (!kTableView) { for (int i = lane; i < n; i += 64) { q.xl[i] = -1e8; q.xu[i] = 1e8; } }   // qp-problem.cpp:118-121 (table views: constants)

  // ---- Hessian ----
  if constexpr (kTableView) {
    // table views: Qb and u as small LDS tables, the 2ns border columns (symmetric) and the diagonal
    for (int d = lane; d < N; d += 64) L.uvec[d] = tb->Uz[d][0];                 // Uz[r][c] = u[r-c]
    if (lane < 2) L.uvec[-1 - lane] = 0.0;
    for (int e = lane; e < n * kGvStride; e += 64) L.Gv[e] = 0.0;
    for (int i = lane; i < 2 * N; i += 64) L.gd[i] = tb->Qb[i % N][i % N];
    WG_WSYNC();
    for (int e = lane; e < N * ns; e += 64) {
      const int i = e % N, j = e / N;
      double p = 0.0;
      for (int k = 0; k < N; k++) { const double v = (L.stepidx[k] == j + 1) ? 1.0 : 0.0; p += tb->Uz[k][i] * v; }
      p *= -m.gamma;
      // kGvOff: the compact view stores both column groups side by side, the element view folds them (wg_ql_herdt.hpp)
      L.Gv[i * kGvStride + j] = 0.0 + p;                                        // x block  x  x-foot column
      L.Gv[(N + i) * kGvStride + kGvOff * ns + j] = 0.0 + p;                    // y block  x  y-foot column
    }
    for (int e = lane; e < ns * ns; e += 64) {
      const int i = e % ns, j = e / ns;
      double p = 0.0;
      for (int k = 0; k < N; k++) {
        const double vi = (L.stepidx[k] == i + 1) ? 1.0 : 0.0, vj = (L.stepidx[k] == j + 1) ? 1.0 : 0.0;
        p += vi * vj;
      }
      p *= m.gamma;
      if (i == j) { L.gd[2 * N + i] = 0.0 + p; L.gd[2 * N + ns + i] = 0.0 + p; }
      else { L.Gv[(2 * N + i) * kGvStride + j] = 0.0 + p; L.Gv[(2 * N + ns + i) * kGvStride + kGvOff * ns + j] = 0.0 + p; }
    }
    // the border rows' Gv entries toward the border columns mirror the lower-left block (symmetry is exact:
    // (VT Uz)(j,i) and (UzT V)(i,j) are the same products in the same order)
  } else {
    for (int e = lane; e < n * n; e += 64) {
      const int i = e % n, j = e / n;
      double v = 0.0;
      if (i < N && j < N) v = tb->Qb[i][j];
      else if (i >= N && i < 2 * N && j >= N && j < 2 * N) v = tb->Qb[i - N][j - N];
      GmL(i, j) = v;
    }
    WG_WSYNC();
    for (int e = lane; e < N * ns; e += 64) {
      const int i = e % N, j = e / N;
      double p = 0.0, pt = 0.0;
      for (int k = 0; k < N; k++) {
        const double v = (L.stepidx[k] == j + 1) ? 1.0 : 0.0;
        p += tb->Uz[k][i] * v;
        pt += v * tb->Uz[k][i];
      }
      p *= -m.gamma; pt *= -m.gamma;
      GmL(i, 2 * N + j) += p; GmL(N + i, 2 * N + ns + j) += p;
      GmL(2 * N + j, i) += pt; GmL(2 * N + ns + j, N + i) += pt;
    }
    for (int e = lane; e < ns * ns; e += 64) {
      const int i = e % ns, j = e / ns;
      double p = 0.0;
      for (int k = 0; k < N; k++) {
        const double vi = (L.stepidx[k] == i + 1) ? 1.0 : 0.0, vj = (L.stepidx[k] == j + 1) ? 1.0 : 0.0;
        p += vi * vj;
      }
      p *= m.gamma;
      GmL(2 * N + i, 2 * N + j) += p; GmL(2 * N + ns + i, 2 * N + ns + j) += p;
    }
  }

  // ---- constraints, build_constraints_cop :393-448, build_constraints_feet :451-474 ----
  for (int r = lane; r < mq; r += 64) {
    const double a = L.rowA[r], bb = L.rowB[r];
    const int kk = L.rowK[r];
    double bacc = 0.0;
    if (r >= 1 && r <= 4 * N) {
      const int i = kk;
      if constexpr (!kTableView) {
        for (int c = 0; c < N; c++) {
          const double u = tb->Uz[i][c];
          const double px = 0.0 + a * u, py = 0.0 + bb * u;
          AmL(r, c) = 0.0 + px * -1.0; AmL(r, N + c) = 0.0 + py * -1.0;
        }
        for (int j = 0; j < ns; j++) {
          const double v = (L.stepidx[i] == j + 1) ? 1.0 : 0.0;
          const double px = 0.0 + a * v, py = 0.0 + bb * v;
          AmL(r, 2 * N + j) = 0.0 + px * 1.0; AmL(r, 2 * N + ns + j) = 0.0 + py * 1.0;
        }
      }
      bacc += L.rowD[r];
      bacc += (0.0 + a * L.szx[i]) * -1.0;
      bacc += (0.0 + bb * L.szy[i]) * -1.0;
      bacc += (0.0 + a * L.VcX[i]) * 1.0;
      bacc += (0.0 + bb * L.VcY[i]) * 1.0;
    } else {
      if constexpr (!kTableView) for (int c = 0; c < n; c++) AmL(r, c) = 0.0;
      if (r > 4 * N && kk >= 0) {
        const int k = kk;
        if constexpr (!kTableView) {
          for (int j = 0; j < ns; j++) {
            const double vf = L.V_f[k * kSMax + j];
            const double px = 0.0 + a * vf, py = 0.0 + bb * vf;
            AmL(r, 2 * N + j) = 0.0 + px * -1.0; AmL(r, 2 * N + ns + j) = 0.0 + py * -1.0;
          }
        }
        bacc += L.rowD[r];
        bacc += (0.0 + a * L.Vc_fX[k]) * 1.0;
        bacc += (0.0 + bb * L.Vc_fY[k]) * 1.0;
      }
    }
    q.b[r] = -bacc;                 // inner sign, qld.cpp:469-475
  }
  WG_WSYNC();
  }   // WG_REP(9)

  // ---- QPProblem::solve -> ql0001_ (eps = 1e-8, qp-problem.cpp:260) ----
#ifdef WG_PROFILE
  tk2 = clock64();
#endif
  QlResult qr;
  if constexpr (kCompactView) {
    if (lane == 0 && fabs(L.gd[n - 1]) == 0.0) L.gd[n - 1] = wg_kconst(1e-8);   // qld.cpp:442-444 (nmax == n)
    WG_WSYNC();
    {
      // the state's LDS copy sits on Z: park it in its HBM slot (L2) for the duration of the solve
      double *dst = reinterpret_cast<double *>(gstate);
      const double *src = reinterpret_cast<const double *>(s);
      for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64) dst[i] = src[i];
    }
    HerdtProb<16> prob;
    prob.Qb = &tb->Qb[0][0]; prob.u = L.uvec; prob.Gv = L.Gv; prob.gd = L.gd;
    prob.rowA = L.rowA; prob.rowB = L.rowB; prob.rowK = L.rowK; prob.stepidx = L.stepidx; prob.V_f = L.V_f;
    prob.R2 = tb->R2; prob.Z2 = tb->Z2; prob.z2sign = tb->z2_cross_sign; prob.diag_b = tb->diag_b; prob.blocks_ok = tb->blocks_ok; prob.ns = ns;
    prob.load_rows(lane, q.b);
    qr = ql_solve(q, prob, 1e-8, hist, hist_cap);
    {
      WG_WSYNC();
      // the stores above completed long ago (the release only waits for them: they are in L2 then, which is where the
      // loads go -- agent-scope atomics: no stale L1 line of the tick's first read, no L1 invalidate that would hit the
      // other resident waves, and no L2 write-back: the same CU reads back what it wrote)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      const double *src = reinterpret_cast<const double *>(gstate);
      double *dst = reinterpret_cast<double *>(s);
      for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      WG_WSYNC();
    }
  } else if constexpr (kElem) {
    if (lane == 0 && fabs(L.gd[n - 1]) == 0.0) L.gd[n - 1] = 1e-8;               // qld.cpp:442-444 (nmax == n)
    WG_WSYNC();
    {
      // the state's LDS copy sits on the solver's area: park it in its HBM slot (L2) for the duration of the solve
      double *dst = reinterpret_cast<double *>(gstate);
      const double *src = reinterpret_cast<const double *>(s);
      for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64) dst[i] = src[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // gd / d (global slot) written above are read by other lanes below
    }
    HerdtElemProbT<(NH == 32) ? 32 : -1> prob;
    typename std::conditional<kRegZ, ZRegs<2 * 32 + 2 * kSMax>, NoZRegs>::type zregs;
    if constexpr (kRegZ) zregs.zt = q.ztile + kZrTile * kZrTs;       // the LDS tail rows behind the tile
    prob.N = N; prob.ns = ns; prob.Qb = &tb->Qb[0][0]; prob.u = L.uvec; prob.Gv = L.Gv; prob.gd = L.gd;
    prob.rowA = L.rowA; prob.rowB = L.rowB; prob.rowK = L.rowK; prob.stepidx = L.stepidx; prob.V_f = L.V_f;
    prob.R2 = tb->R2; prob.Z2 = tb->Z2; prob.z2sign = tb->z2_cross_sign; prob.blocks_ok = tb->blocks_ok;
    QlResume rs;
    qr = ql_solve(q, prob, 1e-8, hist, hist_cap, &rs, &zregs);
    if (WG_UBOOL(qr.ifail == kQlCapHit)) {
      // the active set outgrew the columns of R the LDS holds: the finished columns (packed, qr.nact of them: the capped part
      // and the working column behind it are one contiguous triangle) move to the full-size R of the per-block global slot
      // -- dead since Z = R^-1 was formed -- and the SAME solve goes on there: same arithmetic, slower reads from here on.
      WG_WSYNC();
      double *Rg = extE + eRfull;
      const int cnt = qr.nact * (qr.nact + 1) / 2;
      for (int i = lane; i < cnt; i += 64) Rg[i] = q.R[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      WG_WSYNC();
      q.R = Rg; q.Rf = q.R;
      q.r_tail = n * (n + 1) / 2;
      q.nact_cap = 0;
      rs.valid = 1;
      qr = ql_solve(q, prob, 1e-8, hist, hist_cap, &rs, &zregs);
    }
    {
      WG_WSYNC();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      const double *src = reinterpret_cast<const double *>(gstate);
      double *dst = reinterpret_cast<double *>(s);
      for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      WG_WSYNC();
    }
  } else {
    if (dump) {
      // assemble only: the problem exactly as the tick would hand it to the solver (b in the caller's sign, row 0 the dummy
      // row of qp-problem.cpp:248), zero-padded to the caller's leading dimensions; the gait's state is left alone
      const QpDumpOut &o = *dump;
      for (int e = lane; e < o.nmax * o.nmax; e += 64) { const int i = e % o.nmax, j = e / o.nmax; o.C[e] = (i < n && j < n) ? GmL(i, j) : 0.0; }
      for (int e = lane; e < o.mmax * o.nmax; e += 64) { const int k = e % o.mmax, i = e / o.mmax; o.A[e] = (k < mq && i < n) ? AmL(k, i) : 0.0; }
      for (int i = lane; i < o.nmax; i += 64) { o.d[i] = i < n ? q.d[i] : 0.0; o.xl[i] = i < n ? q.xl[i] : 0.0; o.xu[i] = i < n ? q.xu[i] : 0.0; }
      for (int k = lane; k < o.mmax; k += 64) o.b[k] = k < mq ? -q.b[k] : 0.0;
      if (lane == 0) { *o.n = n; *o.m = mq; }
      TickDiag dg0;
      dg0.ifail = 0; dg0.n_iter = 0; dg0.nact = 0; dg0.n = n; dg0.m = mq; dg0.ns = ns;
      return dg0;
    }
    if (lane == 0 && fabs(GmL(n - 1, n - 1)) == 0.0) GmL(n - 1, n - 1) = 1e-8;   // qld.cpp:442-444 (nmax == n)
    WG_WSYNC();
    DenseProb prob;
    qr = ql_solve(q, prob, 1e-8, hist, hist_cap);
  }
#ifdef WG_PROFILE
  tk3 = clock64();
#endif
  if (hist_len && lane == 0) *hist_len = qr.hist_len;
  // A solve whose iterate became NaN: the reference does not notice (its running comparisons never skip a NaN), adds and drops until
  // maxit = 40 (m + n) and returns ifail = 1 with an all-NaN x, which the tick then integrates -- the gait is lost.  The views'
  // policies follow it there decision by decision (kNanExact, wg_ql_device.hpp: scan_nan_exact); nothing is patched up here.
#if !WG_TICK_NAN_EXACT
  {   // experiment builds only: the views without those forms, one test per tick instead (what round 5 shipped at first; -1.6 % less)
    bool has_nan = false;
    for (int i = lane; i < n; i += 64) { const double xi = q.x[i]; has_nan = has_nan || (xi != xi); }
    if (__ballot(has_nan) != 0ull) {
      for (int i = lane; i < n; i += 64) q.x[i] = __builtin_nan("");
      qr.ifail = 1; qr.n_iter = 40 * (q.m + q.n) + 1;
      WG_WSYNC();
    }
  }
#endif

  // ---- CoM: jerk, 20 interpolated samples, state step (ZMPVelocityReferencedQP.cpp:405-428) ----
  double jx, jy;
  const bool stop_branch = (L.sup0->nb_steps_left == 0) && !(m.flags & WG_FLAG_NO_STOP_CENTERING);
  if (stop_branch) {
    jx = (s->lf[0].x + s->rf[0].x) / 2 - s->front_com_x[0];
    jy = (s->lf[0].y + s->rf[0].y) / 2 - s->front_com_y[0];
    const bool arrived = fabs(jx) < 1e-3 && fabs(jy) < 1e-3;
    const double tf = 0.75;
    jx = 6 / (tf * tf * tf) * (jx - tf * s->front_com_x[1] - (tf * tf / 2) * s->front_com_x[2]);
    jy = 6 / (tf * tf * tf) * (jy - tf * s->front_com_y[1] - (tf * tf / 2) * s->front_com_y[2]);
    WG_WSYNC();
    if (lane == 0 && arrived) s->running = 0;
  } else {
    jx = q.x[0]; jy = q.x[N];
    WG_WSYNC();
    if (lane == 0) s->running = 1;
  }
  {
    const double cx[3] = {s->com_x[0], s->com_x[1], s->com_x[2]}, cy[3] = {s->com_y[0], s->com_y[1], s->com_y[2]};
    const double c02 = -s->com_z / 9.81;
    WG_WSYNC();
    if (lane < K) {                                       // Interpolation :157-227
      const int lk = lane;
      const double t = (lk + 1) * m.Tctrl;
      const double x0 = cx[0] + t * cx[1] + 0.5 * t * t * cx[2] + t * t * t * jx / 6.0;
      const double x1 = cx[1] + t * cx[2] + 0.5 * t * t * jx;
      const double x2 = cx[2] + t * jx;
      const double y0 = cy[0] + t * cy[1] + 0.5 * t * t * cy[2] + t * t * t * jy / 6.0;
      const double y1 = cy[1] + t * cy[2] + 0.5 * t * t * jy;
      const double y2 = cy[2] + t * jy;
      if (out) {
        out->com_x[lk][0] = x0; out->com_x[lk][1] = x1; out->com_x[lk][2] = x2;
        out->com_y[lk][0] = y0; out->com_y[lk][1] = y1; out->com_y[lk][2] = y2;
        out->zmp_x[lk] = 1.0 * x0 + 0.0 * x1 + c02 * x2;
        out->zmp_y[lk] = 1.0 * y0 + 0.0 * y1 + c02 * y2;
      }
      if (lk == 11) {
        s->front_com_x[0] = x0; s->front_com_x[1] = x1; s->front_com_x[2] = x2;
        s->front_com_y[0] = y0; s->front_com_y[1] = y1; s->front_com_y[2] = y2;
      }
    }
    if (lane == 0) {                                      // OneIteration :230-264
      const double A01 = T, A02 = T * T / 2.0, A12 = T;
      const double B0 = T * T * T / 6.0, B1 = T * T / 2.0, B2 = T;
      double nx[3], ny[3];
      nx[0] = 0.0 + 1.0 * cx[0] + A01 * cx[1] + A02 * cx[2];
      nx[1] = 0.0 + 0.0 * cx[0] + 1.0 * cx[1] + A12 * cx[2];
      nx[2] = 0.0 + 0.0 * cx[0] + 0.0 * cx[1] + 1.0 * cx[2];
      ny[0] = 0.0 + 1.0 * cy[0] + A01 * cy[1] + A02 * cy[2];
      ny[1] = 0.0 + 0.0 * cy[0] + 1.0 * cy[1] + A12 * cy[2];
      ny[2] = 0.0 + 0.0 * cy[0] + 0.0 * cy[1] + 1.0 * cy[2];
      s->com_x[0] = nx[0] + jx * B0; s->com_x[1] = nx[1] + jx * B1; s->com_x[2] = nx[2] + jx * B2;
      s->com_y[0] = ny[0] + jy * B0; s->com_y[1] = ny[1] + jy * B1; s->com_y[2] = ny[2] + jy * B2;
      if (out) {
        out->jerk_x = jx; out->jerk_y = jy; out->ifail = qr.ifail; out->n_iter = qr.n_iter; out->nact = qr.nact;
        out->n = n; out->m = mq; out->nb_prw_steps = ns;
      }
    }
  }
  WG_WSYNC();

#ifdef WG_PROFILE
  tkd = clock64();
#endif
  // ---- lane 0: trunk and feet (sequential in k) ----
  if (lane == 0) {
    const Sup cs = *L.sup0;
    const double dt = m.Tctrl;
    // interpolate_trunk_orientation, OrientationsPreview.cpp:368-418
    if (cs.phase == WG_SS && time + 3.0 / 2.0 * T < cs.time_limit) {
      const double a = s->trunk_yaw[1];
      const double c = 3.0 * (s->trunkT_yaw[1] - s->trunk_yaw[1]) / (T * T);
      const double d = -2.0 * c / (3.0 * T);
      const double theta = s->trunk_yaw[0];
      const double yT1 = s->trunkT_yaw[1];
      double y0 = theta, y1 = a, y2 = s->trunk_yaw[2];
      // kept rolled: unrolling makes the compiler hoist all 20 sample times to the top of the kernel and spill them
#pragma unroll 1
      for (int k = 0; k < K; k++) {
        const double tT = (double)(k + 1) * dt;
        // the test uses the yaw rate as updated by the previous sample, like the reference (:390)
        if (fabs(yT1 - y1) - 0.000001 > 0) {
          y0 = (((1.0 / 4.0 * d * tT + 1.0 / 3.0 * c) * tT) * tT + a) * tT + theta;
          y1 = ((d * tT + c) * tT) * tT + a;
          y2 = (3.0 * d * tT + 2.0 * c) * tT;
        } else y0 += dt * yT1;
        if (out) { out->com_yaw[k][0] = y0; out->com_yaw[k][1] = y1; }
      }
      s->trunk_yaw[0] = y0; s->trunk_yaw[1] = y1; s->trunk_yaw[2] = y2;
    } else if (cs.phase == WG_DS || time + 3.0 / 2.0 * T > cs.time_limit) {
      for (int k = 0; k < K; k++)
        if (out) { out->com_yaw[k][0] = s->trunk_yaw[0]; out->com_yaw[k][1] = s->trunk_yaw[1]; }
    } else {
      for (int k = 0; k < K; k++)
        if (out) { out->com_yaw[k][0] = 0.0; out->com_yaw[k][1] = 0.0; }
    }

  }
  WG_WSYNC();
#ifdef WG_PROFILE
  tke = clock64();
#endif

  // ---- feet: one lane per 5 ms sample (interpolate_feet_positions, OnLineFootTrajectoryGeneration.cpp:235-346) ----
  {
    const Sup cs = *L.sup0;
    const double dt = m.Tctrl;
    const wg_foot_sample_t zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int k = lane + 1;                                // sample index 1..20
    const bool mine = lane < K;
    wg_foot_sample_t outl = zero, outr = zero;             // this lane's pair of samples
    wg_foot_sample_t backl = s->lf[2], backr = s->rf[2];   // the queue's newest sample (rewritten in DS)
    if (cs.phase == WG_SS && time + 3.0 / 2.0 * T < cs.time_limit) {
      double FPx, FPy;                                     // interpret_solution :202-232
      {
        const double sign = (cs.foot == WG_LEFT) ? 1.0 : -1.0;
        if (cs.nb_steps_left > 0 && ns > 0) { FPx = q.x[2 * N]; FPy = q.x[2 * N + ns]; }
        else {
          FPx = cs.x + sign * wg_sin(cs.yaw) * m.feet_distance;
          FPy = cs.y - sign * wg_cos(cs.yaw) * m.feet_distance;
        }
      }
      const double local_t = time - (cs.time_limit - (m.t_double + m.t_single));
      const bool left_support = (cs.foot == WG_LEFT);
      const double unlocked = m.t_single * 0.9;
      const double end_lift = (m.t_single - unlocked) * 0.5;
      double swing_passed = 0.0;
      if (local_t > end_lift) swing_passed = local_t - end_lift;
      // pointers, not struct copies: a selected struct value becomes a 144-byte stack object (scratch)
      const wg_foot_sample_t &last = left_support ? s->rf[2] : s->lf[2];      // swing foot, queue back
      const wg_foot_sample_t &sw_prev = left_support ? s->rf[1] : s->lf[1];   // swing foot, [StartIndex-1]
      const wg_foot_sample_t &st_prev = left_support ? s->lf[1] : s->rf[1];   // stance foot, [StartIndex-1]
      const double ti = unlocked - swing_passed;
      double px5[6], py5[6], pth[4], pom[4], pom2[4], pz[5];
      poly5_set(px5, ti, FPx, last.x, last.dx, last.ddx);
      poly5_set(py5, ti, FPy, last.y, last.dy, last.ddy);
      if (cs.state_changed) poly4_set(pz, m.t_single, m.step_height);
      else { for (int e = 0; e < 5; ++e) pz[e] = s->poly_z[e]; }
      poly3_set(pth, ti, L.sup_angles[0] * 180.0 / kPi, last.theta, last.dtheta);
      poly3_set(pom, ti, 0.0 * 180.0 / kPi, last.omega, last.domega);
      poly3_set(pom2, ti, 2 * 0.0 * 180.0 / kPi, last.omega2, last.domega2);
      const double start_landing = end_lift + unlocked;
      const double omega_cmd = 0.0;
      // UpdateFootPosition :50-199.  x, y, theta are frozen before lift-off and after landing ("= previous
      // sample"): samples before the moving window repeat the queue back, samples after it repeat the last
      // moving sample -- so the samples are independent and each lane evaluates its own.
      const double it = (double)k * dt;
      const bool frozen = (local_t + it <= end_lift) || (local_t + it >= start_landing);
      wg_foot_sample_t c = zero;
      {
        const double tt = (local_t < end_lift && local_t + it > end_lift) ? local_t + it - end_lift : it;
        c.x = poly_eval(px5, 5, tt); c.dx = poly_d1(px5, 5, tt); c.ddx = poly_d2(px5, 5, tt);
        c.y = poly_eval(py5, 5, tt); c.dy = poly_d1(py5, 5, tt); c.ddy = poly_d2(py5, 5, tt);
        c.theta = poly_eval(pth, 3, tt); c.dtheta = poly_d1(pth, 3, tt);
      }
      // last moving sample at or before this one (lanes are in time order)
      const unsigned long long moving = __ballot(mine && !frozen);
      const unsigned long long below = moving & ((2ull << lane) - 1ull);          // lanes <= mine
      const int src = below ? 63 - __builtin_clzll(below) : -1;
      const double hx = __shfl(c.x, src < 0 ? 0 : src), hy = __shfl(c.y, src < 0 ? 0 : src), hth = __shfl(c.theta, src < 0 ? 0 : src);
      if (frozen) {
        c = zero;
        c.x = (src < 0) ? last.x : hx; c.y = (src < 0) ? last.y : hy; c.theta = (src < 0) ? last.theta : hth;
      }
      c.z = poly_eval(pz, 4, local_t + it);
      c.dz = poly_d1(pz, 4, local_t + it);
      if (local_t + it < end_lift) {
        c.omega = poly_eval(pom, 3, it); c.domega = poly_d1(pom, 3, it);
      } else if (local_t + it < start_landing) {
        c.omega = omega_cmd - poly_eval(pom2, 3, local_t + it - end_lift) - sw_prev.omega2;
      } else {
        c.omega = poly_eval(pom, 3, local_t + it - start_landing) + sw_prev.omega - omega_cmd;
      }
      {
        // :150-198 keeps the sole above the floor while it pitches; omega == 0 on this path
        // (":omega 0.0"), which makes every term exactly 0 for any ankle geometry
        const double lOmega = c.omega * kPi / 180.0, lTheta = c.theta * kPi / 180.0;
        const double cth = wg_cos(lTheta), sth = wg_sin(lTheta);
        const double Bf = 0.0, Hf = 0.105, Ff = 0.105;
        double dX, dFZ;
        if (lOmega < 0) { dX = -(Bf - Bf * wg_cos(-lOmega) + Hf * wg_sin(-lOmega)); dFZ = Hf * wg_cos(-lOmega) + Bf * wg_sin(-lOmega) - Hf; }
        else { dX = (Ff - Ff * wg_cos(lOmega) + Hf * wg_sin(lOmega)); dFZ = Hf * wg_cos(lOmega) + Ff * wg_sin(lOmega) - Hf; }
        c.x += cth * dX; c.y += sth * dX; c.z += dFZ;
      }
      {
        const wg_foot_sample_t stance = st_prev;
        foot_select(outl, left_support, stance, c);
        foot_select(outr, left_support, c, stance);
      }
      if (lane == 0 && cs.state_changed) for (int e = 0; e < 5; ++e) s->poly_z[e] = pz[e];
    } else if (cs.phase == WG_DS || time + 3.0 / 2.0 * T > cs.time_limit) {
      outl = s->lf[1]; outr = s->rf[1];                    // k = 0 rewrites the queue back (:333-336)
      backl = outl; backr = outr;
    }
    WG_WSYNC();
#ifdef WG_PROFILE
    tkf = clock64();
#endif
    if (mine && out) { out->lf[lane] = outl; out->rf[lane] = outr; }
    if (lane == 0 && out) { out->lf_back = backl; out->rf_back = backr; }
    // the tail that rounds the struct up to whole cache lines: written (zeros) so that the last line leaves the L2 whole and the
    // struct's bytes do not depend on what the caller's buffer held
    if (out && lane >= 32 && lane < 32 + (int)(sizeof(out->pad_) / sizeof(double))) out->pad_[lane - 32] = 0.0;
    if (lane == 11) { s->lf[0] = outl; s->rf[0] = outr; }
    if (lane == 18) { s->lf[1] = outl; s->rf[1] = outr; }
    if (lane == 19) { s->lf[2] = outl; s->rf[2] = outr; }
    if (lane == 0) {
      if (!s->ending_phase) s->time_to_stop = s->upper_time_limit + T * N;     // :446-450
      s->upper_time_limit = s->upper_time_limit + T;
      s->tick_count++;
    }
  }
  WG_WSYNC();
#ifdef WG_PROFILE
  tkg = clock64();
#endif

  // ---- state: LDS -> HBM ----
  {
    double *dst = reinterpret_cast<double *>(gstate);
    const double *src = reinterpret_cast<const double *>(s);
    for (int i = lane; i < (int)(sizeof(wg_gait_state_t) / 8); i += 64) dst[i] = src[i];
  }
#ifdef WG_PROFILE
  if (lane == 0) {
    atomicAdd(&g_prof[21], tk1 - tk0);                 // state load + lane-0 scalar part
    atomicAdd(&g_prof[22], tk2 - tk1);                 // QP assembly
    atomicAdd(&g_prof[23], clock64() - tk3);           // post-processing + state store
    atomicAdd(&g_prof[35], tka - tk0);                 // of 21: state HBM -> LDS
    atomicAdd(&g_prof[36], tkb - tka);                 //        lane 0: support FSM + preview of the support states
    atomicAdd(&g_prof[37], tkc - tkb);                 //        lane 0: orientation preview
    atomicAdd(&g_prof[38], tk1 - tkc);                 //        one instant per lane: selection, rotated references, hull edges
    atomicAdd(&g_prof[39], tkd - tk3);                 // of 23: state fetched back, jerk, CoM samples, LIPM step
    atomicAdd(&g_prof[40], tke - tkd);                 //        lane 0: trunk
    atomicAdd(&g_prof[41], tkf - tke);                 //        feet: polynomials, one lane per sample
    atomicAdd(&g_prof[42], tkg - tkf);                 //        samples into the state's queue (LDS)
  }
#endif
  TickDiag dg;
  dg.ifail = qr.ifail; dg.n_iter = qr.n_iter; dg.nact = qr.nact; dg.n = n; dg.m = mq; dg.ns = ns;
  return dg;
}

}  // namespace wg
