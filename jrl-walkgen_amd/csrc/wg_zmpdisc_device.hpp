// wg_zmpdisc_device.hpp -- ZMP reference queue and feet trajectories of a step sequence, batched over independent gaits.
//
// Device-side replacement for
//   ZMPDiscretization::GetZMPDiscretization      src/ZMPRefTrajectoryGeneration/ZMPDiscretization.cpp:143-173
//     = InitOnLine :319-513, OnLineAddFoot :573-1020 (per step), EndPhaseOfTheWalking :1129-1300,
//       FilterOutValues :1045-1109 (window of InitializeFilter :240-262), UpdateCurrentSupportFootPosition :515-558
//   FootTrajectoryGenerationStandard::UpdateFootPosition  src/FootTrajectoryGeneration/FootTrajectoryGenerationStandard.cpp:411-566
//   Polynome::Compute / Polynome3,4,5::SetParameters      src/Mathematics/Polynome.cpp:44-53, PolynomeFoot.cpp:41-57, 100-120, 174-195
//
// The reference walks one gait sample by sample: every output of the 11-tap (0.05 s) sin^2 filter is an ordered sum, the
// first outputs of each phase read earlier *filtered* outputs (FilterOutValues' `o + r` indexing), the ZMP ramp of a phase
// starts at the last filtered value of the previous one, and step types 3/4 accumulate.  Nothing inside a gait can be
// reordered without changing bits, so the parallel axis is the batch: one lane = one gait, 64 gaits per wave, every
// output array TIME-MAJOR ([sample][gait]) so that a wave stores 512-byte rows -- the layout wg_preview_kernel reads.
// Per lane the last nwin unfiltered and 2 nwin filtered samples live in LDS rings laid out [slot][axis][lane]
// (lane-contiguous: conflict-free).  Operation order follows the oracle restatement (oracle/zmpdisc_oracle.c) line by
// line; sin / cos come from include/wg_trig.h like in the tick kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/wg_mpc.h"
#ifndef WG_TRIG_FN
#define WG_TRIG_FN __host__ __device__ static inline
#endif
#include "../../include/wg_trig.h"

namespace wg {

#define WG_ZD_WIN_MAX 52                 // (nwin + 2 nwin) x 2 axes x 64 lanes x 8 B = 3 KB x nwin <= 160 KB
#define WG_ZD_PI 3.14159265358979323846

constexpr int kZdWinStd = 11;            // floor(0.05 / T) + 1 taps at the reference's T = 5 ms (ZMPDiscretization.cpp:240-262)
struct ZdConst {
  wg_zmpdisc_model_t M;
  int nwin, pad_;
  double win[WG_ZD_WIN_MAX];
};

struct ZdOut {                           // all optional, all time-major
  double *zx, *zy, *ztheta;              // [lcap][B]
  int *ztype;                            // [lcap][B]
  double *left, *right;                  // [lcap][6][B]
  int *ltype, *rtype;                    // [lcap][B]
};

struct ZdFoot {
  double x, y, z, theta, omega, omega2;
  int type;
};

struct ZdPoly {
  double c[6];
};

__device__ __forceinline__ double zd_poly(const ZdPoly &p, int n, double t) {   // Polynome::Compute
  double r = 0.0, pt = 1.0;
#pragma unroll
  for (int i = 0; i < 6; i++)
    if (i < n) {
      r += p.c[i] * pt;
      pt *= t;
    }
  return r;
}
__device__ __forceinline__ void zd_poly3(ZdPoly &p, double FT, double FP) {
  p.c[0] = 0.0; p.c[1] = 0.0; p.c[4] = 0.0; p.c[5] = 0.0;
  double tmp = FT * FT;
  if (FP == 0.0 || FT == 0.0) {
    p.c[2] = 0.0; p.c[3] = 0.0;
  } else {
    p.c[2] = 3.0 * FP / tmp;
    p.c[3] = -2.0 * FP / (tmp * FT);
  }
}
__device__ __forceinline__ void zd_poly4(ZdPoly &p, double FT, double MP) {
  p.c[0] = 0.0; p.c[1] = 0.0; p.c[5] = 0.0;
  double tmp = FT * FT;
  if (MP == 0.0 || tmp == 0.0) {
    p.c[2] = p.c[3] = p.c[4] = 0.0;
  } else {
    p.c[2] = 16.0 * MP / tmp;
    tmp = tmp * FT;
    p.c[3] = -32.0 * MP / tmp;
    tmp = tmp * FT;
    p.c[4] = 16.0 * MP / tmp;
  }
}
__device__ __forceinline__ void zd_poly5(ZdPoly &p, double FT, double FP) {
  p.c[0] = p.c[1] = p.c[2] = 0.0;
  double tmp = FT * FT * FT;
  if (FP == 0.0 || tmp == 0.0) {
    p.c[3] = p.c[4] = p.c[5] = 0.0;
  } else {
    p.c[3] = 10 * FP / tmp;
    tmp *= FT;
    p.c[4] = -15 * FP / tmp;
    tmp *= FT;
    p.c[5] = 6 * FP / tmp;
  }
}

// host + device: number of samples of a sequence (the arithmetic of InitOnLine :390-391, OnLineAddFoot :636-641,
// EndPhaseOfTheWalking :1147-1148, 1248-1249), or a negative code
__host__ __device__ inline int zd_length(const wg_zmpdisc_model_t &M, const wg_rel_step_t *steps, int n_steps) {
  if (n_steps < 2 || n_steps > WG_ZMPDISC_MAX_STEPS || !(M.T > 0)) return WG_ZMPDISC_BAD_INPUT;
  long long n = (int)(2 * M.preview_time / M.T);
  if (n < 3) return WG_ZMPDISC_BAD_INPUT;
  for (int i = 1; i < n_steps; i++) {
    double d = M.t_double, s = M.t_single;
    if (steps[i].ds_time != 0.0) {
      d = steps[i].ds_time;
      s = steps[i].ss_time;
    }
    const double a = round((d + s) / M.T), a1 = round(d / M.T), a2 = round(s / M.T);
    if (!(a >= 1.0 && a < 1048576.0) || a1 < 1.0 || a1 + a2 > a) return WG_ZMPDISC_BAD_INPUT;
    n += (long long)a;
  }
  const double e = round(M.t_double / (2 * M.T)), r = 3.0 * M.preview_time / M.T;
  if (!(e >= 1.0 && e < 1048576.0) || !(r >= 0.0 && r < 16777216.0)) return WG_ZMPDISC_BAD_INPUT;
  n += (long long)e + (int)r;
  return n > (1 << 24) ? WG_ZMPDISC_BAD_INPUT : (int)n;
}

// per-lane view of the two LDS rings
struct ZdRings {
  double *z, *f;      // base of this lane's column in the unfiltered / filtered ring
  int nwin, nf;       // slots
  __device__ __forceinline__ double &Z(int slot, int axis) { return z[(slot * 2 + axis) * 64]; }
  __device__ __forceinline__ double &F(int slot, int axis) { return f[(slot * 2 + axis) * 64]; }
};

// One phase (InitOnLine's rest, one OnLineAddFoot, or the end phase) is described by how its unfiltered samples are
// generated; the filter / output loop is common.
enum { ZD_INIT = 0, ZD_STEP = 1, ZD_END = 2 };

__global__ void __launch_bounds__(64)
wg_zmpdisc_kernel(ZdConst K, int B, int smax, const wg_rel_step_t *__restrict__ steps, const int *__restrict__ n_steps,
                  const double *__restrict__ init_feet, int lcap, ZdOut O, int *__restrict__ length) {
  extern __shared__ __attribute__((aligned(16))) double zd_lds[];
  const int lane = threadIdx.x;
  const int g = blockIdx.x * 64 + lane;
  if (g >= B) return;
  const wg_zmpdisc_model_t &M = K.M;
  const size_t sB = (size_t)B;
  const wg_rel_step_t *st = steps + (size_t)g * smax;
  const int S = n_steps[g];
  int Ltot = (S <= smax) ? zd_length(M, st, S) : WG_ZMPDISC_BAD_INPUT;
  if (Ltot > lcap) Ltot = WG_ZMPDISC_CAPACITY;
  if (length) length[g] = Ltot;
  if (Ltot < 0) return;

  const bool want_feet = O.left || O.right || O.ltype || O.rtype;   // otherwise only each phase's last sample is evaluated
  ZdRings R;
  R.nwin = K.nwin;
  R.nf = 2 * K.nwin;
  R.z = zd_lds + lane;
  R.f = zd_lds + (size_t)K.nwin * 2 * 64 + lane;

  // ---- gait state -------------------------------------------------------------------------------------------------------
  double s00 = 1.0, s01 = 0.0, s02 = 0.0, s10 = 0.0, s11 = 1.0, s12 = 0.0;   // m_CurrentSupportFootPosition rows 0, 1
  double p02 = 0.0, p12 = 0.0;                                               // translation of m_PrevCurrentSupportFootPosition
  double vpre0, vpre1, ang_support, ang_zmp;
  ZdFoot cl, cr;                                                             // back() of the final feet deques
  cl.x = init_feet[(size_t)g * 6 + 0]; cl.y = init_feet[(size_t)g * 6 + 1]; cl.theta = init_feet[(size_t)g * 6 + 2];
  cr.x = init_feet[(size_t)g * 6 + 3]; cr.y = init_feet[(size_t)g * 6 + 4]; cr.theta = init_feet[(size_t)g * 6 + 5];
  cl.z = cl.omega = cl.omega2 = 0.0; cr.z = cr.omega = cr.omega2 = 0.0;
  cl.type = cr.type = 0;
  wg_rel_step_t rel0 = st[0];
  double bpx = 0.0, bpy = 0.0, btheta = 0.0;        // FinalZMPPositions.back()
  double f0x = 0.0, f0y = 0.0;                      // FinalZMPPositions[0]
  int nz = 0;                                       // FinalZMPPositions.size()
  int fslot = 0;                                    // nz mod nf: where the next filtered sample goes in its ring

  auto bookkeeping = [&](double zmp_theta, int &who) {   // :370-383, :619-634
    if (rel0.sy < 0) {
      who = -1;
      vpre0 = cr.x - cl.x; vpre1 = cr.y - cl.y;
      ang_support = cr.theta - cl.theta;
      ang_zmp = cr.theta - zmp_theta;
    } else {
      who = 1;
      vpre0 = -cr.x + cl.x; vpre1 = -cr.y + cl.y;
      ang_support = cl.theta - cr.theta;
      ang_zmp = cl.theta - zmp_theta;
    }
  };
  auto update_support = [&](const wg_rel_step_t &s) {     // :515-558
    p02 = s02; p12 = s12;
    const double c = wg_cos(s.theta * WG_ZD_PI / 180.0), sn = wg_sin(s.theta * WG_ZD_PI / 180.0);
    double o00 = 0.0, o01 = 0.0, o10 = 0.0, o11 = 0.0;
    o00 += c * s00; o00 += -sn * s10;
    o01 += c * s01; o01 += -sn * s11;
    o10 += sn * s00; o10 += c * s10;
    o11 += sn * s01; o11 += c * s11;
    double v0 = 0.0, v1 = 0.0;
    v0 += o00 * s.sx; v0 += o01 * s.sy;
    v1 += o10 * s.sx; v1 += o11 * s.sy;
    s00 = o00; s01 = o01; s10 = o10; s11 = o11;
    s02 += v0; s12 += v1;
  };
  auto zmp_world = [&](double &w0, double &w1) {
    double t = 0.0;
    t += s00 * M.zmp_neutral[0]; t += s01 * M.zmp_neutral[1]; t += s02 * 1.0;
    w0 = t;
    t = 0.0;
    t += s10 * M.zmp_neutral[0]; t += s11 * M.zmp_neutral[1]; t += s12 * 1.0;
    w1 = t;
  };
  auto put_foot = [&](double *base, int *tbase, size_t l, const ZdFoot &f) {
    if (base) {
      double *p = base + l * 6 * sB + g;
      p[0] = f.x; p[sB] = f.y; p[2 * sB] = f.z; p[3 * sB] = f.theta; p[4 * sB] = f.omega; p[5 * sB] = f.omega2;
    }
    if (tbase) tbase[l * sB + g] = f.type;
  };

  {
    int who;
    bookkeeping((cr.theta + cl.theta) / 2.0, who);
  }

  // ---- phases -----------------------------------------------------------------------------------------------------------
  const int n_phases = S + 1;                       // rest, S - 1 steps, end
  for (int ph = 0; ph < n_phases; ph++) {
    const int kind = ph == 0 ? ZD_INIT : (ph == n_phases - 1 ? ZD_END : ZD_STEP);
    int nZ, n1 = 0, n2 = 0, t1 = 0, who = 1, n_end = 0;
    double px0 = 0, py0 = 0, theta0 = 0, dx = 0, dy = 0, w0 = 0, w1 = 0, mod_sst = 0, fin0 = 0, fin1 = 0;
    int type_ss = 0;
    ZdPoly qx, qy, qz, qth, qom, qom2, qzt;
    ZdFoot dsl = cl, dsr = cr;                      // the feet while nothing moves in this phase
    if (kind == ZD_INIT) {
      nZ = (int)(2 * M.preview_time / M.T);
      fin0 = M.zmp_neutral[0]; fin1 = M.zmp_neutral[1];
      dsl.type = dsr.type = 10;
    } else if (kind == ZD_STEP) {
      const wg_rel_step_t rel1 = st[ph];
      double lTdble = M.t_double, lTsingle = M.t_single;
      if (rel1.ds_time != 0.0) {
        lTdble = rel1.ds_time;
        lTsingle = rel1.ss_time;
      }
      bookkeeping(btheta, who);
      nZ = (int)(unsigned)round((lTdble + lTsingle) / M.T);
      update_support(rel0);
      n1 = (int)(unsigned)round(lTdble / M.T);
      n2 = (int)(unsigned)round(lTsingle / M.T);
      px0 = bpx; py0 = bpy; theta0 = btheta;
      zmp_world(w0, w1);
      dx = (w0 - px0) / (unsigned)n1; dy = (w1 - py0) / (unsigned)n1;
      t1 = rel1.step_type;
      if (t1 == 3) { dx = (s02 + M.zmp_shift[0] - px0) / (unsigned)n1; dy = (s12 - py0) / (unsigned)n1; }
      if (t1 == 4) { dx = (s02 + M.zmp_shift[2] - px0) / (unsigned)n1; dy = (s12 - py0) / (unsigned)n1; }
      if (t1 == 5) {
        dx = (s02 - (M.zmp_shift[0] + M.zmp_shift[2] + M.zmp_shift[1] + M.zmp_shift[3]) - px0) / (unsigned)n1;
        dy = (s12 - py0) / (unsigned)n1;
      }
      dsl.z = 0.0; dsr.z = 0.0;
      dsl.type = dsr.type = t1 + 10;
      // second phase set-up, :770-861
      const double next_theta = rel1.theta;
      const double rel_theta = next_theta + ang_support, rel_zmp_theta = next_theta + ang_zmp;
      const double c = wg_cos(next_theta * WG_ZD_PI / 180.0), s = wg_sin(next_theta * WG_ZD_PI / 180.0);
      double o00 = 0.0, o01 = 0.0, o10 = 0.0, o11 = 0.0;
      o00 += c * s00; o00 += -s * s10;
      o01 += c * s01; o01 += -s * s11;
      o10 += s * s00; o10 += c * s10;
      o11 += s * s01; o11 += c * s11;
      double vd0 = 0.0, vd1 = 0.0;
      vd0 += o00 * rel1.sx; vd0 += o01 * rel1.sy;
      vd1 += o10 * rel1.sx; vd1 += o11 * rel1.sy;
      const double vrel0 = vd0 + vpre0, vrel1 = vd1 + vpre1;
      vpre0 = vd0; vpre1 = vd1;
      mod_sst = lTsingle * M.modulation;
      const double end_lift = (lTsingle - mod_sst) * 0.5;
      zd_poly5(qx, mod_sst, vrel0);
      zd_poly5(qy, mod_sst, vrel1);
      zd_poly4(qz, M.t_single, M.step_height);
      zd_poly3(qth, mod_sst, rel_theta);
      zd_poly3(qom, end_lift, M.omega);
      zd_poly3(qom2, mod_sst, 2 * M.omega);
      zd_poly3(qzt, lTsingle, rel_zmp_theta);
      type_ss = who * rel0.step_type;
      rel0 = rel1;                                  // pop_front
    } else {
      update_support(rel0);                         // m_RelativeFootPositions.size() > 0
      n_end = (int)(unsigned)round(M.t_double / (2 * M.T));
      nZ = n_end + (int)(3.0 * M.preview_time / M.T);
      px0 = bpx; py0 = bpy; theta0 = btheta;
      const double pxf = 0.5 * (s02 + p02), pyf = 0.5 * (s12 + p12);
      dx = (pxf - px0) / (double)(unsigned)n_end; dy = (pyf - py0) / (double)(unsigned)n_end;
      dsl.type = dsr.type = 0;
    }

    // unfiltered sample r of this phase (called with r = 0, 1, 2, ... in order: the chains need the previous value)
    double gx = 0.0, gy = 0.0;                      // last generated
    int slot_top = -1;                              // ring slot of the last generated sample
    auto generate = [&](int r) {
      double x, y;
      if (kind == ZD_INIT) {
        const double coef = (double)r / (double)(unsigned)nZ;
        x = 0.0 + (fin0 - 0.0) * coef;
        y = 0.0 + (fin1 - 0.0) * coef;
      } else if (kind == ZD_STEP) {
        if (r < n1) {
          x = px0 + (unsigned)r * dx;
          y = py0 + (unsigned)r * dy;
        } else if (r < n1 + n2) {
          x = w0; y = w1;
          if (t1 == 3 || t1 == 4) {
            const double px02 = px0 + (unsigned)(n1 - 1) * dx, py02 = py0 + (unsigned)(n1 - 1) * dy;
            const double sh = t1 == 3 ? M.zmp_shift[1] : M.zmp_shift[3];
            const double ddx = (s02 + sh - px02) / (unsigned)n2, ddy = (s12 - py02) / (unsigned)n2;
            x = gx + ddx; y = gy + ddy;
          }
        } else {
          x = 0.0; y = 0.0;                         // value-initialised tail of the reference's deque (rounding mismatch)
        }
      } else {
        if (r == 0) { x = px0 + dx; y = py0 + dy; }
        else if (r < n_end) { x = gx + dx; y = gy + dy; }
        else { x = gx; y = gy; }
      }
      gx = x; gy = y;
      slot_top = slot_top + 1 == R.nwin ? 0 : slot_top + 1;
      R.Z(slot_top, 0) = x; R.Z(slot_top, 1) = y;
    };

    double z2x = 0.0, z2y = 0.0;                    // ZMPPositions[lshift] of the rest phase
    int gen = 0;                                    // samples generated so far
    for (; gen < 2 && gen < nZ; gen++) generate(gen);
    for (int i = 0; i < nZ; i++) {
      if (gen < nZ) {
        generate(gen);
        if (gen == 2) { z2x = gx; z2y = gy; }
        gen++;
      }
      // FilterOutValues, :1051-1105.  slot of unfiltered sample q (q <= gen - 1): slot_top - (gen - 1 - q), wrapped
      double l0 = 0.0, l1 = 0.0;
      const int o = nz - 1 - 2;
      if (i + 3 >= K.nwin && i + 2 < nZ) {          // every tap inside the phase: r = i + 2 - j is sample gen - 1 - j
        if (K.nwin == kZdWinStd) {
          // the standard window (T = 5 ms: 11 taps): all 22 ring reads requested before the first add -- the loop below waits for
          // an LDS round trip per tap, which was most of a sample's 2 800 cycles; the same products added in the same order
          double zx[kZdWinStd], zy[kZdWinStd];
          int slot = slot_top;
#pragma unroll
          for (int j = 0; j < kZdWinStd; j++) {
            zx[j] = R.Z(slot, 0); zy[j] = R.Z(slot, 1);
            slot = slot == 0 ? kZdWinStd - 1 : slot - 1;
          }
#pragma unroll
          for (int j = 0; j < kZdWinStd; j++) { const double wj = K.win[j]; l0 += wj * zx[j]; l1 += wj * zy[j]; }
        } else {
        int slot = slot_top;
        for (int j = 0; j < K.nwin; j++) {
          const double wj = K.win[j];
          l0 += wj * R.Z(slot, 0); l1 += wj * R.Z(slot, 1);
          slot = slot == 0 ? R.nwin - 1 : slot - 1;
        }
        }
      } else {
        for (int j = 0; j < K.nwin; j++) {
          int r = i - j + 2;
          const double wj = K.win[j];
          if (r < 0) {
            if (kind == ZD_INIT) {
              l0 += wj * z2x; l1 += wj * z2y;
            } else if (-r < o) {
              const int q = (o + r) % R.nf;
              l0 += wj * R.F(q, 0); l1 += wj * R.F(q, 1);
            } else {
              l0 += wj * f0x; l1 += wj * f0y;
            }
          } else {
            if (r >= nZ) r = nZ - 1;
            int slot = slot_top - (gen - 1 - r);
            if (slot < 0) slot += R.nwin;
            l0 += wj * R.Z(slot, 0); l1 += wj * R.Z(slot, 1);
          }
        }
      }
      // theta, stepType and the feet of sample i
      double th;
      int ty;
      ZdFoot fl = dsl, fr = dsr;
      if (kind == ZD_INIT) {
        th = 0.0; ty = 0;
      } else if (kind == ZD_END) {
        th = theta0; ty = 0;
      } else if (i < n1) {
        th = theta0; ty = t1 + 10;
      } else if (i < n1 + n2) {
        const int k = i - n1;
        th = zd_poly(qzt, 4, (unsigned)k * M.T) + theta0;
        ty = type_ss;
        if (want_feet || i == nZ - 1) {
        // UpdateFootPosition: local index k + 1 (the last double-support sample is the initial one)
        ZdFoot &sup = who == 1 ? fl : fr;
        ZdFoot &non = who == 1 ? fr : fl;
        sup.type = (-1) * t1;
        const ZdFoot i0 = non;
        const double local = (unsigned)(k + 1) * M.T;
        const double end_lift = (M.t_single - mod_sst) * 0.5, start_land = end_lift + mod_sst;
        non.omega2 = 0.0;
        non.type = t1;
        if (local < end_lift) {
          non.x = i0.x; non.y = i0.y; non.theta = i0.theta;
        } else if (local < start_land) {
          non.x = i0.x + zd_poly(qx, 6, local - end_lift);
          non.y = i0.y + zd_poly(qy, 6, local - end_lift);
          non.theta = i0.theta + zd_poly(qth, 4, local - end_lift);
        } else {
          non.x = i0.x + zd_poly(qx, 6, mod_sst);
          non.y = i0.y + zd_poly(qy, 6, mod_sst);
          non.theta = i0.theta + zd_poly(qth, 4, mod_sst);
        }
        non.z = i0.z + zd_poly(qz, 5, local);
        if (local < end_lift)
          non.omega = zd_poly(qom, 4, local);
        else if (local < start_land)
          non.omega = M.omega - zd_poly(qom2, 4, local - end_lift);
        else
          non.omega = zd_poly(qom, 4, local - start_land) - M.omega;
        const double lo = non.omega * WG_ZD_PI / 180.0, lt = non.theta * WG_ZD_PI / 180.0;
        const double ct = wg_cos(lt), stt = wg_sin(lt);
        double dX, dFZ;
        const double Bq = M.foot_b, H = M.foot_h, F = M.foot_f;
        if (lo < 0) {
          const double X1 = Bq * wg_cos(-lo), X2 = H * wg_sin(-lo), Z1 = H * wg_cos(-lo), Z2 = Bq * wg_sin(-lo);
          dX = -(Bq - X1 + X2);
          dFZ = Z1 + Z2 - H;
        } else {
          const double X1 = F * wg_cos(lo), X2 = H * wg_sin(lo), Z1 = H * wg_cos(lo), Z2 = F * wg_sin(lo);
          dX = (F - X1 + X2);
          dFZ = Z1 + Z2 - H;
        }
        non.x += ct * dX;
        non.y += stt * dX;
        non.z += dFZ;
        }
      } else {                                      // value-initialised tail
        th = 0.0; ty = 0;
        fl.x = fl.y = fl.z = fl.theta = fl.omega = fl.omega2 = 0.0; fl.type = 0;
        fr = fl;
      }
      // push
      const size_t l = (size_t)nz;
      if (O.zx) O.zx[l * sB + g] = l0;
      if (O.zy) O.zy[l * sB + g] = l1;
      if (O.ztheta) O.ztheta[l * sB + g] = th;
      if (O.ztype) O.ztype[l * sB + g] = ty;
      put_foot(O.left, O.ltype, l, fl);
      put_foot(O.right, O.rtype, l, fr);
      R.F(fslot, 0) = l0; R.F(fslot, 1) = l1;        // fslot == nz mod nf, kept by increment (no division per sample)
      fslot = fslot + 1 == R.nf ? 0 : fslot + 1;
      if (nz == 0) { f0x = l0; f0y = l1; }
      bpx = l0; bpy = l1; btheta = th;
      cl = fl; cr = fr;
      nz++;
    }
  }
  // a gait at rest after its last sample: lets one preview launch cover a ragged batch
  for (size_t l = (size_t)nz; l < (size_t)lcap; l++) {
    if (O.zx) O.zx[l * sB + g] = bpx;
    if (O.zy) O.zy[l * sB + g] = bpy;
  }
}

}  // namespace wg
