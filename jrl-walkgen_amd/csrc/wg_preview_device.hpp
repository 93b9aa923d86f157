// wg_preview_device.hpp -- Kajita stage-1 preview control, batched over independent gaits.
//
// Device-side replacement for L consecutive calls of
//   PreviewControl::OneIterationOfPreview      src/PreviewControl/PreviewControl.cpp:324-374
//   (and OneIterationOfPreview1D               :376-420, the same arithmetic on one axis)
// with the cart-table matrices and gains the class holds after ComputeOptimalWeights (:198-322) or
// ReadPrecomputedFile (:134-194).
//
// One step of the reference, per axis:   u  = -(Kx x) + Ks s + sum_{i<Nl} F[i] zmp[l+i]      (sum ascending in i, on top of the
//                                                                                            state term)
//                                        x  = A x + u B;   p = C x;   s += zmp[l] - p         (Simulation only)
// The 2 Nl multiply-adds of the window dominate; the additions form one dependent chain that starts from a state-dependent
// value, so within a gait nothing can be reordered without changing bits.  The parallel axis is the batch:
// one lane = one (gait, axis) pair, 64 independent chains per wave, ZMP references stored time-major so that the
// window reads of a wave are single 512-byte rows; F[i] is wave-uniform (scalar loads).
#pragma once
#include <hip/hip_runtime.h>

namespace wg {

struct PreviewConst {
  double A01, A02, A12;   // m_A: [[1, T, T*T/2], [0, 1, T], [0, 0, 1]]
  double B0, B1, B2;      // m_B: [T*T*T/6, T*T/2, T]
  double C2;              // m_C: [1, 0, -zc/9.81]
  double Kx0, Kx1, Kx2, Ks;
  int nl;
};

// zx, zy:  [(L + nl - 1)][B]   time-major ZMP references (entry t of gait g at t*B + g)
// state:   [B][8]              x[3], y[3], sxzmp, syzmp  (read, advanced L steps, written back)
// com:     [L][6][B] or NULL   x[0..2], y[0..2] after each step
// zmp2:    [L][2][B] or NULL   zmpx2, zmpy2 of each step
__global__ void __launch_bounds__(256)
wg_preview_kernel(int B, int L, PreviewConst K, const double *__restrict__ F, const double *__restrict__ zx,
                  const double *__restrict__ zy, double *__restrict__ state, double *__restrict__ com,
                  double *__restrict__ zmp2, int simulation) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int axis = blockIdx.y;
  if (g >= B) return;
  const double *__restrict__ z = axis ? zy : zx;
  double *st = state + (size_t)g * 8;
  double x0 = st[3 * axis], x1 = st[3 * axis + 1], x2 = st[3 * axis + 2], s = st[6 + axis];
  const size_t sB = (size_t)B;
  const int nl = K.nl;
  for (int l = 0; l < L; ++l) {
    // r = MAL_RET_A_by_B(m_Kx, x): 1x3 by 3x1, products summed k ascending from 0.0 (:338)
    double r = 0.0;
    r += K.Kx0 * x0; r += K.Kx1 * x1; r += K.Kx2 * x2;
    double u = -r + K.Ks * s;                                       // :339
    const double *zl = z + (size_t)l * sB + g;
    int i = 0;
    for (; i + 8 <= nl; i += 8) {                                   // :346-347, the loads run ahead of the add chain
      const double z0 = zl[(size_t)(i + 0) * sB], z1 = zl[(size_t)(i + 1) * sB], z2 = zl[(size_t)(i + 2) * sB],
                   z3 = zl[(size_t)(i + 3) * sB], z4 = zl[(size_t)(i + 4) * sB], z5 = zl[(size_t)(i + 5) * sB],
                   z6 = zl[(size_t)(i + 6) * sB], z7 = zl[(size_t)(i + 7) * sB];
      u += F[i + 0] * z0; u += F[i + 1] * z1; u += F[i + 2] * z2; u += F[i + 3] * z3;
      u += F[i + 4] * z4; u += F[i + 5] * z5; u += F[i + 6] * z6; u += F[i + 7] * z7;
    }
    for (; i < nl; ++i) u += F[i] * zl[(size_t)i * sB];
    // x = MAL_RET_A_by_B(m_A, x) + ux * m_B  (:355): every product of the 3x3 by 3x1 prod is formed, zeros included
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    a0 += 1.0 * x0; a0 += K.A01 * x1; a0 += K.A02 * x2;
    a1 += 0.0 * x0; a1 += 1.0 * x1;   a1 += K.A12 * x2;
    a2 += 0.0 * x0; a2 += 0.0 * x1;   a2 += 1.0 * x2;
    x0 = a0 + u * K.B0; x1 = a1 + u * K.B1; x2 = a2 + u * K.B2;
    double p = 0.0;                                                  // :358-360
    p += 1.0 * x0; p += 0.0 * x1; p += K.C2 * x2;
    if (simulation) s += (zl[0] - p);                                // :367-368
    if (com) {
      double *c = com + ((size_t)l * 6 + 3 * axis) * sB + g;
      c[0] = x0; c[sB] = x1; c[2 * sB] = x2;
    }
    if (zmp2) zmp2[((size_t)l * 2 + axis) * sB + g] = p;
  }
  st[3 * axis] = x0; st[3 * axis + 1] = x1; st[3 * axis + 2] = x2; st[6 + axis] = s;
}

// The same iteration with the preview window staged in LDS.  wg_preview_kernel re-reads the whole window from L2 at every
// step (nl x 512 B per wave-step; measured: the L2 rate, not the ALUs, sets its speed).  Consecutive steps share all but one
// sample, so this kernel keeps the most recent R samples of its 64 gait-axes in an LDS ring (R x 512 B, up to 144 KB: one
// wave per CU) and fetches one new row per step; taps beyond R (none when nl <= R) still come from L2.  The taps are
// visited in the same order (i ascending: the ring is read from the current slot to the end, then from the start), so
// the sums are the same bits.
__global__ void __launch_bounds__(64)
wg_preview_ring_kernel(int B, int L, PreviewConst K, int R, const double *__restrict__ F, const double *__restrict__ zx,
                       const double *__restrict__ zy, double *__restrict__ state, double *__restrict__ com,
                       double *__restrict__ zmp2, int simulation) {
  extern __shared__ __attribute__((aligned(16))) double wg_ring[];      // [R][64]
  const int lane = threadIdx.x;
  const int g0 = blockIdx.x * 64 + lane;
  const bool valid = g0 < B;
  const int g = valid ? g0 : B - 1;                                     // surplus lanes shadow the last gait (no stores)
  const int axis = blockIdx.y;
  const double *__restrict__ z = axis ? zy : zx;
  const size_t sB = (size_t)B;
  const int nl = K.nl;
  const int Lz = L + nl - 1;
  for (int t = 0; t < R; ++t) wg_ring[t * 64 + lane] = (t < Lz) ? z[(size_t)t * sB + g] : 0.0;
  double *st = state + (size_t)g * 8;
  double x0 = st[3 * axis], x1 = st[3 * axis + 1], x2 = st[3 * axis + 2], s = st[6 + axis];
  int k = 0;                                                            // ring slot of sample l
  for (int l = 0; l < L; ++l) {
    const int tn = l + R;                                               // the sample that replaces slot k after this step
    const double znew = (tn < Lz) ? z[(size_t)tn * sB + g] : 0.0;       // issued early: needed at the end of the step
    double r = 0.0;
    r += K.Kx0 * x0; r += K.Kx1 * x1; r += K.Kx2 * x2;
    double u = -r + K.Ks * s;
    const double *zl = z + (size_t)l * sB + g;
    const double z_l = wg_ring[k * 64 + lane];
    int i = 0;
    const int nring = nl < R ? nl : R;                                  // taps served by the ring
    const int first = (R - k) < nring ? (R - k) : nring;                // taps in slots k .. R-1
    {
      const double *rp = wg_ring + k * 64 + lane;
      int c = 0;
      for (; c + 8 <= first; c += 8, i += 8, rp += 8 * 64) {
        const double a0 = rp[0], a1 = rp[64], a2 = rp[128], a3 = rp[192], a4 = rp[256], a5 = rp[320], a6 = rp[384], a7 = rp[448];
        u += F[i + 0] * a0; u += F[i + 1] * a1; u += F[i + 2] * a2; u += F[i + 3] * a3;
        u += F[i + 4] * a4; u += F[i + 5] * a5; u += F[i + 6] * a6; u += F[i + 7] * a7;
      }
      for (; c < first; ++c, ++i, rp += 64) u += F[i] * rp[0];
    }
    {
      const double *rp = wg_ring + lane;                                // wrapped part: slots 0 .. k-1
      const int second = nring - first;
      int c = 0;
      for (; c + 8 <= second; c += 8, i += 8, rp += 8 * 64) {
        const double a0 = rp[0], a1 = rp[64], a2 = rp[128], a3 = rp[192], a4 = rp[256], a5 = rp[320], a6 = rp[384], a7 = rp[448];
        u += F[i + 0] * a0; u += F[i + 1] * a1; u += F[i + 2] * a2; u += F[i + 3] * a3;
        u += F[i + 4] * a4; u += F[i + 5] * a5; u += F[i + 6] * a6; u += F[i + 7] * a7;
      }
      for (; c < second; ++c, ++i, rp += 64) u += F[i] * rp[0];
    }
    for (; i < nl; ++i) u += F[i] * zl[(size_t)i * sB];                 // taps beyond the ring (nl > R)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    a0 += 1.0 * x0; a0 += K.A01 * x1; a0 += K.A02 * x2;
    a1 += 0.0 * x0; a1 += 1.0 * x1;   a1 += K.A12 * x2;
    a2 += 0.0 * x0; a2 += 0.0 * x1;   a2 += 1.0 * x2;
    x0 = a0 + u * K.B0; x1 = a1 + u * K.B1; x2 = a2 + u * K.B2;
    double p = 0.0;
    p += 1.0 * x0; p += 0.0 * x1; p += K.C2 * x2;
    if (simulation) s += (z_l - p);
    if (valid) {
      if (com) {
        double *c = com + ((size_t)l * 6 + 3 * axis) * sB + g;
        c[0] = x0; c[sB] = x1; c[2 * sB] = x2;
      }
      if (zmp2) zmp2[((size_t)l * 2 + axis) * sB + g] = p;
    }
    wg_ring[k * 64 + lane] = znew;
    k = (k + 1 == R) ? 0 : k + 1;
  }
  if (valid) { st[3 * axis] = x0; st[3 * axis + 1] = x1; st[3 * axis + 2] = x2; st[6 + axis] = s; }
}

// The same iteration with each chain split over K lanes.  Per step the reference's sum is ONE chain of nl dependent
// additions (8 cycles each): 2560 cycles at nl = 320 whatever the kernel does -- but the kernels above spend far more than
// that re-reading the window (L2) or keeping a 144 KB ring (one wave per CU).  Here K = 8 consecutive lanes own one
// (gait, axis): lane k keeps taps [kT, (k+1)T) of the window, T = 40 -- its gains F in registers, its samples in an LDS ring
// of T slots per lane (20 KB per wave, so the register budget, not LDS, sets the residency: two waves per SIMD).  Every step
// each lane forms its T products at once (all 64 lanes useful), then the chain runs through the K lanes in order: pass p
// adds lane p's products to the partial sum handed over from lane p-1 (one DPP row shift per pass).  The additions are the
// same additions in the same order, so the bits are the same; the window slides by passing each lane's oldest sample to
// its left neighbour (DPP) and fetching ONE new sample per (gait, axis) and step from memory.
// Eight gait-axes per wave instead of 64: the adds of a pass are useful in one lane of eight, but nothing is re-read.
// Instantiated for T = 16, 24, 32, 40, 48: a window of nl taps uses the smallest T with K T >= nl and ceil(nl / T) lanes of
// each group (T = 40, all eight lanes for the standard 1.6 s / 5 ms window); nl > 384 or < 64 use the kernels above.
template <int CTRL>
__device__ __forceinline__ double pv_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

template <int T, int K, bool FULL>     // FULL: nl is a multiple of T, the last used lane has T taps like the others
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
wg_preview_split_kernel(int B, int L, PreviewConst Kc, const double *__restrict__ F, const double *__restrict__ zx,
                        const double *__restrict__ zy, double *__restrict__ state, double *__restrict__ com,
                        double *__restrict__ zmp2, int simulation) {
  static_assert(K == 8, "the DPP controls below are written for groups of eight lanes");
  extern __shared__ __attribute__((aligned(16))) double pv_ring[];      // [T][64]
  constexpr int G = 64 / K;
  const int lane = threadIdx.x;
  const int k = lane & (K - 1);
  const int g0 = blockIdx.x * G + lane / K;
  const bool valid = g0 < B;
  const int g = valid ? g0 : B - 1;                                     // surplus groups shadow the last gait (no stores)
  const int axis = blockIdx.y;
  const double *__restrict__ z = axis ? zy : zx;
  const size_t sB = (size_t)B;
  const int nl = Kc.nl;
  const int Ku = (nl + T - 1) / T;                                      // lanes of the group that hold taps, 1..K
  const int Tl = nl - (Ku - 1) * T;                                     // taps of the last of them, 1..T
  const int Lz = L + nl - 1;
  const bool last = k == Ku - 1, first = k == 0;
  double Fk[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int i = k * T + t;
    Fk[t] = i < nl ? F[i] : 0.0;
    pv_ring[t * 64 + lane] = (i < nl && i < Lz) ? z[(size_t)i * sB + g] : 0.0;
  }
  const double *st = state + (size_t)g * 8;
  double x0 = st[3 * axis], x1 = st[3 * axis + 1], x2 = st[3 * axis + 2], s = st[6 + axis];
  int head = 0;                                                         // ring slot of each lane's first tap
  for (int l = 0; l < L; ++l) {
    const int tn = l + nl;                                              // the sample that enters the window after this step
    const double znew = (last && tn < Lz) ? z[(size_t)tn * sB + g] : 0.0;
    double prod[T], zold = 0.0;
    {
      // slot of tap t: head + t, wrapped.  Two lane bases (unwrapped / wrapped) and a constant offset per tap, so that
      // a tap costs one select, not an address computation
      const char *b1 = reinterpret_cast<const char *>(pv_ring) + (head * 64 + lane) * 8;
      const char *b2 = b1 - T * 512;
      const int nowrap = T - head;                                      // taps 0 .. nowrap-1 sit at slots head .. T-1
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const char *b = t < nowrap ? b1 : b2;
        const double zt = *reinterpret_cast<const double *>(b + t * 512);
        if (t == 0) zold = zt;
        prod[t] = Fk[t] * zt;
      }
    }
    double r = 0.0;
    r += Kc.Kx0 * x0; r += Kc.Kx1 * x1; r += Kc.Kx2 * x2;
    double acc = -r + Kc.Ks * s;                                        // the chain starts in lane 0 of the group
#pragma unroll
    for (int p = 0; p < K - 1; ++p) {
      if (p < Ku - 1) {                                                 // wave-uniform: lanes p hold T taps each
#pragma unroll
        for (int t = 0; t < T; ++t) acc += prod[t];
        acc = pv_dpp<0x111>(acc);                                       // row_shr:1 -- lane p+1 takes over
      }
    }
    if (FULL) {
#pragma unroll
      for (int t = 0; t < T; ++t) acc += prod[t];
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (t < Tl) acc += prod[t];
    }
    double u = acc;                                                     // the sum sits in lane Ku-1: back to lane 0 (row_shl)
    switch (Ku) {
      case 8: u = pv_dpp<0x107>(acc); break;
      case 7: u = pv_dpp<0x106>(acc); break;
      case 6: u = pv_dpp<0x105>(acc); break;
      case 5: u = pv_dpp<0x104>(acc); break;
      case 4: u = pv_dpp<0x103>(acc); break;
      case 3: u = pv_dpp<0x102>(acc); break;
      case 2: u = pv_dpp<0x101>(acc); break;
      default: break;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    a0 += 1.0 * x0; a0 += Kc.A01 * x1; a0 += Kc.A02 * x2;
    a1 += 0.0 * x0; a1 += 1.0 * x1;    a1 += Kc.A12 * x2;
    a2 += 0.0 * x0; a2 += 0.0 * x1;    a2 += 1.0 * x2;
    const double n0 = a0 + u * Kc.B0, n1 = a1 + u * Kc.B1, n2 = a2 + u * Kc.B2;
    double pz = 0.0;
    pz += 1.0 * n0; pz += 0.0 * n1; pz += Kc.C2 * n2;
    if (first) {                                                        // only lane 0 carries the gait's state
      x0 = n0; x1 = n1; x2 = n2;
      if (simulation) s += (zold - pz);
      if (valid) {
        if (com) {
          double *c = com + ((size_t)l * 6 + 3 * axis) * sB + g;
          c[0] = x0; c[sB] = x1; c[2 * sB] = x2;
        }
        if (zmp2) zmp2[((size_t)l * 2 + axis) * sB + g] = pz;
      }
    }
    // the window slides: every lane hands its oldest sample to the left, the last lane takes the new one
    double incoming = pv_dpp<0x101>(zold);                              // row_shl:1 -- lane k takes lane k+1's oldest
    int wslot = head;
    if (last) {
      incoming = znew;
      wslot = head + Tl;
      if (wslot >= T) wslot -= T;
    }
    pv_ring[wslot * 64 + lane] = incoming;
    head = head + 1 == T ? 0 : head + 1;
  }
  if (valid && first) {
    double *so = state + (size_t)g * 8;
    so[3 * axis] = x0; so[3 * axis + 1] = x1; so[3 * axis + 2] = x2; so[6 + axis] = s;
  }
}

// [B][cols] (gait-major) <-> [cols][B] (time-major) on the device, for the host-pointer entry point
__global__ void wg_transpose_kernel(int rows, int cols, const double *__restrict__ in, double *__restrict__ out) {
  __shared__ double tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = in[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[threadIdx.x][j];
  }
}

}  // namespace wg
