// wg_gramian_device.hpp -- the horizon-condensing Gramian on the matrix cores, batched over models.
//
// Device-side counterpart of GeneratorVelRef::build_invariant_part (src/ZMPRefTrajectoryGeneration/generator-vel-ref.cpp:587-614)
// on the maps of RigidBodySystem::compute_dyn_cjerk (src/PreviewControl/rigid-body-system.cpp:377-452):
//     Q_b = beta * I'I + alpha * Uv' Uv + gamma * Uz' Uz,        Uv, Uz lower-triangular Toeplitz in (T, h)
// -- the only GEMM-shaped work of the Herdt path (3 x 2N^3 flop per model).  The single-model set-up
// (wg_mpc_configure) keeps the host loop with the reference's summation order, because the tick must reproduce the
// reference bit for bit; THIS kernel is for fleets whose gaits do not share one model (their own sampling period /
// CoM height): one wavefront per model, operands generated in registers from (T, h) -- no matrix is read -- and the two
// products run as chains of v_mfma_f64_16x16x4_f64 (or v_mfma_f32_16x16x4_f32).  MFMA accumulates with fused
// multiply-adds in its own order, so the result agrees with the host loop to rounding (tests: 1e-14 / 1e-6 relative),
// not bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace wg {

typedef double gram_d4 __attribute__((ext_vector_type(4)));
typedef float gram_f4 __attribute__((ext_vector_type(4)));

// entries of the velocity / CoP maps, evaluated like rigid-body-system.cpp:404-441 (left to right)
__device__ __forceinline__ double gram_uv(unsigned k, unsigned i, unsigned N, double T) {
  return (i <= k && k < N) ? (2 * (k - i) + 1) * T * T * 0.5 : 0.0;
}
__device__ __forceinline__ double gram_uz(unsigned k, unsigned i, unsigned N, double T, double h) {
  return (i <= k && k < N) ? (1 + 3 * (k - i) + 3 * (k - i) * (k - i)) * T * T * T / 6.0 - T * h / 9.81 : 0.0;
}

// One wave per model.  N <= 32: up to 2 x 2 output tiles of 16 x 16.
// f64: lane l feeds A[row l&15][k l>>4], B[k l>>4][col l&15]; result register r of lane l is C[(l>>4) + 4r][l&15].
// f32: same A/B maps; result register r of lane l is C[4(l>>4) + r][l&15].
template <bool F32>
__global__ void __launch_bounds__(64)
wg_gramian_kernel(int B, int N, const double *__restrict__ Ts, const double *__restrict__ hs, double alpha, double beta,
                  double gamma, double *__restrict__ Qb) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  if (b >= B) return;
  const double T = Ts[b], h = hs[b];
  const unsigned uN = (unsigned)N;
  const int nt = (N + 15) / 16;
  const unsigned r16 = lane & 15, kq = lane >> 4;
  for (int ti = 0; ti < nt; ++ti)
    for (int tj = 0; tj < nt; ++tj) {
      const unsigned i = 16 * ti + r16, j = 16 * tj + r16;    // A row index i (of U'), B column index j
      double pv[4], pz[4];
      if constexpr (F32) {
        gram_f4 av = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f};
        for (unsigned k0 = 0; k0 < uN; k0 += 4) {
          const unsigned k = k0 + kq;
          const bool in = i < uN && j < uN;
          const float a_v = (float)((i < uN) ? gram_uv(k, i, uN, T) : 0.0), b_v = (float)((j < uN) ? gram_uv(k, j, uN, T) : 0.0);
          const float a_z = (float)((i < uN) ? gram_uz(k, i, uN, T, h) : 0.0), b_z = (float)((j < uN) ? gram_uz(k, j, uN, T, h) : 0.0);
          (void)in;
          av = __builtin_amdgcn_mfma_f32_16x16x4f32(a_v, b_v, av, 0, 0, 0);
          az = __builtin_amdgcn_mfma_f32_16x16x4f32(a_z, b_z, az, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) { pv[r] = (double)av[r]; pz[r] = (double)az[r]; }
      } else {
        gram_d4 av = {0.0, 0.0, 0.0, 0.0}, az = {0.0, 0.0, 0.0, 0.0};
        for (unsigned k0 = 0; k0 < uN; k0 += 4) {
          const unsigned k = k0 + kq;
          const double a_v = (i < uN) ? gram_uv(k, i, uN, T) : 0.0, b_v = (j < uN) ? gram_uv(k, j, uN, T) : 0.0;
          const double a_z = (i < uN) ? gram_uz(k, i, uN, T, h) : 0.0, b_z = (j < uN) ? gram_uz(k, j, uN, T, h) : 0.0;
          av = __builtin_amdgcn_mfma_f64_16x16x4f64(a_v, b_v, av, 0, 0, 0);
          az = __builtin_amdgcn_mfma_f64_16x16x4f64(a_z, b_z, az, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) { pv[r] = av[r]; pz[r] = az[r]; }
      }
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + (F32 ? 4 * (int)kq + r : (int)kq + 4 * r), col = 16 * tj + (int)r16;
        if (row < N && col < N) {
          double q = 0.0;                                       // generator-vel-ref.cpp:592-613: += in call order
          q += ((row == col) ? 1.0 : 0.0) * beta;
          q += pv[r] * alpha;
          q += pz[r] * gamma;
          Qb[((size_t)b * N + row) * N + col] = q;
        }
      }
    }
}

}  // namespace wg
