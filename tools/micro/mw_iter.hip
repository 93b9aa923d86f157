// Prototype of ONE active-set iteration of ql0002 at N = 32 (n = 72, nact = 41) in the layout docs/HISTORY.md 7 only estimated (DESIGN 4.3 has the result): a
// workgroup of W waves per gait, Z (72 x 72, leading dimension 73: conflict-free column walks) in LDS, three gaits per CU
// (53 KB of LDS each), the lane-parallel phases split over the waves, the two serial chains on waves 0 and 1 (side by side
// when W >= 2), a workgroup barrier at every hand-over.  Same instruction shapes as the solver's phases -- the chain of
// rotation norms IS the solver's (wg::givens_norm_fast from csrc/wg_ql_device.hpp), the rest restates their loops on LDS
// operands -- on synthetic, well-conditioned data (rotations are exact Givens pairs, so Z stays orthogonal-ish for any
// number of iterations).  What it leaves out makes it an OPTIMISTIC bound: the step products Z s, the drops (a second sweep
// with rotations of R), the residual refresh, the tick around the solve (the 45 % "everything else" row of
// profiles/round4_phase_attribution_n32.txt).
//
// Phases of an iteration (barrier after each):
//   A  Z^T a        72 columns, one per lane, 72-term ordered sums        (qld.cpp:1421-1470)
//   B  chains       wave 0: 30 rotation norms (:1992-2004); wave min(1, W-1): back substitution of 41 rows (:1824-1851)
//   C  coefficients (ga, gb) of the 30 rotations, one lane each            (:2005-2014)
//   D  rotations    72 rows of Z, one per lane, 30 dependent steps each    (:2015-2029)
//   E  sums         products lane-parallel, then three ordered 72-term sums (:1491-1532), one per wave when W >= 3
//   F  updates      ratio test (wave min), x += step Z(:, nact), lambda, the ordered |x| sum (:1715-1786)
//   G  scan         149 rows, one per lane (W = 1: three passes), 36-term sums; arg-max in the wave, across waves through LDS
// Reported: ns and shader cycles per iteration with every CU holding three workgroups, and what that is in MPC ticks/s at
// 44 iterations per tick -- against the shipped one-wave-per-gait kernel (12 gaits per CU, Z in its L2 slot).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/micro/mw_iter tools/micro/mw_iter.hip && tools/micro/mw_iter
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../jrl-walkgen_amd/csrc/wg_ql_device.hpp"

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

constexpr int kN = 72, kLd = 73, kNact = 41, kRot = kN - 1 - kNact, kM = 149, kScanTerms = 36;
struct Lds {
  double Z[kN * kLd];
  double R[kNact * (kNact + 1) / 2];
  double ww[kN], s[kN], x[kN], chain[kN], ga[kN], gb[kN], lam[kN], bw[kN];   // chain / ga / gb double as the sums' scratch, ga | gb as
                                                                              // the back substitution's two 64-entry product buffers
  double red[8];
};
static_assert(sizeof(Lds) <= 42 * 1280, "three workgroups per CU");

__device__ __forceinline__ double rlane(double v, int src) { return wg::rl(v, src); }

template <int W, int PHASES>   // PHASES: bit mask of the phases that run (timing one phase = all minus all-but-one)
__global__ __launch_bounds__(64 * W) void mw_iter(double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  Lds &L = *reinterpret_cast<Lds *>(raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // ---- synthetic, well-conditioned state
  for (int k = tid; k < kN * kLd; k += 64 * W) { const int i = k % kLd, j = k / kLd; L.Z[k] = (i == j) ? 1.0 : 1e-3 * ((i * 7 + j * 3) % 11 - 5); }
  for (int k = tid; k < kNact * (kNact + 1) / 2; k += 64 * W) L.R[k] = 0.01 * ((k * 5) % 7 - 3);
  for (int k = tid; k < kNact; k += 64 * W) L.R[k * (k + 1) / 2 + k] = 2.0 + 0.01 * k;
  for (int k = tid; k < kN; k += 64 * W) { L.ww[k] = 0.1 + 0.01 * k; L.x[k] = 0.01 * k; L.lam[k] = 1.0; L.bw[k] = 0.0; L.s[k] = 1.0; L.chain[k] = 1.0; L.ga[k] = 1.0; L.gb[k] = 0.0; }
  __syncthreads();
  // row / column owner: item c belongs to wave c % W, lane c / W (W = 1: items 64..71 are a second pass of lanes 0..7)
  const bool own0 = (W == 1) ? lane < 64 : lane < (kN + W - 1 - wave) / W;
  const int item0 = (W == 1) ? lane : lane * W + wave;
  const bool own1 = (W == 1) && lane < kN - 64;
  const int item1 = 64 + lane;
  double keep = 0.0;
  for (int it = 0; it < iters; ++it) {
    // ---- A: Z^T a
    if (PHASES & 1) {
      auto col = [&](int c) {
        const double *z = L.Z + c * kLd;
        double acc = 0.0;
#pragma unroll 1
        for (int j = 0; j < kN; j += 8) {
          double u[8], w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { u[e] = z[j + e]; w[e] = L.ww[j + e]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += u[e] * w[e];
        }
        L.s[c] = acc;
      };
      if (own0) col(item0);
      if (own1) col(item1);
    }
    __syncthreads();
    // ---- B: the two serial chains
    if (PHASES & 2) {
      if (wave == 0 && !(PHASES & 128)) {                    // chain of rotation norms: the solver's own arithmetic and loop shape
        double cur = L.s[kN - 1] + 1.0;                      // (sweep(): pairs of rotations, the next operand fetched ahead of the chain)
        double pa = L.s[kN - 2], pb;
        int iv = kN - 4;
        asm volatile("" : "+v"(iv));
#pragma unroll 1
        for (int k = kRot >> 1; k > 0; --k) {
          pb = L.s[iv + 1];
          cur = wg::givens_norm_fast(pa, cur); L.chain[iv + 2] = cur;
          pa = L.s[iv];
          cur = wg::givens_norm_fast(pb, cur); L.chain[iv + 1] = cur;
          iv -= 2;
        }
        if (kRot & 1) { cur = wg::givens_norm_fast(pa, cur); L.chain[iv + 2] = cur; }
      }
      if (wave == (W > 1 ? 1 : 0) && !(PHASES & 256)) {      // back substitution: the solver's own LDS-pipelined form
        wg::QlView q;
        q.R = L.R; q.ww = L.bw;
        wg::backsub_lds<64>(q, L.s, kNact, lane, L.ga);
        if (lane < kNact) L.lam[lane] = L.lam[lane] * 0.5 + 1e-3 * L.bw[lane];
      }
    }
    __syncthreads();
    // ---- C: (ga, gb) of every rotation, one lane each (wave 0)
    if (PHASES & 4) {
      if (wave == 0 && lane < kRot) {
        const int c = kNact + 1 + lane;
        const double P = L.s[c - 1], Q = (c == kN - 1) ? L.s[kN - 1] + 1.0 : L.chain[c], Nn = L.chain[c - 1];
        L.ga[c] = P / Nn; L.gb[c] = Q / Nn;
      }
    }
    __syncthreads();
    // ---- D: the rotations through the rows of Z
    if (PHASES & 8) {
      auto row = [&](int i) {
        double *z = L.Z + i;
        double carry = z[(kN - 1) * kLd];
#pragma unroll 1
        for (int c = kN - 1; c > kNact; c -= 6) {                                // kRot = 30 = 5 groups of 6
          double zl[6], ga[6], gb[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) { zl[k] = z[(c - 1 - k) * kLd]; ga[k] = L.ga[c - k]; gb[k] = L.gb[c - k]; }
#pragma unroll
          for (int k = 0; k < 6; ++k) { const double t = ga[k] * zl[k] + gb[k] * carry; z[(c - k) * kLd] = ga[k] * carry - gb[k] * zl[k]; carry = t; }
        }
        z[kNact * kLd] = carry;
      };
      if (own0) row(item0);
      if (own1) row(item1);
    }
    __syncthreads();
    // ---- E: the three dependence sums
    if (PHASES & 16) {
      auto prod = [&](int i) { const double zi = L.Z[i + kNact * kLd], wi = L.ww[i]; L.chain[i] = wi * zi; L.ga[i] = fabs(wi * zi); L.gb[i] = zi * zi; };
      if (own0) prod(item0);
      if (own1) prod(item1);
      __syncthreads();
      const int which = (W >= 3) ? wave : lane;               // W >= 3: one sum per wave (lane 0); else lanes 0..2 of wave 0
      if ((W >= 3) ? (wave < 3 && lane == 0) : (wave == 0 && lane < 3)) {
        const double *src = which == 0 ? L.chain : (which == 1 ? L.ga : L.gb);
        double acc = 0.0;
#pragma unroll 1
        for (int i = 0; i < kN; i += 12) {
          double t[12];
#pragma unroll
          for (int e = 0; e < 12; ++e) t[e] = src[i + e];
#pragma unroll
          for (int e = 0; e < 12; ++e) acc += t[e];
        }
        L.red[which] = acc;
      }
    }
    __syncthreads();
    // ---- F: ratio test, x and lambda updates, |x| sum
    if (PHASES & 32) {
      const double sumc = L.red[2];
      if (wave == 0) {
        double r = (lane < kNact && L.lam[lane] > 0.0) ? L.lam[lane] : 1e30;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r = fmin(r, __shfl_xor(r, o));
        if (lane < kNact) L.lam[lane] = fmax(0.0, L.lam[lane] - 1e-6 * r);
      }
      const double step = 1e-9 * sumc;
      if (own0) L.x[item0] += step * L.Z[item0 + kNact * kLd];
      if (own1) L.x[item1] += step * L.Z[item1 + kNact * kLd];
      __syncthreads();
      if (wave == W - 1 && lane == 0) {
        double acc = 0.0;
#pragma unroll 1
        for (int i = 0; i < kN; i += 12) {
          double t[12];
#pragma unroll
          for (int e = 0; e < 12; ++e) t[e] = fabs(L.x[i + e]);
#pragma unroll
          for (int e = 0; e < 12; ++e) acc += t[e];
        }
        L.red[3] = acc;
      }
    }
    __syncthreads();
    // ---- G: violation scan
    if (PHASES & 64) {
      double best = -1.0; int bidx = -1;
      for (int k0 = 0; k0 < kM; k0 += 64 * W) {
        const int k = k0 + tid;
        const double cf = 1e-3 * ((k % 13) - 6);
        double sum = -0.5, tmp = 0.5;
#pragma unroll 1
        for (int j = 0; j < kScanTerms; j += 12) {
          double xs[12];
#pragma unroll
          for (int e = 0; e < 12; ++e) xs[e] = L.x[j + e];
#pragma unroll
          for (int e = 0; e < 12; ++e) { const double t = cf * xs[e]; sum += t; tmp += fabs(t); }
        }
        const double v = (k < kM) ? -sum * (1.0 + 1e-3 * tmp) : -1.0;
        if (v > best) { best = v; bidx = k; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
      }
      if (W > 1) {
        if (lane == 0) { L.red[4 + wave] = best; }
        __syncthreads();
        double b2 = L.red[4];
#pragma unroll
        for (int w = 1; w < W; ++w) b2 = fmax(b2, L.red[4 + w]);
        best = b2;
      }
      keep += best;
      if (tid == 0) L.ww[it % kN] = 0.1 + 1e-3 * (bidx % 7);    // the "new normal" of the next iteration depends on the scan
    }
    __syncthreads();
  }
  if (tid < kN) out[(size_t)blockIdx.x * kN + tid] = L.x[tid] + L.s[tid] + keep + L.Z[tid * kLd + tid] + L.lam[tid] + L.red[3];
}

template <int W, int PH>
double run(int grid, int iters, double *out) {
  const size_t lds = sizeof(Lds);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mw_iter<W, PH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  mw_iter<W, PH><<<grid, 64 * W, lds>>>(out, 8);
  CHECK(hipEventRecord(e0));
  mw_iter<W, PH><<<grid, 64 * W, lds>>>(out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return 1e6 * ms / iters;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, grid = 3 * cus, iters = 3000;
  const double ghz = prop.clockRate * 1e-6;
  double *out;
  CHECK(hipMalloc(&out, (size_t)grid * kN * 8));
  printf("# %s, %d CUs, %.2f GHz; %d workgroups (three per CU: %zu B of LDS each), %d iterations each\n", prop.name, cus, ghz, grid, sizeof(Lds), iters);
  printf("# n = %d, nact = %d, %d rotations, %d scan rows; per iteration of ONE gait (three gaits per CU run side by side)\n", kN, kNact, kRot, kM);
  const char *names[7] = {"A  Z^T a", "B  chains (norms | back substitution)", "C  rotation coefficients", "D  rotations of Z's rows",
                          "E  dependence sums", "F  ratio test, x / lambda, |x| sum", "G  violation scan + arg-max"};
#define ALL(W) do { \
    const double full = run<W, 127>(grid, iters, out), none = run<W, 0>(grid, iters, out); \
    const double ph[7] = {run<W, 127 - 1>(grid, iters, out), run<W, 127 - 2>(grid, iters, out), run<W, 127 - 4>(grid, iters, out), \
                          run<W, 127 - 8>(grid, iters, out), run<W, 127 - 16>(grid, iters, out), run<W, 127 - 32>(grid, iters, out), \
                          run<W, 127 - 64>(grid, iters, out)}; \
    printf("W = %d waves per gait: %9.1f ns = %7.0f cycles per iteration; barriers alone (no phase): %7.1f ns = %5.0f cycles (%d barriers)\n", \
           W, full, full * ghz, none, none * ghz, W > 1 ? 11 : 10); \
    for (int p = 0; p < 7; ++p) printf("      without phase %-40s %9.1f ns -> the phase costs %7.0f cycles\n", names[p], ph[p], (full - ph[p]) * ghz); \
    { const double nb = ph[1], cn = run<W, 127 + 256>(grid, iters, out), bs = run<W, 127 + 128>(grid, iters, out); \
      printf("      phase B taken apart: the chain of norms alone %7.0f cycles, the back substitution alone %7.0f cycles, both %7.0f\n", \
             (cn - nb) * ghz, (bs - nb) * ghz, (full - nb) * ghz); } \
    printf("      => %.3f M ticks/s at 44 iterations per tick with %d gaits resident (optimistic: see the header)\n", grid / (44.0 * full * 1e-9) * 1e-6, grid); \
  } while (0)
  ALL(1);
  ALL(2);
  ALL(4);
  const double slot = cus / (44.0 * 1.32e6) * ghz * 1e9;          // cycles per gait-iteration and per resident gait of a CU
  printf("# shipped wg_mpc_run_xcd_kernel<32>: 12 gaits per CU at 1.31 - 1.33 M ticks/s = an iteration of a gait every %.0f cycles, %.0f per resident gait;\n"
         "# three gaits per CU must finish an iteration in %.0f cycles to match it, in %.0f for 1.6 M ticks/s\n",
         12.0 * slot, slot, 3.0 * slot, 3.0 * slot * 1.32 / 1.6);
  return 0;
}
