// isolated cost of the Givens sweep variants (one wave per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../jrl-walkgen_amd/csrc/wg_ql_device.hpp"
#define Zm(i, j) q.Z[(i) + (j) * q.ldz]
#define Rp(i, j) q.R[(j) * ((j) + 1) / 2 + (i)]
__device__ __forceinline__ void sweep_flat_timed(const wg::QlView &q, double *s, int nu, int nact, int lane, unsigned long long *tt) {
  using namespace wg;
  unsigned long long t0 = clock64();
  const int n = q.n;
  if (nu - 1 <= nact) return;
  double myP = 0.0, myQ = 0.0, myN = 0.0;
  {
    double cur = s[nu - 1];
    double p = s[nu - 2];
    for (int c = nu - 1; c > nact; --c) {
      const int nx = (c - 2 >= 0) ? c - 2 : 0;
      const double p_next = s[nx];                          // operand of the next rotation, off the chain
      const bool z = (cur == 0.0);
      const double nrmc = givens_norm(p, z ? 1.0 : cur);
      const double nrm = z ? 0.0 : nrmc;
      const bool me = (lane == c);
      myP = me ? p : myP; myQ = me ? cur : myQ; myN = me ? nrm : myN;
      cur = z ? p : nrmc;
      p = p_next;
    }
  }
  unsigned long long t1 = clock64();
  double *gab = q.sc0;                                      // pairs {ga, gb}; sc0 and sc1 are adjacent (2n doubles)
  {
    const bool mine = lane > nact && lane < nu;
    const bool rot = mine && myN != 0.0;
    const double den = rot ? myN : 1.0;
    const double ga = rot ? myP / den : 1.0;
    const double gb = rot ? myQ / den : 0.0;                // gb == 0 marks a skipped rotation (q was 0)
    const int cl = mine ? lane : nu - 1;                    // lanes without a rotation shadow lane nu-1 ... with ITS values
    const double ga_w = mine ? ga : rl(ga, nu - 1), gb_w = mine ? gb : rl(gb, nu - 1);
    gab[2 * cl] = ga_w; gab[2 * cl + 1] = gb_w;
    if (rot) s[lane - 1] = myN;
  }
  WG_WSYNC();
  unsigned long long t2 = clock64();
  {
    const int i = lane < n ? lane : n - 1;                  // surplus lanes shadow row n-1
    const int ldz = q.ldz;
    double *zp = q.Z + i + (nu - 1) * ldz;                  // Z(i, c)
    double carry = zp[0];
    double zl = zp[-ldz];
    double ga = gab[2 * (nu - 1)], gb = gab[2 * (nu - 1) + 1];
    int c1 = nu - 2 > nact ? nu - 2 : nact + 1;             // rotation after the current one (clamped)
    double zl1 = q.Z[i + (c1 - 1) * ldz], ga1 = gab[2 * c1], gb1 = gab[2 * c1 + 1];
    for (int c = nu - 1; c > nact; --c) {
      const int c2 = c - 2 > nact ? c - 2 : nact + 1;       // two rotations ahead (clamped: value unused past the end)
      const double zl2 = q.Z[i + (c2 - 1) * ldz], ga2 = gab[2 * c2], gb2 = gab[2 * c2 + 1];
      const bool skip = (gb == 0.0);
      const double t_r = ga * zl + gb * carry;
      const double z_r = ga * carry - gb * zl;
      zp[0] = skip ? carry : z_r;
      carry = skip ? zl : t_r;
      zp -= ldz;
      zl = zl1; ga = ga1; gb = gb1;
      zl1 = zl2; ga1 = ga2; gb1 = gb2;
    }
    zp[0] = carry;
  }
  WG_WSYNC();
  unsigned long long t3 = clock64();
  tt[0] += t1 - t0; tt[1] += t2 - t1; tt[2] += t3 - t2;
}


__global__ __launch_bounds__(64) void k(unsigned long long *cyc, double *sink, int nact, int variant) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x, n = 36, m = 75;
  wg::QlDims D(n, m, m, false);
  wg::QlView q; q.carve(lds, D, 0);
  for (int e = lane; e < n * q.ldz; e += 64) q.Z[e] = 0.01 * ((e * 37) % 101) - 0.3;
  double *s = q.R + nact * (nact + 1) / 2;
  __syncthreads();
  unsigned long long tot = 0; unsigned long long tt[3] = {0, 0, 0};
  for (int rep = 0; rep < 50; ++rep) {
    for (int i = lane; i < n; i += 64) s[i] = 0.1 * ((i * 13 + rep) % 17) - 0.7;
    __syncthreads();
    unsigned long long t0 = clock64();
    if (variant == 0) wg::sweep(q, s, n, nact, lane); else sweep_flat_timed(q, s, n, nact, lane, tt);
    tot += clock64() - t0;
  }
  if (lane == 0) { cyc[variant] = tot / 50; if (variant) { cyc[2] = tt[0] / 50; cyc[3] = tt[1] / 50; cyc[4] = tt[2] / 50; } }
  sink[lane] = q.Z[lane] + s[nact];
}
int main() {
  unsigned long long *cyc, h[5]; double *sink; hipMalloc(&cyc, 64); hipMalloc(&sink, 512);
  for (int nact : {0, 10, 20}) {
    k<<<1, 64, 48 * 1024>>>(cyc, sink, nact, 0); k<<<1, 64, 48 * 1024>>>(cyc, sink, nact, 1);
    hipDeviceSynchronize(); hipMemcpy(h, cyc, 40, hipMemcpyDeviceToHost);
    printf("nact=%2d rotations=%2d  readlane-sweep %llu cyc  flat %llu cyc (%.0f / rotation)\n", nact, 35 - nact, h[0], h[1], (double)h[1] / (35 - nact)); printf("    phases: norms %llu (%.0f/rot)  coeff %llu  rows %llu (%.0f/rot)\n", h[2], (double)h[2]/(35-nact), h[3], h[4], (double)h[4]/(35-nact));
  }
  return 0;
}
