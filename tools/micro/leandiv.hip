// Does the unscaled Newton division (rcp + two refinements + residual correction) give the correctly rounded quotient that
// the compiler's full sequence (v_div_scale / v_div_fmas / v_div_fixup) gives, for |a| <= |b| in a moderate exponent range?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline double lean_div(double a, double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double q = a * y;
  const double r = __builtin_fma(-b, q, a);
  return __builtin_fma(r, y, q);
}
__device__ inline uint64_t rng(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__global__ void k(unsigned long long *bad, unsigned long long *first, int iters, int expspan, int general) {
  uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long nb = 0;
  for (int i = 0; i < iters; ++i) {
    // random mantissas, exponents of b in [-expspan, expspan], a = b * (random in [-1,1]) re-randomised in the low bits
    const uint64_t m1 = rng(s) & 0xFFFFFFFFFFFFFull, m2 = rng(s) & 0xFFFFFFFFFFFFFull;
    const int eb = (int)(rng(s) % (2 * expspan + 1)) - expspan;
    const int ea = general ? eb + (int)(rng(s) % 121) - 60 : eb - (int)(rng(s) % 60);
    double b = __longlong_as_double((long long)(((uint64_t)(1023 + eb) << 52) | m1));
    double a = __longlong_as_double((long long)(((uint64_t)(1023 + ea) << 52) | m2));
    if (rng(s) & 1) a = -a;
    if (rng(s) & 1) b = -b;
    if (!general && fabs(a) > fabs(b)) { const double t = a; a = b; b = t; }
    const double q1 = a / b, q2 = lean_div(a, b);
    if (__double_as_longlong(q1) != __double_as_longlong(q2)) { if (!nb) { first[0] = __double_as_longlong(a); first[1] = __double_as_longlong(b); } ++nb; }
  }
  if (nb) atomicAdd(bad, nb);
}
int main() {
  unsigned long long *bad, *first, h = 0, hf[2] = {0, 0};
  hipMalloc(&bad, 8); hipMalloc(&first, 16);
  for (int general = 0; general < 2; ++general)
  for (int span : {8, 100, 500, 900}) {
    hipMemset(bad, 0, 8);
    k<<<1024, 256>>>(bad, first, 4000, span, general);
    hipDeviceSynchronize();
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 16, hipMemcpyDeviceToHost);
    printf("%s exponent span +-%d: %llu mismatches in %.2e divisions (first a=%016llx b=%016llx)\n", general ? "any ratio 2^+-60," : "|a|<=|b|,", span, h, 1024.0 * 256 * 4000, hf[0], hf[1]);
  }
  return 0;
}
