// fastexp.hpp -- fp64 exp for non-positive arguments on gfx950: 32-entry table (held in LDS, 256 bytes;
// lanes that pick the same entry are served by a broadcast) + degree-6 polynomial, ~1 ulp, ~17 instructions
// (the library exp is ~30).  Shared by the sampler (gibbs_kernel.hip) and the evaluation kernels
// (evaluate.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace kdehip {

// 2^(j/32), j = 0..31, correctly rounded
static __constant__ double kExp2Tab[32] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};

// exp(x) for x <= 0 (NaN in -> NaN out).  x = (32k + j) * ln2/32 + r, |r| <= ln2/64:
// exp(x) = 2^k * 2^(j/32) * (1 + r + r^2/2 + ... + r^6/720); the truncation error is < 4e-18.
// Two halves, so that a caller can put independent work between the table lookup and its use.
struct ExpSplit {
  double r, t;  // reduced argument; 2^(j/32) from the table
  int ki;       // 32k + j
};
__device__ __forceinline__ ExpSplit exp_nonpos_begin(double x, const double *__restrict__ tab /* LDS */) {
  x = fmax(x, -800.0);  // exp(-800) already underflows to 0; keeps the reduction finite
  const double kf = rint(x * 0x1.71547652b82fep+5);            // 32/ln2
  double r = fma(kf, -0x1.62e42fee00000p-6, x);                // ln2/32, high part (32 bits)
  r = fma(kf, -0x1.a39ef35793c76p-38, r);                      // low part
  const int ki = static_cast<int>(kf);
  return {r, tab[ki & 31], ki};
}
__device__ __forceinline__ double exp_nonpos_end(const ExpSplit &s) {
  const double r = s.r;
  double p = fma(r, 0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7);  // 1/720, 1/120
  p = fma(p, r, 0x1.5555555555555p-5);                              // 1/24
  p = fma(p, r, 0x1.5555555555555p-3);                              // 1/6
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = p * r;                                                        // exp(r) - 1
  return ldexp(fma(s.t, p, s.t), s.ki >> 5);
}
__device__ __forceinline__ double exp_nonpos(double x, const double *__restrict__ tab /* LDS */) {
  return exp_nonpos_end(exp_nonpos_begin(x, tab));
}

}  // namespace kdehip
