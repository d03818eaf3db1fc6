// philox.hpp -- Philox4x32-10 counter RNG, identical on host and device.
//
// Replaces the reference's `rand(...)` / `randn(...)` keyword defaults (src/MSGibbs01.jl:661-662):
// every random number of a product is a pure function of (seed, global sample index, draw index),
// so chains can be split over wavefronts, calls and GPUs without changing any result.
//   stream 0, block b -> uniforms 2b and 2b+1 of the sample   (select call c uses uniform c)
//   stream 1, block b -> normals  2b and 2b+1 of the sample   (Box-Muller pair)
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define KDEHIP_HD __host__ __device__ inline
#else
#define KDEHIP_HD inline
#endif

namespace kdehip {

struct Philox4 {
  uint32_t v[4];
};

KDEHIP_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32);
#endif
}

KDEHIP_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 out;
  out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
  return out;
}

// 64 random bits -> double in (0,1): 53 bits, offset by half an ulp so 0 and 1 never occur.
KDEHIP_HD double bits_to_unit(uint32_t lo, uint32_t hi) {
  const uint64_t x = (static_cast<uint64_t>(hi) << 32) | lo;
  return (static_cast<double>(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

KDEHIP_HD Philox4 philox_block(uint64_t seed, uint64_t sample, uint32_t block, uint32_t stream) {
  return philox4x32_10(static_cast<uint32_t>(sample), static_cast<uint32_t>(sample >> 32), block,
                       stream, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
}

// uniform `c` of a sample
KDEHIP_HD double philox_uniform(uint64_t seed, uint64_t sample, uint32_t c) {
  const Philox4 r = philox_block(seed, sample, c >> 1, 0u);
  return (c & 1u) ? bits_to_unit(r.v[2], r.v[3]) : bits_to_unit(r.v[0], r.v[1]);
}

// normal `r` of a sample: Box-Muller on the block's two uniforms (even r -> cosine, odd r -> sine)
KDEHIP_HD double philox_normal(uint64_t seed, uint64_t sample, uint32_t r) {
  const Philox4 b = philox_block(seed, sample, r >> 1, 1u);
  const double u1 = bits_to_unit(b.v[0], b.v[1]);
  const double u2 = bits_to_unit(b.v[2], b.v[3]);
  const double rad = sqrt(-2.0 * log(u1));
  const double ang = 6.283185307179586476925286766559 * u2;
  return (r & 1u) ? rad * sin(ang) : rad * cos(ang);
}

// both normals of block `b` (normals 2b and 2b+1) from ONE Philox block, logarithm and sine/cosine pair: the same
// numbers as philox_normal(.., 2b) and philox_normal(.., 2b+1)
KDEHIP_HD void philox_normal_pair(uint64_t seed, uint64_t sample, uint32_t b, double &even, double &odd) {
  const Philox4 blk = philox_block(seed, sample, b, 1u);
  const double u1 = bits_to_unit(blk.v[0], blk.v[1]);
  const double u2 = bits_to_unit(blk.v[2], blk.v[3]);
  const double rad = sqrt(-2.0 * log(u1));
  const double ang = 6.283185307179586476925286766559 * u2;
  even = rad * cos(ang);
  odd = rad * sin(ang);
}

}  // namespace kdehip
