// selftest.hip -- measures, on the device it runs on, the accuracy of the three hardware fp32 approximations the fp32
// screen's error bound takes as a premise (screen_device.hpp: v_rcp_f32, v_rsq_f32 and v_exp_f32 "within 1 ulp", i.e. a
// relative error of at most 2 u, u = 2^-24).  The reference has no counterpart (src/MSGibbs01.jl:250-351 is fp64 only):
// this is what turns the premise from documentation into a measurement -- tests/test_gpu_ulp.py sweeps every fp32 input
// of the ranges the screen can feed the instructions and asserts the budget.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "kdehip_internal.hpp"

namespace kdehip {
namespace {

// hardware result for input bit pattern `b` (the builtins the screen's evaluators use: gibbs_device.hpp Num<float>,
// gibbs_lean.hip step_screen)
template <int WHICH>
__device__ __forceinline__ float hw(float x) {
  if constexpr (WHICH == 0) return __builtin_amdgcn_rcpf(x);
  else if constexpr (WHICH == 1) return __builtin_amdgcn_rsqf(x);
  else return __builtin_amdgcn_exp2f(x);
}
// the error of one input: relative, in units of u (WHICH 0..2); WHICH 3: |result - 2^x| in units of 2^-126 (the zone where
// the exact value is below the smallest normal: the bound only needs "what fp32 flushes or holds as a denormal is off by
// less than 2^-126")
template <int WHICH>
__device__ __forceinline__ float err_of(uint32_t bits) {
  const float x = __uint_as_float(bits);
  const double xd = static_cast<double>(x);
  double ref;
  if constexpr (WHICH == 0) ref = 1.0 / xd;
  else if constexpr (WHICH == 1) ref = 1.0 / sqrt(xd);
  else ref = exp2(xd);
  const double r = static_cast<double>(hw<(WHICH == 3 ? 2 : WHICH)>(x));
  double e;
  if constexpr (WHICH == 3) e = fabs(r - ref) * 0x1p126;
  else e = fabs(r - ref) / fabs(ref) * 0x1p24;
  if (!(e == e)) e = 0x1p60;  // a NaN where a number was due is an unbounded error
  return static_cast<float>(e);
}

template <int WHICH>
__global__ __launch_bounds__(256) void ulp_sweep_kernel(uint32_t first, uint64_t count, unsigned long long *best) {
  unsigned long long key = 0ull;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < count; i += 256ull * gridDim.x) {
    const uint32_t b = first + static_cast<uint32_t>(i);
    const float e = err_of<WHICH>(b);
    const unsigned long long k = (static_cast<unsigned long long>(__float_as_uint(e)) << 32) | b;  // (e >= 0: bits are monotone)
    key = k > key ? k : key;
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const unsigned long long o = __shfl_xor(key, s);
    key = o > key ? o : key;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(best, key);
}
template <int WHICH>
__global__ void ulp_one_kernel(uint32_t bits, uint32_t *out) {
  out[0] = __float_as_uint(hw<(WHICH == 3 ? 2 : WHICH)>(__uint_as_float(bits)));
}

template <int WHICH>
hipError_t run(uint32_t first, uint64_t count, unsigned long long *d_best, uint32_t *d_res, unsigned long long *h_best,
               uint32_t *h_res) {
  hipError_t e = hipMemsetAsync(d_best, 0, sizeof(unsigned long long), nullptr);
  if (e != hipSuccess) return e;
  const int blocks = count < (1u << 20) ? 64 : 256 * 8;
  ulp_sweep_kernel<WHICH><<<blocks, 256, 0, nullptr>>>(first, count, d_best);
  e = hipMemcpy(h_best, d_best, sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  ulp_one_kernel<WHICH><<<1, 1, 0, nullptr>>>(static_cast<uint32_t>(*h_best & 0xffffffffull), d_res);
  return hipMemcpy(h_res, d_res, sizeof(uint32_t), hipMemcpyDeviceToHost);
}

}  // namespace
}  // namespace kdehip

extern "C" int kdehip_selftest_fp32(int which, uint32_t first_bits, uint64_t count, int device, double *max_err,
                                    uint32_t *worst_bits, uint32_t *worst_result_bits) {
  using namespace kdehip;
  if (which < 0 || which > 3 || count == 0 || count > (1ull << 32) || !max_err)
    return set_error(KDEHIP_ERR_ARG, "kdehip_selftest_fp32: which in 0..3, 1 <= count <= 2^32, max_err required");
  DeviceGuard guard;
  if (int rc = guard.enter(device)) return rc;
  void *blk = nullptr;
  hipError_t e = cached_malloc(&blk, 64);
  if (e != hipSuccess) return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_selftest_fp32: ") + hipGetErrorString(e));
  auto *d_best = static_cast<unsigned long long *>(blk);
  auto *d_res = reinterpret_cast<uint32_t *>(d_best + 1);
  unsigned long long best = 0;
  uint32_t res = 0;
  switch (which) {
    case 0: e = run<0>(first_bits, count, d_best, d_res, &best, &res); break;
    case 1: e = run<1>(first_bits, count, d_best, d_res, &best, &res); break;
    case 2: e = run<2>(first_bits, count, d_best, d_res, &best, &res); break;
    default: e = run<3>(first_bits, count, d_best, d_res, &best, &res); break;
  }
  cached_free(blk, 64);
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("kdehip_selftest_fp32: ") + hipGetErrorString(e));
  const uint32_t ebits = static_cast<uint32_t>(best >> 32);
  float ef;
  __builtin_memcpy(&ef, &ebits, sizeof ef);
  *max_err = static_cast<double>(ef);
  if (worst_bits) *worst_bits = static_cast<uint32_t>(best & 0xffffffffull);
  if (worst_result_bits) *worst_result_bits = res;
  return KDEHIP_OK;
}
