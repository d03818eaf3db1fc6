// Issue-rate microbenchmark for the vector instructions the samplers are made of (gfx950): shader cycles per
// wave-instruction and SIMD at 1, 2 and 4 wavefronts per SIMD (s_memtime around an unrolled loop of independent
// operations; the clock the chip held = d s_memtime / d s_memrealtime x 100 MHz), and ns from HIP events.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void rate_kernel(float *out, unsigned long long *cycles, int iters) {
  // cycles[2*b] = shader cycles, cycles[2*b+1] = 100 MHz ticks of workgroup b's first wavefront
  // 8 independent accumulators: no dependency stalls
  float a[8];
  f2 p[8];
  double d[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * 1e-3f + k; p[k] = f2{a[k], a[k] + 1.0f}; d[k] = a[k]; }
  const float m = 0.999f, c = 1e-3f;
  const f2 pm = {m, m}, pc = {c, c};
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pm), "v"(pc));
        if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
        if constexpr (OP == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
        if constexpr (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"((double)m), "v"((double)c));
        if constexpr (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pm));
        if constexpr (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
        if constexpr (OP == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"((double)m));
        if constexpr (OP == 8) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
        if constexpr (OP == 9) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));
        if constexpr (OP == 10) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"((double)c));
        if constexpr (OP == 11) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[k]) : "v"(1));
        if constexpr (OP == 12) asm volatile("v_mov_b32 %0, %0" : "+v"(a[k]));
        if constexpr (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(m));
        if constexpr (OP == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[k]) : "v"(m) : "s20", "s21");
        if constexpr (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if constexpr (OP == 16) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[k]), "v"((double)m) : "vcc");
        if constexpr (OP == 17) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        if constexpr (OP == 18) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[k]) : "s20");
        if constexpr (OP == 20) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(m));
        if constexpr (OP == 21) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(m) : "vcc");
        if constexpr (OP == 22) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[k]) : "v"(m) : "s20", "s21");
        if constexpr (OP == 23) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(m) : "vcc");
        if constexpr (OP == 24) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[k]) : "v"((double)m));
        if constexpr (OP == 25) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
        if constexpr (OP == 26) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if constexpr (OP == 27) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(d[k]));
        if constexpr (OP == 19) asm volatile("v_exp_f32 %0, %0\n v_fma_f64 %1, %1, %2, %3" : "+v"(a[k]), "+v"(d[k]) : "v"((double)m), "v"((double)c));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y + (float)d[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cycles[2 * blockIdx.x] = t1 - t0; cycles[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP>
void run(const char *name, int waves_per_simd) {
  const int iters = 20000;
  const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU, its waves spread over the 4 SIMDs
  const int blocks = 256;
  float *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<OP><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e0);
  rate_kernel<OP><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = double(iters) * 32 * waves_per_simd;  // wave-instructions each SIMD executed
  // kernel time -> ns per wave-instruction per SIMD; x 2.4 GHz = cycles (the clock the chip holds is not known exactly)
  const double ns = ms * 1e6 / insts_per_simd;
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
  std::vector<double> cy(blocks), ghz(blocks);
  for (int b = 0; b < blocks; ++b) { cy[b] = double(h[2 * b]) / insts_per_simd; ghz[b] = double(h[2 * b]) / double(h[2 * b + 1]) * 0.1; }
  std::sort(cy.begin(), cy.end()); std::sort(ghz.begin(), ghz.end());
  printf("%-14s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD at %.2f GHz (medians over workgroups); %.3f ns by HIP events\n",
         name, waves_per_simd, cy[blocks / 2], ghz[blocks / 2], ns);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 4}) {
    run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<5>("v_pk_mul_f32", w); run<6>("v_pk_add_f32", w);
    run<2>("v_exp_f32", w); run<3>("v_rsq_f32", w);
    run<4>("v_fma_f64", w); run<7>("v_mul_f64", w); run<10>("v_add_f64", w); run<8>("v_rcp_f64", w); run<9>("v_rsq_f64", w);
    run<11>("v_ldexp_f64", w); run<12>("v_mov_b32", w); run<13>("v_cndmask vcc", w); run<14>("v_cndmask sgpr", w);
    run<15>("v_add_u32", w); run<16>("v_cmp_lt_f64", w); run<17>("v_mov_b32 dpp", w); run<18>("v_readlane_b32", w);
    run<19>("exp_f32+fma_f64", w); run<20>("cndmask e64 vcc", w); run<21>("cmp+cndmask vcc", w); run<22>("cmp+cndmask sgpr", w);
    run<23>("v_cmp_lt_f32 vcc", w); run<24>("v_max_f64", w); run<25>("v_cvt_f64_i32", w); run<26>("v_and_b32", w); run<27>("v_lshlrev_b64", w);
  }
  return 0;
}
