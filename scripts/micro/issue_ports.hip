// Can a SIMD of gfx950 issue a vector, a scalar and an LDS instruction from different wavefronts in the same issue
// slot?  4 wavefronts per SIMD; the odd ones run a second instruction stream beside the fp64 FMA stream of the even
// ones (or every wavefront interleaves the two streams itself).  ns per FMA wave-instruction and SIMD from HIP events.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/issue_ports.hip -o /tmp/issue_ports && /tmp/issue_ports
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: even waves FMA, odd waves exit        1: odd waves SALU       2: odd waves LDS reads
//      3: all waves FMA only                    4: all waves FMA + SALU interleaved 1:1   5: all waves FMA + LDS 4:1
//      6: all waves FMA + SALU 1:1 + LDS 4:1    7: FMA : SALU = 4 : 1      8: FMA : SALU : LDS = 8 : 2 : 1 (the samplers' mix)
template <int MODE>
__global__ void k(double *out, int iters) {
  __shared__ double sm[1024];
  sm[threadIdx.x & 1023] = threadIdx.x;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const bool odd = (wave >> 2) & 1;  // waves w, w+4, w+8, w+12 share SIMD w & 3
  double d[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = threadIdx.x + j;
  const double m = 0.999, c = 1e-3;
  int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
  const __attribute__((address_space(3))) double *lp = (const __attribute__((address_space(3))) double *)sm + (threadIdx.x & 63);
  double acc = 0;
  if (MODE <= 2 && odd) {
    if (MODE == 0) return;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        if (MODE == 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc");
        if (MODE == 2) { double v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(lp)); asm volatile("" :: "v"(v)); }
      }
      if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s1;
    return;
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(m), "v"(c));
        if (MODE == 4 || MODE == 6 || ((MODE == 7 || MODE == 8) && (j & 3) == 1))
          asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc");
        if (((MODE == 5 || MODE == 6) && (j & 3) == 3) || (MODE == 8 && j == 7)) { double v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(lp)); asm volatile("" :: "v"(v)); }
      }
    }
    if (MODE >= 5) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  for (int j = 0; j < 8; ++j) acc += d[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s1 + s3 + s0;
}

template <int MODE>
void run(const char *name, int threads = 1024) {
  const int iters = 20000, blocks = 256;
  double *out;
  (void)hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(out, iters);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fma_waves = (MODE <= 2) ? 2 : threads / 256;  // FMA wavefronts per SIMD
  const double ns = ms * 1e6 / (double(iters) * 32 * fma_waves);
  printf("%-52s %.3f ms, %.3f ns per FMA wave-instruction and SIMD\n", name, ms, ns);
  (void)hipFree(out);
}

int main() {
  run<0>("2 FMA wavefronts per SIMD alone");
  run<1>("2 FMA + 2 SALU wavefronts per SIMD");
  run<2>("2 FMA + 2 LDS-read wavefronts per SIMD");
  run<3>("4 FMA wavefronts per SIMD");
  run<4>("4 wavefronts, each FMA : SALU = 1 : 1");
  run<5>("4 wavefronts, each FMA : LDS = 4 : 1");
  run<6>("4 wavefronts, each FMA : SALU : LDS = 4 : 4 : 1");
  run<7>("4 wavefronts, each FMA : SALU = 4 : 1");
  run<8>("4 wavefronts, each FMA : SALU : LDS = 8 : 2 : 1");
  run<3>("2 wavefronts per SIMD, FMA only", 512);
  run<7>("2 wavefronts, each FMA : SALU = 4 : 1", 512);
  run<8>("2 wavefronts, each FMA : SALU : LDS = 8 : 2 : 1", 512);
  run<3>("1 wavefront per SIMD, FMA only", 256);
  run<8>("1 wavefront, FMA : SALU : LDS = 8 : 2 : 1", 256);
  return 0;
}
