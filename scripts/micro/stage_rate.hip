// How fast does a workgroup stage an L2-resident tile into LDS with direct-to-LDS loads (buffer_load_dwordx4 ... lds,
// the sampler's stage_tile)?  Every workgroup of the grid (one per CU) copies the SAME `bytes`-sized image `iters`
// times; a copy is issued by all wavefronts (1 KiB pieces, round robin) and waited for (vmcnt(0) + barrier) before the
// next one starts -- so the time per copy is the issue -> landed time of one tile, not a pipelined rate -- or, with
// DEPTH = 2, two copies (the two pool halves) are kept in flight.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/stage_rate.hip -o /tmp/stage_rate && /tmp/stage_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using LdsVoidPtr = __attribute__((address_space(3))) void *;

template <int WAVES>
__device__ __forceinline__ void stage(const unsigned char *src, unsigned char *dst, int bytes, int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(src), 0, bytes, 0x00020000);
  const int pieces = bytes >> 10;
  for (int c = wave; c < pieces; c += WAVES)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsVoidPtr)(dst + (c << 10)), 16, lane << 4, c << 10, 0, 0);
}

template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void k(const unsigned char *src, int bytes, int iters, double *out) {
  __shared__ __attribute__((aligned(1024))) unsigned char pool[120 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (DEPTH == 2) stage<WAVES>(src, pool, bytes, wave, lane);
  for (int i = 0; i < iters; ++i) {
    if (DEPTH == 2) {
      stage<WAVES>(src, pool + ((i + 1) & 1) * 60 * 1024, bytes, wave, lane);
      // wait for the older of the two copies: this wavefront's share of it = ceil/floor(pieces / WAVES) loads
      const int mine = ((bytes >> 10) - wave + WAVES - 1) / WAVES;
      if (mine >= 8) __builtin_amdgcn_s_waitcnt(0x0F70 | 8); else if (mine == 7) __builtin_amdgcn_s_waitcnt(0x0F70 | 7);
      else __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
    } else {
      stage<WAVES>(src, pool, bytes, wave, lane);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __syncthreads();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = reinterpret_cast<double *>(pool)[5];
}

template <int WAVES, int DEPTH>
void run(int bytes) {
  const int iters = 2000, blocks = 256;
  unsigned char *src; double *out;
  (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 1, 1 << 20);
  (void)hipMalloc(&out, sizeof(double) * blocks);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WAVES, DEPTH>), dim3(blocks), dim3(WAVES * 64), 0, 0, src, bytes, 10, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<WAVES, DEPTH>), dim3(blocks), dim3(WAVES * 64), 0, 0, src, bytes, iters, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;
  std::printf("waves %2d  depth %d  tile %6d B : %7.3f us per copy = %6.1f GB/s per CU (all 256 CUs busy)\n", WAVES, DEPTH, bytes, us,
              bytes / us * 1e-3);
  (void)hipFree(src); (void)hipFree(out);
}

int main() {
  for (int bytes : {8 * 1024, 16 * 1024, 28 * 1024, 57 * 1024}) {
    run<8, 1>(bytes);
    run<16, 1>(bytes);
    run<4, 1>(bytes);
  }
  run<8, 2>(57 * 1024);
  run<16, 2>(57 * 1024);
  return 0;
}
