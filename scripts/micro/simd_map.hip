// Which SIMD does wavefront w of a workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // id 4, offset 0, size 32
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = hw;
}
int main() {
  for (int waves : {8, 16}) {
    unsigned *out, h[16 * 8];
    (void)hipMalloc(&out, sizeof(h));
    hipLaunchKernelGGL(k, dim3(8), dim3(waves * 64), 0, 0, out);
    (void)hipMemcpy(h, out, sizeof(unsigned) * waves * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < 8; ++b) {
      std::printf("waves/WG %2d block %d: simd of wave 0..: ", waves, b);
      for (int w = 0; w < waves; ++w) std::printf("%u ", (h[b * waves + w] >> 4) & 3);
      std::printf("\n");
    }
    (void)hipFree(out);
  }
}
