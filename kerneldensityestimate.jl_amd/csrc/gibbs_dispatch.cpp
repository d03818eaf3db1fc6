// gibbs_dispatch.cpp -- routes a product run to the translation unit that holds the kernels of its
// dimension count (gibbs_kernel.hip is compiled once per D, see the Makefile).
#include <hip/hip_runtime.h>

#include <atomic>

#include "kdehip_internal.hpp"

namespace kdehip {

int device_cu_count() {
  static std::atomic<int> cache[64];  // per device ordinal; 0 = not asked yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

#define KDEHIP_DECL(d) int launch_gibbs_d##d(int, int, const PlanDev &, const RunArgs &, void *);
KDEHIP_DECL(1) KDEHIP_DECL(2) KDEHIP_DECL(3) KDEHIP_DECL(4) KDEHIP_DECL(5) KDEHIP_DECL(6) KDEHIP_DECL(7) KDEHIP_DECL(8)
#undef KDEHIP_DECL

int launch_gibbs(int precision, int mode, const PlanDev &plan, const RunArgs &args, void *stream) {
  switch (plan.D) {
    case 1: return launch_gibbs_d1(precision, mode, plan, args, stream);
    case 2: return launch_gibbs_d2(precision, mode, plan, args, stream);
    case 3: return launch_gibbs_d3(precision, mode, plan, args, stream);
    case 4: return launch_gibbs_d4(precision, mode, plan, args, stream);
    case 5: return launch_gibbs_d5(precision, mode, plan, args, stream);
    case 6: return launch_gibbs_d6(precision, mode, plan, args, stream);
    case 7: return launch_gibbs_d7(precision, mode, plan, args, stream);
    case 8: return launch_gibbs_d8(precision, mode, plan, args, stream);
    default: return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  }
}

}  // namespace kdehip
