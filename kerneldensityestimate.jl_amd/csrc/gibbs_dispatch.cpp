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

int chains_per_workgroup(int64_t Np, int variant) {
  // fewer wavefronts per SIMD run each chain faster, more hide each other's latencies; the relative costs of one
  // round are measured ones (config 3: 0.59, 0.79, 1.10, 1.37 ms for 4, 8, 12, 16 chains per workgroup) and
  // differ little between shapes
  const int v = variant % 1000;
  if (v == 8 || v == 16) return v;
  if (v == 2) return 4;
  // (12 chains per workgroup, cost 1.86, was dropped in round 3: it won only between 2048 and 3072 chains, by 20 %,
  // and cost a quarter of the library's kernels)
  static const int kWidth[3] = {4, 8, 16};
  static const double kCost[3] = {1.0, 1.34, 2.31};
  const int64_t cus = device_cu_count();
  int waves = 16;
  double best = 0.0;
  for (int i = 0; i < 3; ++i) {
    const int64_t wgs = (Np + kWidth[i] - 1) / kWidth[i];
    const double t = static_cast<double>((wgs + cus - 1) / cus) * kCost[i];
    if (i == 0 || t < best) { best = t; waves = kWidth[i]; }
  }
  return waves;
}

int lean_waves(int64_t Np, int variant) {
  const int vt = (variant / 1000) % 10;
  if (vt == 6) return 16;
  if (vt == 8) return 8;
  return chains_per_workgroup(Np, variant);
}

#define KDEHIP_DECL(d)                                                                  \
  int launch_gibbs_d##d(int, int, const PlanDev &, const RunArgs &, void *);           \
  int launch_lean_d##d(int, int, const PlanDev &, const RunArgs &, void *);             \
  int launch_lean_hi_d##d(int, int, const PlanDev &, const RunArgs &, void *);          \
  int launch_lean_batch_d##d(int, const PlanDev &, const RunArgs &, void *);            \
  int launch_tables_batch_d##d(const PlanDev &, const RunArgs &, void *);               \
  int launch_lean_f32_d##d(int, int, const PlanDev &, const RunArgs &, void *);
KDEHIP_DECL(1) KDEHIP_DECL(2) KDEHIP_DECL(3) KDEHIP_DECL(4) KDEHIP_DECL(5) KDEHIP_DECL(6) KDEHIP_DECL(7) KDEHIP_DECL(8)
#undef KDEHIP_DECL

int launch_gibbs(int precision, int mode, const PlanDev &plan, const RunArgs &args_in, void *stream) {
  RunArgs args = args_in;
  const int v = args.variant % 1000;
  const bool generic_only = (v >= kVariantGenericBase && v < kVariantGenericBase + 20);
  if (generic_only) args.variant -= kVariantGenericBase;
  if (!generic_only && plan.M == 8 && precision == 64) {  // 8 densities, fp64 (BASELINE config 4): its own translation unit
    int rc = kLeanNotCovered;
    switch (plan.D) {
      case 1: rc = launch_lean_hi_d1(precision, mode, plan, args, stream); break;
      case 2: rc = launch_lean_hi_d2(precision, mode, plan, args, stream); break;
      case 3: rc = launch_lean_hi_d3(precision, mode, plan, args, stream); break;
      case 4: rc = launch_lean_hi_d4(precision, mode, plan, args, stream); break;
      case 5: rc = launch_lean_hi_d5(precision, mode, plan, args, stream); break;
      case 6: rc = launch_lean_hi_d6(precision, mode, plan, args, stream); break;
      case 7: rc = launch_lean_hi_d7(precision, mode, plan, args, stream); break;
      case 8: rc = launch_lean_hi_d8(precision, mode, plan, args, stream); break;
      default: break;
    }
    if (rc != kLeanNotCovered) return rc;
  } else if (!generic_only && plan.M <= 4) {  // products of 2..4 densities, all dimensions active: the register-resident kernel  // products of 2..4 densities, all dimensions active: the register-resident kernel
    int rc = kLeanNotCovered;
    const bool f32 = (precision == 32);
    switch (plan.D) {
      case 1: rc = f32 ? launch_lean_f32_d1(precision, mode, plan, args, stream) : launch_lean_d1(precision, mode, plan, args, stream); break;
      case 2: rc = f32 ? launch_lean_f32_d2(precision, mode, plan, args, stream) : launch_lean_d2(precision, mode, plan, args, stream); break;
      case 3: rc = f32 ? launch_lean_f32_d3(precision, mode, plan, args, stream) : launch_lean_d3(precision, mode, plan, args, stream); break;
      case 4: rc = f32 ? launch_lean_f32_d4(precision, mode, plan, args, stream) : launch_lean_d4(precision, mode, plan, args, stream); break;
      case 5: rc = f32 ? launch_lean_f32_d5(precision, mode, plan, args, stream) : launch_lean_d5(precision, mode, plan, args, stream); break;
      case 6: rc = f32 ? launch_lean_f32_d6(precision, mode, plan, args, stream) : launch_lean_d6(precision, mode, plan, args, stream); break;
      case 7: rc = f32 ? launch_lean_f32_d7(precision, mode, plan, args, stream) : launch_lean_d7(precision, mode, plan, args, stream); break;
      case 8: rc = f32 ? launch_lean_f32_d8(precision, mode, plan, args, stream) : launch_lean_d8(precision, mode, plan, args, stream); break;
      default: break;
    }
    if (rc != kLeanNotCovered) return rc;
  }
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

// One launch for a group of fp64 products of M (2..4) densities and D dimensions (kdehip_prod_philox_batch): the BATCH
// instantiations of gibbs_lean.hip, 16 chains per workgroup.  `plan` = any member's (the kernel takes each workgroup's
// own from args.batch); args.Np = workgroups x 16.
int launch_gibbs_batch(int D, int M, const PlanDev &plan, const RunArgs &args, void *stream) {
  switch (D) {
    case 1: return launch_lean_batch_d1(M, plan, args, stream);
    case 2: return launch_lean_batch_d2(M, plan, args, stream);
    case 3: return launch_lean_batch_d3(M, plan, args, stream);
    case 4: return launch_lean_batch_d4(M, plan, args, stream);
    case 5: return launch_lean_batch_d5(M, plan, args, stream);
    case 6: return launch_lean_batch_d6(M, plan, args, stream);
    case 7: return launch_lean_batch_d7(M, plan, args, stream);
    case 8: return launch_lean_batch_d8(M, plan, args, stream);
    default: return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  }
}

// the conditional tables of such a group, one launch (args.batch / batch_map list the products that have tables; args.Np =
// workgroups x 4)
int launch_tables_batch(int D, const PlanDev &plan, const RunArgs &args, void *stream) {
  switch (D) {
    case 1: return launch_tables_batch_d1(plan, args, stream);
    case 2: return launch_tables_batch_d2(plan, args, stream);
    case 3: return launch_tables_batch_d3(plan, args, stream);
    case 4: return launch_tables_batch_d4(plan, args, stream);
    case 5: return launch_tables_batch_d5(plan, args, stream);
    case 6: return launch_tables_batch_d6(plan, args, stream);
    case 7: return launch_tables_batch_d7(plan, args, stream);
    case 8: return launch_tables_batch_d8(plan, args, stream);
    default: return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  }
}

}  // namespace kdehip
