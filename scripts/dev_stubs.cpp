// Stubs for the sampler translation units a MINI development library leaves out (scripts/dev_lean.sh MINI=1): every
// launch entry of the dimension counts other than -DKDEHIP_DIM reports "not covered" / unsupported.
#include "../kerneldensityestimate.jl_amd/csrc/kdehip_internal.hpp"
namespace kdehip {
#define STUB(d)                                                                                                       \
  int launch_gibbs_d##d(int, int, const PlanDev &, const RunArgs &, void *) { return set_error(KDEHIP_ERR_UNSUPPORTED, "mini development library: dimension count not built"); } \
  int launch_tables_batch_d##d(const PlanDev &, const RunArgs &, void *) { return set_error(KDEHIP_ERR_UNSUPPORTED, "mini development library: dimension count not built"); } \
  int launch_lean_d##d(int, int, const PlanDev &, const RunArgs &, void *) { return kLeanNotCovered; }                 \
  int launch_lean_hi_d##d(int, int, const PlanDev &, const RunArgs &, void *) { return kLeanNotCovered; }              \
  int launch_lean_f32_d##d(int, int, const PlanDev &, const RunArgs &, void *) { return kLeanNotCovered; }
#define BSTUB(d) \
  int launch_lean_batch_d##d(int, const PlanDev &, const RunArgs &, void *) { return set_error(KDEHIP_ERR_UNSUPPORTED, "mini development library: no batched kernels"); }
#if KDEHIP_DIM != 1
STUB(1) BSTUB(1)
#endif
#if KDEHIP_DIM != 2
STUB(2) BSTUB(2)
#endif
#if KDEHIP_DIM != 3
STUB(3) BSTUB(3)
#endif
#if KDEHIP_DIM != 4
STUB(4) BSTUB(4)
#endif
#if KDEHIP_DIM != 5
STUB(5) BSTUB(5)
#endif
#if KDEHIP_DIM != 6
STUB(6) BSTUB(6)
#endif
#if KDEHIP_DIM != 7
STUB(7) BSTUB(7)
#endif
#if KDEHIP_DIM != 8
STUB(8) BSTUB(8)
#endif
}  // namespace kdehip
