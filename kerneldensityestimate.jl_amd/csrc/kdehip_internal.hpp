// kdehip_internal.hpp -- shared declarations of libkdehip.so (host side + PODs passed to kernels).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/kdehip.h"

namespace kdehip {

// ---- error reporting (thread-local message behind kdehip_last_error) ---------------------------
int set_error(int code, const std::string &msg);
const char *last_error_cstr();

// ---- packed per-level layout ("pack_levels") ---------------------------------------------------
// The frontier of density j after l levelDown! calls (reference src/MSGibbs01.jl:500-523) is
// data-independent, so it is materialised once on the host.  A frontier of n nodes is stored as a
// tile of B*64 positions, B = ceil(n/64): frontier entry z (0-based, reference order) lives at
// position (z % B)*64 + z / B, i.e. lane `ln` of a wavefront owns the CONTIGUOUS entries
// z = ln*B .. ln*B+B-1 and reads them with fully coalesced loads (row i of the tile holds entry
// ln*B+i of every lane).  Fields are struct-of-arrays with leading dimension ld = B*64:
//   [0, D)      mean of dim d          [D, 2D)  bandwidth (variance) of dim d      [2D]  weight
// Padding positions (z >= n) carry weight 0, variance 1, mean 0 and never win a draw.
struct LevelDesc {
  int32_t n;          // frontier size n_{j,l}
  int32_t B;          // entries per lane
  int64_t data_off;   // element offset (units of T) of the tile in the plan's data buffer
  int64_t perm_off;   // offset of the tile's int32 permutation row (position-indexed, 0 = internal)
  int32_t uniform_bw; // 1: every node of the frontier has the same bandwidth vector
  int32_t pad_;
};

struct PlanDev {
  const void *data;          // T[...]
  const int32_t *perm;       // int32[...]
  const LevelDesc *levels;   // [M][L+1], level 0 = root
  int32_t M, L, D, pad_;
  uint32_t mask_bits[KDEHIP_MAX_DENS];    // bit d: density j informs dimension d (partialDimMask)
  uint32_t others_bits[KDEHIP_MAX_DENS];  // bit d: some density k != j informs dimension d
};

struct RunArgs {
  int64_t Np;
  int32_t Niter;
  int32_t addEntropy;
  int32_t rng_philox;   // 0: read d_randU/d_randN, 1: on-device Philox
  int32_t variant;
  const double *randU;
  const double *randN;
  int64_t K, R;         // per-sample consumption
  uint64_t seed;
  int64_t sample_offset;
  double *points;
  int64_t *indices;
  int32_t *labels;
};

// Host result of packing one product (precision-independent description + fp64 payload; the fp32
// payload is a rounding of it).
struct PackedProduct {
  int M = 0, D = 0, L = 0;
  std::vector<LevelDesc> levels;   // [M][L+1]
  std::vector<double> data;        // fp64 payload
  std::vector<int32_t> perm;
  int64_t nodes_per_sweep = 0;     // sum_j sum_{l>=1} n_{j,l}
  bool fast_ok_f64 = true, fast_ok_f32 = true;
  uint32_t mask_bits[KDEHIP_MAX_DENS] = {0};
  uint32_t others_bits[KDEHIP_MAX_DENS] = {0};
  bool masked = false;
};

// Validates the densities and builds the packed layout.  Returns KDEHIP_OK or an error code.
int pack_levels(int Ndens, const kdehip_density *trees, int ndims, const uint8_t *mask,
                PackedProduct &out);

// floor(log(maxNp)/log(2) + 1), reference src/MSGibbs01.jl:568
int nlevels_for(int64_t maxNp);

// ---- kernel launch (gibbs_kernel.hip) ----------------------------------------------------------
// precision 64/32, fast = product/rsqrt evaluation, otherwise the per-dimension divide+log form.
int launch_gibbs(int precision, bool fast, const PlanDev &plan, const RunArgs &args, void *stream);

}  // namespace kdehip
