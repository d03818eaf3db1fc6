// device_density.hpp -- densities that live in HBM (pack_device.hip) and what the packers share.
#pragma once

#include <cstdint>
#include <vector>

#include "kdehip_internal.hpp"

namespace kdehip {

// The frontiers of ONE density for levels 0..L (levelDown!, reference src/MSGibbs01.jl:500-523): data independent,
// so they are expanded once -- per product by the host packer, per density for densities kept on the device.
struct Frontiers {
  std::vector<int32_t> ids;      // node ids (1-based), level l at [off[l], off[l+1])
  std::vector<int64_t> off;      // L + 2 entries
  std::vector<uint8_t> uniform;  // L + 1: every node of the frontier has the bandwidth vector of its first node
  bool bad = false;              // look = true: a mean beyond 1e100, a non-positive / non-finite variance or weight was seen
  double lo[KDEHIP_MAX_DIMS], hi[KDEHIP_MAX_DIMS];  // look = true: range of the variances per dimension
  int64_t nodes = 0;             // sum_{l >= 1} n_l
};
// `look`: also examine every node once (the conditions of the fast arithmetic form, pack_levels.cpp).
int expand_frontiers(const kdehip_density &t, int D, int L, bool look, Frontiers &out);

// The shape of one tile: all pack_layout needs once the frontiers are known.
struct TileShape {
  int64_t n;
  bool uniform;
};
// Phases 2-4 of pack_layout (tile geometry, staging modes, conditional tables) from the shapes [M][L+1] alone.
int pack_layout_shapes(int M, int D, int L, const TileShape *shapes, const uint8_t *mask, int precision, bool fast,
                       PackedProduct &out);
// whether D variances in [lo, 2*hi] keep the product/rsqrt arithmetic inside the range of the precision
bool variances_in_range(const double *bw_lo, const double *bw_hi, int D, int precision);

// One tile of the plan image as the gather kernel of pack_device.hip sees it (what pack_layout_shapes decided).
struct FillJob {
  int64_t hdr_off;    // element offset of the tile header in the plan's data
  int64_t perm_off;   // offset of its permutation row
  int64_t front_off;  // offset of the frontier's node ids in the density's `front` array
  int32_t n, B, F, uniform;
  int32_t dens;       // which density
  int32_t pad_;
};
static_assert(sizeof(FillJob) == 48, "FillJob layout");
struct FillArgs {
  const double *means[KDEHIP_MAX_DENS];
  const double *bandwidth[KDEHIP_MAX_DENS];
  const double *weights[KDEHIP_MAX_DENS];
  const int64_t *perm[KDEHIP_MAX_DENS];
  const int32_t *front[KDEHIP_MAX_DENS];
  const FillJob *jobs;
  void *data;  // T[...]
  int32_t *perm_out;
  int32_t D;
};
int launch_fill_tiles(int precision, const FillArgs &a, int ntiles, int maxB, void *stream);

}  // namespace kdehip

// A BallTreeDensity resident on one device (include/kdehip.h "densities in HBM").
struct kdehip_device_density {
  int device = 0;
  int64_t N = 0;
  int D = 0;
  int Lown = 0;               // levels of its own tree: floor(log(N)/log 2 + 1); deeper frontiers repeat the last one (all leaves)
  kdehip::Frontiers fr;       // sizes, offsets and flags (the ids themselves live on the device: `front`)
  void *d_blob = nullptr;     // the one device allocation; the pointers below point into it
  size_t blob_bytes = 0;
  const double *means = nullptr, *bandwidth = nullptr, *weights = nullptr;
  const int64_t *perm = nullptr;
  const int32_t *front = nullptr;
};
