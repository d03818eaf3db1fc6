// device_density.hpp -- densities that live in HBM (pack_device.hip) and what the packers share.
#pragma once

#include <atomic>
#include <cstdint>
#include <vector>

#include "kdehip_internal.hpp"

namespace kdehip {

// The frontiers of ONE density for levels 0..L (levelDown!, reference src/MSGibbs01.jl:500-523): data independent,
// so they are expanded once -- per product by the host packer, per density for densities kept on the device.
struct Frontiers {
  std::vector<int32_t> ids;      // node ids (1-based), level l at [off[l], off[l+1])
  std::vector<uint8_t> fresh;    // per id: 1 where the node enters the frontier (a leaf stays as its own child: 0 from then on)
  std::vector<int64_t> off;      // L + 2 entries
  std::vector<uint8_t> uniform;  // L + 1: every node of the frontier has the bandwidth vector of its first node
  std::vector<double> uratio;    // L + 1, uniform frontiers: max over nodes and dimensions of |mean_d| / sqrt(2 bandwidth_d)
                                 // (what the shared-bandwidth evaluator's rounding error grows with: pack_layout_shapes)
  bool bad = false;              // look = true: a mean beyond 1e100, a non-positive / non-finite variance or weight was seen
  double lo[KDEHIP_MAX_DIMS], hi[KDEHIP_MAX_DIMS];  // look = true: range of the variances per dimension
  int64_t nodes = 0;             // sum_{l >= 1} n_l
};
// `look`: also examine every node once (the conditions of the fast arithmetic form, pack_levels.cpp).
int expand_frontiers(const kdehip_density &t, int D, int L, bool look, Frontiers &out);
// The same in two steps: the ids depend on the child arrays only, the flags (shared bandwidth per level, ranges, finiteness)
// on the values -- kde!(points) of a device matrix expands the ids while the GPU still searches the bandwidth.
int expand_frontier_ids(const kdehip_density &t, int D, int L, Frontiers &out);
void examine_frontiers(const kdehip_density &t, int D, int L, bool look, Frontiers &out);

// The shape of one tile: all pack_layout needs once the frontiers are known.
struct TileShape {
  int64_t n;
  bool uniform;
  double uratio;  // Frontiers.uratio of a uniform frontier
};
// Phases 2-4 of pack_layout (tile geometry, staging modes, conditional tables) from the shapes [M][L+1] alone.
int pack_layout_shapes(int M, int D, int L, const TileShape *shapes, const uint8_t *mask, int precision, bool fast,
                       PackedProduct &out);
// whether D variances in [lo, 2*hi] keep the product/rsqrt arithmetic inside the range of the precision
bool variances_in_range(const double *bw_lo, const double *bw_hi, int D, int precision);

// One tile of a plan image as the gather kernel of pack_device.hip sees it (what pack_layout_shapes decided): where its
// frontier's nodes come from (the density's arrays in HBM) and where header, rows and permutation row go.  Absolute
// pointers: one launch fills the tiles of any number of products (kdehip_prod_philox_batch).
struct FillJob {
  const double *means, *bandwidth, *weights;
  const int64_t *perm;
  const int32_t *front;  // the frontier's node ids (1-based)
  void *hdr;             // T*: the tile header
  int32_t *perm_out;     // the tile's permutation row
  int32_t n, B, F, uniform, D, pad_;
};
static_assert(sizeof(FillJob) == 80, "FillJob layout");
int launch_fill_tiles(int precision, const FillJob *d_jobs, int ntiles, int maxB, void *stream);
// The fp32 screen tiles of an fp64 plan (kdehip_internal.hpp "fp32 screening") from its fp64 tiles, one workgroup per
// (density, level); enqueue only.
int launch_screen_build(const PlanDev &plan, void *stream);

// kdehip_density_set_bandwidth (balltree.cpp) that also reports what the packers' examination of the nodes would find
struct NodeStats { bool bad; double lo[KDEHIP_MAX_DIMS], hi[KDEHIP_MAX_DIMS]; };
// (`order`: the internal nodes, every node AFTER its descendants when read back to front -- children_first_order -- when the
// caller has worked it out already; it depends on the child arrays only)
int set_bandwidth_examined(int64_t D, int64_t N, const double *ks, int64_t nks, const double *weights, const int64_t *left_child,
                           const int64_t *right_child, const double *means, double *bandwidth, double *bandwidthMin,
                           double *bandwidthMax, NodeStats *st, const std::vector<int64_t> *order = nullptr);
int children_first_order(int64_t N, const int64_t *left_child, const int64_t *right_child, std::vector<int64_t> &order);

}  // namespace kdehip

namespace kdehip {
// The densities a batched call builds (kdehip_mul_device_batch) live in ONE device block and ONE pinned mirror -- one
// allocation, two uploads for the whole batch instead of three small transfers per density --, released when the last of
// them is freed.
struct SharedBlock {
  std::atomic<int> refs{0};
  void *d_blob = nullptr;
  size_t blob_bytes = 0;
  void *mirror = nullptr;
  size_t mirror_bytes = 0;
};
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
  // A density the library built itself (kdehip_density_from_device_points) keeps the reference's twelve arrays as a
  // host mirror for kdehip_density_download, in ONE pinned block from the library's cache (a fresh 1 MB heap block per
  // density is an mmap of its own: 250 page faults to fill it and -- in a process with a live HIP runtime, whose driver
  // hooks every unmap -- a munmap of ~2 ms to drop it: the "host tree alone 2.1 ms" of profiles/r04p): the head of the
  // block IS the image of the device block ([means | bandwidth | weights | permutation | frontier ids], uploaded
  // straight from here), the other eight arrays follow
  bool built = false;
  void *mirror = nullptr;
  size_t mirror_bytes = 0;
  struct Mirror {
    double *centers = nullptr, *ranges = nullptr, *means = nullptr, *bandwidth = nullptr, *bwmin = nullptr, *bwmax = nullptr,
           *weights = nullptr;
    int64_t *left = nullptr, *right = nullptr, *lowest = nullptr, *highest = nullptr, *perm = nullptr;
  } m;
  double bw[KDEHIP_MAX_DIMS] = {};  // its LOOCV bandwidth (standard deviations)
  kdehip::SharedBlock *shared = nullptr;  // set: d_blob / mirror above are null, the arrays live in the batch's blocks
};

namespace kdehip {
// kdehip_prod_philox_device with everything -- preparation and sampling -- on ONE stream (product.hip): what a blocking
// caller that waits for the product anyway uses (kdehip_mul_device).
int prod_philox_device_blocking_stream(int Ndens, kdehip_device_density *const *trees, int64_t Np, int Niter, uint64_t seed,
                                       int64_t sample_offset, int addEntropy, const uint8_t *partialDimMask, int precision,
                                       double *d_points, int64_t *d_indices, void *stream);
}  // namespace kdehip
