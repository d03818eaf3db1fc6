// balltree.cpp -- host-side construction of a BallTreeDensity (product code, part of libkdehip.so).
//
// Replaces, for hosts that do not have the Julia reference at hand, the chain
//   kde!(points, ks, weights)        reference src/KDE01.jl:34-57
//   -> makeBallTreeDensity           reference src/BallTreeDensity01.jl:192-231
//   -> makeBallTree / buildTree!     reference src/BallTree01.jl:415-463
// and must reproduce the reference's node numbering and leaf order exactly, because the level lists
// the Gibbs sampler walks (and therefore which label a given uniform draw selects) depend on them.
// Layout produced: 1-based node ids, internal nodes 1..N-1 (root 1), slot N unused, leaves N+1..2N.
//
// Compiled with -ffp-contract=off: the moment-matching expressions must not be fused.
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/kdehip.h"
#include "kdehip_internal.hpp"

namespace kdehip {
namespace {

class DensityBuilder {
 public:
  DensityBuilder(int64_t D, int64_t N, double *centers, double *ranges, double *weights,
                 int64_t *left, int64_t *right, int64_t *lo, int64_t *hi, int64_t *perm,
                 double *means, double *bw)
      : D_(D), N_(N), centers_(centers), ranges_(ranges), weights_(weights), left_(left),
        right_(right), lo_(lo), hi_(hi), perm_(perm), means_(means), bw_(bw), next_id_(2) {}

  void build() { build_node(N_ + 1, 2 * N_, 1); }

 private:
  // row pointers of node `id` (1-based)
  double *ctr(int64_t id) { return centers_ + (id - 1) * D_; }
  double *rng(int64_t id) { return ranges_ + (id - 1) * D_; }
  double *mu(int64_t id) { return means_ + (id - 1) * D_; }
  double *var(int64_t id) { return bw_ + (id - 1) * D_; }
  bool valid(int64_t id) const { return id > 0 && id <= 2 * N_; }  // BallTree01.jl:83

  // Exchange two leaves: weight, permutation, centre (swapBall!, BallTree01.jl:109-138) and
  // mean, bandwidth (swapDensity!, BallTreeDensity01.jl:112-139; uniform-bandwidth case).
  void exchange(int64_t a, int64_t b) {
    if (a == b) return;
    std::swap(weights_[a - 1], weights_[b - 1]);
    std::swap(perm_[a - 1], perm_[b - 1]);
    double *ca = ctr(a), *cb = ctr(b), *ma = mu(a), *mb = mu(b), *va = var(a), *vb = var(b);
    for (int64_t k = 0; k < D_; ++k) {
      std::swap(ca[k], cb[k]);
      std::swap(ma[k], mb[k]);
      std::swap(va[k], vb[k]);
    }
  }

  // Dimension of largest spread over leaves first..last (most_spread_coord, BallTree01.jl:142-173).
  // The reference leaves the last leaf out of both sums while scaling by 1/(last-first); ties and
  // the all-equal case resolve to the lowest dimension (strict '>').
  int64_t widest_dim(int64_t first, int64_t last) {
    const double scale = 1.0 / static_cast<double>(last - first);
    int64_t best = 0;
    double best_var = 0.0;
    for (int64_t k = 0; k < D_; ++k) {
      double m = 0.0;
      for (int64_t id = first; id < last; ++id) m = m + scale * ctr(id)[k];
      double v = 0.0;
      for (int64_t id = first; id < last; ++id) {
        const double dlt = ctr(id)[k] - m;
        v += dlt * dlt;
      }
      if (v > best_var) { best_var = v; best = k; }
    }
    return best;
  }

  // Quick-select (select!, BallTree01.jl:223-242): afterwards leaves first..pos are <= those after.
  void quick_select(int64_t k, int64_t pos, int64_t first, int64_t last) {
    while (first < last) {
      exchange((first + last) / 2, first);  // pivot to the front
      int64_t store = first;
      for (int64_t id = first; id <= last; ++id) {
        // the pivot value is re-read each time: it stays at `first` for the whole scan
        if (ctr(id)[k] - ctr(first)[k] < 0.0) {
          ++store;
          exchange(store, id);
        }
      }
      exchange(first, store);
      if (store <= pos) first = store + 1;
      if (store >= pos) last = store - 1;
    }
  }

  // Bounding box + weight (calcStatsBall!, BallTree01.jl:282-336) and moment-matched Gaussian
  // (calcStatsDensity!, BallTreeDensity01.jl:141-187) of an internal node from its two children.
  void summarize(int64_t id) {
    const int64_t a = left_[id - 1], b = right_[id - 1];
    if (!valid(a) || !valid(b)) return;
    for (int64_t k = 0; k < D_; ++k) {
      const double upA = ctr(a)[k] + rng(a)[k], upB = ctr(b)[k] + rng(b)[k];
      const double dnA = ctr(a)[k] - rng(a)[k], dnB = ctr(b)[k] - rng(b)[k];
      const double top = (upA > upB) ? upA : upB;
      const double bottom = (dnA < dnB) ? dnA : dnB;
      const double half = (top - bottom) / 2.0;
      rng(id)[k] = half;
      ctr(id)[k] = bottom + half;
    }
    weights_[id - 1] = (a != b) ? weights_[a - 1] + weights_[b - 1] : weights_[a - 1];

    double wa = weights_[a - 1], wb = weights_[b - 1];
    const double wt = wa + wb + DBL_EPSILON;  // eps(Float64), BallTreeDensity01.jl:161
    wa /= wt;
    wb /= wt;
    for (int64_t k = 0; k < D_; ++k) {
      const double ma = mu(a)[k], mb = mu(b)[k];
      const double m = wa * ma + wb * mb;
      mu(id)[k] = m;
      var(id)[k] = wa * (var(a)[k] + ma * ma) + wb * (var(b)[k] + mb * mb) - m * m;
    }
  }

  // buildBall!, BallTree01.jl:342-411.  Child ids are handed out (left, then right) before either
  // subtree is built; a one-leaf side points straight at the leaf.
  void build_node(int64_t first, int64_t last, int64_t id) {
    lo_[id - 1] = first;
    hi_[id - 1] = last;
    if (first == last) {  // single-point density, :351-362
      left_[id - 1] = first;
      right_[id - 1] = last;
      summarize(id);
      right_[id - 1] = -1;
      return;
    }
    const int64_t k = widest_dim(first, last);
    const int64_t mid = (first + last) / 2;
    quick_select(k, mid, first, last);
    const int64_t a = (mid <= first) ? first : next_id_++;
    const int64_t b = (mid + 1 >= last) ? last : next_id_++;
    left_[id - 1] = a;
    right_[id - 1] = b;
    if (a != first) build_node(first, mid, a);
    if (b != last) build_node(mid + 1, last, b);
    summarize(id);
  }

  const int64_t D_, N_;
  double *centers_, *ranges_, *weights_;
  int64_t *left_, *right_, *lo_, *hi_, *perm_;
  double *means_, *bw_;
  int64_t next_id_;
};

}  // namespace
}  // namespace kdehip

extern "C" int kdehip_make_density(int64_t D, int64_t N, const double *points, const double *ks,
                                   int64_t nks, const double *weights_in, double *centers,
                                   double *ranges, double *weights, int64_t *left_child,
                                   int64_t *right_child, int64_t *lowest_leaf, int64_t *highest_leaf,
                                   int64_t *permutation, double *means, double *bandwidth,
                                   double *bandwidthMin, double *bandwidthMax) {
  using namespace kdehip;
  if (D < 1 || N < 1) return set_error(KDEHIP_ERR_ARG, "kdehip_make_density: need D >= 1 and N >= 1");
  if (nks != 1 && nks != D)
    return set_error(KDEHIP_ERR_ARG, "kdehip_make_density: ks must have 1 or D entries");
  if (!points || !ks || !centers || !ranges || !weights || !left_child || !right_child ||
      !lowest_leaf || !highest_leaf || !permutation || !means || !bandwidth || !bandwidthMin ||
      !bandwidthMax)
    return set_error(KDEHIP_ERR_ARG, "kdehip_make_density: null pointer");

  const size_t nd = static_cast<size_t>(2 * N * D);
  std::memset(centers, 0, nd * sizeof(double));
  std::memset(ranges, 0, nd * sizeof(double));
  std::memset(means, 0, nd * sizeof(double));
  std::memset(bandwidth, 0, nd * sizeof(double));
  // internal slots: weights 0, children/leaf bounds 1, permutation 0 (makeBallTree, :447-456)
  for (int64_t i = 0; i < N; ++i) {
    weights[i] = 0.0;
    left_child[i] = right_child[i] = lowest_leaf[i] = highest_leaf[i] = 1;
    permutation[i] = 0;
  }
  double total = 0.0;
  for (int64_t i = 0; i < N; ++i) total += weights_in ? weights_in[i] : 1.0;
  for (int64_t i = 0; i < N; ++i) {
    const int64_t id = N + 1 + i;  // leaf of input point i (buildTree!, :419-429)
    weights[id - 1] = (weights_in ? weights_in[i] : 1.0) / total;
    left_child[id - 1] = lowest_leaf[id - 1] = highest_leaf[id - 1] = id;
    right_child[id - 1] = -1;
    permutation[id - 1] = i + 1;
    for (int64_t k = 0; k < D; ++k) {
      const double x = points[i * D + k];
      const double sd = (nks == 1) ? ks[0] : ks[k];
      const double v = sd * sd;  // ks.^2, KDE01.jl:45
      centers[(id - 1) * D + k] = x;
      means[(id - 1) * D + k] = x;
      bandwidth[(id - 1) * D + k] = v;
      bandwidthMin[i * D + k] = v;
      bandwidthMax[i * D + k] = v;
    }
  }
  DensityBuilder(D, N, centers, ranges, weights, left_child, right_child, lowest_leaf, highest_leaf,
                 permutation, means, bandwidth)
      .build();
  return KDEHIP_OK;
}
