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
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <string>
#include <utility>
#include <vector>

#include "../../include/kdehip.h"
#include "device_density.hpp"
#include "host_pool.hpp"
#include "kdehip_internal.hpp"

namespace kdehip {
namespace {

// The reference moves every leaf's payload (centre, mean, bandwidth, weight, permutation) on each
// quick-select swap.  All of it is a function of "which input point sits in which leaf slot", so
// this builder permutes only an index array `slot_` (leaf slot -> input point) with the reference's
// exact swap sequence, materialises the leaves once, and then computes the node statistics in the
// reference's post-order.  Same arrays, bit for bit, at a fraction of the memory traffic.
//
// Subtrees are independent once their node's quick-select is done: the leaf range, and -- because the reference hands
// out ids left subtree first and a subtree of n leaves holds n-1 internal nodes -- the id range of each side are known
// before either is built (the same closed form the device builder uses, treebuild.hip).  The top two or three levels
// of a large density therefore build their left side on another host thread; every node is still computed by the same
// expressions from the same operands, so the arrays do not depend on the number of threads (host_pool.hpp).
class DensityBuilder {
 public:
  DensityBuilder(int64_t D, int64_t N, const double *points, double *centers, double *ranges,
                 double *weights, int64_t *left, int64_t *right, int64_t *lo, int64_t *hi, int64_t *perm,
                 double *means, double *bw)
      : D_(D), N_(N), pts_(points), centers_(centers), ranges_(ranges), weights_(weights), left_(left),
        right_(right), lo_(lo), hi_(hi), perm_(perm), means_(means), bw_(bw) {
    slot_.resize(static_cast<size_t>(N));
    for (int64_t i = 0; i < N; ++i) slot_[static_cast<size_t>(i)] = i;  // buildTree!, :419-429
  }

  // wnorm: normalised weight of every input point; var: the D leaf variances
  void build(const double *wnorm, const double *var) {
    wnorm_ = wnorm;
    leaf_var_ = var;
    Scratch s(D_);
    int depth = 0;
    if (N_ >= kSplitLeaves) {  // (small densities never start the workers)
      const int w = HostPool::get().workers();
      depth = w >= 15 ? 4 : (w >= 7 ? 3 : (w >= 3 ? 2 : (w >= 1 ? 1 : 0)));
    }
    subtree(N_ + 1, 2 * N_, 1, 2, depth, s);
    if (N_ == 1) right_[0] = -1;  // single-point density, :358-360
  }

 private:
  static constexpr int64_t kSplitLeaves = 512;  // a node with fewer leaves is built by the thread that reached it

  struct Scratch {  // per host thread
    explicit Scratch(int64_t D) : acc(static_cast<size_t>(2 * D)) {}
    std::vector<double> acc, keys;
    std::vector<int64_t> order;
  };

  // the leaves first..last in their final order
  void materialise(int64_t first, int64_t last) {
    for (int64_t id = first; id <= last; ++id) {
      const int64_t src = slot_[static_cast<size_t>(id - N_ - 1)];
      weights_[id - 1] = wnorm_[src];
      perm_[id - 1] = src + 1;
      for (int64_t k = 0; k < D_; ++k) {
        const double x = pts_[src * D_ + k];
        centers_[(id - 1) * D_ + k] = x;
        means_[(id - 1) * D_ + k] = x;
        bw_[(id - 1) * D_ + k] = leaf_var_[k];
      }
    }
  }

  // Everything below internal node `id` (leaves first..last; `next` = the first id its descendants take): topology,
  // leaves, statistics.  `depth` = how many more levels may hand their left side to another thread.
  void subtree(int64_t first, int64_t last, int64_t id, int64_t next, int depth, Scratch &s) {
    if (depth <= 0 || last - first + 1 < kSplitLeaves) {
      s.order.clear();
      build_node(first, last, id, next, s);
      materialise(first, last);
      const std::vector<int64_t> order = s.order;  // (summarize does not touch the scratch; the copy keeps that obvious)
      for (int64_t node : order) summarize(node);
      return;
    }
    lo_[id - 1] = first;
    hi_[id - 1] = last;
    const int64_t k = widest_dim(first, last, s);
    const int64_t mid = (first + last) / 2;
    quick_select(k, mid, first, last, s);
    const int64_t a = next++, b = next++;      // (kSplitLeaves >= 4: both sides are internal nodes)
    left_[id - 1] = a;
    right_[id - 1] = b;
    const int64_t next_right = next + (mid - first + 1) - 2;  // the left side's n-1 internal nodes, `a` among them
    HostPool &pool = HostPool::get();
    HostPool::Ticket left = pool.submit([=] {
      Scratch mine(D_);
      subtree(first, mid, a, next, depth - 1, mine);
    });
    try {
      subtree(mid + 1, last, b, next_right, depth - 1, s);
    } catch (...) {
      try { pool.join(left); } catch (...) {}  // (the left side must not outlive this frame)
      throw;
    }
    pool.join(left);  // (runs it here if no worker has started it)
    summarize(id);
  }

  // coordinate k of the point currently in leaf `id` (N+1 .. 2N)
  double key(int64_t id, int64_t k) const { return pts_[slot_[static_cast<size_t>(id - N_ - 1)] * D_ + k]; }
  void exchange(int64_t a, int64_t b) {  // swapBall!/swapDensity!, BallTree01.jl:109-138
    std::swap(slot_[static_cast<size_t>(a - N_ - 1)], slot_[static_cast<size_t>(b - N_ - 1)]);
  }
  double *ctr(int64_t id) { return centers_ + (id - 1) * D_; }
  double *rng(int64_t id) { return ranges_ + (id - 1) * D_; }
  double *mu(int64_t id) { return means_ + (id - 1) * D_; }
  double *var(int64_t id) { return bw_ + (id - 1) * D_; }
  bool valid(int64_t id) const { return id > 0 && id <= 2 * N_; }  // BallTree01.jl:83

  // Dimension of largest spread over leaves first..last (most_spread_coord, BallTree01.jl:142-173).
  // The reference leaves the last leaf out of both sums while scaling by 1/(last-first); ties and
  // the all-equal case resolve to the lowest dimension (strict '>').
  int64_t widest_dim(int64_t first, int64_t last, Scratch &s) {
    // every dimension keeps the reference's own sequential sums; the dimensions are interleaved in
    // the inner loop only to give the CPU D independent dependency chains over contiguous memory
    const double scale = 1.0 / static_cast<double>(last - first);
    double *m = s.acc.data(), *v = s.acc.data() + D_;
    for (int64_t k = 0; k < D_; ++k) m[k] = v[k] = 0.0;
    for (int64_t id = first; id < last; ++id) {
      const double *x = pts_ + slot_[static_cast<size_t>(id - N_ - 1)] * D_;
      for (int64_t k = 0; k < D_; ++k) m[k] = m[k] + scale * x[k];
    }
    for (int64_t id = first; id < last; ++id) {
      const double *x = pts_ + slot_[static_cast<size_t>(id - N_ - 1)] * D_;
      for (int64_t k = 0; k < D_; ++k) {
        const double dlt = x[k] - m[k];
        v[k] += dlt * dlt;
      }
    }
    int64_t best = 0;
    double best_var = 0.0;
    for (int64_t k = 0; k < D_; ++k)
      if (v[k] > best_var) { best_var = v[k]; best = k; }
    return best;
  }

  // Quick-select (select!, BallTree01.jl:223-242): afterwards leaves first..pos are <= those after.
  // The scan is the reference's single forward pass ("if less than the pivot: ++store, swap(store, i)"),
  // written branch-free on a contiguous copy of the keys: a not-less element swaps with itself.
  void quick_select(int64_t k, int64_t pos, int64_t first, int64_t last, Scratch &s) {
    if (first >= last) return;
    const int64_t base = first;
    const int64_t n0 = last - first + 1;
    if (static_cast<int64_t>(s.keys.size()) < n0) s.keys.resize(static_cast<size_t>(n0));
    double *kk = s.keys.data();                                  // kk[i]  = key of leaf base+i
    int64_t *sl = slot_.data() + (base - N_ - 1);                // sl[i]  = input point in leaf base+i
    for (int64_t i = 0; i < n0; ++i) kk[i] = pts_[sl[i] * D_ + k];
    int64_t lo = 0, hi = n0 - 1;
    const int64_t p = pos - base;
    while (lo < hi) {
      const int64_t r = (lo + base + hi + base) / 2 - base;      // floor((low+high)/2) on 1-based ids
      std::swap(kk[r], kk[lo]);
      std::swap(sl[r], sl[lo]);
      const double pivot = kk[lo];
      int64_t store = lo;
      for (int64_t i = lo; i <= hi; ++i) {
        const bool lt = (kk[i] - pivot < 0.0);
        store += lt ? 1 : 0;
        const double ka = kk[store], kb = kk[i];
        const int64_t sa = sl[store], sb = sl[i];
        kk[store] = lt ? kb : ka;
        kk[i] = lt ? ka : kb;
        sl[store] = lt ? sb : sa;
        sl[i] = lt ? sa : sb;
      }
      std::swap(kk[lo], kk[store]);
      std::swap(sl[lo], sl[store]);
      if (store <= p) lo = store + 1;
      if (store >= p) hi = store - 1;
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
  // subtree is built; a one-leaf side points straight at the leaf.  Statistics are deferred: nodes
  // are recorded in the order the reference computes them (children before parents).  `next`: the
  // next free id, shared by the whole recursion below one subtree() call.
  void build_node(int64_t first, int64_t last, int64_t id, int64_t &next, Scratch &s) {
    lo_[id - 1] = first;
    hi_[id - 1] = last;
    if (first == last) {  // single-point density, :351-362 (right child fixed up after the stats)
      left_[id - 1] = first;
      right_[id - 1] = last;
      s.order.push_back(id);
      return;
    }
    const int64_t k = widest_dim(first, last, s);
    const int64_t mid = (first + last) / 2;
    quick_select(k, mid, first, last, s);
    const int64_t a = (mid <= first) ? first : next++;
    const int64_t b = (mid + 1 >= last) ? last : next++;
    left_[id - 1] = a;
    right_[id - 1] = b;
    if (a != first) build_node(first, mid, a, next, s);
    if (b != last) build_node(mid + 1, last, b, next, s);
    s.order.push_back(id);
  }

  const int64_t D_, N_;
  const double *pts_;
  double *centers_, *ranges_, *weights_;
  int64_t *left_, *right_, *lo_, *hi_, *perm_;
  double *means_, *bw_;
  const double *wnorm_ = nullptr, *leaf_var_ = nullptr;
  std::vector<int64_t> slot_;
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
  std::vector<double> wnorm(static_cast<size_t>(N)), var(static_cast<size_t>(D));
  for (int64_t i = 0; i < N; ++i) wnorm[static_cast<size_t>(i)] = (weights_in ? weights_in[i] : 1.0) / total;  // KDE01.jl:46
  for (int64_t k = 0; k < D; ++k) {
    const double sd = (nks == 1) ? ks[0] : ks[k];
    var[static_cast<size_t>(k)] = sd * sd;  // ks.^2, KDE01.jl:45
  }
  for (int64_t i = 0; i < N; ++i) {
    const int64_t id = N + 1 + i;  // leaf slots (buildTree!, :419-429)
    left_child[id - 1] = lowest_leaf[id - 1] = highest_leaf[id - 1] = id;
    right_child[id - 1] = -1;
    for (int64_t k = 0; k < D; ++k) {
      bandwidthMin[i * D + k] = var[static_cast<size_t>(k)];
      bandwidthMax[i * D + k] = var[static_cast<size_t>(k)];
    }
  }
  try {
    DensityBuilder(D, N, points, centers, ranges, weights, left_child, right_child, lowest_leaf, highest_leaf,
                   permutation, means, bandwidth)
        .build(wnorm.data(), var.data());
  } catch (const std::exception &e) {  // (scratch allocation on this or a worker thread: nothing may cross the C boundary)
    return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_make_density: ") + e.what());
  }
  return KDEHIP_OK;
}

// The bandwidth-dependent half of kde!(points, ks, weights) on an EXISTING tree: the topology (splits, leaf order),
// the bounding boxes, the weights and the means of makeBallTreeDensity do not depend on the bandwidth -- only
// `bandwidth` (leaf: ks^2, src/KDE01.jl:45; internal: the moment matching of calcStatsDensity!,
// src/BallTreeDensity01.jl:141-187) and bandwidthMin/Max do.  `kde!(points)` can therefore build its tree while the
// GPU is still searching the LOOCV bandwidth, and fill the variances in afterwards: the same expressions on the same
// operands in an order that keeps children before parents, i.e. bit-identical to kdehip_make_density with that ks.
extern "C" int kdehip_density_set_bandwidth(int64_t D, int64_t N, const double *ks, int64_t nks, const double *weights,
                                            const int64_t *left_child, const int64_t *right_child, const double *means,
                                            double *bandwidth, double *bandwidthMin, double *bandwidthMax) {
  return kdehip::set_bandwidth_examined(D, N, ks, nks, weights, left_child, right_child, means, bandwidth, bandwidthMin,
                                        bandwidthMax, nullptr, nullptr);
}

// internal nodes 1 .. N-1 in pre-order from the root (read back to front: every node after its descendants)
int kdehip::children_first_order(int64_t N, const int64_t *left_child, const int64_t *right_child, std::vector<int64_t> &order) {
  std::vector<int64_t> stack;
  stack.reserve(64);
  order.clear();
  order.reserve(static_cast<size_t>(N));
  if (N < 2) return KDEHIP_OK;
  stack.push_back(1);
  while (!stack.empty()) {
    const int64_t id = stack.back();
    stack.pop_back();
    if (id < 1 || id >= N) return set_error(KDEHIP_ERR_ARG, "kdehip_density_set_bandwidth: malformed tree");
    order.push_back(id);
    const int64_t a = left_child[id - 1], b = right_child[id - 1];
    if (a <= N && a >= 1 && a != id) stack.push_back(a);
    if (b <= N && b >= 1 && b != id) stack.push_back(b);
    if (static_cast<int64_t>(order.size()) > N) return set_error(KDEHIP_ERR_ARG, "kdehip_density_set_bandwidth: malformed tree");
  }
  return KDEHIP_OK;
}

// The same, and -- on the values it walks through anyway -- what examine_frontiers' `look` would find on this tree:
// the range of the variances per dimension and whether every mean, variance and weight is fit for the fast arithmetic
// form (every node of a well-formed tree is on some frontier, so "all nodes" is "all frontier nodes").
int kdehip::set_bandwidth_examined(int64_t D, int64_t N, const double *ks, int64_t nks, const double *weights,
                                   const int64_t *left_child, const int64_t *right_child, const double *means,
                                   double *bandwidth, double *bandwidthMin, double *bandwidthMax, NodeStats *st,
                                   const std::vector<int64_t> *order_in) {
  if (D < 1 || N < 1) return set_error(KDEHIP_ERR_ARG, "kdehip_density_set_bandwidth: need D >= 1 and N >= 1");
  if (nks != 1 && nks != D) return set_error(KDEHIP_ERR_ARG, "kdehip_density_set_bandwidth: ks must have 1 or D entries");
  if (!ks || !weights || !left_child || !right_child || !means || !bandwidth || !bandwidthMin || !bandwidthMax)
    return set_error(KDEHIP_ERR_ARG, "kdehip_density_set_bandwidth: null pointer");
  std::vector<double> var(static_cast<size_t>(D));
  for (int64_t k = 0; k < D; ++k) {
    const double sd = (nks == 1) ? ks[0] : ks[k];
    var[static_cast<size_t>(k)] = sd * sd;  // ks.^2, KDE01.jl:45
  }
  for (int64_t i = 0; i < N; ++i)
    for (int64_t k = 0; k < D; ++k) {
      bandwidth[(N + i) * D + k] = var[static_cast<size_t>(k)];
      bandwidthMin[i * D + k] = var[static_cast<size_t>(k)];
      bandwidthMax[i * D + k] = var[static_cast<size_t>(k)];
    }
  if (N == 1) {  // single-point density (BallTree01.jl:351-362): the root summarises the leaf with itself
    const double wa0 = weights[N], wt = wa0 + wa0 + DBL_EPSILON;
    const double wa = wa0 / wt, wb = wa0 / wt;
    for (int64_t k = 0; k < D; ++k) {
      const double ma = means[N * D + k], m = wa * ma + wb * ma;
      bandwidth[k] = wa * (bandwidth[N * D + k] + ma * ma) + wb * (bandwidth[N * D + k] + ma * ma) - m * m;
    }
    return KDEHIP_OK;
  }
  // internal nodes 1 .. N-1, children before parents
  std::vector<int64_t> own;
  if (!order_in) {
    const int rc = children_first_order(N, left_child, right_child, own);
    if (rc != KDEHIP_OK) return rc;
  }
  const std::vector<int64_t> &order = order_in ? *order_in : own;
  // (with `st`: the range of the variances is kept on the values as they are formed)
  double vlo[KDEHIP_MAX_DIMS], vhi[KDEHIP_MAX_DIMS];
  for (int64_t k = 0; k < KDEHIP_MAX_DIMS; ++k) { vlo[k] = k < D ? var[static_cast<size_t>(k)] : INFINITY; vhi[k] = k < D ? var[static_cast<size_t>(k)] : 0.0; }
  bool vbad = false;
  for (size_t t = order.size(); t-- > 0;) {  // reverse pre-order: every node after its descendants
    const int64_t id = order[t];
    const int64_t a = left_child[id - 1], b = right_child[id - 1];
    if (a < 1 || a > 2 * N || b < 1 || b > 2 * N) continue;
    double wa = weights[a - 1], wb = weights[b - 1];
    const double wt = wa + wb + DBL_EPSILON;  // eps(Float64), BallTreeDensity01.jl:161
    wa /= wt;
    wb /= wt;
    for (int64_t k = 0; k < D; ++k) {
      const double ma = means[(a - 1) * D + k], mb = means[(b - 1) * D + k];
      const double m = wa * ma + wb * mb;
      const double v = wa * (bandwidth[(a - 1) * D + k] + ma * ma) + wb * (bandwidth[(b - 1) * D + k] + mb * mb) - m * m;
      bandwidth[(id - 1) * D + k] = v;
      if (st && k < KDEHIP_MAX_DIMS) {
        vbad |= !(v > 0.0) | !(v < INFINITY);
        vlo[k] = v < vlo[k] ? v : vlo[k];
        vhi[k] = v > vhi[k] ? v : vhi[k];
      }
    }
  }
  if (st) {
    // what is left of the examination: every mean below 1e100 in magnitude, every weight finite and not negative (flat
    // sweeps over contiguous arrays: the nodes 1 .. N-1 and N+1 .. 2N; slot N is not a node), the leaf variances
    bool bad = vbad;
    for (int64_t k = 0; k < D; ++k) bad = bad || !(var[static_cast<size_t>(k)] > 0.0) || !(var[static_cast<size_t>(k)] < INFINITY);
    // (the leaves decide: an internal node's mean is a convex combination of leaf means -- it cannot leave their range
    // by more than rounding, and a NaN or an infinity among them reaches the root -- and its weight a sum of leaf weights)
    double macc = 0.0;
    bool nan = false;
    for (int64_t i = N * D; i < 2 * N * D; ++i) {
      const double a = std::fabs(means[i]);
      nan |= !(a == a);
      macc = a > macc ? a : macc;
    }
    for (int64_t k = 0; k < D; ++k) { const double a = std::fabs(means[k]); nan |= !(a == a); macc = a > macc ? a : macc; }  // the root
    bad = bad || nan || !(macc < 1e100);
    for (int64_t id = N + 1; id <= 2 * N; ++id) {
      const double w = weights[id - 1];
      bad |= !(w >= 0.0) | !(w < INFINITY);
    }
    bad = bad || !(weights[0] >= 0.0) || !(weights[0] < INFINITY);
    st->bad = bad;
    for (int k = 0; k < KDEHIP_MAX_DIMS; ++k) { st->lo[k] = vlo[k]; st->hi[k] = vhi[k]; }
  }
  return KDEHIP_OK;
}
