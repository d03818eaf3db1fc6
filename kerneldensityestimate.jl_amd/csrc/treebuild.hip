// treebuild.hip -- construction of a BallTreeDensity on the GPU: the device form of
// balltree.cpp, bit-identical to it (and so to the reference's kde!(points, ks, weights), src/KDE01.jl:34-57 ->
// makeBallTreeDensity, src/BallTreeDensity01.jl:192-231 -> buildTree!/buildBall!, src/BallTree01.jl:342-434):
// same node numbering, same leaf order, same statistics.
//
// One workgroup (16 wavefronts) builds one density, level-synchronously: the leaf ranges of one depth are
// independent, so they are processed side by side, and a batch of densities occupies as many CUs as it has members.
// Per depth:
//   * widest dimension of every range (most_spread_coord, :142-173): the reference's own SEQUENTIAL sums (the argmax
//     decides the split, so the summation order is part of the contract) -- one lane per (range, dimension) walks the
//     range through an LDS copy of the points in current leaf order;
//   * quick-select around the median (select!, :223-242): the reference's single forward scan ("if less than the
//     pivot: ++store, swap(store, i)") moves the not-less elements like a queue whose front goes to the back at every
//     less element.  Written as a tape (push = append, rotate = append a copy of the entry at the head), the final
//     arrangement is a pointer chase that pointer jumping resolves in a few parallel rounds -- the exact permutation
//     of the sequential scan, computed by a wavefront.  Ranges of at most kSeqMax leaves replay the scan itself, one
//     lane per range;
//   * child ids in closed form (the reference hands them out depth first; the subtree sizes are known from the
//     range lengths), children of two or more leaves become the ranges of the next depth.
// Then the leaves are written in their final order and the node statistics (calcStatsBall!, :282-336;
// calcStatsDensity!, src/BallTreeDensity01.jl:141-187) are computed bottom-up, one depth at a time.
//
// Compiled with -ffp-contract=off (the moment-matching expressions must not be fused).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstring>
#include <string>
#include <vector>

#include "kdehip_internal.hpp"
#include "phase_timer.hpp"

namespace kdehip {
namespace {

#define KDEHIP_CHECK(expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return set_error(KDEHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int kTB = 1024;    // threads of the workgroup
constexpr int kSeqMax = 32;  // ranges up to this many leaves: one lane replays the scan
constexpr int kMaxDepth = 40;

struct TreeJob {
  int64_t N;
  int D;
  const double *pts;    // [N][D], input order
  const double *wnorm;  // [N] normalised weights
  double var[KDEHIP_MAX_DIMS];
  double *centers, *ranges, *weights, *means, *bw;  // [2N][D] / [2N]
  int64_t *left, *right, *lo, *hi, *perm;           // [2N]
  int32_t *nodes_by_depth;                          // [N] scratch
};
struct TreeBatch { TreeJob job[KDEHIP_MAX_DENS]; };

// A leaf range [first, last] (0-based leaf positions) with the id of its node and the next free internal id at the
// time buildBall! enters it.
struct Range { uint16_t first, last, node, next; };

// LDS carve-up for (N, D); sizes in bytes
struct TreeLds {
  size_t off_u, off_keys, off_slot, off_cur, off_nxt, off_dim, total;
  size_t u_bytes;
};
inline TreeLds tree_lds(int64_t N, int D) {
  auto al = [](size_t x) { return (x + 15) & ~static_cast<size_t>(15); };
  TreeLds l;
  const size_t pts_bytes = static_cast<size_t>(N) * D * 8;
  const size_t sel_bytes = al(static_cast<size_t>(N) * 8) + al(static_cast<size_t>(N) * 2) + al(static_cast<size_t>(N) * 4) * 2;
  l.u_bytes = al(pts_bytes > sel_bytes ? pts_bytes : sel_bytes);
  l.off_u = 0;
  l.off_keys = l.off_u + l.u_bytes;
  l.off_slot = l.off_keys + al(static_cast<size_t>(N) * 8);
  l.off_cur = l.off_slot + al(static_cast<size_t>(N) * 2);
  l.off_nxt = l.off_cur + al((static_cast<size_t>(N) / 2 + 1) * sizeof(Range));
  l.off_dim = l.off_nxt + al((static_cast<size_t>(N) / 2 + 1) * sizeof(Range));
  l.total = l.off_dim + al(static_cast<size_t>(N) / 2 + 1);
  return l;
}

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- quick-select of one range by one wavefront --------------------------------------------------------------
// K/S: keys and slots of the range (LDS, indexed from the range start), K2/S2: scratch of the same size,
// tp/tq: tape arrays (parent pointer, element) of 2n entries.  pos = target position (the median).
__device__ void wave_quick_select(double *K, uint16_t *S, double *K2, uint16_t *S2, uint16_t *tp, uint16_t *tq, int n,
                                  int pos, int lane) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int r = (lo + hi) >> 1;  // floor((low+high)/2), BallTree01.jl:228 (the range start shifts both ids alike)
    if (lane == 0) {
      const double tk = K[r]; K[r] = K[lo]; K[lo] = tk;
      const uint16_t ts = S[r]; S[r] = S[lo]; S[lo] = ts;
    }
    wave_fence();
    const double pivot = K[lo];
    const int m = hi - lo;  // scanned elements after the pivot: e = 1..m at position lo + e
    // pass 1: totals of "less" / "not less" and of effective rotations (a less element met while the queue of
    // not-less elements is non-empty)
    int Ltot = 0, first_ge = m + 1;
    for (int e0 = 1; e0 <= m; e0 += 64) {
      const int e = e0 + lane;
      const bool in = e <= m;
      const bool less = in && (K[lo + e] - pivot < 0.0);
      const unsigned long long bl = __ballot(less), bg = __ballot(in && !less);
      if (first_ge > m && bg) first_ge = e0 + (__ffsll(static_cast<long long>(bg)) - 1);
      Ltot += __popcll(bl);
    }
    const int Gtot = m - Ltot;
    // effective rotations = less elements after the first not-less one (every element in front of it is less)
    const int Rtot = (Gtot > 0) ? Ltot - (first_ge - 1) : 0;
    const int T = Gtot + Rtot;
    // pass 2: placement of the less elements (stable: rank j -> position lo + j; the last one ends at lo through the
    // final swap with the pivot) and the tape of the not-less ones
    int Lrun = 0, Grun = 0, Rrun = 0;
    for (int e0 = 1; e0 <= m; e0 += 64) {
      const int e = e0 + lane;
      const bool in = e <= m;
      const double key = in ? K[lo + e] : 0.0;
      const uint16_t sl = in ? S[lo + e] : 0;
      const bool less = in && (key - pivot < 0.0);
      const bool ge = in && !less;
      const bool rot = less && e > first_ge;
      const unsigned long long bl = __ballot(less), bg = __ballot(ge), br = __ballot(rot);
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int L = Lrun + __popcll(bl & below) + (less ? 1 : 0);  // inclusive counts at this element
      const int G = Grun + __popcll(bg & below) + (ge ? 1 : 0);
      const int R = Rrun + __popcll(br & below) + (rot ? 1 : 0);
      if (less) {
        const int dst = (L < Ltot) ? lo + L : lo;
        K2[dst] = key;
        S2[dst] = sl;
      }
      if (ge) { const int idx = G + R - 1; tp[idx] = static_cast<uint16_t>(idx); tq[idx] = static_cast<uint16_t>(e); }
      if (rot) { const int idx = G + R - 1; tp[idx] = static_cast<uint16_t>(R - 1); }
      Lrun += __popcll(bl);
      Grun += __popcll(bg);
      Rrun += __popcll(br);
    }
    wave_fence();
    // the queue in its final order goes behind the pivot: every surviving tape entry Q[Rtot..T) is followed back to
    // the pushed element it is a copy of (a rotation's source lies in front of it; chains are a dozen hops at most
    // for random keys, since every hop skips the rotations in between)
    for (int q0 = Rtot; q0 < T; q0 += 64) {
      const int q = q0 + lane;
      if (q < T) {
        int p = q;
        uint16_t pp = tp[p];
        while (pp != p) { p = pp; pp = tp[p]; }
        const int e = tq[p];
        const int dst = lo + Ltot + 1 + (q - Rtot);
        K2[dst] = K[lo + e];
        S2[dst] = S[lo + e];
      }
    }
    if (lane == 0) { K2[lo + Ltot] = pivot; S2[lo + Ltot] = S[lo]; }  // (Ltot = 0: the pivot stays where it is)
    wave_fence();
    for (int i0 = lo; i0 <= hi; i0 += 64) {
      const int i = i0 + lane;
      if (i <= hi) { K[i] = K2[i]; S[i] = S2[i]; }
    }
    wave_fence();
    const int store = lo + Ltot;
    const int nlo = (store <= pos) ? store + 1 : lo;
    const int nhi = (store >= pos) ? store - 1 : hi;
    lo = nlo;
    hi = nhi;
  }
}

// ---- the same scan replayed by one lane (small ranges); identical to balltree.cpp's quick_select ----------------
__device__ void lane_quick_select(double *K, uint16_t *S, int n, int pos) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int r = (lo + hi) >> 1;
    { const double tk = K[r]; K[r] = K[lo]; K[lo] = tk; const uint16_t ts = S[r]; S[r] = S[lo]; S[lo] = ts; }
    const double pivot = K[lo];
    int store = lo;
    for (int i = lo; i <= hi; ++i) {
      if (K[i] - pivot < 0.0) {
        ++store;
        const double tk = K[store]; K[store] = K[i]; K[i] = tk;
        const uint16_t ts = S[store]; S[store] = S[i]; S[i] = ts;
      }
    }
    { const double tk = K[lo]; K[lo] = K[store]; K[store] = tk; const uint16_t ts = S[lo]; S[lo] = S[store]; S[store] = ts; }
    if (store <= pos) lo = store + 1;
    if (store >= pos) hi = store - 1;
  }
}

__global__ __launch_bounds__(kTB) void tree_build_kernel(const TreeBatch batch, const TreeLds L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_nr, s_nnext, s_depth_off[kMaxDepth + 1];
#ifdef KDEHIP_TREE_STAMPS
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#define TSTAMP(k) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[k] += t_now - t_last; t_last = t_now; } while (0)
#else
#define TSTAMP(k) do {} while (0)
#endif
  const TreeJob &J = batch.job[blockIdx.x];
  const int N = static_cast<int>(J.N), D = J.D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *U = reinterpret_cast<double *>(smem + L.off_u);
  double *K = reinterpret_cast<double *>(smem + L.off_keys);
  uint16_t *S = reinterpret_cast<uint16_t *>(smem + L.off_slot);
  Range *cur = reinterpret_cast<Range *>(smem + L.off_cur);
  Range *nxt = reinterpret_cast<Range *>(smem + L.off_nxt);
  uint8_t *dimv = smem + L.off_dim;
  // select-phase view of the union region
  auto al = [](size_t x) { return (x + 15) & ~static_cast<size_t>(15); };
  double *K2 = U;
  uint16_t *S2 = reinterpret_cast<uint16_t *>(smem + L.off_u + al(static_cast<size_t>(N) * 8));
  uint16_t *tp = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(S2) + al(static_cast<size_t>(N) * 2));
  uint16_t *tq = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(tp) + al(static_cast<size_t>(N) * 4));

  // ---- makeBallTree / buildTree! initial state (:437-463, :415-434) ----
  for (int i = tid; i < 2 * N * D; i += kTB) { J.centers[i] = 0.0; J.ranges[i] = 0.0; J.means[i] = 0.0; J.bw[i] = 0.0; }
  for (int i = tid; i < N; i += kTB) {
    J.weights[i] = 0.0;
    J.left[i] = J.right[i] = J.lo[i] = J.hi[i] = 1;  // children arrays start as ones, permutation as zeros
    J.perm[i] = 0;
    const int64_t id = N + 1 + i;
    J.left[id - 1] = J.lo[id - 1] = J.hi[id - 1] = id;  // a leaf is its own left child; right = NO_CHILD
    J.right[id - 1] = -1;
    S[i] = static_cast<uint16_t>(i);
  }
  if (tid == 0) {
    cur[0] = Range{0, static_cast<uint16_t>(N - 1), 1, 2};
    s_nr = 1;
    s_depth_off[0] = 0;
  }
  __syncthreads();
  TSTAMP(0);

  int depth = 0;
  for (;; ++depth) {
    const int nr = s_nr;
    if (nr == 0) break;
    if (tid == 0) { s_depth_off[depth + 1] = s_depth_off[depth] + nr; s_nnext = 0; }
    for (int r = tid; r < nr; r += kTB) J.nodes_by_depth[s_depth_off[depth] + r] = cur[r].node;
    // points in current leaf order
    for (int i = tid; i < N * D; i += kTB) {
      const int p = i / D, k = i - p * D;
      U[i] = J.pts[static_cast<int64_t>(S[p]) * D + k];
    }
    __syncthreads();
    TSTAMP(1);
    // widest dimension (most_spread_coord, :142-173): 8 lanes per range, lane k sums dimension k sequentially; the
    // reference leaves the LAST leaf of the range out of both sums while scaling by 1/(last - first)
    for (int r = tid >> 3; r < ((nr + 127) / 128) * 128; r += kTB / 8) {
      const int k = tid & 7;
      double v = -1.0;
      if (r < nr && k < D) {
        const int first = cur[r].first, last = cur[r].last;
        const double scale = 1.0 / static_cast<double>(last - first);
        double m = 0.0;
        const double *x = U + static_cast<size_t>(first) * D + k;
#pragma unroll 8
        for (int i = 0; i < last - first; ++i) m = m + scale * x[static_cast<size_t>(i) * D];
        v = 0.0;
#pragma unroll 8
        for (int i = 0; i < last - first; ++i) {
          const double dlt = x[static_cast<size_t>(i) * D] - m;
          v += dlt * dlt;
        }
      }
      // argmax over the 8 lanes of the group with the reference's scan: best = 0; take k if v_k > best so far
      int best = 0;
      double best_var = 0.0;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const double vk = __shfl(v, (lane & ~7) + kk);
        if (vk > best_var) { best_var = vk; best = kk; }
      }
      if (r < nr && k == 0) dimv[r] = static_cast<uint8_t>(best);
    }
    __syncthreads();
    TSTAMP(2);
    // keys of every range along its dimension
    for (int r = wave; r < nr; r += kTB / 64) {
      const int first = cur[r].first, n = cur[r].last - first + 1, k = dimv[r];
      for (int i = lane; i < n; i += 64) K[first + i] = U[static_cast<size_t>(first + i) * D + k];
    }
    __syncthreads();  // (the union region now belongs to the select phase)
    TSTAMP(3);
    // quick-select around the median leaf (select!, :223-242): large ranges by wavefronts, small ones by lanes
    for (int r = wave; r < nr; r += kTB / 64) {
      const int first = cur[r].first, n = cur[r].last - first + 1;
      if (n > kSeqMax) {
        const int mid = (first + cur[r].last) >> 1;  // floor((low+high)/2) on ids = on positions (same offset twice)
        wave_quick_select(K + first, S + first, K2 + first, S2 + first, tp + 2 * first, tq + 2 * first, n, mid - first, lane);
      }
    }
    for (int r = tid; r < nr; r += kTB) {
      const int first = cur[r].first, n = cur[r].last - first + 1;
      if (n <= kSeqMax) lane_quick_select(K + first, S + first, n, ((first + cur[r].last) >> 1) - first);
    }
    __syncthreads();
    TSTAMP(4);
    // buildBall! bookkeeping (:342-411): child ids handed out left, then right, before either subtree is built; a
    // one-leaf side points straight at the leaf; the left subtree uses (leaves - 2) more ids before the right one
    for (int r = tid; r < nr; r += kTB) {
      const Range R = cur[r];
      const int first = R.first, last = R.last, mid = (first + last) >> 1;
      const int nl = mid - first + 1, nrr = last - mid;
      int nxtid = R.next;
      const int64_t leaf_lo = static_cast<int64_t>(N) + 1 + first, leaf_mid = static_cast<int64_t>(N) + 1 + mid;
      const int64_t leaf_hi = static_cast<int64_t>(N) + 1 + last;
      const int64_t a = (nl >= 2) ? nxtid++ : leaf_lo;
      const int64_t b = (nrr >= 2) ? nxtid++ : leaf_hi;
      J.left[R.node - 1] = a;
      J.right[R.node - 1] = b;
      J.lo[R.node - 1] = leaf_lo;
      J.hi[R.node - 1] = leaf_hi;
      (void)leaf_mid;
      if (nl >= 2) {
        const int slot = atomicAdd(&s_nnext, 1);
        nxt[slot] = Range{static_cast<uint16_t>(first), static_cast<uint16_t>(mid), static_cast<uint16_t>(a), static_cast<uint16_t>(nxtid)};
      }
      if (nrr >= 2) {
        const int slot = atomicAdd(&s_nnext, 1);
        const int after_left = nxtid + (nl >= 2 ? nl - 2 : 0);
        nxt[slot] = Range{static_cast<uint16_t>(mid + 1), static_cast<uint16_t>(last), static_cast<uint16_t>(b), static_cast<uint16_t>(after_left)};
      }
    }
    __syncthreads();
    if (tid == 0) s_nr = s_nnext;
    Range *t = cur; cur = nxt; nxt = t;
    __syncthreads();
    TSTAMP(5);
  }

  // ---- leaves in their final order (buildTree! :419-429 after all swaps) ----
  for (int p = tid; p < N; p += kTB) {
    const int64_t id = static_cast<int64_t>(N) + 1 + p, src = S[p];
    J.weights[id - 1] = J.wnorm[src];
    J.perm[id - 1] = src + 1;
    for (int k = 0; k < D; ++k) {
      const double x = J.pts[src * D + k];
      J.centers[(id - 1) * D + k] = x;
      J.means[(id - 1) * D + k] = x;
      J.bw[(id - 1) * D + k] = J.var[k];
    }
  }
  __syncthreads();
  TSTAMP(6);
  // ---- node statistics bottom-up (calcStatsBall! :282-336, calcStatsDensity! BallTreeDensity01.jl:141-187) ----
  for (int dd = depth - 1; dd >= 0; --dd) {
    const int begin = s_depth_off[dd], end = s_depth_off[dd + 1];
    for (int t = begin + tid; t < end; t += kTB) {
      const int64_t id = J.nodes_by_depth[t];
      const int64_t a = J.left[id - 1], b = J.right[id - 1];
      for (int k = 0; k < D; ++k) {
        const double ca = J.centers[(a - 1) * D + k], ra = J.ranges[(a - 1) * D + k];
        const double cb = J.centers[(b - 1) * D + k], rb = J.ranges[(b - 1) * D + k];
        const double upA = ca + ra, upB = cb + rb, dnA = ca - ra, dnB = cb - rb;
        const double top = (upA > upB) ? upA : upB, bottom = (dnA < dnB) ? dnA : dnB;
        const double half = (top - bottom) / 2.0;
        J.ranges[(id - 1) * D + k] = half;
        J.centers[(id - 1) * D + k] = bottom + half;
      }
      double wa = J.weights[a - 1], wb = J.weights[b - 1];
      J.weights[id - 1] = (a != b) ? wa + wb : wa;
      const double wt = wa + wb + DBL_EPSILON;  // eps(Float64), BallTreeDensity01.jl:161
      wa /= wt;
      wb /= wt;
      for (int k = 0; k < D; ++k) {
        const double ma = J.means[(a - 1) * D + k], mb = J.means[(b - 1) * D + k];
        const double m = wa * ma + wb * mb;
        J.means[(id - 1) * D + k] = m;
        J.bw[(id - 1) * D + k] = wa * (J.bw[(a - 1) * D + k] + ma * ma) + wb * (J.bw[(b - 1) * D + k] + mb * mb) - m * m;
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  TSTAMP(7);
#ifdef KDEHIP_TREE_STAMPS
  if (tid == 0 && blockIdx.x == 0)
    printf("tree stamps (shader cycles): init %llu | gather %llu | widest %llu | keys %llu | select %llu | children %llu | leaves %llu | stats %llu | depths %d\n",
           t_acc[0], t_acc[1], t_acc[2], t_acc[3], t_acc[4], t_acc[5], t_acc[6], t_acc[7], depth);
#endif
}

}  // namespace
}  // namespace kdehip

using namespace kdehip;

// LDS bytes one workgroup may declare on gfx950 (160 KiB) minus the kernel's static variables
static constexpr size_t kTreeLdsLimit = 160 * 1024 - 1024;

extern "C" int kdehip_make_density_device_supported(int64_t D, int64_t N) {
  if (D < 1 || D > KDEHIP_MAX_DIMS || N < 2 || N > 16384) return 0;
  return tree_lds(N, static_cast<int>(D)).total <= kTreeLdsLimit ? 1 : 0;
}

extern "C" int kdehip_make_densities_device(int nb, int64_t D, const int64_t *Ns, const double *const *points,
                                            const double *const *ks, int64_t nks, const double *const *weights_in,
                                            double *const *centers, double *const *ranges, double *const *weights,
                                            int64_t *const *left_child, int64_t *const *right_child,
                                            int64_t *const *lowest_leaf, int64_t *const *highest_leaf,
                                            int64_t *const *permutation, double *const *means,
                                            double *const *bandwidth, double *const *bandwidthMin,
                                            double *const *bandwidthMax, int device) {
  if (nb < 1 || nb > KDEHIP_MAX_DENS) return set_error(KDEHIP_ERR_ARG, "batch size outside 1..KDEHIP_MAX_DENS");
  if (!Ns || !points || !ks || !centers || !ranges || !weights || !left_child || !right_child || !lowest_leaf ||
      !highest_leaf || !permutation || !means || !bandwidth || !bandwidthMin || !bandwidthMax)
    return set_error(KDEHIP_ERR_ARG, "kdehip_make_densities_device: null pointer");
  if (nks != 1 && nks != D) return set_error(KDEHIP_ERR_ARG, "kdehip_make_densities_device: ks must have 1 or D entries");
  int64_t maxN = 0;
  for (int j = 0; j < nb; ++j) {
    if (!kdehip_make_density_device_supported(D, Ns[j]))
      return set_error(KDEHIP_ERR_UNSUPPORTED, "density too large for the device builder (use kdehip_make_density)");
    if (Ns[j] > maxN) maxN = Ns[j];
  }
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;

  // one device block and one pinned block for the whole batch:
  //   in:  per density [points N*D | wnorm N]          out: [centers, ranges, means, bw: 2N*D each | weights 2N |
  //        left, right, lo, hi, perm: 2N int64 each] + scratch N int32
  auto al = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  struct Off { size_t pts, wn, cen, rng, mea, bw, w, l, r, lo, hi, pm, scratch, in_end, out_begin, out_end; };
  std::vector<Off> off(nb);
  size_t in_total = 0;
  for (int j = 0; j < nb; ++j) {
    off[j].pts = in_total; in_total = al(in_total + sizeof(double) * Ns[j] * D);
    off[j].wn = in_total; in_total = al(in_total + sizeof(double) * Ns[j]);
  }
  size_t total = in_total;
  const size_t out_begin = total;
  for (int j = 0; j < nb; ++j) {
    const size_t nd = sizeof(double) * 2 * Ns[j] * D, n2 = sizeof(double) * 2 * Ns[j];
    off[j].cen = total; total = al(total + nd);
    off[j].rng = total; total = al(total + nd);
    off[j].mea = total; total = al(total + nd);
    off[j].bw = total; total = al(total + nd);
    off[j].w = total; total = al(total + n2);
    off[j].l = total; total = al(total + n2);
    off[j].r = total; total = al(total + n2);
    off[j].lo = total; total = al(total + n2);
    off[j].hi = total; total = al(total + n2);
    off[j].pm = total; total = al(total + n2);
  }
  const size_t out_end = total;
  for (int j = 0; j < nb; ++j) { off[j].scratch = total; total = al(total + sizeof(int32_t) * Ns[j]); }

  void *d_base = nullptr, *h_base = nullptr;
  KDEHIP_CHECK(cached_malloc(&d_base, total));
  struct Free { void *d, *h; size_t nd, nh; ~Free() { if (d) cached_free(d, nd); if (h) cached_host_free(h, nh); } } fr{d_base, nullptr, total, out_end};
  KDEHIP_CHECK(cached_host_malloc(&h_base, out_end));
  fr.h = h_base;
  unsigned char *hb = static_cast<unsigned char *>(h_base), *db = static_cast<unsigned char *>(d_base);

  TreeBatch batch{};
  for (int j = 0; j < nb; ++j) {
    const int64_t N = Ns[j];
    std::memcpy(hb + off[j].pts, points[j], sizeof(double) * N * D);
    double *wn = reinterpret_cast<double *>(hb + off[j].wn);
    const double *win = weights_in ? weights_in[j] : nullptr;
    double tot = 0.0;
    for (int64_t i = 0; i < N; ++i) tot += win ? win[i] : 1.0;
    for (int64_t i = 0; i < N; ++i) wn[i] = (win ? win[i] : 1.0) / tot;  // KDE01.jl:46
    TreeJob &J = batch.job[j];
    J.N = N; J.D = static_cast<int>(D);
    J.pts = reinterpret_cast<const double *>(db + off[j].pts);
    J.wnorm = reinterpret_cast<const double *>(db + off[j].wn);
    for (int64_t k = 0; k < D; ++k) {
      const double sd = (nks == 1) ? ks[j][0] : ks[j][k];
      J.var[k] = sd * sd;  // ks.^2, KDE01.jl:45
    }
    J.centers = reinterpret_cast<double *>(db + off[j].cen);
    J.ranges = reinterpret_cast<double *>(db + off[j].rng);
    J.means = reinterpret_cast<double *>(db + off[j].mea);
    J.bw = reinterpret_cast<double *>(db + off[j].bw);
    J.weights = reinterpret_cast<double *>(db + off[j].w);
    J.left = reinterpret_cast<int64_t *>(db + off[j].l);
    J.right = reinterpret_cast<int64_t *>(db + off[j].r);
    J.lo = reinterpret_cast<int64_t *>(db + off[j].lo);
    J.hi = reinterpret_cast<int64_t *>(db + off[j].hi);
    J.perm = reinterpret_cast<int64_t *>(db + off[j].pm);
    J.nodes_by_depth = reinterpret_cast<int32_t *>(db + off[j].scratch);
  }
  KDEHIP_CHECK(hipMemcpyAsync(d_base, h_base, in_total, hipMemcpyHostToDevice, hipStreamPerThread));
  const TreeLds L = tree_lds(maxN, static_cast<int>(D));
  // (per call: the attribute belongs to the function ON THE CURRENT DEVICE, and concurrent host threads get here)
  KDEHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(tree_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(kTreeLdsLimit)));
  PhaseTimer timer(kPhaseTreeBuild, hipStreamPerThread);  // (kdehip_profile_phase_read(2): the kernel alone)
  hipLaunchKernelGGL(tree_build_kernel, dim3(nb), dim3(kTB), L.total, hipStreamPerThread, batch, L);
  KDEHIP_CHECK(hipGetLastError());
  timer.stop();
  KDEHIP_CHECK(hipMemcpyAsync(hb + out_begin, db + out_begin, out_end - out_begin, hipMemcpyDeviceToHost, hipStreamPerThread));
  KDEHIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
  timer.collect();
  for (int j = 0; j < nb; ++j) {
    const int64_t N = Ns[j];
    const size_t nd = sizeof(double) * 2 * N * D, n2 = sizeof(double) * 2 * N;
    std::memcpy(centers[j], hb + off[j].cen, nd);
    std::memcpy(ranges[j], hb + off[j].rng, nd);
    std::memcpy(means[j], hb + off[j].mea, nd);
    std::memcpy(bandwidth[j], hb + off[j].bw, nd);
    std::memcpy(weights[j], hb + off[j].w, n2);
    std::memcpy(left_child[j], hb + off[j].l, n2);
    std::memcpy(right_child[j], hb + off[j].r, n2);
    std::memcpy(lowest_leaf[j], hb + off[j].lo, n2);
    std::memcpy(highest_leaf[j], hb + off[j].hi, n2);
    std::memcpy(permutation[j], hb + off[j].pm, n2);
    for (int64_t i = 0; i < N; ++i)
      for (int64_t k = 0; k < D; ++k) bandwidthMin[j][i * D + k] = bandwidthMax[j][i * D + k] = batch.job[j].var[k];
  }
  return KDEHIP_OK;
}
