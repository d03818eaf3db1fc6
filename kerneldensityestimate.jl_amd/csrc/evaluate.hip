// evaluate.hip -- direct KDE evaluation and LOOCV bandwidth selection on gfx950.
//
// Replaces, for the default configuration of the reference (FORCE_EVAL_DIRECT = true,
// src/KernelDensityEstimate.jl:54, so `evaluate` always ends in `evalDirect`):
//   evaluateDualTree(bd, pos) / bd(pos)         src/DualTree01.jl:370-446 -> evaluate :303-346 -> evalDirect :130-162
//   kde!(points)  (automatic bandwidth)         src/KDE01.jl:3-27 -> ksize, golden, nLOO_LL, src/CrossValidation.jl:15-120
// The all-pairs Gaussian sum is the GPU part: one lane per query point, source points staged through
// LDS in chunks and read as broadcasts, partial sums per (source chunk, query) reduced in a fixed order
// by a second kernel (deterministic, no atomics).  The golden-section search runs on the host and
// advances the D independent 1-D searches of kde!(points) in lock step, one launch per round.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fastexp.hpp"
#include "kdehip_internal.hpp"

namespace kdehip {
namespace {

constexpr int kEvalThreads = 256;  // queries per block
constexpr int kEvalChunk = 128;    // source points per staged chunk
constexpr int kEvalMaxGroups = 64; // at most this many partial sums per query (scratch = 64 * Nq doubles)

// Source chunks are dealt to groups of consecutive chunks: as many groups as it takes to give every CU a few
// blocks (small problems: one chunk per group, the most parallel split), never more than kEvalMaxGroups.
struct GroupSplit { int64_t chunks_per_group; int ngroups; };
inline GroupSplit split_chunks(int64_t N, int64_t Nq, int nprob) {
  const int64_t nchunks = (N + kEvalChunk - 1) / kEvalChunk;
  const int64_t qblocks = ((Nq + kEvalThreads - 1) / kEvalThreads) * (nprob > 0 ? nprob : 1);
  int64_t want = (int64_t(8) * device_cu_count() + qblocks - 1) / qblocks;  // groups for ~8 blocks per CU
  if (want < 1) want = 1;
  if (want > kEvalMaxGroups) want = kEvalMaxGroups;
  if (want > nchunks) want = nchunks;
  GroupSplit g;
  g.chunks_per_group = (nchunks + want - 1) / want;
  g.ngroups = static_cast<int>((nchunks + g.chunks_per_group - 1) / g.chunks_per_group);
  return g;
}

#define KDEHIP_CHECK(expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return set_error(KDEHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct DevBuf {  // device scratch of one call, from the library's allocation cache (devmem.cpp)
  void *p = nullptr;
  size_t n = 0;
  ~DevBuf() { if (p) cached_free(p, n); }
  hipError_t alloc(size_t bytes) { n = bytes ? bytes : 1; return cached_malloc(&p, n); }
  template <typename T> T *as() { return static_cast<T *>(p); }
};

// One problem of a batch: N source points (tree/leaf order), Nq queries.
struct EvalProblem {
  const double *src;   // [N][D]
  const double *w;     // [N]
  const double *qry;   // [Nq][D]
  double *partial;     // [ngroups][Nq]
  double nhib[KDEHIP_MAX_DIMS];  // -1/(2 bw_k)
  int64_t N, Nq;
  int64_t chunks_per_group;  // consecutive 128-point source chunks summed by one block
};

// A launch handles up to KDEHIP_MAX_DIMS independent problems (the D one-dimensional searches of
// kde!(points) in one round); their descriptors travel as kernel arguments, not through memory.
struct EvalBatch { EvalProblem p[KDEHIP_MAX_DIMS]; };

// partial[g][q] = sum over the source chunks c of group g, in chunk order, of
//   sum_{i in chunk c, (i != q if loo)} w_i exp(-1/2 sum_k (x_qk - c_ik)^2 / bw_k)
// (the kernel value of distGauss!, src/DualTree01.jl:14-47, with leaf ranges 0 and uniform bandwidth).
// A block owns kEvalThreads queries and ONE group of consecutive 128-point source chunks, which it walks in
// order with the running sum in a register: the scratch is [ngroups][Nq] with ngroups <= kEvalMaxGroups
// whatever N is (grid.y stays far below the 65535 limit), and the summation order is fixed by (N, ngroups).
template <int D>
__global__ __launch_bounds__(kEvalThreads) void eval_partial_kernel(const EvalBatch batch, int loo) {
  __shared__ double sSrc[2][kEvalChunk * (D + 1)];
  __shared__ double sExpTab[32];
  if (threadIdx.x < 32) sExpTab[threadIdx.x] = kExp2Tab[threadIdx.x];
  const EvalProblem &pb = batch.p[blockIdx.z];
  const int64_t q = static_cast<int64_t>(blockIdx.x) * kEvalThreads + threadIdx.x;
  const int64_t c_begin = static_cast<int64_t>(blockIdx.y) * pb.chunks_per_group;
  int64_t c_end = c_begin + pb.chunks_per_group;
  const int64_t nchunks = (pb.N + kEvalChunk - 1) / kEvalChunk;
  if (c_end > nchunks) c_end = nchunks;
  if (c_begin >= c_end || static_cast<int64_t>(blockIdx.x) * kEvalThreads >= pb.Nq) return;  // block-uniform
  double x[D];
#pragma unroll
  for (int k = 0; k < D; ++k) x[k] = (q < pb.Nq) ? pb.qry[q * D + k] : 0.0;
  auto stage = [&](int64_t c, int buf) {
    const int64_t i0 = c * kEvalChunk;
    const int cnt = static_cast<int>((pb.N - i0 < kEvalChunk) ? (pb.N - i0) : kEvalChunk);
    for (int t = threadIdx.x; t < cnt * (D + 1); t += kEvalThreads) {
      const int i = t / (D + 1), f = t % (D + 1);
      sSrc[buf][t] = (f < D) ? pb.src[(i0 + i) * D + f] : pb.w[i0 + i];
    }
  };
  stage(c_begin, 0);
  double total = 0.0;
  for (int64_t c = c_begin; c < c_end; ++c) {
    const int buf = static_cast<int>((c - c_begin) & 1);
    __syncthreads();  // chunk c is staged; the other buffer is free again
    if (c + 1 < c_end) stage(c + 1, buf ^ 1);
    const int64_t i0 = c * kEvalChunk;
    const int cnt = static_cast<int>((pb.N - i0 < kEvalChunk) ? (pb.N - i0) : kEvalChunk);
    double sum = 0.0;
    for (int i = 0; i < cnt; ++i) {
      const double *s = sSrc[buf] + i * (D + 1);
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double d = x[k] - s[k];
        acc = fma(d * d, pb.nhib[k], acc);
      }
      double v = s[D] * exp_nonpos(acc, sExpTab);  // acc <= 0
      if (loo && i0 + i == q) v = 0.0;  // leave-one-out: skip the self term (:141)
      sum += v;
    }
    total += sum;
  }
  if (q < pb.Nq) pb.partial[static_cast<int64_t>(blockIdx.y) * pb.Nq + q] = total;
}

struct FinishProblem {
  const double *partial;   // [nchunks][Nq] (nchunks = groups of source chunks, see split_chunks)
  const double *w;         // [N] (loo: 1 - w_q)
  const int64_t *out_idx;  // optional: output position of query q (loo: permutation - 1), or null
  double *out;
  double inv_norm;
  int64_t Nq;
  int nchunks;
};

struct FinishBatch { FinishProblem p[KDEHIP_MAX_DIMS]; };

// p[q] = (sum over chunks, in chunk order) / norm [/ (1 - w_q)]   (src/DualTree01.jl:325-340)
__global__ void eval_finish_kernel(const FinishBatch batch, int loo) {
  const FinishProblem &pb = batch.p[blockIdx.y];
  const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= pb.Nq) return;
  double s = 0.0;
  for (int c = 0; c < pb.nchunks; ++c) s += pb.partial[static_cast<int64_t>(c) * pb.Nq + q];
  double p = s * pb.inv_norm;
  if (loo) p = p / (1.0 - pb.w[q]);
  pb.out[pb.out_idx ? pb.out_idx[q] : q] = p;
}

// Leave-one-out finish fused with the entropy reduction of nLOO_LL: block b of problem y writes
// hpart[y][b] = sum over its queries of W_q * log(p_q)  (evalAvgLogL, src/DualTree01.jl:450-474;
// a zero likelihood that carries weight makes the log-likelihood -Inf, :460-463).
constexpr int kFinishThreads = 256;
__global__ __launch_bounds__(kFinishThreads) void loo_entropy_kernel(const FinishBatch batch,
                                                                   double *__restrict__ hpart, int nblocks) {
  __shared__ double red[kFinishThreads];
  const FinishProblem &pb = batch.p[blockIdx.y];
  const int64_t q = static_cast<int64_t>(blockIdx.x) * kFinishThreads + threadIdx.x;
  double term = 0.0;
  if (q < pb.Nq) {
    double s = 0.0;
    for (int c = 0; c < pb.nchunks; ++c) s += pb.partial[static_cast<int64_t>(c) * pb.Nq + q];
    const double w = pb.w[q];
    const double p = s * pb.inv_norm / (1.0 - w);
    if (p == 0.0) term = (w != 0.0) ? -INFINITY : 0.0;
    else term = log(p) * w;
  }
  red[threadIdx.x] = term;
  __syncthreads();
  for (int off = kFinishThreads / 2; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) hpart[blockIdx.y * nblocks + blockIdx.x] = red[0];
}

template <int D>
void launch_partial(const EvalBatch &d_problems, int nprob, int64_t maxNq, int ngroups, int loo,
                    hipStream_t st) {
  dim3 grid(static_cast<unsigned>((maxNq + kEvalThreads - 1) / kEvalThreads),
            static_cast<unsigned>(ngroups), static_cast<unsigned>(nprob));
  hipLaunchKernelGGL((eval_partial_kernel<D>), grid, dim3(kEvalThreads), 0, st, d_problems, loo);
}

int launch_partial_dims(int D, const EvalBatch &d_problems, int nprob, int64_t maxNq, int ngroups, int loo,
                        hipStream_t st) {
  switch (D) {
    case 1: launch_partial<1>(d_problems, nprob, maxNq, ngroups, loo, st); break;
    case 2: launch_partial<2>(d_problems, nprob, maxNq, ngroups, loo, st); break;
    case 3: launch_partial<3>(d_problems, nprob, maxNq, ngroups, loo, st); break;
    case 4: launch_partial<4>(d_problems, nprob, maxNq, ngroups, loo, st); break;
    case 5: launch_partial<5>(d_problems, nprob, maxNq, ngroups, loo, st); break;
    case 6: launch_partial<6>(d_problems, nprob, maxNq, ngroups, loo, st); break;
    case 7: launch_partial<7>(d_problems, nprob, maxNq, ngroups, loo, st); break;
    case 8: launch_partial<8>(d_problems, nprob, maxNq, ngroups, loo, st); break;
    default: return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("eval launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

// (2 pi)^(D/2) * prod_k sqrt(bw_k)   (src/DualTree01.jl:325-330)
double gauss_norm(const double *bw, int D) {
  double norm = std::pow(2.0 * M_PI, D / 2.0);
  for (int k = 0; k < D; ++k) norm *= std::sqrt(bw[k]);
  return norm;
}

}  // namespace
}  // namespace kdehip

using namespace kdehip;

extern "C" int kdehip_evaluate(const kdehip_density *bd, const double *pos, int64_t Nq, int leave_one_out,
                               double *p_out, int device) {
  if (!bd || !p_out) return set_error(KDEHIP_ERR_ARG, "null argument");
  const int D = static_cast<int>(bd->ndim);
  const int64_t N = bd->npts;
  if (D < 1 || D > KDEHIP_MAX_DIMS) return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  if (N < 1 || !bd->means || !bd->bandwidth || !bd->weights || !bd->permutation)
    return set_error(KDEHIP_ERR_ARG, "malformed density");
  if (leave_one_out) Nq = N;
  else if (!pos || Nq < 0) return set_error(KDEHIP_ERR_ARG, "pos must hold Nq >= 0 points");
  if (Nq == 0) return KDEHIP_OK;
  // the reference's evalDirect reads ONE bandwidth vector (bandwidthMin[1..D], BallTreeDensity01.jl:98)
  const double *leaf_pts = bd->means + N * D;  // leaf centres == leaf means == the points, tree order
  const double *bw = bd->bandwidth + N * D;
  for (int64_t i = 0; i < N; ++i)
    for (int k = 0; k < D; ++k)
      if (bd->bandwidth[(N + i) * D + k] != bw[k])
        return set_error(KDEHIP_ERR_UNSUPPORTED, "per-point bandwidths are not supported (the reference's kde! never builds them)");
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;

  const GroupSplit gs = split_chunks(N, Nq, 1);
  const int nchunks = gs.ngroups;  // partial sums per query
  DevBuf d_src, d_w, d_q, d_part, d_out, d_idx;
  KDEHIP_CHECK(d_src.alloc(sizeof(double) * N * D));
  KDEHIP_CHECK(d_w.alloc(sizeof(double) * N));
  KDEHIP_CHECK(d_part.alloc(sizeof(double) * nchunks * Nq));
  KDEHIP_CHECK(d_out.alloc(sizeof(double) * Nq));
  KDEHIP_CHECK(hipMemcpy(d_src.p, leaf_pts, sizeof(double) * N * D, hipMemcpyHostToDevice));
  KDEHIP_CHECK(hipMemcpy(d_w.p, bd->weights + N, sizeof(double) * N, hipMemcpyHostToDevice));
  std::vector<int64_t> idx;
  if (leave_one_out) {  // p[getIndexOf(locations, j)] (:335): results in the caller's original order
    idx.resize(N);
    for (int64_t i = 0; i < N; ++i) idx[i] = bd->permutation[N + i] - 1;
    KDEHIP_CHECK(d_idx.alloc(sizeof(int64_t) * N));
    KDEHIP_CHECK(hipMemcpy(d_idx.p, idx.data(), sizeof(int64_t) * N, hipMemcpyHostToDevice));
  } else {
    KDEHIP_CHECK(d_q.alloc(sizeof(double) * Nq * D));
    KDEHIP_CHECK(hipMemcpy(d_q.p, pos, sizeof(double) * Nq * D, hipMemcpyHostToDevice));
  }
  EvalBatch eb{};
  FinishBatch fb{};
  EvalProblem &pb = eb.p[0];
  pb.src = d_src.as<double>(); pb.w = d_w.as<double>();
  pb.qry = leave_one_out ? d_src.as<double>() : d_q.as<double>();
  pb.partial = d_part.as<double>(); pb.N = N; pb.Nq = Nq; pb.chunks_per_group = gs.chunks_per_group;
  for (int k = 0; k < D; ++k) pb.nhib[k] = -0.5 / bw[k];
  FinishProblem &fp = fb.p[0];
  fp.partial = d_part.as<double>(); fp.w = d_w.as<double>();
  fp.out_idx = leave_one_out ? d_idx.as<int64_t>() : nullptr;
  fp.out = d_out.as<double>(); fp.inv_norm = 1.0 / gauss_norm(bw, D); fp.Nq = Nq; fp.nchunks = nchunks;
  rc = launch_partial_dims(D, eb, 1, Nq, gs.ngroups, leave_one_out ? 1 : 0, nullptr);
  if (rc != KDEHIP_OK) return rc;
  hipLaunchKernelGGL(eval_finish_kernel, dim3(static_cast<unsigned>((Nq + 255) / 256), 1), dim3(256), 0, nullptr,
                     fb, leave_one_out ? 1 : 0);
  KDEHIP_CHECK(hipGetLastError());
  KDEHIP_CHECK(hipDeviceSynchronize());
  KDEHIP_CHECK(hipMemcpy(p_out, d_out.p, sizeof(double) * Nq, hipMemcpyDeviceToHost));
  return KDEHIP_OK;
}

// ---- kde!(points): per-dimension LOOCV bandwidth ------------------------------------------------------

namespace {

// Host state of one 1-D golden-section search (golden, src/CrossValidation.jl:44-98).
struct Golden {
  double x0, x1, x2, x3, f1, f2;
  double minm, maxm;
  double bcur;        // current leaf variance of the search density (drifts like the reference's (b*a)/a)
  int phase;          // 0: needs f1, 1: needs f2, 2: iterating, 3: done
  int pending;        // which of f1/f2 the evaluation in flight fills (1 or 2)
  double alpha;       // argument of the evaluation in flight
  int nevals;
  double result;
};

// Bounding interval (centre, half-range) of the 1-D ball-tree node that covers the sorted ranks
// [a, b], computed bottom-up exactly as calcStatsBall! does (src/BallTree01.jl:282-336): in one
// dimension the median splits of buildBall! (:371-394) make every node a rank interval, so the
// tree's `ranges` -- all neighborMinMax needs (src/CrossValidation.jl:100-108) -- follow from a sort.
// `low`/`high` are the reference's 1-based leaf ids of the interval ends; min2r collects the minimum
// of sqrt((2*range)^2) over internal nodes.
void interval_stats(const double *xs, int64_t low, int64_t high, int64_t leaf0, double &centre, double &half,
                    double &min2r) {
  if (low == high) { centre = xs[low - leaf0]; half = 0.0; return; }
  const int64_t split = (low + high) / 2;
  double cL, rL, cR, rR;
  interval_stats(xs, low, split, leaf0, cL, rL, min2r);
  interval_stats(xs, split + 1, high, leaf0, cR, rR, min2r);
  const double upA = cL + rL, upB = cR + rR, dnA = cL - rL, dnB = cR - rR;
  const double top = (upA > upB) ? upA : upB, bottom = (dnA < dnB) ? dnA : dnB;
  half = (top - bottom) / 2.0;
  centre = bottom + half;
  const double v = std::sqrt((2.0 * half) * (2.0 * half));
  if (v < min2r) min2r = v;
}

}  // namespace

extern "C" int kdehip_auto_bandwidth(int64_t D64, int64_t N, const double *points, double *bw_out,
                                     int32_t *nevals_out, int device) {
  if (!points || !bw_out) return set_error(KDEHIP_ERR_ARG, "null argument");
  if (D64 < 1 || D64 > KDEHIP_MAX_DIMS) return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  if (N < 2) return set_error(KDEHIP_ERR_ARG, "kde!(points) needs at least two points");
  const int D = static_cast<int>(D64);
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;
  const bool timing = std::getenv("KDEHIP_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_begin = tnow();

  // Per dimension d: the marginal's tree ranges give neighborMinMax; the search density is
  // kde!(x_d, (minm+maxm)/2) (ksize, src/CrossValidation.jl:110-120).  The GPU evaluates the
  // leave-one-out likelihood over the points in their ORIGINAL order (the tree order only fixes the
  // reference's summation order), so no tree is built here.
  std::vector<Golden> g(D);
  std::vector<double> xo(static_cast<size_t>(D) * N), wts(static_cast<size_t>(D) * N);
  {
    // weights: ones -> /N (kde!(points,[1.0])) -> renormalised by the marginal's kde! (src/KDE01.jl:46,152)
    std::vector<double> w0(N), w1(N);
    for (int64_t i = 0; i < N; ++i) w0[i] = 1.0 / static_cast<double>(N);
    double t = 0.0;
    for (int64_t i = 0; i < N; ++i) t += w0[i];
    for (int64_t i = 0; i < N; ++i) w1[i] = w0[i] / t;
    // the D marginals are independent: one host thread each (the sort dominates this phase)
    auto prep = [&](int d) {
      std::vector<double> xs(static_cast<size_t>(N));
      for (int64_t i = 0; i < N; ++i) {
        xo[static_cast<size_t>(d) * N + i] = points[i * D + d];
        wts[static_cast<size_t>(d) * N + i] = w1[i];
        xs[i] = points[i * D + d];
      }
      std::sort(xs.begin(), xs.end());
      double centre, half, minm = INFINITY;
      interval_stats(xs.data(), N + 1, 2 * N, N + 1, centre, half, minm);
      const double maxm = std::sqrt((2.0 * half) * (2.0 * half));  // root
      if (minm < 1e-6) minm = 1e-6;
      Golden &s = g[d];
      const double mid = (minm + maxm) / 2.0;
      s.minm = minm; s.maxm = maxm; s.bcur = mid * mid;
      const double ax = 2.0 * minm / (minm + maxm), bx = 1.0, cx = 2.0 * maxm / (minm + maxm);
      const double C = (3.0 - std::sqrt(5.0)) / 2.0;
      s.x0 = ax; s.x3 = cx;
      if (std::fabs(cx - bx) > std::fabs(bx - ax)) { s.x1 = bx; s.x2 = bx + C * (cx - bx); }
      else { s.x1 = bx - C * (bx - ax); s.x2 = bx; }
      s.phase = 0; s.nevals = 0; s.pending = 0; s.alpha = 0; s.f1 = s.f2 = 0; s.result = 0;
    };
    if (D > 1 && N >= 512) {
      std::vector<std::thread> th;
      for (int d = 1; d < D; ++d) th.emplace_back(prep, d);
      prep(0);
      for (auto &t2 : th) t2.join();
    } else {
      for (int d = 0; d < D; ++d) prep(d);
    }
  }

  auto t_host = tnow();
  const GroupSplit gs = split_chunks(N, N, D);
  const int nchunks = gs.ngroups;  // partial sums per query
  const int nfb = static_cast<int>((N + kFinishThreads - 1) / kFinishThreads);
  DevBuf d_x, d_w, d_part;
  KDEHIP_CHECK(d_x.alloc(sizeof(double) * D * N));
  KDEHIP_CHECK(d_w.alloc(sizeof(double) * D * N));
  KDEHIP_CHECK(d_part.alloc(sizeof(double) * D * nchunks * N));
  // the per-round result (D x nfb partial log-likelihoods) is written by the kernel straight into
  // pinned host memory: no device buffer, no copy, one stream synchronisation per round
  struct Pinned {
    double *p = nullptr;
    size_t n = 0;
    ~Pinned() { if (p) cached_host_free(p, n); }
  } h_pin;
  h_pin.n = sizeof(double) * D * nfb;
  KDEHIP_CHECK(cached_host_malloc(reinterpret_cast<void **>(&h_pin.p), h_pin.n));
  KDEHIP_CHECK(hipMemcpy(d_x.p, xo.data(), sizeof(double) * D * N, hipMemcpyHostToDevice));
  KDEHIP_CHECK(hipMemcpy(d_w.p, wts.data(), sizeof(double) * D * N, hipMemcpyHostToDevice));

  auto t_upload = tnow();
  int rounds = 0;
  EvalBatch eb{};
  FinishBatch fb{};
  const double C = (3.0 - std::sqrt(5.0)) / 2.0, R = 1.0 - C;
  const double tol = 1e-2;  // ksize, src/CrossValidation.jl:116

  for (;;) {
    // decide what every unfinished search evaluates in this round
    std::vector<int> active;
    for (int d = 0; d < D; ++d) {
      Golden &s = g[d];
      if (s.phase == 3) continue;
      if (s.phase == 0) { s.alpha = s.x1; s.pending = 1; }
      else if (s.phase == 1) { s.alpha = s.x2; s.pending = 2; }
      else {
        if (!(std::fabs(s.x3 - s.x0) > tol * (std::fabs(s.x1) + std::fabs(s.x2)))) {
          s.result = (s.f1 < s.f2) ? s.x1 : s.x2;
          s.phase = 3;
          continue;
        }
        if (s.f2 < s.f1) { s.x0 = s.x1; s.x1 = s.x2; s.x2 = R * s.x1 + C * s.x3; s.f1 = s.f2; s.alpha = s.x2; s.pending = 2; }
        else { s.x3 = s.x2; s.x2 = s.x1; s.x1 = R * s.x2 + C * s.x0; s.f2 = s.f1; s.alpha = s.x1; s.pending = 1; }
      }
      active.push_back(d);
    }
    if (active.empty()) break;
    ++rounds;
    // nLOO_LL (src/CrossValidation.jl:15-24): bandwidth *= alpha^2 for the evaluation, /= alpha^2 after
    const int na = static_cast<int>(active.size());
    std::vector<double> a2(na), bw_eval(na);
    for (int a = 0; a < na; ++a) {
      const int d = active[a];
      a2[a] = g[d].alpha * g[d].alpha;
      bw_eval[a] = g[d].bcur * a2[a];
      EvalProblem &pb = eb.p[a];
      std::memset(&pb, 0, sizeof(pb));
      pb.src = d_x.as<double>() + static_cast<size_t>(d) * N;
      pb.qry = pb.src;
      pb.w = d_w.as<double>() + static_cast<size_t>(d) * N;
      pb.partial = d_part.as<double>() + static_cast<size_t>(d) * nchunks * N;
      pb.nhib[0] = -0.5 / bw_eval[a];
      pb.N = N; pb.Nq = N; pb.chunks_per_group = gs.chunks_per_group;
      FinishProblem &fp = fb.p[a];
      std::memset(&fp, 0, sizeof(fp));
      fp.partial = pb.partial; fp.w = pb.w;
      fp.inv_norm = 1.0 / gauss_norm(&bw_eval[a], 1);
      fp.Nq = N; fp.nchunks = nchunks;
    }
    rc = launch_partial_dims(1, eb, na, N, gs.ngroups, 1, nullptr);
    if (rc != KDEHIP_OK) return rc;
    hipLaunchKernelGGL(loo_entropy_kernel, dim3(static_cast<unsigned>(nfb), static_cast<unsigned>(na)),
                       dim3(kFinishThreads), 0, nullptr, fb, h_pin.p, nfb);
    KDEHIP_CHECK(hipGetLastError());
    KDEHIP_CHECK(hipStreamSynchronize(nullptr));
    const double *h_host = h_pin.p;
    for (int a = 0; a < na; ++a) {
      const int d = active[a];
      Golden &s = g[d];
      double ll = 0.0;
      for (int b = 0; b < nfb; ++b) ll += h_host[static_cast<size_t>(a) * nfb + b];
      const double H = -ll;  // entropy = -evalAvgLogL (src/DualTree01.jl:505-508); -(-Inf) = +Inf
      s.bcur = (s.bcur * a2[a]) / a2[a];
      s.nevals += 1;
      if (s.pending == 1) s.f1 = H; else s.f2 = H;
      if (s.phase < 2) s.phase += 1;
    }
  }
  int total = 0;
  for (int d = 0; d < D; ++d) {
    double ks = g[d].result * (g[d].minm + g[d].maxm) / 2.0;  // ksize, src/CrossValidation.jl:117
    bw_out[d] = std::sqrt(ks * ks);                            // getBW of kde!(.., [ks]) (src/KDE01.jl:45,118)
    total += g[d].nevals;
  }
  if (nevals_out) *nevals_out = total;
  if (timing) {
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::fprintf(stderr, "kdehip_auto_bandwidth D=%d N=%lld: host prep %.0f us, alloc+upload %.0f us, %d rounds %.0f us\n", D,
                 static_cast<long long>(N), us(t_begin, t_host), us(t_host, t_upload), rounds, us(t_upload, tnow()));
  }
  return KDEHIP_OK;
}
