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
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "fastexp.hpp"
#include "host_pool.hpp"
#include "kdehip_internal.hpp"
#include "loocv_search.hpp"
#include "phase_timer.hpp"

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
  hipStream_t st = hipStreamPerThread;  // the calling thread's own stream, like every blocking entry point (kdehip.h)

  const GroupSplit gs = split_chunks(N, Nq, 1);
  const int nchunks = gs.ngroups;  // partial sums per query
  // ONE pinned image [points | weights | queries or output positions] goes up in one DMA, the results come back in
  // one; everything is enqueued on the calling thread's stream and the host waits once (pageable hipMemcpy calls, one per
  // array, cost more than the kernel for anything below ~10^8 kernel evaluations).
  const size_t o_src = 0;
  const size_t o_w = o_src + sizeof(double) * N * D;
  const size_t o_q = o_w + sizeof(double) * N;
  const size_t up_bytes = o_q + (leave_one_out ? sizeof(int64_t) * N : sizeof(double) * Nq * D);
  const size_t o_out = (up_bytes + 255) & ~size_t(255);
  const size_t pin_bytes = o_out + sizeof(double) * Nq;
  struct Pinned {
    void *p = nullptr; size_t n = 0;
    ~Pinned() { if (p) cached_host_free(p, n); }
  } pin;
  pin.n = pin_bytes;
  KDEHIP_CHECK(cached_host_malloc(&pin.p, pin.n));
  unsigned char *h = static_cast<unsigned char *>(pin.p);
  std::memcpy(h + o_src, leaf_pts, sizeof(double) * N * D);
  std::memcpy(h + o_w, bd->weights + N, sizeof(double) * N);
  if (leave_one_out) {  // p[getIndexOf(locations, j)] (:335): results in the caller's original order
    int64_t *idx = reinterpret_cast<int64_t *>(h + o_q);
    for (int64_t i = 0; i < N; ++i) idx[i] = bd->permutation[N + i] - 1;
  } else {
    std::memcpy(h + o_q, pos, sizeof(double) * Nq * D);
  }
  DevBuf d_up, d_part, d_out;
  KDEHIP_CHECK(d_up.alloc(up_bytes));
  KDEHIP_CHECK(d_part.alloc(sizeof(double) * nchunks * Nq));
  KDEHIP_CHECK(d_out.alloc(sizeof(double) * Nq));
  unsigned char *du = d_up.as<unsigned char>();
  KDEHIP_CHECK(hipMemcpyAsync(du, h, up_bytes, hipMemcpyHostToDevice, st));
  const double *d_src = reinterpret_cast<const double *>(du + o_src);
  const double *d_w = reinterpret_cast<const double *>(du + o_w);
  EvalBatch eb{};
  FinishBatch fb{};
  EvalProblem &pb = eb.p[0];
  pb.src = d_src; pb.w = d_w;
  pb.qry = leave_one_out ? d_src : reinterpret_cast<const double *>(du + o_q);
  pb.partial = d_part.as<double>(); pb.N = N; pb.Nq = Nq; pb.chunks_per_group = gs.chunks_per_group;
  for (int k = 0; k < D; ++k) pb.nhib[k] = -0.5 / bw[k];
  FinishProblem &fp = fb.p[0];
  fp.partial = d_part.as<double>(); fp.w = d_w;
  fp.out_idx = leave_one_out ? reinterpret_cast<const int64_t *>(du + o_q) : nullptr;
  fp.out = d_out.as<double>(); fp.inv_norm = 1.0 / gauss_norm(bw, D); fp.Nq = Nq; fp.nchunks = nchunks;
  PhaseTimer timer(kPhaseEvaluate, st);
  rc = launch_partial_dims(D, eb, 1, Nq, gs.ngroups, leave_one_out ? 1 : 0, st);
  if (rc != KDEHIP_OK) { (void)hipStreamSynchronize(st); return rc; }
  hipLaunchKernelGGL(eval_finish_kernel, dim3(static_cast<unsigned>((Nq + 255) / 256), 1), dim3(256), 0, st,
                     fb, leave_one_out ? 1 : 0);
  hipError_t le = hipGetLastError();
  timer.stop();
  if (le == hipSuccess) le = hipMemcpyAsync(h + o_out, d_out.p, sizeof(double) * Nq, hipMemcpyDeviceToHost, st);
  const hipError_t se = hipStreamSynchronize(st);  // (also before the scratch goes back to the cache on an error)
  KDEHIP_CHECK(le);
  KDEHIP_CHECK(se);
  timer.collect();
  std::memcpy(p_out, h + o_out, sizeof(double) * Nq);
  return KDEHIP_OK;
}

// ---- kde!(points): per-dimension LOOCV bandwidth ------------------------------------------------------
// The whole search runs on the device: a preparation kernel (sort of every marginal + the bottom-up interval
// arithmetic that stands in for the marginal's ball tree), then rounds of two launches -- the all-pairs
// leave-one-out sums of all D one-dimensional problems, and the fused finish + log-likelihood reduction -- that are
// enqueued back to back WITHOUT host synchronisation: the golden-section state machine of every dimension
// (golden, src/CrossValidation.jl:44-98) lives in device memory and is advanced in the prologue of the round's
// first kernel (every block redoes the few dozen scalar operations; block 0 of a dimension stores the result in
// the other half of a double-buffered state).  The host only looks at the state after a batch of rounds.

namespace {

constexpr double kGoldenTol = 1e-2;  // ksize, src/CrossValidation.jl:116
constexpr int kCounterStride = 32;   // loo_round_pairs_kernel's slot counters: a 128-byte line each (neighbours in one line
                                     // make it bounce between the XCDs' L2s)
constexpr int kPrepThreads = 1024;
constexpr int64_t kPrepMaxN = kLoocvPrepMaxN;  // marginals up to this size are prepared on the device (LDS: 48 bytes per point
                                               // of the next power of two: 96 KiB; 4096 points would need more than the CU has)

// State of one 1-D golden-section search (golden, src/CrossValidation.jl:44-98) + what ksize needs around it.
struct Golden {
  double x0, x1, x2, x3, f1, f2;
  double minm, maxm;
  double bcur;        // current leaf variance of the search density (drifts like the reference's (b*a)/a)
  double alpha;       // argument of the evaluation in flight
  double bw_eval;     // bcur * alpha^2: the variance that evaluation uses
  double result;
  int phase;          // 0: needs f1, 1: needs f2, 2: iterating, 3: done
  int pending;        // which of f1/f2 the evaluation in flight fills (1 or 2), 0 = none
  int nevals;
  int spec;           // speculative rounds: the launch that evaluated `pending` also evaluated BOTH candidates of the decision after it
};

// Initial bracket of ksize (src/CrossValidation.jl:110-120) from neighborMinMax (:100-108).
__host__ __device__ inline void golden_init(Golden &s, double minm, double maxm) {
  if (minm < 1e-6) minm = 1e-6;
  const double mid = (minm + maxm) / 2.0;
  s.minm = minm; s.maxm = maxm; s.bcur = mid * mid;
  const double ax = 2.0 * minm / (minm + maxm), bx = 1.0, cx = 2.0 * maxm / (minm + maxm);
  const double C = (3.0 - sqrt(5.0)) / 2.0;
  s.x0 = ax; s.x3 = cx;
  if (fabs(cx - bx) > fabs(bx - ax)) { s.x1 = bx; s.x2 = bx + C * (cx - bx); }
  else { s.x1 = bx - C * (bx - ax); s.x2 = bx; }
  s.phase = 0; s.nevals = 0; s.pending = 0; s.alpha = 0; s.bw_eval = 0; s.f1 = s.f2 = 0; s.result = 0; s.spec = 0;
}

// The state machine in two halves.  book: the evaluation that was in flight has finished (its block partials of
// W*log p are in `hpart`, summed in block order); decide: what to evaluate next, or finish.
__host__ __device__ inline void golden_book(Golden &s, const double *hpart, int nfb) {
  if (s.phase == 3 || !s.pending) return;
  double ll = 0.0;
  for (int b = 0; b < nfb; ++b) ll += hpart[b];
  const double H = -ll;  // entropy = -evalAvgLogL (src/DualTree01.jl:505-508); -(-Inf) = +Inf
  const double a2 = s.alpha * s.alpha;
  s.bcur = (s.bcur * a2) / a2;  // nLOO_LL: bandwidth *= alpha^2 ... /= alpha^2 (src/CrossValidation.jl:15-24)
  s.nevals += 1;
  if (s.pending == 1) s.f1 = H; else s.f2 = H;
  if (s.phase < 2) s.phase += 1;
  s.pending = 0;
}
// (the two halves of an iteration of golden, :70-90: has the bracket closed; move it to the right or to the left)
__host__ __device__ inline bool golden_closed(const Golden &s) {
  return !(fabs(s.x3 - s.x0) > kGoldenTol * (fabs(s.x1) + fabs(s.x2)));
}
__host__ __device__ inline void golden_shift(Golden &s, bool right) {
  const double C = (3.0 - sqrt(5.0)) / 2.0, R = 1.0 - C;
  if (right) { s.x0 = s.x1; s.x1 = s.x2; s.x2 = R * s.x1 + C * s.x3; s.f1 = s.f2; s.alpha = s.x2; s.pending = 2; }
  else { s.x3 = s.x2; s.x2 = s.x1; s.x1 = R * s.x2 + C * s.x0; s.f2 = s.f1; s.alpha = s.x1; s.pending = 1; }
}
__host__ __device__ inline void golden_decide(Golden &s) {
  if (s.phase == 3) return;
  if (s.phase == 0) { s.alpha = s.x1; s.pending = 1; }
  else if (s.phase == 1) { s.alpha = s.x2; s.pending = 2; }
  else {
    if (golden_closed(s)) {
      s.result = (s.f1 < s.f2) ? s.x1 : s.x2;
      s.phase = 3;
      return;
    }
    golden_shift(s, s.f2 < s.f1);
  }
  s.bw_eval = s.bcur * (s.alpha * s.alpha);
}
// SPECULATIVE rounds (loo_round_spec_kernel).  Which point golden evaluates after the one in flight depends on that
// evaluation only through ONE comparison, so both candidates are known beforehand and a launch can evaluate three points
// -- the one in flight's successor is then already there, whichever way the comparison goes -- and the search advances two
// evaluations per launch.  golden_book_blind: the booking of the evaluation in flight as far as it does not need the
// result (the drift of the leaf variance, the counter).  golden_candidate: the state as it will be when the comparison
// comes out `right`; phase 3 = the bracket closes first, there is no such candidate.
__host__ __device__ inline void golden_book_blind(Golden &s) {
  const double a2 = s.alpha * s.alpha;
  s.bcur = (s.bcur * a2) / a2;
  s.nevals += 1;
  if (s.phase < 2) s.phase += 1;
  s.pending = 0;
}
__host__ __device__ inline Golden golden_candidate(Golden s, bool right) {
  golden_book_blind(s);
  if (s.phase < 2) { golden_decide(s); return s; }  // (the opening: x2 follows x1 whatever x1 gave)
  if (golden_closed(s)) { s.phase = 3; return s; }
  golden_shift(s, right);
  s.bw_eval = s.bcur * (s.alpha * s.alpha);
  return s;
}
// The prologue of a speculative launch: book what the previous launch evaluated -- the point in flight (shares in part[0])
// and, when it evaluated candidates too, the one the comparison picks (part[1]: right, or the second opening probe;
// part[2]: left) -- and decide again: `s` leaves with the next certain evaluation pending, or finished.
__host__ __device__ inline void golden_advance_spec(Golden &s, const double *part0, const double *part1, const double *part2,
                                                    int nfb) {
  if (s.phase == 3) return;
  if (s.pending) {
    const int ph = s.phase;
    golden_book(s, part0, nfb);
    const bool right = s.f2 < s.f1;  // (what golden_decide is about to branch on, once the opening is over)
    golden_decide(s);
    if (s.phase == 3) return;
    if (s.spec) {
      // (ph == 0: the opening, x2 follows x1 whatever x1 gave -- part1; from then on the candidate the comparison picks.  A
      // launch that evaluates candidates with x2 in flight (ph == 1) books by `right` like any other: ADVICE round 5)
      golden_book(s, (ph == 0 || right) ? part1 : part2, nfb);
      golden_decide(s);
      if (s.phase == 3) return;
    }
  } else {
    golden_decide(s);  // (a batch starts: nothing in flight)
  }
  s.spec = 1;
}

// Bounding interval (centre, half-range) of the 1-D ball-tree node that covers the sorted ranks
// [a, b], computed bottom-up exactly as calcStatsBall! does (src/BallTree01.jl:282-336): in one
// dimension the median splits of buildBall! (:371-394) make every node a rank interval, so the
// tree's `ranges` -- all neighborMinMax needs (src/CrossValidation.jl:100-108) -- follow from a sort.
// `low`/`high` are the reference's 1-based leaf ids of the interval ends; min2r collects the minimum
// of sqrt((2*range)^2) over internal nodes.  (Host form, for marginals beyond kPrepMaxN points.)
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

#ifdef KDEHIP_PREP_STAMPS  // (diagnostic builds: s_memtime at the phase boundaries of block 0, scripts/prep_stamps.py)
__device__ unsigned long long g_prep_stamps[8];
#define PSTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_prep_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(k) do {} while (0)
#endif
// One block per marginal: x_d in original order to `xo`, sort in LDS, interval arithmetic, initial search state.  `points`
// = nm / D matrices of D x N (column-major, one behind the other): block m prepares dimension m % D of matrix m / D -- the
// D searches of one kde!(points), or those of a whole batch of them (kdehip_mul_device_batch) in the same launch.
__global__ __launch_bounds__(kPrepThreads) void loocv_prep_kernel(const double *__restrict__ points, int64_t N, int D, int nm,
                                                                 double *__restrict__ xo, Golden *__restrict__ state,
                                                                 unsigned *__restrict__ arrivals, int ntiles) {
  extern __shared__ double sm[];
  const int m = blockIdx.x, mat = m / D, d = m - mat * D;
  points += static_cast<int64_t>(mat) * N * D;
  xo += static_cast<int64_t>(m - d) * N;  // (xo[d * N + e] below: marginal m at xo[m * N])
  PSTAMP(0);
  if (static_cast<int>(threadIdx.x) < ntiles)  // (the rounds' slot counters: up to three probes per launch)
    for (int p = 0; p < 3; ++p) arrivals[((p * nm + m) * ntiles + threadIdx.x) * kCounterStride] = 0;
  int64_t P = 1;
  while (P < N) P <<= 1;
  double *xs = sm;              // [P] sorted marginal (padded with +inf)
  double *cen = sm + P;         // [2][P] centre of the nodes of two consecutive depths
  double *hal = sm + 3 * P;     // [2][P] half-range
  // rank interval [low, high] of every node of the median-split tree, heap order (node t of depth dd at 2^dd + t; low >
  // high: the node does not exist), laid out TOP-DOWN once -- it depends on N only -- instead of being walked down from the
  // root by every node of every depth (node_interval: a dependent chain of up to 11 64-bit steps per node and depth, 12 of
  // the kernel's 24 us at 1000 points; profiles/r05_experiments.md section 13)
  unsigned short *ilo = reinterpret_cast<unsigned short *>(sm + 5 * P);  // [2P]
  unsigned short *ihi = ilo + 2 * P;                                     // [2P]
  __shared__ double s_min[kPrepThreads / 64];
  // every thread keeps its element(s) of the marginal in registers: element tid, and tid + 1024 when P = 2048
  const int Pi = static_cast<int>(P), tid = threadIdx.x;
  const bool two = Pi > kPrepThreads;
  const int e0 = tid, e1 = tid + kPrepThreads;
  double v0 = INFINITY, v1 = INFINITY;  // (padding: +inf sorts to the end)
  if (e0 < N) { v0 = points[static_cast<int64_t>(e0) * D + d]; xo[static_cast<int64_t>(d) * N + e0] = v0; }
  if (two && e1 < N) { v1 = points[static_cast<int64_t>(e1) * D + d]; xo[static_cast<int64_t>(d) * N + e1] = v1; }
  PSTAMP(1);
  // Bitonic sort, ascending.  A stage pairs element i with i ^ j.  For j <= 32 the partner sits in the same wavefront, 32
  // lanes away at most: the two exchange through the cross-lane network and each keeps the smaller or the larger (45 of
  // the 55 stages at 1024 points; through LDS such a stage was two dependent reads and two divergent writes, ~450 cycles
  // against ~200).  j = 1024 pairs a thread's own two elements.  The stages in between (j = 64 .. 512) go through LDS, one
  // thread per pair, one barrier each.
  auto lane_stage = [&](double &v, int e, int k, int j) {
    const double p = __shfl_xor(v, j);
    const bool up = (e & k) == 0, lower = (e & j) == 0;
    v = (up == lower) ? fmin(v, p) : fmax(v, p);
  };
  for (int k = 2; k <= Pi; k <<= 1) {
    if (k > 64) {
      if (k > kPrepThreads) {  // j = 1024 (the last phase of 2048 elements: ascending everywhere)
        const double lo = fmin(v0, v1), hi = fmax(v0, v1);
        v0 = lo; v1 = hi;
      }
      if (e0 < Pi) xs[e0] = v0;
      if (two) xs[e1] = v1;
      for (int j = (k >> 1) < 512 ? (k >> 1) : 512; j >= 64; j >>= 1) {
        __syncthreads();
        for (int c = tid; c < Pi / 2; c += kPrepThreads) {
          const int i = ((c & ~(j - 1)) << 1) | (c & (j - 1));
          const int l = i | j;
          const double a = xs[i], b = xs[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { xs[i] = b; xs[l] = a; }
        }
      }
      __syncthreads();
      if (e0 < Pi) v0 = xs[e0];
      if (two) v1 = xs[e1];
    }
    for (int j = (k >> 1) < 32 ? (k >> 1) : 32; j >= 1; j >>= 1) {
      lane_stage(v0, e0, k, j);
      if (two) lane_stage(v1, e1, k, j);
    }
  }
  if (e0 < Pi) xs[e0] = v0;
  if (two) xs[e1] = v1;
  __syncthreads();
  PSTAMP(2);
  int depth = 0;
  while ((int64_t(1) << depth) < N) ++depth;  // the deepest level that can hold a node
  {
    const int n = static_cast<int>(N);
    if (threadIdx.x == 0) { ilo[1] = 0; ihi[1] = static_cast<unsigned short>(n - 1); }
    __syncthreads();
    for (int dd = 0; dd < depth; ++dd) {
      const int cnt = 1 << dd;
      for (int t = threadIdx.x; t < cnt; t += kPrepThreads) {
        const int low = ilo[cnt + t], high = ihi[cnt + t];
        const int c = 2 * (cnt + t);
        if (low >= high) {  // a leaf, or no node: no children
          ilo[c] = ilo[c + 1] = 1; ihi[c] = ihi[c + 1] = 0;
        } else {
          const int split = ((low + n + 1) + (high + n + 1)) / 2 - (n + 1);  // same rounding as on the 1-based ids (:371)
          ilo[c] = static_cast<unsigned short>(low); ihi[c] = static_cast<unsigned short>(split);
          ilo[c + 1] = static_cast<unsigned short>(split + 1); ihi[c + 1] = static_cast<unsigned short>(high);
        }
      }
      // (up to 64 children: written and read by wavefront 0 alone, whose LDS accesses execute in order -- no block barrier)
      if (2 * cnt > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
  double vmin = INFINITY;
  for (int dd = depth; dd >= 0; --dd) {
    double *c0 = cen + (dd & 1) * P, *h0 = hal + (dd & 1) * P;
    const double *c1 = cen + ((dd + 1) & 1) * P, *h1 = hal + ((dd + 1) & 1) * P;
    const int cnt = 1 << dd;
    for (int t = threadIdx.x; t < cnt && t < P; t += kPrepThreads) {
      const int low = ilo[cnt + t], high = ihi[cnt + t];
      if (low > high) continue;  // no such node (an ancestor is a single leaf)
      if (low == high) { c0[t] = xs[low]; h0[t] = 0.0; continue; }
      const double cL = c1[2 * t], rL = h1[2 * t], cR = c1[2 * t + 1], rR = h1[2 * t + 1];
      const double upA = cL + rL, upB = cR + rR, dnA = cL - rL, dnB = cR - rR;
      const double top = (upA > upB) ? upA : upB, bottom = (dnA < dnB) ? dnA : dnB;
      const double half = (top - bottom) / 2.0;
      h0[t] = half;
      c0[t] = bottom + half;
      const double v = sqrt((2.0 * half) * (2.0 * half));
      if (v < vmin) vmin = v;
    }
    if (cnt > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier();  // (as above: the levels of wavefront 0 alone)
  }
  PSTAMP(3);
  for (int off = 32; off > 0; off >>= 1) vmin = fmin(vmin, __shfl_down(vmin, off));
  if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = vmin;
  __syncthreads();
  if (threadIdx.x == 0) {
    double mn = s_min[0];
    for (int i = 1; i < kPrepThreads / 64; ++i) mn = fmin(mn, s_min[i]);
    const double half_root = hal[0];
    Golden g;
    golden_init(g, mn, sqrt((2.0 * half_root) * (2.0 * half_root)));
    state[blockIdx.x] = g;
  }
  PSTAMP(4);
}

constexpr int kLooThreads = 256;  // queries per block
constexpr int kLooChunk = 128;    // source points per staged chunk

struct LooRound {
  const double *x;        // [D][N] marginals, original order
  double *partial;        // [D][ngroups][N]; one launch per round: [D][T][T][64], (tile, source tile) slots
  unsigned *arrivals;     // [D][T] slots delivered per tile (one launch per round; zero between rounds)
  int joint;              // one launch per round: the first launch evaluates BOTH opening probes of every search (a second
                          // set of slots, counters and shares behind the first), the second launch books both
  double *hpart;          // [2][D][nfb] block partials of W*log p of the evaluation in flight (by round parity)
  Golden *state;          // [2][D]
  int64_t N;
  double w;               // the common weight of every point (kde!(points) has none of its own)
  double sqrt_2pi;        // pow(2 pi, 1/2) as the host's libm rounds it
  int chunks_per_group, ngroups, nfb, D, round;
  int spec;               // speculative rounds (loo_round_spec_kernel): three sets of slots / counters, shares [2][3][D][nfb]
};

// Round, first launch: advance the search of this block's dimension, then the all-pairs leave-one-out sums
// partial[g][q] = sum_{i in group g, i != q} exp(-1/2 (x_q - x_i)^2 / bw)   (weights are uniform: applied later)
__global__ __launch_bounds__(kLooThreads) void loo_round_partial_kernel(const LooRound r) {
  __shared__ double sSrc[2][kLooChunk];
  __shared__ double sExpTab[32];
  __shared__ Golden sh;
  const int d = blockIdx.z;
  if (threadIdx.x < 32) sExpTab[threadIdx.x] = kExp2Tab[threadIdx.x];
  if (threadIdx.x == 0) {
    Golden s = r.state[(r.round & 1) * r.D + d];
    golden_book(s, r.hpart + (static_cast<int64_t>(r.round & 1) * r.D + d) * r.nfb, r.nfb);
    golden_decide(s);
    sh = s;
    if (blockIdx.x == 0 && blockIdx.y == 0) r.state[((r.round + 1) & 1) * r.D + d] = s;
  }
  __syncthreads();
  if (sh.phase == 3) return;  // this dimension's search is over
  const double nhib = -0.5 / sh.bw_eval;
  const double *x = r.x + static_cast<int64_t>(d) * r.N;
  const int64_t q = static_cast<int64_t>(blockIdx.x) * kLooThreads + threadIdx.x;
  const int64_t nchunks = (r.N + kLooChunk - 1) / kLooChunk;
  const int64_t c_begin = static_cast<int64_t>(blockIdx.y) * r.chunks_per_group;
  int64_t c_end = c_begin + r.chunks_per_group;
  if (c_end > nchunks) c_end = nchunks;
  const double xq = q < r.N ? x[q] : 0.0;
  auto stage = [&](int64_t c, int buf) {
    const int64_t i = c * kLooChunk + threadIdx.x;
    if (threadIdx.x < kLooChunk) sSrc[buf][threadIdx.x] = i < r.N ? x[i] : INFINITY;  // (a point at infinity contributes exp(-inf) = 0)
  };
  if (c_begin < c_end) stage(c_begin, 0);
  double total = 0.0;
  for (int64_t c = c_begin; c < c_end; ++c) {
    const int buf = static_cast<int>((c - c_begin) & 1);
    __syncthreads();
    if (c + 1 < c_end) stage(c + 1, buf ^ 1);
    const int64_t i0 = c * kLooChunk;
    double sum = 0.0;
#pragma unroll 4
    for (int i = 0; i < kLooChunk; ++i) {
      const double dlt = xq - sSrc[buf][i];
      double v = exp_nonpos((dlt * dlt) * nhib, sExpTab);
      if (i0 + i == q) v = 0.0;  // leave-one-out: skip the self term (:141)
      sum += v;
    }
    total += sum;
  }
  if (q < r.N) r.partial[(static_cast<int64_t>(d) * r.ngroups + blockIdx.y) * r.N + q] = total;
}

// A whole round in ONE launch (marginals up to kFusedMaxN points), every kernel value computed ONCE: exp(-(x_i-x_j)^2/2bw)
// is the same for (i, j) and (j, i), so a wavefront that holds the 64 points of tile I in its lanes and lets the 64 points
// of tile J travel around them (DPP wave rotate, one lane per step -- no LDS traffic besides the exp table) adds every
// value to two sums: the row sum of its own point and a column sum that travels with the visiting point and is home
// again after 64 steps.  The unordered tile pairs are dealt out in a circle: the items of tile I are the diagonal
// (I, I) and (I, I+k mod T) for k = 1 .. T/2 (for even T the offset T/2 pairs every tile with one partner only, so the
// tiles of the lower half take it).  An item leaves the 64 sums it holds for tile I in part[I][src = J] and those for
// tile J in part[J][src = I]: every (tile, source tile) slot is written exactly once, so the total of a query is the
// sum of its T slots in source order -- fixed, whatever the order the items finish in.  That total, W*log p and the
// tile's share of the log-likelihood (evalAvgLogL, src/DualTree01.jl:450-474) are the work of whichever wavefront
// delivers a tile's LAST slot (a counter per tile): no second launch, and the search state of the next round is
// advanced in the next launch's prologue.  (Measured at 6 x 2048, scripts in profiles/r03_loocv.md: the arithmetic of a
// round fell from 21 us to 4 items of 2.9 us per SIMD = 11.6 us (3264 items on 1024 SIMDs: 3.2 each would do); the
// hand-over costs 0.4 us per step and 3.5 us for the last tile's loads: 25.8 -> 17.9 us a round.)
constexpr int kTile = 64;
constexpr int kPairWaves = 16;  // (items of 16 wavefronts fill a CU evenly -- 4 per SIMD; workgroups of 4 were dealt 2-6 to a CU)
constexpr int64_t kFusedMaxN = 4096;
__device__ __forceinline__ double wave_rotate(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x13C, 0xF, 0xF, false);  // wave_ror:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// The slots cross workgroups (and XCDs, each with an L2 of its own) inside one launch.  A release fence at device scope
// writes the whole L2 back (buffer_wbl2: measured 4x the round's arithmetic); instead every slot is stored and loaded
// as a device-scope relaxed atomic -- written through to, and read from, the level all XCDs share -- and a wavefront
// counts only after its stores have been acknowledged (s_waitcnt vmcnt(0)).
// This hand-over leans on gfx950 behaviour beyond the HIP memory model's guarantees for relaxed atomics (sc1 stores are
// written through to the level all XCDs share and acknowledged only then; sc1 loads miss the XCD's own L2): it is
// compiled for that target only, the finishing wavefront additionally starts with an agent-scope ACQUIRE fence (one per
// tile: cheap, unlike a release per item), and tests/test_gpu_bandwidth.py pins the result against the two-launch rounds
// (no cross-workgroup hand-over inside a launch; KDEHIP_LOOCV_TWO_LAUNCH=1) at many sizes, odd and even tile counts.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "loo_round_pairs_kernel's slot hand-over is written for gfx950 (MI355X)"
#endif
__device__ __forceinline__ void slot_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double slot_load(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void slots_delivered() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
struct PairSet {  // where one probe of a launch keeps its slots [D][T][T][64], counters [D][T] and tile shares [T]
  double *slots;
  unsigned *arrivals;
  double *shares;
};
// all T slots of `tile` are in place: total per query in source order, W*log p, the tile's share of the log-likelihood
__device__ __forceinline__ void pairs_finish_tile(const LooRound &r, const PairSet &ps, int d, int tile, int lane, double bw_eval) {
  const int T = r.ngroups;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (after the counter that said "all T slots are in", before the slot loads)
  if (lane == 0) ps.arrivals[(d * T + tile) * kCounterStride] = 0;  // every slot is in: nobody counts on this tile again before the next round
  const double *slots = ps.slots + (static_cast<int64_t>(d) * T + tile) * T * kTile + lane;
  double tot = 0.0;
  for (int s0 = 0; s0 < T; s0 += 32) {  // (32 loads in flight: they come from beyond the L2, 3 us a trip)
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = s0 + u < T ? slot_load(slots + static_cast<int64_t>(s0 + u) * kTile) : 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u)
      if (s0 + u < T) tot += v[u];
  }
  const int64_t q = static_cast<int64_t>(tile) * kTile + lane;
  const double inv_norm = 1.0 / (r.sqrt_2pi * sqrt(bw_eval));  // norm = (2 pi)^(1/2) * sqrt(bw), :325-330
  double term = 0.0;
  if (q < r.N) {
    const double w = r.w;
    const double p = (tot * w) * inv_norm / (1.0 - w);
    if (p == 0.0) term = (w != 0.0) ? -INFINITY : 0.0;  // a zero likelihood that carries weight: -Inf (:460-463)
    else term = log(p) * w;
  }
  for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off);  // fixed order
  if (lane == 0) ps.shares[tile] = term;
}
// This wavefront's slots of tiles I and J (J < 0: of tile I only) are in place: count them -- lane 0 for I, lane 1 for J,
// one atomic instruction -- and finish every tile whose T slots are complete with that.
__device__ __forceinline__ void pairs_arrive(const LooRound &r, const PairSet &ps, int d, int I, int J, int lane, double bw_eval) {
  const int T = r.ngroups;
  unsigned old = 0;
  if (lane == 0 || (lane == 1 && J >= 0))
    old = __hip_atomic_fetch_add(ps.arrivals + (d * T + (lane == 0 ? I : J)) * kCounterStride, 1u, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
  const unsigned oldI = __builtin_amdgcn_readlane(old, 0), oldJ = __builtin_amdgcn_readlane(old, 1);
  if (oldI == static_cast<unsigned>(T - 1)) pairs_finish_tile(r, ps, d, I, lane, bw_eval);
  if (J >= 0 && oldJ == static_cast<unsigned>(T - 1)) pairs_finish_tile(r, ps, d, J, lane, bw_eval);
}

// One tile pair (I, I + k mod T) of dimension d by one wavefront: the 64 x 64 kernel values, each computed once and added to
// a row sum (own point) and a column sum (visiting point), left in the (tile, source tile) slots; then the arrival.
__device__ __forceinline__ void pairs_item(const LooRound &r, const PairSet &ps, int d, int e, int lane, double bw_eval,
                                           const double *sExpTab) {
  const int T = r.ngroups, K = T / 2;
  const int I = e / (K + 1), k = e - I * (K + 1);
  if (I >= T) return;
  if (2 * k == T && I >= K) return;  // even T, offset T/2: the partner tile holds this pair
  const int J = I + k < T ? I + k : I + k - T;
  const double nhib = -0.5 / bw_eval;
  const double *x = r.x + static_cast<int64_t>(d) * r.N;
  const int64_t qi = static_cast<int64_t>(I) * kTile + lane, qj = static_cast<int64_t>(J) * kTile + lane;
  const double xi = qi < r.N ? x[qi] : INFINITY;
  double xj = qj < r.N ? x[qj] : -INFINITY;  // (a point at infinity contributes exp(-inf) = 0; opposite signs: no inf - inf)
  double *slot_row = ps.slots + ((static_cast<int64_t>(d) * T + I) * T + J) * kTile + lane;
  double row = 0.0;
  if (k == 0) {  // the own tile: the first rotation skips the self term (:141); both orders of a pair are met
    xj = wave_rotate(xj);
#pragma unroll 4
    for (int s = 1; s < kTile; ++s) {
      const double dlt = xi - xj;
      row += exp256_nonpos((dlt * dlt) * nhib, sExpTab);
      xj = wave_rotate(xj);
    }
    slot_store(slot_row, row);
    slots_delivered();
    pairs_arrive(r, ps, d, I, -1, lane, bw_eval);
    return;
  }
  double col = 0.0;
#pragma unroll 4
  for (int s = 0; s < kTile; ++s) {
    const double dlt = xi - xj;
    const double v = exp256_nonpos((dlt * dlt) * nhib, sExpTab);
    row += v;
    col = wave_rotate(col + v);
    xj = wave_rotate(xj);
  }
  slot_store(slot_row, row);
  slot_store(ps.slots + ((static_cast<int64_t>(d) * T + J) * T + I) * kTile + lane, col);  // (64 rotations: home again)
  slots_delivered();
  pairs_arrive(r, ps, d, I, J, lane, bw_eval);
}
// OPENING: 0 = a round of one evaluation per search; 1 = the first launch, both opening probes; 2 = the launch after it
// (reqd_work_group_size: where the compiler keeps a thread's temporaries in LDS it indexes them by the flat thread id,
// and without the sizes it reads them from the dispatch packet -- in host memory: 2-15 us on every workgroup's path)
template <int OPENING>
__global__ __launch_bounds__(kTile *kPairWaves) void loo_round_pairs_kernel(const LooRound r) {
  __shared__ double sExpTab[256];
  __shared__ double sPart[OPENING == 2 ? 2 : 1][kFusedMaxN / kTile];
  __shared__ double sBw;   // what the workgroup needs of the advanced search state: the variance of this evaluation,
  __shared__ int sPhase;   // and whether the search is over
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // golden's first two evaluations (x1, x2: src/CrossValidation.jl:57-66) do not depend on each other: the first launch
  // runs both, blockIdx.z >= D being the second probe of dimension z - D on its own slots, counters and shares
  const int probe = OPENING == 1 && static_cast<int>(blockIdx.z) >= r.D ? 1 : 0, d = blockIdx.z - probe * r.D;
  PairSet ps;
  ps.slots = r.partial + probe * (static_cast<int64_t>(r.D) * r.ngroups * r.ngroups * kTile);
  ps.arrivals = r.arrivals + probe * (r.D * r.ngroups * kCounterStride);
  // the tile shares of this launch: plane (round+1)&1 of [4][D][nfb]; the second probe's land in plane 3
  ps.shares = r.hpart + (static_cast<int64_t>(probe ? 3 : ((r.round + 1) & 1)) * r.D + d) * r.nfb;
  constexpr bool follow = OPENING == 2;  // the launch after the joint one: two evaluations to book
  if (threadIdx.x < 256) sExpTab[threadIdx.x] = kExp2Tab256[threadIdx.x];
  // the tile shares of the evaluation in flight: one load per thread (not T dependent ones by thread 0)
  if (static_cast<int>(threadIdx.x) < r.nfb) {
    sPart[0][threadIdx.x] = r.hpart[(static_cast<int64_t>(r.round & 1) * r.D + d) * r.nfb + threadIdx.x];
    if constexpr (follow) sPart[1][threadIdx.x] = r.hpart[(static_cast<int64_t>(3) * r.D + d) * r.nfb + threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Golden s = r.state[(r.round & 1) * r.D + d];
    golden_book(s, sPart[0], r.nfb);
    golden_decide(s);
    if constexpr (follow) {
      golden_book(s, sPart[1], r.nfb);
      golden_decide(s);
    }
    if (blockIdx.x == 0 && !probe) r.state[((r.round + 1) & 1) * r.D + d] = s;
    if (probe) {  // the second probe: the state as it will be once the first is booked (the booking needs no result)
      const double a2 = s.alpha * s.alpha;
      s.bcur = (s.bcur * a2) / a2;
      s.nevals += 1;
      s.phase = 1;
      s.pending = 0;
      golden_decide(s);
    }
    sBw = s.bw_eval;
    sPhase = s.phase;
  }
  __syncthreads();
  if (sPhase == 3) return;  // this dimension's search is over
  pairs_item(r, ps, d, blockIdx.x * kPairWaves + wave, lane, sBw, sExpTab);
}

// A SPECULATIVE round (golden_advance_spec above): blockIdx.z = probe * D + dimension; probe 0 evaluates the point that is
// certain, probes 1 and 2 the two points one of which golden will ask for next (FIRST: the opening, probes 0 and 1 = x1 and
// x2).  Every probe has slots, counters and shares of its own; the shares are double-buffered by launch parity.  Chosen by
// the host when three evaluations still fit the chip a few wavefronts deep (small marginals: a round is then mostly its
// fixed ~12 us of launch, prologue and hand-over, and two evaluations per launch nearly halve the search: 6 x 1000 points
// 0.36 -> 0.2x ms); the numbers golden sees are those of the plain rounds, bit for bit (same tiles, same order).
template <bool FIRST>
__global__ __launch_bounds__(kTile *kPairWaves) void loo_round_spec_kernel(const LooRound r) {
  __shared__ double sExpTab[256];
  __shared__ double sPart[3][kFusedMaxN / kTile];
  __shared__ double sBw;
  __shared__ int sPhase;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int probe = static_cast<int>(blockIdx.z) / r.D, d = static_cast<int>(blockIdx.z) - probe * r.D;
  const int64_t plane = static_cast<int64_t>(r.D) * r.nfb;
  PairSet ps;
  ps.slots = r.partial + probe * (static_cast<int64_t>(r.D) * r.ngroups * r.ngroups * kTile);
  ps.arrivals = r.arrivals + probe * (r.D * r.ngroups * kCounterStride);
  ps.shares = r.hpart + (static_cast<int64_t>((r.round + 1) & 1) * 3 + probe) * plane + static_cast<int64_t>(d) * r.nfb;
  if (threadIdx.x < 256) sExpTab[threadIdx.x] = kExp2Tab256[threadIdx.x];
  if constexpr (!FIRST) {
    if (static_cast<int>(threadIdx.x) < r.nfb)
      for (int p = 0; p < 3; ++p)
        sPart[p][threadIdx.x] = r.hpart[(static_cast<int64_t>(r.round & 1) * 3 + p) * plane + static_cast<int64_t>(d) * r.nfb + threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Golden s = r.state[(r.round & 1) * r.D + d];
    if constexpr (FIRST) {
      golden_decide(s);  // x1
      s.spec = 1;
    } else {
      golden_advance_spec(s, sPart[0], sPart[1], sPart[2], r.nfb);
    }
    if (blockIdx.x == 0 && probe == 0) r.state[((r.round + 1) & 1) * r.D + d] = s;
    if (probe > 0 && s.phase != 3) s = golden_candidate(s, probe == 1);
    sBw = s.bw_eval;
    sPhase = s.phase;
  }
  __syncthreads();
  if (sPhase == 3) return;  // this dimension's search is over, or the bracket closes before this candidate
  pairs_item(r, ps, d, blockIdx.x * kPairWaves + wave, lane, sBw, sExpTab);
}

// Round, second launch: p_q = w * (sum over groups) / norm / (1 - w); block partial of W_q * log p_q
// (evalAvgLogL, src/DualTree01.jl:450-474; a zero likelihood that carries weight makes it -Inf, :460-463).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void loo_round_entropy_kernel(const LooRound r) {
  __shared__ double red[THREADS];
  const int d = blockIdx.y;
  const Golden &s = r.state[((r.round + 1) & 1) * r.D + d];  // as advanced by this round's first launch
  if (s.phase == 3) return;
  const double inv_norm = 1.0 / (r.sqrt_2pi * sqrt(s.bw_eval));  // norm = (2 pi)^(1/2) * sqrt(bw), :325-330
  const int64_t q = static_cast<int64_t>(blockIdx.x) * THREADS + threadIdx.x;
  double term = 0.0;
  if (q < r.N) {
    double acc = 0.0;
    for (int g = 0; g < r.ngroups; ++g) acc += r.partial[(static_cast<int64_t>(d) * r.ngroups + g) * r.N + q];
    const double w = r.w;
    const double p = (acc * w) * inv_norm / (1.0 - w);
    if (p == 0.0) term = (w != 0.0) ? -INFINITY : 0.0;
    else term = log(p) * w;
  }
  red[threadIdx.x] = term;
  __syncthreads();
  for (int off = THREADS / 2; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) r.hpart[(static_cast<int64_t>((r.round + 1) & 1) * r.D + d) * r.nfb + blockIdx.x] = red[0];
}

// After the last round of a batch: book its evaluation; a search that thereby converges is finished here, one that
// goes on keeps the booked state (no evaluation pending) and decides again in the next batch's first round.
__global__ void loo_finalize_kernel(const LooRound r) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= r.D) return;
  Golden s = r.state[(r.round & 1) * r.D + d];
  if (r.spec) {  // speculative rounds: up to two evaluations to book (golden_advance_spec, stopping short of a new evaluation)
    const int64_t plane = static_cast<int64_t>(r.D) * r.nfb;
    const double *part = r.hpart + static_cast<int64_t>(r.round & 1) * 3 * plane + static_cast<int64_t>(d) * r.nfb;
    if (s.phase != 3 && s.pending) {
      const int ph = s.phase;
      golden_book(s, part, r.nfb);
      const bool right = s.f2 < s.f1;
      Golden next = s;
      golden_decide(next);
      if (next.phase != 3 && s.spec) {
        s = next;
        golden_book(s, part + ((ph == 0 || right) ? 1 : 2) * plane, r.nfb);
        next = s;
        golden_decide(next);
      }
      if (next.phase == 3) s = next;
    }
    s.spec = 0;
    r.state[(r.round & 1) * r.D + d] = s;
    return;
  }
  golden_book(s, r.hpart + (static_cast<int64_t>(r.round & 1) * r.D + d) * r.nfb, r.nfb);
  Golden next = s;
  golden_decide(next);
  r.state[(r.round & 1) * r.D + d] = (next.phase == 3) ? next : s;
}

}  // namespace

// The search itself, on stream `st` of the current device (the caller holds the DeviceGuard), for `nb` matrices of D x N
// at once: nm = nb * D one-dimensional searches advance in the SAME launches (a launch indexes marginals, not dimensions:
// blockIdx.z, the state, the shares and the counters are all per marginal, and every marginal's tiles and summation order
// depend on N only -- so each search is, bit for bit, the one a call of its own would run).  `points` is the host's copy
// of the (single) matrix, `d_points` the matrices already in HBM, one behind the other -- the product(s) a resident chain
// has just sampled (kdehip_density_from_device_points, kdehip_mul_device_batch): the marginals are then prepared straight
// from them, nothing is uploaded.  Marginals beyond kPrepMaxN points are prepared on the host and need `points` (nb = 1).
// Protocol: begin (everything up to the first batch of rounds enqueued) -> the caller synchronises `st` -> poll (done, or
// the next batch enqueued) -> ... -> finish (bandwidths).  Several searches (different N) can be in flight on one stream.
class kdehip::LoocvSearch {
 public:
  int begin(int nb, int D, int64_t N, const double *points, const double *d_points, hipStream_t st);
  int poll(bool *done);
  int finish(double *bw_out, int32_t *nevals_out);  // bw_out: nb * D standard deviations; nevals_out: nb counts
  int rounds() const { return rounds_; }
  int batches() const { return batches_; }
  ~LoocvSearch() {
    // an early error return waits for whatever has been enqueued before the device block and the pinned block go back
    // to the caches (where another thread may be handed them at once); on the regular path the stream is already idle
    if (armed_) (void)hipStreamSynchronize(st_);
    if (pin_) cached_host_free(pin_, pin_bytes_);
  }

 private:
  int enqueue_batch(int batch);
  LooRound r_{};
  DevBuf dev_;
  void *pin_ = nullptr;
  size_t pin_bytes_ = 0;
  Golden *h_state_ = nullptr;
  hipStream_t st_ = nullptr;
  bool armed_ = false, pairs_ = false;
  std::unique_ptr<PhaseTimer> timer_;  // kdehip_profile_phase_read(0): the batch of rounds in flight
  int nb_ = 0, D_ = 0, nm_ = 0, rounds_ = 0, batches_ = 0, pair_items_ = 0;
  int64_t qblocks_ = 0;
};

int kdehip::LoocvSearch::begin(int nb, int D, int64_t N, const double *points, const double *d_points, hipStream_t st) {
  st_ = st; nb_ = nb; D_ = D;
  const int nm = nm_ = nb * D;
  if (nb < 1 || nm > kLoocvMaxMarginals) return set_error(KDEHIP_ERR_UNSUPPORTED, "bandwidth search: too many marginals for one launch");
  if (!points && !(d_points && N <= kPrepMaxN)) return set_error(KDEHIP_ERR_ARG, "auto_bandwidth_run: no host copy of the points");
  if (nb > 1 && !(d_points && N <= kPrepMaxN)) return set_error(KDEHIP_ERR_ARG, "bandwidth search: a batch needs device matrices of at most 2048 points");

  // weights: ones -> /N (kde!(points,[1.0])) -> renormalised by the marginal's kde! (src/KDE01.jl:46,152): every
  // point ends up with the same weight w1
  const double w0 = 1.0 / static_cast<double>(N);
  double t = 0.0;
  for (int64_t i = 0; i < N; ++i) t += w0;
  const double w1 = w0 / t;

  const int64_t nchunks = (N + kLooChunk - 1) / kLooChunk;
  const int64_t qblocks = qblocks_ = (N + kLooThreads - 1) / kLooThreads;
  int64_t want = (int64_t(8) * device_cu_count() + qblocks * nm - 1) / (qblocks * nm);  // ~8 blocks per CU
  if (want < 1) want = 1;
  if (want > kEvalMaxGroups) want = kEvalMaxGroups;
  if (want > nchunks) want = nchunks;
  LooRound &r = r_;
  // one fused launch per round (KDEHIP_LOOCV_TWO_LAUNCH=1: the two-launch rounds at every size -- what the tests pin the
  // fused hand-over against)
  static const bool two_launch = [] { const char *e = std::getenv("KDEHIP_LOOCV_TWO_LAUNCH"); return e && e[0] == '1'; }();
  const bool pairs = pairs_ = N <= kFusedMaxN && !two_launch;
  const int ntiles = static_cast<int>((N + kTile - 1) / kTile);
  r.chunks_per_group = static_cast<int>((nchunks + want - 1) / want);
  r.ngroups = pairs ? ntiles : static_cast<int>((nchunks + r.chunks_per_group - 1) / r.chunks_per_group);
  r.nfb = pairs ? ntiles : static_cast<int>(qblocks);  // blocks of the log-likelihood reduction (64 / 256 queries each)
  r.N = N; r.D = nm; r.w = w1; r.round = 0;
  r.sqrt_2pi = std::pow(2.0 * M_PI, 1 / 2.0);

  // one device block: [points N*D | xo nm*N | partial nm*ngroups*N | hpart nm*nfb | state 2*nm]
  auto al = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t off_x = al(d_points ? 0 : sizeof(double) * N * D);
  const size_t off_part = al(off_x + sizeof(double) * N * nm);
  r.joint = pairs ? 1 : 0;
  // speculative rounds (three evaluations per launch, two of them booked: loo_round_spec_kernel) while three evaluations
  // are still only a few wavefronts per SIMD: nm T (T/2 + 1) tile pairs per evaluation on 4 SIMDs per CU
  // (KDEHIP_LOOCV_SPEC=<k>: up to k tile pairs per CU and evaluation; 0 = never; default 8)
  static const int spec_per_cu = [] { const char *e = std::getenv("KDEHIP_LOOCV_SPEC"); return e && e[0] ? std::atoi(e) : 8; }();
  const int pair_items = pair_items_ = ntiles * (ntiles / 2 + 1);  // per marginal: the diagonal and the offsets 1 .. T/2 of every tile
  r.spec = (pairs && static_cast<int64_t>(nm) * pair_items <= int64_t(spec_per_cu) * device_cu_count()) ? 1 : 0;
  const size_t off_h = al(off_part + sizeof(double) * nm * r.ngroups * (pairs ? 3 * int64_t(ntiles) * kTile : N));
  const size_t off_state = al(off_h + sizeof(double) * 6 * nm * r.nfb);
  const size_t off_arr = al(off_state + sizeof(Golden) * 2 * nm);
  const size_t total = off_arr + (pairs ? sizeof(unsigned) * 3 * nm * ntiles * kCounterStride : 0);
  KDEHIP_CHECK(dev_.alloc(total));
  unsigned char *base = dev_.as<unsigned char>();
  double *d_pts = reinterpret_cast<double *>(base);
  r.x = reinterpret_cast<double *>(base + off_x);
  r.partial = reinterpret_cast<double *>(base + off_part);
  r.hpart = reinterpret_cast<double *>(base + off_h);
  r.state = reinterpret_cast<Golden *>(base + off_state);
  r.arrivals = reinterpret_cast<unsigned *>(base + off_arr);
  pin_bytes_ = std::max(d_points ? size_t(0) : sizeof(double) * N * D, sizeof(Golden) * 2 * nm);
  KDEHIP_CHECK(cached_host_malloc(&pin_, pin_bytes_));
  h_state_ = static_cast<Golden *>(pin_);
  armed_ = true;
  timer_.reset(new PhaseTimer(kPhaseLoocv, st));

  if (N <= kPrepMaxN) {
    if (!d_points) {
      std::memcpy(pin_, points, sizeof(double) * N * D);
      KDEHIP_CHECK(hipMemcpyAsync(d_pts, pin_, sizeof(double) * N * D, hipMemcpyHostToDevice, st));
    }
    int64_t P = 1;
    while (P < N) P <<= 1;
    KDEHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(loocv_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(sizeof(double) * 6 * P)));  // up to 96 KiB; per call = per device
    hipLaunchKernelGGL(loocv_prep_kernel, dim3(nm), dim3(kPrepThreads), sizeof(double) * 6 * P, st, d_points ? d_points : d_pts, N, D,
                       nm, const_cast<double *>(r.x), r.state, r.arrivals, ntiles);
    KDEHIP_CHECK(hipGetLastError());
  } else {
    // large marginals: sort and interval arithmetic on the host (one thread per dimension), same state afterwards
    std::vector<double> xo(static_cast<size_t>(D) * N);
    std::vector<Golden> g(D);
    auto prep = [&](int d) {
      std::vector<double> xs(static_cast<size_t>(N));
      for (int64_t i = 0; i < N; ++i) xs[i] = xo[static_cast<size_t>(d) * N + i] = points[i * D + d];
      std::sort(xs.begin(), xs.end());
      double centre, half, minm = INFINITY;
      interval_stats(xs.data(), N + 1, 2 * N, N + 1, centre, half, minm);
      golden_init(g[d], minm, std::sqrt((2.0 * half) * (2.0 * half)));
    };
    std::vector<std::thread> th;
    for (int d = 1; d < D; ++d) th.emplace_back(prep, d);
    prep(0);
    for (auto &t2 : th) t2.join();
    KDEHIP_CHECK(hipMemcpyAsync(const_cast<double *>(r.x), xo.data(), sizeof(double) * D * N, hipMemcpyHostToDevice, st));
    KDEHIP_CHECK(hipMemcpyAsync(r.state, g.data(), sizeof(Golden) * D, hipMemcpyHostToDevice, st));
    if (pairs) KDEHIP_CHECK(hipMemsetAsync(r.arrivals, 0, sizeof(unsigned) * 3 * D * ntiles * kCounterStride, st));
    KDEHIP_CHECK(hipStreamSynchronize(st));  // (xo and g are pageable and leave scope)
  }
  // (20 evaluations in the first batch: the first launch of `pairs` runs two; speculative launches book two each)
  return enqueue_batch(r.spec ? 10 : (r.joint ? 19 : 20));
}

// (Round 5 built the whole search as ONE persistent launch -- workgroups that stay, a tile counter per dimension between
// the rounds: bit-identical and SLOWER, 457 against 410 us at 6 x 2048: noticing a counter from another XCD and fetching
// the shares behind it cost more than the 3.5 us launch gap they replace.  Removed; profiles/r05_experiments.md section 8.)
int kdehip::LoocvSearch::enqueue_batch(int batch) {
  LooRound &r = r_;
  hipStream_t st = st_;
  const unsigned nm = static_cast<unsigned>(nm_);
  const dim3 gridA(static_cast<unsigned>(qblocks_), static_cast<unsigned>(r.ngroups), nm);
  const dim3 gridB(static_cast<unsigned>(r.nfb), nm);
  const dim3 gridP(static_cast<unsigned>((pair_items_ + kPairWaves - 1) / kPairWaves), 1, nm);
  const dim3 gridP2(gridP.x, 1, 2 * nm);  // the joint first launch
  const dim3 gridS2(gridP.x, 1, 2 * nm), gridS3(gridP.x, 1, 3 * nm);
  for (int k = 0; k < batch; ++k) {
    if (r.spec) {
      if (r.round == 0) hipLaunchKernelGGL(loo_round_spec_kernel<true>, gridS2, dim3(kTile * kPairWaves), 0, st, r);
      else hipLaunchKernelGGL(loo_round_spec_kernel<false>, gridS3, dim3(kTile * kPairWaves), 0, st, r);
    } else if (pairs_) {
      if (r.joint && r.round == 0) hipLaunchKernelGGL(loo_round_pairs_kernel<1>, gridP2, dim3(kTile * kPairWaves), 0, st, r);
      else if (r.joint && r.round == 1) hipLaunchKernelGGL(loo_round_pairs_kernel<2>, gridP, dim3(kTile * kPairWaves), 0, st, r);
      else hipLaunchKernelGGL(loo_round_pairs_kernel<0>, gridP, dim3(kTile * kPairWaves), 0, st, r);
    } else {
      hipLaunchKernelGGL(loo_round_partial_kernel, gridA, dim3(kLooThreads), 0, st, r);
      hipLaunchKernelGGL(loo_round_entropy_kernel<kLooThreads>, gridB, dim3(kLooThreads), 0, st, r);
    }
    ++r.round;
    ++rounds_;
  }
  hipLaunchKernelGGL(loo_finalize_kernel, dim3((nm + 63) / 64), dim3(64), 0, st, r);
  KDEHIP_CHECK(hipGetLastError());
  if (timer_) timer_->stop();
  KDEHIP_CHECK(hipMemcpyAsync(h_state_, r.state + (r.round & 1) * nm_, sizeof(Golden) * nm_, hipMemcpyDeviceToHost, st));
  return KDEHIP_OK;
}

// After the caller has synchronised the stream: have all searches converged?  If not, the next batch of rounds is enqueued.
int kdehip::LoocvSearch::poll(bool *done) {
  ++batches_;
  if (timer_) { timer_->collect(); timer_.reset(); }
  bool all = true;
  for (int m = 0; m < nm_; ++m) all = all && h_state_[m].phase == 3;
  *done = all;
  if (all) { armed_ = false; return KDEHIP_OK; }
  if (batches_ >= 16) return set_error(KDEHIP_ERR_HIP, "bandwidth search did not converge");
  timer_.reset(new PhaseTimer(kPhaseLoocv, st_));
  return enqueue_batch(r_.spec ? 4 : 8);
}

int kdehip::LoocvSearch::finish(double *bw_out, int32_t *nevals_out) {
  for (int b = 0; b < nb_; ++b) {
    int total_evals = 0;
    for (int d = 0; d < D_; ++d) {
      const Golden &g = h_state_[b * D_ + d];
      if (g.phase != 3) return set_error(KDEHIP_ERR_HIP, "bandwidth search did not converge");
      const double ks = g.result * (g.minm + g.maxm) / 2.0;  // ksize, src/CrossValidation.jl:117
      bw_out[b * D_ + d] = std::sqrt(ks * ks);               // getBW of kde!(.., [ks]) (src/KDE01.jl:45,118)
      total_evals += g.nevals;
    }
    if (nevals_out) nevals_out[b] = total_evals;
  }
  return KDEHIP_OK;
}

kdehip::LoocvSearch *kdehip::loocv_new() { return new (std::nothrow) LoocvSearch(); }
void kdehip::loocv_delete(LoocvSearch *s) { delete s; }
int kdehip::loocv_begin(LoocvSearch *s, int nb, int D, int64_t N, const double *d_points, void *stream) {
  return s->begin(nb, D, N, nullptr, d_points, static_cast<hipStream_t>(stream));
}
int kdehip::loocv_poll(LoocvSearch *s, bool *done) { return s->poll(done); }
int kdehip::loocv_finish(LoocvSearch *s, double *bw_out, int32_t *nevals_out) { return s->finish(bw_out, nevals_out); }

int kdehip::auto_bandwidth_run(int D, int64_t N, const double *points, const double *d_points, void *stream,
                               double *bw_out, int32_t *nevals_out, const std::function<void()> *overlap) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool timing = std::getenv("KDEHIP_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_begin = tnow();
  LoocvSearch search;
  int rc = search.begin(1, D, N, points, d_points, st);
  if (rc != KDEHIP_OK) return rc;
  auto t_prep = tnow();
  if (overlap) (*overlap)();  // (the first batch is in flight: the caller's host work runs under it)
  for (bool done = false; !done;) {
    KDEHIP_CHECK(hipStreamSynchronize(st));
    rc = search.poll(&done);
    if (rc != KDEHIP_OK) return rc;
  }
  rc = search.finish(bw_out, nevals_out);
  if (rc != KDEHIP_OK) return rc;
  if (timing) {
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::fprintf(stderr, "kdehip_auto_bandwidth D=%d N=%lld: upload + prep + first batch enqueue %.0f us, %d rounds in %d batches %.0f us\n", D,
                 static_cast<long long>(N), us(t_begin, t_prep), search.rounds(), search.batches(), us(t_prep, tnow()));
  }
  return KDEHIP_OK;
}

extern "C" int kdehip_auto_bandwidth(int64_t D64, int64_t N, const double *points, double *bw_out,
                                     int32_t *nevals_out, int device) {
  if (!points || !bw_out) return set_error(KDEHIP_ERR_ARG, "null argument");
  if (D64 < 1 || D64 > KDEHIP_MAX_DIMS) return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  if (N < 2) return set_error(KDEHIP_ERR_ARG, "kde!(points) needs at least two points");
  DeviceGuard guard;
  const int rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;
  try {  // (host-prepared marginals use std::vector / std::thread: nothing may throw out of an extern "C" entry point)
    return auto_bandwidth_run(static_cast<int>(D64), N, points, nullptr, hipStreamPerThread, bw_out, nevals_out);
  } catch (const std::exception &e) {
    return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_auto_bandwidth: ") + e.what());
  }
}

// kde!(points) (src/KDE01.jl:3-27) with the tree built under the bandwidth search: the builder is a task of the host
// pool (and hands its own subtrees on from there), the search keeps this thread until the GPU is done.
extern "C" int kdehip_make_density_auto(int64_t D, int64_t N, const double *points, double *bw_out, int32_t *nevals, int device,
                                        double *centers, double *ranges, double *weights, int64_t *left_child,
                                        int64_t *right_child, int64_t *lowest_leaf, int64_t *highest_leaf,
                                        int64_t *permutation, double *means, double *bandwidth, double *bandwidthMin,
                                        double *bandwidthMax) {
  using namespace kdehip;
  if (D < 1 || N < 2) return set_error(KDEHIP_ERR_ARG, "kdehip_make_density_auto: need D >= 1 and N >= 2");
  if (!points || !bw_out || !centers || !ranges || !weights || !left_child || !right_child || !lowest_leaf ||
      !highest_leaf || !permutation || !means || !bandwidth || !bandwidthMin || !bandwidthMax)
    return set_error(KDEHIP_ERR_ARG, "kdehip_make_density_auto: null pointer");
  int tree_rc = KDEHIP_OK, rc = KDEHIP_OK;
  try {
    TaskGroup group(HostPool::get());
    group.run([&] {
      const double one = 1.0;  // (placeholder bandwidth: only `bandwidth`, bandwidthMin/Max depend on it)
      tree_rc = kdehip_make_density(D, N, points, &one, 1, nullptr, centers, ranges, weights, left_child, right_child,
                                    lowest_leaf, highest_leaf, permutation, means, bandwidth, bandwidthMin, bandwidthMax);
    });
    rc = kdehip_auto_bandwidth(D, N, points, bw_out, nevals, device);
    group.wait();
  } catch (const std::exception &e) {
    return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_make_density_auto: ") + e.what());
  }
  if (rc != KDEHIP_OK) return rc;  // (message set by the search, on this thread)
  if (tree_rc != KDEHIP_OK) return set_error(tree_rc, "kdehip_make_density_auto: the tree build failed");
  return kdehip_density_set_bandwidth(D, N, bw_out, D, weights, left_child, right_child, means, bandwidth, bandwidthMin,
                                      bandwidthMax);
}

#ifdef KDEHIP_PREP_STAMPS
extern "C" int kdehip_debug_prep_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prep_stamps), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -5;
}
#endif
