// gibbs_kernel.hip -- the multiscale-Gibbs product sampler on gfx950 (MI355X), hand-written HIP.
//
// Replaces the sample loop of `gibbs1` (reference src/MSGibbs01.jl:581-626) together with
// makeFasterSampleIndex! (:250-328), selectLabelOnLevel (:330-351), gaussianProductMeanCov!
// (:176-216), samplePoint! (:440-463), sampleIndices! (:364-385), sampleIndex (:404-429) and the
// label bookkeeping of levelDown! (:500-523; the frontiers themselves are pre-expanded by
// pack_levels.cpp).
//
// Mapping (MI355X-first, not a translation of the Julia loop nest):
//   * one 64-lane wavefront owns one output sample (one Gibbs chain); chains never communicate;
//   * inside a (level, density) step the lanes are the frontier nodes: lane `ln` owns the
//     contiguous entries ln*B .. ln*B+B-1 (B = ceil(n/64)), reads them with coalesced loads from the
//     lane-blocked tile, keeps a private running sum, and ONE DPP wavefront prefix scan turns the
//     64 lane sums into the cumulative weights the categorical draw needs; the winning lane's
//     block is then re-evaluated by the whole wavefront (one more pass over <= 64 nodes) and
//     scanned again to find the node.  No p[] array is ever materialised;
//   * between steps the lanes are the DIMENSIONS: lane d keeps 1/variance and mean/variance of the
//     currently selected kernel of every density for dimension d (LDS, per wavefront), forms the
//     leave-one-out Gaussian product for its dimension, and the D results are broadcast to scalar
//     registers with v_readlane;
//   * random numbers come from the caller's streams (bit-for-bit the reference's consumption
//     order) or from an on-device Philox4x32-10 keyed by (seed, global sample, draw).
//
// Two arithmetic forms of the kernel evaluation:
//   FAST    p = w * rsqrt(prod_d c_d) * exp(-1/2 * sum_d delta_d^2 / c_d), with the D reciprocals
//           obtained from ONE rsqrt via prefix/suffix products -- no divide, no log;
//   GENERIC the reference's own per-dimension divide + log with its NaN rules (:287-303), used for
//           partialDimMask products and for inputs whose variance products could leave the range of T.
//
// Compiled with -ffp-contract=off; fused multiply-adds are written explicitly where wanted.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kdehip_internal.hpp"
#include "philox.hpp"

namespace kdehip {

constexpr int kWavesPerBlock = 4;

// ---- small device helpers ------------------------------------------------------------------------

template <typename T> struct Num;
template <> struct Num<double> {
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return ::rsqrt(x); }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
  static __device__ __forceinline__ double tiny_total() { return 1e-99; }  // :311
};
template <> struct Num<float> {
  static __device__ __forceinline__ float exp(float x) { return __expf(x); }
  static __device__ __forceinline__ float log(float x) { return __logf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return ::rsqrtf(x); }
  static __device__ __forceinline__ float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
  static __device__ __forceinline__ float tiny_total() { return 1e-37f; }  // 1e-99 is not a float
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fetch(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

// Inclusive prefix sum over the 64 lanes of a wavefront with DPP row shifts / row broadcasts
// (lanes without a source read 0).
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
  v += dpp_fetch<0x111, 0xF>(v);  // row_shr:1
  v += dpp_fetch<0x112, 0xF>(v);  // row_shr:2
  v += dpp_fetch<0x114, 0xF>(v);  // row_shr:4
  v += dpp_fetch<0x118, 0xF>(v);  // row_shr:8
  v += dpp_fetch<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3
  v += dpp_fetch<0x143, 0xC>(v);  // row_bcast:31 -> rows 2 and 3
  return v;
}

// Orders this wavefront's LDS traffic (written by some lanes, read by others) in the compiler;
// the hardware executes one wavefront's LDS instructions in order.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double lane_read(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_read(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// ---- kernel evaluation at one frontier position ------------------------------------------------

// FAST form.  center[d] / cov[d] are wave-uniform.  Returns w * N(center; mean, bw + cov).
template <typename T, int D>
__device__ __forceinline__ T eval_fast(const T *__restrict__ tile, int ld, int pos,
                                       const T (&center)[D], const T (&cov)[D]) {
  T c[D], d2[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const T mu = tile[d * ld + pos];
    const T bw = tile[(D + d) * ld + pos];
    c[d] = bw + cov[d];
    const T dl = mu - center[d];
    d2[d] = dl * dl;
  }
  const T w = tile[2 * D * ld + pos];
  // others[d] = prod_{k != d} c[k] from prefix and suffix products; P = prod_k c[k]
  T pre[D], suf[D];
  pre[0] = T(1);
#pragma unroll
  for (int d = 1; d < D; ++d) pre[d] = pre[d - 1] * c[d - 1];
  suf[D - 1] = T(1);
#pragma unroll
  for (int d = D - 2; d >= 0; --d) suf[d] = suf[d + 1] * c[d + 1];
  const T P = pre[D - 1] * c[D - 1];
  T num = T(0);
#pragma unroll
  for (int d = 0; d < D; ++d) num = Num<T>::fma(d2[d], pre[d] * suf[d], num);
  const T r = Num<T>::rsqrt(P);
  const T q = num * r * r;  // = sum_d delta_d^2 / c_d
  const T p = w * r * Num<T>::exp(T(-0.5) * q);
  return (p != p) ? T(0) : p;  // suppress NaNs, :302
}

// GENERIC form: literally the reference's accumulation (:280-303) incl. inactive dimensions.
template <typename T, int D>
__device__ __forceinline__ T eval_generic(const T *__restrict__ tile, int ld, int pos,
                                          const T (&center)[D], const T (&cov)[D], uint32_t act) {
  T acc = T(0);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if ((act >> d) & 1u) {
      const T c = tile[(D + d) * ld + pos] + cov[d];
      const T dl = tile[d * ld + pos] - center[d];
      const T distr = (dl * dl) / c;
      if (distr == distr) {
        acc += distr;
        acc += Num<T>::log(c);
      }
    }
  }
  const T p = Num<T>::exp(T(-0.5) * acc) * tile[2 * D * ld + pos];
  return (p != p) ? T(0) : p;
}

template <typename T, int D, bool FAST>
__device__ __forceinline__ T eval_node(const T *__restrict__ tile, int ld, int pos,
                                       const T (&center)[D], const T (&cov)[D], uint32_t act) {
  if constexpr (FAST) return eval_fast<T, D>(tile, ld, pos, center, cov);
  else return eval_generic<T, D>(tile, ld, pos, center, cov, act);
}

// ---- one categorical label draw over a frontier -------------------------------------------------
// Evaluates every node of the frontier against (center, cov), and returns the 0-based frontier
// entry selected by the uniform draw `u`: the first z with u <= cdf[z], else the last
// (selectLabelOnLevel :330-351 applied to the CDF of makeFasterSampleIndex! :318-325).
template <typename T, int D, bool FAST>
__device__ __forceinline__ int draw_label(const T *__restrict__ tile, const LevelDesc &ds, int lane,
                                          const T (&center)[D], const T (&cov)[D], uint32_t act,
                                          double u) {
  const int n = ds.n, B = ds.B, ld = B * 64;
  // pass 1: private sum over the lane's contiguous entries (rows of the tile are coalesced)
  T S = T(0);
  for (int i = 0; i < B; ++i) S += eval_node<T, D, FAST>(tile, ld, i * 64 + lane, center, cov, act);
  const T incl = wave_inclusive_scan(S);
  const T total = lane_read(incl, 63);

  if (total < Num<T>::tiny_total()) {
    // "stick with selection of others": uniform over the frontier (:311-315); with a zero/NaN
    // last weight the reference's CDF is all-NaN and the last entry is taken.
    const int zl = n - 1;
    const T wl = tile[2 * D * ld + (zl % B) * 64 + zl / B];
    if (!(wl > T(0))) return n - 1;
    int z = static_cast<int>(ceil(u * static_cast<double>(n))) - 1;
    return z < 0 ? 0 : (z > n - 1 ? n - 1 : z);
  }

  const T target = static_cast<T>(u) * total;
  const unsigned long long hit = __ballot(target <= incl);
  const int last_lane = (n - 1) / B;
  int lstar = hit ? (__ffsll(hit) - 1) : last_lane;
  if (lstar > last_lane) lstar = last_lane;
  if (B == 1) return lstar;

  // pass 2: narrow inside the winning lane's block until a single node is left
  T base = lane_read(incl - S, lstar);  // exclusive prefix of the block
  int r0 = 0;
  int len = n - lstar * B;
  if (len > B) len = B;
  while (len > 64) {  // only for frontiers beyond 4096 nodes
    const int b2 = (len + 63) / 64;
    T S2 = T(0);
    for (int i = 0; i < b2; ++i) {
      const int r = lane * b2 + i;
      if (r < len) S2 += eval_node<T, D, FAST>(tile, ld, (r0 + r) * 64 + lstar, center, cov, act);
    }
    const T inc2 = wave_inclusive_scan(S2);
    const unsigned long long h2 = __ballot(target <= base + inc2);
    const int lastl = (len - 1) / b2;
    int l2 = h2 ? (__ffsll(h2) - 1) : lastl;
    if (l2 > lastl) l2 = lastl;
    base += lane_read(inc2 - S2, l2);
    r0 += l2 * b2;
    len = (len - l2 * b2 < b2) ? (len - l2 * b2) : b2;
  }
  T p2 = T(0);
  if (lane < len) p2 = eval_node<T, D, FAST>(tile, ld, (r0 + lane) * 64 + lstar, center, cov, act);
  const T inc3 = wave_inclusive_scan(p2);
  const unsigned long long h3 = __ballot((target <= base + inc3) && (lane < len));
  const int istar = h3 ? (__ffsll(h3) - 1) : (len - 1);
  return lstar * B + r0 + istar;
}

// ---- the sampler ----------------------------------------------------------------------------------

template <typename T, int D, bool FAST>
__global__ __launch_bounds__(kWavesPerBlock * 64) void gibbs_product_kernel(PlanDev plan, RunArgs a) {
  __shared__ T sLam[kWavesPerBlock][KDEHIP_MAX_DENS * D];  // 1/variance of the selected kernels
  __shared__ T sLmu[kWavesPerBlock][KDEHIP_MAX_DENS * D];  // mean/variance
  __shared__ int sZ[kWavesPerBlock][KDEHIP_MAX_DENS];      // selected frontier entry per density

  const int lane = threadIdx.x & 63;
  // readfirstlane makes the wave id (and everything derived from it: sample index, RNG counters,
  // descriptor addresses) provably wave-uniform, so it lives in SGPRs / runs on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t s = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  if (s >= a.Np) return;  // wave-uniform; no block-wide barriers are used below
  const uint64_t gs = static_cast<uint64_t>(a.sample_offset + s);

  const int M = plan.M, L = plan.L;
  const T *__restrict__ data = static_cast<const T *>(plan.data);
  T *lam = sLam[wave];
  T *lmu = sLmu[wave];
  int *zsel = sZ[wave];
  const int dl = lane < D ? lane : D - 1;  // this lane's dimension in the "lanes = dimensions" phases

  uint32_t any_bits = 0;  // dimensions informed by at least one density
  for (int j = 0; j < M; ++j) any_bits |= plan.mask_bits[j];

  // selected kernel of density j <- frontier entry z of level descriptor ds
  // (updateGlbParticlesVariance!, :89-115; masked dimensions carry no information)
  auto set_particle = [&](int j, const LevelDesc &ds, int z) {
    const int B = ds.B, ld = B * 64;
    const int pos = (z % B) * 64 + z / B;
    const T *tile = data + ds.data_off;
    const T mu = tile[dl * ld + pos];
    const T var = tile[(D + dl) * ld + pos];
    const bool on = (plan.mask_bits[j] >> dl) & 1u;
    const T l = on ? T(1) / var : T(0);
    if (lane < D) {
      lam[j * D + dl] = l;
      lmu[j * D + dl] = on ? mu * l : T(0);
    }
    if (lane == 0) zsel[j] = z;
  };

  // Gaussian product of the selected kernels without density `skip` for this lane's dimension
  // (gaussianProductMeanCov!, :176-216): cov = 1/sum(lambda), mean = cov * sum(mu*lambda).
  auto product_dim = [&](int skip, uint32_t info_bits, T &mean, T &cov) {
    T ls = T(0), ms = T(0);
    for (int k = 0; k < M; ++k) {
      if (k != skip) {
        ls += lam[k * D + dl];
        ms += lmu[k * D + dl];
      }
    }
    const bool on = (info_bits >> dl) & 1u;
    cov = on ? T(1) / ls : T(0);
    mean = on ? cov * ms : T(0);
  };

  // init: frontier = {root}, label = root (levelInit!/initIndices!/calcIndices!, :587-589)
  for (int j = 0; j < M; ++j) set_particle(j, plan.levels[j * (L + 1)], 0);
  wave_sync();

  uint32_t c = static_cast<uint32_t>(M);  // select-call counter; the M init calls read nothing
  auto next_uniform = [&]() -> double {
    double u;
    if (a.rng_philox) u = philox_uniform(a.seed, gs, c);
    else u = a.randU[s * a.K + static_cast<int64_t>(c) - 1];
    ++c;
    return u;
  };
  auto normal_for_lane = [&](int q) -> double {  // normal (q, dl) of this sample
    const uint32_t r = static_cast<uint32_t>(q * D + dl);
    if (a.rng_philox) return philox_normal(a.seed, gs, r);
    return a.randN[s * a.R + r];
  };

  T xs[D], zero[D];
#pragma unroll
  for (int d = 0; d < D; ++d) zero[d] = T(0);

  for (int l = 1; l <= L; ++l) {
    // samplePoint! (:440-463): x = mean + sqrt(cov) * randn, all densities included
    {
      T mean, cov;
      product_dim(-1, any_bits, mean, cov);
      const T x = mean + Num<T>::sqrt(cov) * static_cast<T>(normal_for_lane(l - 1));
#pragma unroll
      for (int d = 0; d < D; ++d) xs[d] = lane_read(x, d);
    }
    // sampleIndices! (:364-385): every density draws a label on the new frontier against x
    for (int j = 0; j < M; ++j) {
      const LevelDesc ds = plan.levels[j * (L + 1) + l];
      const uint32_t act = plan.mask_bits[j] & plan.others_bits[j];
      const double u = next_uniform();
      const int z = draw_label<T, D, FAST>(data + ds.data_off, ds, lane, xs, zero, act, u);
      if (lane == 0) zsel[j] = z;
    }
    wave_sync();
    for (int j = 0; j < M; ++j) set_particle(j, plan.levels[j * (L + 1) + l], zsel[j]);
    wave_sync();

    // sequential Gibbs sweeps (:604-609): leave density j out, redraw its label
    for (int it = 0; it < a.Niter; ++it) {
      for (int j = 0; j < M; ++j) {
        const LevelDesc ds = plan.levels[j * (L + 1) + l];
        T mean, cov;
        product_dim(j, plan.others_bits[j], mean, cov);
        T mc[D], cc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
          mc[d] = lane_read(mean, d);
          cc[d] = lane_read(cov, d);
        }
        const uint32_t act = plan.mask_bits[j] & plan.others_bits[j];
        const double u = next_uniform();
        const int z = draw_label<T, D, FAST>(data + ds.data_off, ds, lane, mc, cc, act, u);
        wave_sync();
        set_particle(j, ds, z);
        wave_sync();
      }
    }
    if (a.labels && lane == 0) {
      for (int j = 0; j < M; ++j) {
        const LevelDesc ds = plan.levels[j * (L + 1) + l];
        const int z = zsel[j];
        a.labels[(s * M + j) * L + (l - 1)] = plan.perm[ds.perm_off + (z % ds.B) * 64 + z / ds.B];
      }
    }
  }

  // final labels (:612-616) and final point (:625)
  if (lane == 0) {
    for (int j = 0; j < M; ++j) {
      const LevelDesc ds = plan.levels[j * (L + 1) + L];
      const int z = zsel[j];
      a.indices[s * M + j] =
          static_cast<int64_t>(plan.perm[ds.perm_off + (z % ds.B) * 64 + z / ds.B]) + 1;
    }
  }
  {
    T mean, cov;
    product_dim(-1, any_bits, mean, cov);
    T x = mean;
    if (a.addEntropy) x = mean + Num<T>::sqrt(cov) * static_cast<T>(normal_for_lane(L));
    if (lane < D) a.points[s * D + lane] = static_cast<double>(x);
  }
}

// ---- launcher --------------------------------------------------------------------------------------

template <typename T, int D, bool FAST>
static int launch_one(const PlanDev &plan, const RunArgs &args, hipStream_t stream) {
  const int64_t blocks = (args.Np + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks <= 0) return KDEHIP_OK;
  hipLaunchKernelGGL((gibbs_product_kernel<T, D, FAST>), dim3(static_cast<unsigned>(blocks)),
                     dim3(kWavesPerBlock * 64), 0, stream, plan, args);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return set_error(KDEHIP_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

template <typename T, bool FAST>
static int launch_dims(const PlanDev &plan, const RunArgs &args, hipStream_t stream) {
  switch (plan.D) {
    case 1: return launch_one<T, 1, FAST>(plan, args, stream);
    case 2: return launch_one<T, 2, FAST>(plan, args, stream);
    case 3: return launch_one<T, 3, FAST>(plan, args, stream);
    case 4: return launch_one<T, 4, FAST>(plan, args, stream);
    case 5: return launch_one<T, 5, FAST>(plan, args, stream);
    case 6: return launch_one<T, 6, FAST>(plan, args, stream);
    case 7: return launch_one<T, 7, FAST>(plan, args, stream);
    case 8: return launch_one<T, 8, FAST>(plan, args, stream);
    default: return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  }
}

int launch_gibbs(int precision, bool fast, const PlanDev &plan, const RunArgs &args, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (precision == 64)
    return fast ? launch_dims<double, true>(plan, args, st) : launch_dims<double, false>(plan, args, st);
  if (precision == 32)
    return fast ? launch_dims<float, true>(plan, args, st) : launch_dims<float, false>(plan, args, st);
  return set_error(KDEHIP_ERR_ARG, "precision must be 64 or 32");
}

}  // namespace kdehip
