// gibbs_kernel.hip -- the multiscale-Gibbs product sampler on gfx950 (MI355X), hand-written HIP.
//
// Replaces the sample loop of `gibbs1` (reference src/MSGibbs01.jl:581-626) together with
// makeFasterSampleIndex! (:250-328), selectLabelOnLevel (:330-351), gaussianProductMeanCov!
// (:176-216), samplePoint! (:440-463), sampleIndices! (:364-385), sampleIndex (:404-429) and the
// label bookkeeping of levelDown! (:500-523; the frontiers themselves are pre-expanded by
// pack_levels.cpp).
//
// Mapping (MI355X-first, not a translation of the Julia loop nest):
//   * one 64-lane wavefront owns one output sample (one Gibbs chain); chains never communicate.
//     4, 8, 12 or 16 chains (picked per launch, see launch_one) share a workgroup, one workgroup per CU,
//     and walk the data-independent (level, pass, density) schedule in lock step so that the tile every wavefront
//     is about to read is staged ONCE per workgroup into LDS with direct-to-LDS loads
//     (buffer_load_dwordx4 ... lds): levels whose tiles all fit the 120 KiB pool stay resident for the
//     whole level; larger ones are streamed one tile per step through a double buffer (the copy of
//     step t+1 overlaps the evaluation of step t, one barrier per step); tiles beyond half the pool
//     are streamed through the two halves a few rows at a time (one barrier per chunk);
//   * inside a (level, density) step the lanes are the frontier nodes: lane `ln` owns the
//     contiguous entries ln*B .. ln*B+B-1 (B = ceil(n/64)), reads them row by row with coalesced
//     loads (one address per row, fields at constant offsets; the next row is requested while the
//     current one is evaluated, two rows interleaved per trip), keeps a private running sum, and
//     ONE DPP wavefront prefix scan turns the 64 lane sums into the cumulative weights the
//     categorical draw needs; the winning lane's block is then re-evaluated by the whole wavefront
//     (one more pass over <= 64 nodes) and scanned again to find the node.  No p[] array is ever
//     materialised;
//   * between steps the lanes are the DIMENSIONS: lane d keeps 1/variance and mean/variance of the
//     currently selected kernel of every density for dimension d (LDS, per wavefront), forms the
//     leave-one-out Gaussian product for its dimension, and the D results are broadcast to scalar
//     registers with v_readlane;
//   * random numbers come from the caller's streams (bit-for-bit the reference's consumption
//     order) or from an on-device Philox4x32-10 keyed by (seed, global sample, draw), which the
//     compiler runs on the scalar unit because every input is wave-uniform.
//
// Arithmetic forms of the kernel evaluation p_z = w_z * N(center; mean_z, bw_z + cov) (kernel
// template MODE: generic, fast, fast with inactive dimensions):
//   UNIFORM  levels whose nodes share one bandwidth vector (every leaf level): the D reciprocals and
//            the normalisation are wave-uniform and hoisted; per node D subtracts, D multiplies,
//            D fused multiply-adds and one exp;
//   FAST     p = w * rsqrt(prod_d c_d) * exp(-1/2 * sum_d delta_d^2 / c_d), the D reciprocals
//            obtained from ONE rsqrt via prefix/suffix products -- no divide, no log;
//   GENERIC  the reference's own per-dimension divide + log with its NaN rules (:287-303), used for
//            inputs whose variance products could leave the range of T or are not finite/positive.
// partialDimMask products (and one-density "products") run UNIFORM/FAST with the inactive dimensions
// contributing c = 1, delta = 0 (MODE = fast-masked).
// fp64 exp on the fast forms is a 32-entry-table (256 bytes of LDS) + degree-6
// polynomial, ~1 ulp; the GENERIC form calls the library exp/log.
//
// Compiled with -ffp-contract=off; fused multiply-adds are written explicitly where wanted.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "fastexp.hpp"
#include "kdehip_internal.hpp"
#include "philox.hpp"

namespace kdehip {

// ---- small device helpers ------------------------------------------------------------------------

// 1/x for positive, finite, normal x (the fast paths guarantee that at pack time): hardware
// reciprocal + two Newton steps, ~half the dependent latency of the IEEE division sequence.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return fmaf(r, fmaf(-x, r, 1.0f), r);
}

template <typename T> struct Num;
template <> struct Num<double> {
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  // the fast forms hand exp_fast an exponent already multiplied by kExpArg (fp32: log2 e, so that the
  // hardware's base-2 exponential needs no extra multiply; fp64: 1)
  static constexpr double kExpArg = 1.0;
  static __device__ __forceinline__ double exp_fast(double x, const double *tab) { return exp_nonpos(x, tab); }
  using ExpMid = ExpSplit;  // exp_fast in two halves (table lookup issued / result formed)
  static __device__ __forceinline__ ExpMid exp_begin(double x, const double *tab) { return exp_nonpos_begin(x, tab); }
  static __device__ __forceinline__ double exp_end(const ExpMid &m) { return exp_nonpos_end(m); }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  // 1/sqrt(x) for positive, finite, normal x (guaranteed on the fast forms at pack time): the library's
  // refinement of v_rsq_f64 without its zero/infinity handling -- same result, three instructions fewer
  static __device__ __forceinline__ double rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
  }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
  static __device__ __forceinline__ double tiny_total() { return 1e-99; }  // :311
  static constexpr int kOffsetSteps = 0;   // fp64 holds pT < 1e-99 directly: one attempt
  static constexpr double kOffsetStep = 0.0;
  static __device__ __forceinline__ double final_total() { return 1e-99; }
};
template <> struct Num<float> {
  static __device__ __forceinline__ float exp(float x) { return __expf(x); }
  static constexpr float kExpArg = 1.44269504088896340736f;
  static __device__ __forceinline__ float exp_fast(float x, const double *) { return __builtin_amdgcn_exp2f(x); }
  struct ExpMid { float x; };
  static __device__ __forceinline__ ExpMid exp_begin(float x, const double *) { return {x}; }
  static __device__ __forceinline__ float exp_end(const ExpMid &m) { return __builtin_amdgcn_exp2f(m.x); }
  static __device__ __forceinline__ float log(float x) { return __logf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }  // normal x: no denormal scaling
  static __device__ __forceinline__ float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
  // The reference's "pT < 1e-99 -> uniform draw" (:311-315) in a type whose exp underflows at 2^-126: the sum is
  // formed with every exponent raised by a wave-uniform offset o = 0, 110, 220, 330 binades (first o whose sum
  // reaches 2^-100, so that every term within 2^-26 of the largest one is still a normal number); the selection
  // only needs relative values, and at o = 330 the reference's threshold 1e-99 * 2^330 = 2.19 is representable:
  // the fallback fires exactly where the fp64 sum would be below 1e-99 (up to fp32 rounding of the sum).
  static __device__ __forceinline__ float tiny_total() { return 0x1p-100f; }
  static constexpr int kOffsetSteps = 3;
  static constexpr float kOffsetStep = 110.0f;
  static __device__ __forceinline__ float final_total() { return static_cast<float>(1e-99 * 0x1p330); }
};

// Two fp32 entries per lane in one register pair: gfx950 executes v_pk_add/mul/fma_f32 on both halves at
// the rate of one scalar fp32 instruction, so the fp32 first pass evaluates the rows two at a time.
typedef float kdehip_f2 __attribute__((ext_vector_type(2)));
template <> struct Num<kdehip_f2> {
  static __device__ __forceinline__ kdehip_f2 exp_fast(kdehip_f2 x, const double *) {
    return {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
  }
  static __device__ __forceinline__ kdehip_f2 rsqrt(kdehip_f2 x) { return {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
  static __device__ __forceinline__ kdehip_f2 fma(kdehip_f2 a, kdehip_f2 b, kdehip_f2 c) {
    return __builtin_elementwise_fma(a, b, c);
  }
};

// min(x, hi) that keeps a NaN a NaN (the fast forms detect a NaN centre/cov on the total)
__device__ __forceinline__ float clamp_hi(float x, float hi) { return x > hi ? hi : x; }
__device__ __forceinline__ double clamp_hi(double x, double hi) { return x > hi ? hi : x; }
__device__ __forceinline__ kdehip_f2 clamp_hi(kdehip_f2 x, kdehip_f2 hi) {
  return {x.x > hi.x ? hi.x : x.x, x.y > hi.y ? hi.y : x.y};
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  constexpr bool kBound = (ROW_MASK == 0xF);  // full row mask: bound_ctrl supplies the zeros
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, kBound);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, kBound);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fetch(float v) {
  constexpr bool kBound = (ROW_MASK == 0xF);
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, kBound));
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

// ---- phase stamps (diagnostic build only, -DKDEHIP_STAMPS; never part of the product library) ----
#ifdef KDEHIP_STAMPS
static __device__ unsigned long long g_stamp_acc[16];
#define KSTAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define KSTAMP_ARGS , stamp_acc, stamp_on
#define KSTAMP_ADD(slot, t0, t1) do { if (stamp_on) stamp_acc[slot] += (t1) - (t0); } while (0)
#else
#define KSTAMP(var) do {} while (0)
#define KSTAMP_ARGS
#define KSTAMP_ADD(slot, t0, t1) do {} while (0)
#endif

template <typename P> constexpr bool kIsLdsPointer = false;
template <typename T> constexpr bool kIsLdsPointer<const __attribute__((address_space(3))) T *> = true;

// ---- kernel evaluation of one frontier entry -----------------------------------------------------
// `e` points at (row, field 0, lane) of the entry (LDS or global pointer); field f is at e[f*64].

// A load the compiler may not merge with its neighbours: the two halves of a packed fp32 pair come from two
// rows; merged ds_read2 loads of two FIELDS of one row would land in the wrong register pairing and cost a
// v_mov per value to untangle.
template <typename P>
__device__ __forceinline__ float load_single(P p) {
  using E = std::remove_pointer_t<P>;
  using VP = std::conditional_t<kIsLdsPointer<P>, volatile __attribute__((address_space(3))) const float *, volatile const float *>;
  (void)sizeof(E);
  return *(VP)(p);
}

// UNIFORM: the level has one bandwidth vector; ninv[d] = -1/(2 c_d), scale = rsqrt(prod_d c_d).
// OFF (fp32 retries only): every exponent is raised by `xoff` (base-2 units) and clamped below the overflow
// of exp2 -- see Num<float>::tiny_total.
template <typename T, int D, bool OFF = false>
struct EvalUniform {
  T center[D], ninv[D], scale;
  T xoff;
  const double *tab;
  __device__ __forceinline__ EvalUniform<T, D, true> with_offset(T o) const {
    EvalUniform<T, D, true> e;
#pragma unroll
    for (int d = 0; d < D; ++d) { e.center[d] = center[d]; e.ninv[d] = ninv[d]; }
    e.scale = scale; e.tab = tab; e.xoff = o;
    return e;
  }
  template <typename V> struct RowT { V m[D], w; };  // the fields of one entry (V = T) or of two (packed pair)
  using Row = RowT<T>;
  template <typename P>
  __device__ __forceinline__ Row load(P e) const {
    Row r;
#pragma unroll
    for (int d = 0; d < D; ++d) r.m[d] = e[d * 64];
    r.w = e[D * 64];
    return r;
  }
  // value = front * exp(exponent)
  template <typename V>
  __device__ __forceinline__ V exponent(const RowT<V> &r, V &front) const {
    V acc = OFF ? V(xoff) : V(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const V dl = r.m[d] - center[d];
      acc = Num<V>::fma(dl * dl, V(ninv[d]), acc);
    }
    if constexpr (OFF) acc = clamp_hi(acc, V(T(126)));
    // (no per-entry NaN test: on the fast paths every tile value is finite and positive, so a NaN can
    // only come from the wave-uniform centre/cov and then hits every entry -- handled on the total)
    front = r.w * scale;
    return acc;
  }
  template <typename V>
  __device__ __forceinline__ V eval(const RowT<V> &r) const {
    V front;
    const V x = exponent<V>(r, front);
    return front * Num<V>::exp_fast(x, tab);
  }
  // the same value in two halves: arg() ends by issuing the exp table lookup, fin() uses it
  struct Mid { T front; typename Num<T>::ExpMid e; };
  __device__ __forceinline__ Mid arg(const Row &r) const {
    Mid m;
    const T x = exponent<T>(r, m.front);
    m.e = Num<T>::exp_begin(x, tab);
    return m;
  }
  __device__ __forceinline__ T fin(const Mid &m) const { return m.front * Num<T>::exp_end(m.e); }
  __device__ __forceinline__ T operator()(const Row &r) const { return eval<T>(r); }
  template <typename P>
  __device__ __forceinline__ T operator()(P e) const { return (*this)(load(e)); }
  static constexpr bool kPairs = true;
  template <typename P>
  __device__ __forceinline__ kdehip_f2 pair(P e, int RS) const {  // entries at e and e + RS
    RowT<kdehip_f2> r;
#pragma unroll
    for (int d = 0; d < D; ++d) r.m[d] = kdehip_f2{load_single(e + d * 64), load_single(e + RS + d * 64)};
    r.w = kdehip_f2{load_single(e + D * 64), load_single(e + RS + D * 64)};
    return eval<kdehip_f2>(r);
  }
};

// FAST: per-node bandwidths; one rsqrt instead of D divides and D logs.
template <typename T, int D, bool MASKED, bool OFF = false>
struct EvalFast {
  T center[D], cov[D];
  T xoff;
  const double *tab;
  uint32_t act;  // MASKED: dimensions that take part (:282); an inactive one contributes c = 1, delta = 0
  __device__ __forceinline__ EvalFast<T, D, MASKED, true> with_offset(T o) const {
    EvalFast<T, D, MASKED, true> e;
#pragma unroll
    for (int d = 0; d < D; ++d) { e.center[d] = center[d]; e.cov[d] = cov[d]; }
    e.tab = tab; e.act = act; e.xoff = o;
    return e;
  }
  template <typename V> struct RowT { V m[D], v[D], w; };
  using Row = RowT<T>;
  template <typename P>
  __device__ __forceinline__ Row load(P e) const {
    Row r;
#pragma unroll
    for (int d = 0; d < D; ++d) { r.m[d] = e[d * 64]; r.v[d] = e[(D + d) * 64]; }
    r.w = e[2 * D * 64];
    return r;
  }
  template <typename V>
  __device__ __forceinline__ V exponent(const RowT<V> &row, V &front) const {
    V c[D], d2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      c[d] = row.v[d] + cov[d];
      const V dl = row.m[d] - center[d];
      d2[d] = dl * dl;
      if constexpr (MASKED) {
        const bool on = (act >> d) & 1u;
        c[d] = on ? c[d] : V(1);
        d2[d] = on ? d2[d] : V(0);
      }
    }
    const V w = row.w;
    // pre[d]*suf[d] = prod_{k != d} c[k]; P = prod_k c[k]
    V pre[D], suf[D];
    pre[0] = V(1);
#pragma unroll
    for (int d = 1; d < D; ++d) pre[d] = pre[d - 1] * c[d - 1];
    suf[D - 1] = V(1);
#pragma unroll
    for (int d = D - 2; d >= 0; --d) suf[d] = suf[d + 1] * c[d + 1];
    const V prod = pre[D - 1] * c[D - 1];
    V num = V(0);
#pragma unroll
    for (int d = 0; d < D; ++d) num = Num<V>::fma(d2[d], pre[d] * suf[d], num);
    const V r = Num<V>::rsqrt(prod);
    const V q = num * r * r;  // = sum_d delta_d^2 / c_d
    front = w * r;
    if constexpr (OFF)
      return clamp_hi(Num<V>::fma(q, V(T(-0.5) * T(Num<T>::kExpArg)), V(xoff)), V(T(126)));
    else
      return V(T(-0.5) * T(Num<T>::kExpArg)) * q;
  }
  template <typename V>
  __device__ __forceinline__ V eval(const RowT<V> &row) const {
    V front;
    const V x = exponent<V>(row, front);
    return front * Num<V>::exp_fast(x, tab);
  }
  struct Mid { T front; typename Num<T>::ExpMid e; };
  __device__ __forceinline__ Mid arg(const Row &row) const {
    Mid m;
    const T x = exponent<T>(row, m.front);
    m.e = Num<T>::exp_begin(x, tab);
    return m;
  }
  __device__ __forceinline__ T fin(const Mid &m) const { return m.front * Num<T>::exp_end(m.e); }
  __device__ __forceinline__ T operator()(const Row &row) const { return eval<T>(row); }
  template <typename P>
  __device__ __forceinline__ T operator()(P e) const { return (*this)(load(e)); }
  static constexpr bool kPairs = true;
  template <typename P>
  __device__ __forceinline__ kdehip_f2 pair(P e, int RS) const {  // entries at e and e + RS
    RowT<kdehip_f2> r;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      r.m[d] = kdehip_f2{load_single(e + d * 64), load_single(e + RS + d * 64)};
      r.v[d] = kdehip_f2{load_single(e + (D + d) * 64), load_single(e + RS + (D + d) * 64)};
    }
    r.w = kdehip_f2{load_single(e + 2 * D * 64), load_single(e + RS + 2 * D * 64)};
    return eval<kdehip_f2>(r);
  }
};

// GENERIC: literally the reference's accumulation (:280-303) incl. inactive dimensions.
template <typename T, int D, bool OFF = false>
struct EvalGeneric {
  T center[D], cov[D];
  T xoff;
  uint32_t act;
  __device__ __forceinline__ EvalGeneric<T, D, true> with_offset(T o) const {
    EvalGeneric<T, D, true> e;
#pragma unroll
    for (int d = 0; d < D; ++d) { e.center[d] = center[d]; e.cov[d] = cov[d]; }
    e.act = act; e.xoff = o;
    return e;
  }
  struct Row { T m[D], v[D], w; };
  template <typename P>
  __device__ __forceinline__ Row load(P e) const {
    Row r;
#pragma unroll
    for (int d = 0; d < D; ++d) { r.m[d] = e[d * 64]; r.v[d] = e[(D + d) * 64]; }
    r.w = e[2 * D * 64];
    return r;
  }
  __device__ __forceinline__ T operator()(const Row &row) const {
    T acc = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if ((act >> d) & 1u) {
        const T c = row.v[d] + cov[d];
        const T dl = row.m[d] - center[d];
        const T distr = (dl * dl) / c;
        if (distr == distr) {
          acc += distr;
          acc += Num<T>::log(c);
        }
      }
    }
    T arg = T(-0.5) * acc;
    if constexpr (OFF) {  // xoff binades = xoff * ln 2 in the natural exponent; stay below exp's overflow
      arg = arg + xoff * T(0.6931471805599453);
      arg = clamp_hi(arg, T(87));
    }
    const T p = Num<T>::exp(arg) * row.w;
    return (p != p) ? T(0) : p;
  }
  template <typename P>
  __device__ __forceinline__ T operator()(P e) const { return (*this)(load(e)); }
  struct Mid { T p; };
  __device__ __forceinline__ Mid arg(const Row &row) const { return {(*this)(row)}; }
  __device__ __forceinline__ T fin(const Mid &m) const { return m.p; }
  static constexpr bool kPairs = false;
};

// ---- one categorical label draw over a frontier -------------------------------------------------
// Evaluates every node of the frontier with `ev`, and returns the tile position (row*64 + lane) of
// the entry selected by the uniform draw `u`: the first z with u <= cdf[z], else the last
// (selectLabelOnLevel :330-351 applied to the CDF of makeFasterSampleIndex! :318-325).
// `rows` points at row 0, field 0, lane 0 of the tile (LDS or global pointer type P).
// pass 1 over rows held at `rows` (LDS or global): the lane's private sum over its contiguous entries
template <typename P> constexpr bool kIsLdsPtr = false;
template <typename T> constexpr bool kIsLdsPtr<const __attribute__((address_space(3))) T *> = true;

template <typename T, typename P, typename Eval, bool PREFETCH = true>
__device__ __forceinline__ T lane_sum_rows(P rows, int nrows, int RS, int lane, const Eval &ev) {
  constexpr bool kUsePairs = sizeof(T) == 4 && Eval::kPairs;
  if constexpr (kUsePairs) {
    // fp32: two rows per trip through the packed-math pipe (their 2F loads are in flight together)
    kdehip_f2 S2 = {0.0f, 0.0f};
    P e2 = rows + lane;
    int i2 = 0;
    for (; i2 + 2 <= nrows; i2 += 2, e2 += 2 * RS) S2 += ev.pair(e2, RS);
    T Sp = S2.x + S2.y;
    if (i2 < nrows) Sp += ev(e2);
    return Sp;
  }
  if constexpr (!PREFETCH) {
    T S0 = T(0);
    P e0 = rows + lane;
#pragma unroll 2
    for (int i = 0; i < nrows; ++i, e0 += RS) S0 += ev(e0);
    return S0;
  }
  // software pipelined, two rows per trip with ping-pong register sets (no copies): the fields of the
  // next row are requested before the current row is evaluated, so the LDS (or L2) round trip overlaps
  // ~40-100 fp64 instructions instead of stalling in front of each of them.
  T S = T(0);
  P e = rows + lane;
  typename Eval::Row ra = ev.load(e);
  // LDS tiles: have row 0 landed before the loop, otherwise the compiler's wait-count bookkeeping merges
  // "row 0 pending" into the loop head and waits for every prefetch right after issuing it
  if constexpr (kIsLdsPtr<P>) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
  int i = 0;
  for (; i + 2 <= nrows; i += 2) {
    const typename Eval::Row rb = ev.load(e + RS);  // row i+1
    __builtin_amdgcn_sched_barrier(0);              // keep the requests above the arithmetic
    const typename Eval::Mid ma = ev.arg(ra);       // ... ends by issuing row i's exp table lookup
    __builtin_amdgcn_sched_barrier(0);
    const typename Eval::Mid mb = ev.arg(rb);       // hides the latency of row i's lookup
    e += (i + 2 < nrows) ? 2 * RS : RS;             // row i+2, or row i+1 again (never past the tile)
    ra = ev.load(e);
    __builtin_amdgcn_sched_barrier(0);
    S += ev.fin(ma);
    S += ev.fin(mb);
  }
  if (i < nrows) S += ev(ra);
  return S;
}

// Uniform draws over the frontier that replaced an underflowed conditional (:311-315) are counted in the 8 bytes
// in front of the plan's level table (a layout contract with product.hip: the table's address is live in scalar
// registers anyway, so the counter costs the common path nothing): how tests compare the fp32 path's fallback
// behaviour with fp64's and the oracle's.  `fb` = the level table's address.
__device__ __forceinline__ void count_fallback(const void *fb, int lane) {
  if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(const_cast<void *>(fb)) - 1, 1ull);
}

template <typename T, typename P, typename Eval>
__device__ __forceinline__ int select_from_scan(T incl, T S, P rows, const LevelDesc &ds, int lane, const Eval &ev,
                                                double u, T thr, bool final, const void *fb
#ifdef KDEHIP_STAMPS
                                                , unsigned long long *stamp_acc, bool stamp_on
#endif
);

// Selection from the lane sums S: wavefront scan, winning lane, then pass 2 over the winning lane's
// block read through `rows` (row 0, field 0, lane 0 of the whole tile; LDS or global).
template <typename T, typename P, typename Eval>
__device__ __forceinline__ int select_label(T S, P rows, const LevelDesc &ds, int lane, const Eval &ev, double u,
                                            T thr, bool final, const void *fb
#ifdef KDEHIP_STAMPS
                                            , unsigned long long *stamp_acc, bool stamp_on
#endif
) {
  return select_from_scan<T, P>(wave_inclusive_scan(S), S, rows, ds, lane, ev, u, thr, final, fb KSTAMP_ARGS);
}

// The selection proper, from the inclusive wavefront scan `incl` of the lane sums S.  Also entered
// directly with a scan that was computed ahead of time (conditional tables, single-row frontiers).
// A total below `thr` (or NaN) is the reference's underflow case: with `final` the uniform fallback is taken,
// otherwise -1 is returned and the caller repeats the evaluation with raised exponents (fp32 only).
template <typename T, typename P, typename Eval>
__device__ __forceinline__ int select_from_scan(T incl, T S, P rows, const LevelDesc &ds, int lane, const Eval &ev,
                                                double u, T thr, bool final, const void *fb
#ifdef KDEHIP_STAMPS
                                                , unsigned long long *stamp_acc, bool stamp_on
#endif
) {
  const int n = ds.n, B = ds.B, F = ds.F;
  const int RS = F * 64 + 1;
  KSTAMP(tp1);
  const T total = lane_read(incl, 63);

  if (!(total >= thr)) {  // also taken when every weight is NaN (:302 zeroes them all)
    if (!final) return -1;
    count_fallback(fb, lane);
    // "stick with selection of others": uniform over the frontier (:311-315); with a zero/NaN
    // last weight the reference's CDF is all-NaN and the last entry is taken.
    const int zl = n - 1;
    const T wl = rows[(zl % B) * RS + (F - 1) * 64 + zl / B];
    int z = n - 1;
    if (wl > T(0)) {
      z = static_cast<int>(ceil(u * static_cast<double>(n))) - 1;
      z = z < 0 ? 0 : (z > n - 1 ? n - 1 : z);
    }
    return (z % B) * 64 + z / B;
  }

  const T target = static_cast<T>(u) * total;
  const unsigned long long hit = __ballot(target <= incl);
  const int last_lane = ds.last_lane;
  int lstar = hit ? (__ffsll(hit) - 1) : last_lane;
  if (lstar > last_lane) lstar = last_lane;
  KSTAMP(tp2);
  KSTAMP_ADD(3, tp1, tp2);
  if (B == 1) return lstar;

  // pass 2: narrow inside the winning lane's block until a single node is left
  T base = lane_read(incl - S, lstar);  // exclusive prefix of the block
  int r0 = 0;
  int len = n - lstar * B;
  if (len > B) len = B;
  P col = rows + lstar;
  while (len > 64) {  // only for frontiers beyond 4096 nodes
    const int b2 = (len + 63) / 64;
    T S2 = T(0);
    for (int i = 0; i < b2; ++i) {
      const int r = lane * b2 + i;
      if (r < len) S2 += ev(col + (r0 + r) * RS);
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
  if (lane < len) p2 = ev(col + (r0 + lane) * RS);
  const T inc3 = wave_inclusive_scan(p2);
  const unsigned long long h3 = __ballot((target <= base + inc3) && (lane < len));
  const int istar = h3 ? (__ffsll(h3) - 1) : (len - 1);
  KSTAMP(tp3);
  KSTAMP_ADD(4, tp2, tp3);
  return (r0 + istar) * 64 + lstar;
}

// whole tile readable through one pointer (resident / streamed LDS image, or global memory)
// Frontiers of 2 .. BMAX rows per lane whose tile is in LDS, in builds with registers to spare: the lane keeps
// the BMAX values of its block from the first pass, so the second pass needs no re-evaluation, no LDS gather
// and no second wavefront scan -- every lane forms the running sums of its own block (the same sequential
// sums the first pass accumulates), finds the first row that reaches the target, and the answer is read from
// the winning lane.  (Rows beyond B are padding of weight 0 in the tile; they are not even evaluated.)
template <typename T, typename P, typename Eval, int BMAX>
__device__ __forceinline__ int draw_label_kept(P rows, const LevelDesc &ds, int lane, const Eval &ev, double u,
                                               const void *fb) {
  const int n = ds.n, B = ds.B, F = ds.F;
  const int RS = F * 64 + 1;
  T v[BMAX];
  P e = rows + lane;
  // first pass: the two-rows-per-trip schedule of lane_sum_rows (next row requested early, the two rows
  // interleaved around their exp table lookups), fully unrolled so that the values stay in registers
  typename Eval::Row ra = ev.load(e);
#pragma unroll
  for (int i = 0; i < BMAX; i += 2) {
    if (i + 2 <= B) {  // wave-uniform
      const typename Eval::Row rb = ev.load(e + (i + 1) * RS);
      __builtin_amdgcn_sched_barrier(0);
      const typename Eval::Mid ma = ev.arg(ra);
      if (i + 2 < B) ra = ev.load(e + (i + 2) * RS);
      __builtin_amdgcn_sched_barrier(0);
      const typename Eval::Mid mb = ev.arg(rb);
      __builtin_amdgcn_sched_barrier(0);
      v[i] = ev.fin(ma);
      v[i + 1] = ev.fin(mb);
    } else if (i < B) {
      v[i] = ev(ra);
      v[i + 1] = T(0);
    } else {
      v[i] = T(0);
      v[i + 1] = T(0);
    }
  }
  T S = T(0);
#pragma unroll
  for (int i = 0; i < BMAX; ++i) S += v[i];  // (+0 for the rows beyond B: the first pass's sequential sum)
  const T incl = wave_inclusive_scan(S);
  const T total = lane_read(incl, 63);
  if (!(total >= Num<T>::tiny_total())) {  // uniform fallback (:311-315), as in select_from_scan
    count_fallback(fb, lane);
    const int zl = n - 1;
    const T wl = rows[(zl % B) * RS + (F - 1) * 64 + zl / B];
    int z = n - 1;
    if (wl > T(0)) {
      z = static_cast<int>(ceil(u * static_cast<double>(n))) - 1;
      z = z < 0 ? 0 : (z > n - 1 ? n - 1 : z);
    }
    return (z % B) * 64 + z / B;
  }
  const T target = static_cast<T>(u) * total;
  const unsigned long long hit = __ballot(target <= incl);
  const int last_lane = ds.last_lane;
  int lstar = hit ? (__ffsll(hit) - 1) : last_lane;
  if (lstar > last_lane) lstar = last_lane;
  // second pass, in every lane on its own block: first row r with target <= exclusive prefix + v[0..r]
  // (the running sums never decrease, so the first row that reaches the target = the number of rows below it)
  T run = incl - S;
  int first = 0;
#pragma unroll
  for (int r = 0; r < BMAX; ++r) {
    run += v[r];
    first += (target <= run) ? 0 : 1;
  }
  int len = n - lstar * B;
  if (len > B) len = B;
  int istar = __builtin_amdgcn_readlane(first, lstar);
  if (istar > len - 1) istar = len - 1;  // no row reached the target (rounding), or only padding rows did
  return istar * 64 + lstar;
}

// fp32: the evaluation is repeated with every exponent raised by 110, 220, 330 binades while the sum stays below
// 2^-100 (Num<float>::tiny_total); the last attempt applies the reference's threshold.  Rare (densities far
// apart), so these passes are the plain ones: no prefetch, any readable pointer (LDS image or global memory).
template <typename T, typename P, typename Eval>
__device__ __forceinline__ int draw_label_raised(P rows, const LevelDesc &ds, int lane, const Eval &ev, double u,
                                              const void *fb) {
  int pos = -1;
  for (int k = 1; k <= Num<T>::kOffsetSteps && pos < 0; ++k) {
    const auto evo = ev.with_offset(T(Num<T>::kOffsetStep) * T(k));
    const T S = lane_sum_rows<T, P, std::decay_t<decltype(evo)>, false>(rows, ds.B, ds.F * 64 + 1, lane, evo);
    const bool final = (k == Num<T>::kOffsetSteps);
    pos = select_from_scan<T, P>(wave_inclusive_scan(S), S, rows, ds, lane, evo, u,
                                 final ? Num<T>::final_total() : Num<T>::tiny_total(), final, fb
#ifdef KDEHIP_STAMPS
                                 , nullptr, false
#endif
    );
  }
  return pos;
}

// selection from first-pass lane sums S formed at offset 0, then the raised attempts if the sum underflowed
template <typename T, typename P, typename Eval>
__device__ __forceinline__ int select_or_raise(T S, P rows, const LevelDesc &ds, int lane, const Eval &ev, double u,
                                               const void *fb
#ifdef KDEHIP_STAMPS
                                               , unsigned long long *stamp_acc, bool stamp_on
#endif
) {
  constexpr bool kOneAttempt = (Num<T>::kOffsetSteps == 0);
  const int pos = select_label<T, P>(S, rows, ds, lane, ev, u, Num<T>::tiny_total(), kOneAttempt, fb KSTAMP_ARGS);
  if constexpr (kOneAttempt) return pos;
  else {
    if (__builtin_expect(pos >= 0, 1)) return pos;
    return draw_label_raised<T, P>(rows, ds, lane, ev, u, fb);
  }
}

template <typename T, typename P, bool PREFETCH, bool kKeptRows, typename Eval>
__device__ __forceinline__ int draw_label(P rows, const LevelDesc &ds, int lane, const Eval &ev, double u,
                                          const void *fb
#ifdef KDEHIP_STAMPS
                                          , unsigned long long *stamp_acc, bool stamp_on
#endif
) {
#if !defined(KDEHIP_STAMPS) && !defined(KDEHIP_NO_KEPT)
  // PREFETCH marks the builds with registers to spare (see kPrefetchRows); fp32 has its packed-pair first pass
  if constexpr (PREFETCH && kIsLdsPtr<P> && sizeof(T) == 8 && kKeptRows) {
    if (ds.B > 1 && ds.B <= 4) return draw_label_kept<T, P, Eval, 4>(rows, ds, lane, ev, u, fb);
    if (ds.B > 4 && ds.B <= 8) return draw_label_kept<T, P, Eval, 8>(rows, ds, lane, ev, u, fb);
  }
#endif
  KSTAMP(tp0);
  const T S = lane_sum_rows<T, P, Eval, PREFETCH>(rows, ds.B, ds.F * 64 + 1, lane, ev);
  KSTAMP(tp1);
  KSTAMP_ADD(2, tp0, tp1);
  return select_or_raise<T, P>(S, rows, ds, lane, ev, u, fb KSTAMP_ARGS);
}

// ---- the sampler ----------------------------------------------------------------------------------

// LDS of one workgroup (ONE object, so the compiler keeps direct-to-LDS loads asynchronous):
//   [exp table 256 B][per-wave chain state][tile pool kLdsPoolBytes]
template <typename T, int D, int WAVES>
struct LdsLayout {
  static constexpr int kStatePerWave = (2 * KDEHIP_MAX_DENS * D) * int(sizeof(T)) + KDEHIP_MAX_DENS * int(sizeof(int));
  static constexpr int kStateOff = 256;
  static constexpr int kPoolOff = (kStateOff + WAVES * kStatePerWave + 1023) / 1024 * 1024;
  static constexpr int kBytes = kPoolOff + kLdsPoolBytes;
  static_assert(kBytes <= 160 * 1024, "LDS budget of one CU exceeded");
};

template <typename T> using LdsPtr = const __attribute__((address_space(3))) T *;
// read-only, wave-uniform tables are read through the constant address space so that the compiler
// uses scalar loads (s_load_*, lgkmcnt) and never drains the direct-to-LDS copies in flight (vmcnt)
typedef int kdehip_v16i __attribute__((ext_vector_type(16)));
struct LevelTable {
  const __attribute__((address_space(4))) kdehip_v16i *p;
  __device__ __forceinline__ LevelDesc operator[](int idx) const {
    const kdehip_v16i raw = p[idx];  // one s_load_dwordx16
    LevelDesc d;
    __builtin_memcpy(&d, &raw, sizeof(LevelDesc));
    return d;
  }
};
// conditional-table descriptors (32 B each) through the constant address space, like LevelTable
typedef int kdehip_v8i __attribute__((ext_vector_type(8)));
struct TabTable {
  const __attribute__((address_space(4))) kdehip_v8i *p;
  __device__ __forceinline__ TabDesc operator[](int idx) const {
    const kdehip_v8i raw = p[idx];  // one s_load_dwordx8
    TabDesc d;
    __builtin_memcpy(&d, &raw, sizeof(TabDesc));
    return d;
  }
};

using LdsVoidPtr = __attribute__((address_space(3))) void *;

// Cooperative, asynchronous copy of one tile image (bytes is a multiple of 1 KiB) into the pool: every
// wavefront issues direct-to-LDS loads for its share of 1-KiB pieces (16 bytes per lane).
// The MUBUF form (buffer_load_dwordx4 ... lds) is used rather than global_load_lds: the compiler counts the
// FLAT-encoded form against lgkmcnt as well and, while such a copy is in flight, turns EVERY LDS wait into
// s_waitcnt lgkmcnt(0) -- which would serialise the row prefetches of the first pass behind each other.
// Both forms are tracked by vmcnt in hardware.
template <int WAVES>
__device__ __forceinline__ void stage_tile(const unsigned char *__restrict__ src, unsigned char *dst,
                                           int bytes, int wave, int lane) {
  // raw buffer over exactly this image: base = src, stride 0, num_records = bytes, gfx9 dword 3
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(src), 0, bytes, 0x00020000);
  const int pieces = bytes >> 10;
  for (int c = wave; c < pieces; c += WAVES)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsVoidPtr)(dst + (c << 10)), 16, lane << 4, c << 10, 0, 0);
}

template <typename T, int D, int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gibbs_product_kernel(PlanDev plan, RunArgs a) {
  constexpr bool FAST = (MODE != kModeGeneric);      // product/rsqrt + uniform-bandwidth forms
  constexpr bool MASKED = (MODE == kModeFastMasked);  // ... with inactive dimensions
  constexpr bool kAllDimsOn = (MODE == kModeFast);    // the plan checked it: no mask tests in this build
  // pass 1 prefetches the next row's fields while it evaluates the current one; the 16-wavefront fp64
  // builds have 128 VGPRs and would spill from D = 6 on
  constexpr bool kPrefetchRows = (WAVES <= 12) || sizeof(T) == 4 || D <= 4;
  using Lay = LdsLayout<T, D, WAVES>;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[Lay::kBytes];

  double *sExpTab = reinterpret_cast<double *>(smem);
  if (threadIdx.x < 32) sExpTab[threadIdx.x] = kExp2Tab[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  // readfirstlane makes the wave id (and everything derived from it: sample index, RNG counters,
  // descriptor addresses) provably wave-uniform, so it lives in SGPRs / runs on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  int64_t s = static_cast<int64_t>(blockIdx.x) * WAVES + wave;
  // surplus wavefronts of the last workgroup keep taking part in staging and barriers: they replay
  // the last chain and store nothing
  const bool live = s < a.Np;
  if (!live) s = a.Np - 1;
  const uint64_t gs = static_cast<uint64_t>(a.sample_offset + s);

  const int M = plan.M, L = plan.L;
  const T *__restrict__ data = static_cast<const T *>(plan.data);
  const LevelTable levels{(const __attribute__((address_space(4))) kdehip_v16i *)(plan.levels)};
  unsigned char *state = smem + Lay::kStateOff + wave * Lay::kStatePerWave;
  T *lam = reinterpret_cast<T *>(state);                 // 1/variance of the selected kernels
  T *lmu = lam + KDEHIP_MAX_DENS * D;                    // mean/variance
  int *psel = reinterpret_cast<int *>(lmu + KDEHIP_MAX_DENS * D);  // selected tile position per density
  unsigned char *pool = smem + Lay::kPoolOff;
  const int dl = lane < D ? lane : D - 1;  // this lane's dimension in the "lanes = dimensions" phases

  // variant: 0 default; 1 = read every tile from global memory (no LDS staging); 2 / 8 / 12 / 16 = 4 / 8 / 12 / 16 chains per workgroup.
  // Diagnostic builds (-DKDEHIP_EXPERIMENTS, scripts/) add level cut-offs and ablation flags.
  const int vlev = a.variant % 1000;
#ifdef KDEHIP_EXPERIMENTS
  const int vflags = a.variant / 1000;
#else
  constexpr int vflags = 0;
#endif
#ifdef KDEHIP_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool stamp_on = false;
#endif
  uint32_t any_bits = 0;  // dimensions informed by at least one density
  for (int j = 0; j < M; ++j) any_bits |= levels[j * (L + 1)].mask_bits;

  // selected kernel of density j <- entry `pos` of the tile whose header is at `hdr` (LDS or global)
  // (updateGlbParticlesVariance!, :89-115; masked dimensions carry no information)
  auto set_particle = [&](int j, const LevelDesc &ds, auto hdr, int pos) {
    auto e = hdr + kTileHeader + (pos >> 6) * (ds.F * 64 + 1) + (pos & 63);
    const T mu = e[dl * 64];
    const T var = ds.uniform_bw ? hdr[dl] : e[(D + dl) * 64];
    const bool on = kAllDimsOn || ((ds.mask_bits >> dl) & 1u);
    const T l = on ? (FAST ? fast_rcp(var) : T(1) / var) : T(0);
    if (lane < D) {
      lam[j * D + dl] = l;
      lmu[j * D + dl] = on ? mu * l : T(0);
    }
    if (lane == 0) psel[j] = pos;
  };

  // Gaussian product of the selected kernels without density `skip` for this lane's dimension
  // (gaussianProductMeanCov!, :176-216): cov = 1/sum(lambda), mean = cov * sum(mu*lambda).
  auto product_dim = [&](int skip, uint32_t info_bits, T &mean, T &cov) {
    // The density left out contributes an exact +0 to the reference's sequential sums (:199-213): its slot
    // is zeroed instead of being masked out of every term (every caller with skip >= 0 adopts a new kernel
    // for that density right after the draw, or never set the slot at all).  All LDS reads are issued
    // before the first add (one LDS round trip instead of M).
    if (skip >= 0) {
      if (lane < D) { lam[skip * D + dl] = T(0); lmu[skip * D + dl] = T(0); }
      wave_sync();
    }
    T ls = T(0), ms = T(0);
    int k = 0;
    for (; k + 4 <= M; k += 4) {
      T l4[4], m4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { l4[i] = lam[(k + i) * D + dl]; m4[i] = lmu[(k + i) * D + dl]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { ls += l4[i]; ms += m4[i]; }
    }
    for (; k < M; ++k) {
      ls += lam[k * D + dl];
      ms += lmu[k * D + dl];
    }
    const bool on = kAllDimsOn || ((info_bits >> dl) & 1u);
    cov = on ? (FAST ? fast_rcp(ls) : T(1) / ls) : T(0);
    mean = on ? cov * ms : T(0);
  };

  // One label draw of a density on level descriptor ds against the per-dimension (mean, cov) held by
  // the dimension lanes.  `hdr` = tile header (LDS or global pointer); `run(ev)` evaluates the frontier
  // with the functor it is handed and returns the selected tile position.
  auto draw = [&](const LevelDesc &ds, auto hdr, T mean, T cov, auto &&run) -> int {
    if constexpr (FAST) {
      const uint32_t act = ds.mask_bits & ds.others_bits;
      if (ds.uniform_bw) {
        EvalUniform<T, D> ev;
        ev.tab = sExpTab;
        T c = hdr[dl] + cov;
        T ni = (T(-0.5) * T(Num<T>::kExpArg)) * fast_rcp(c);
        if constexpr (MASKED) {  // an inactive dimension contributes nothing: c = 1, weight 0
          const bool on = (act >> dl) & 1u;
          c = on ? c : T(1);
          ni = on ? ni : T(0);
        }
        T Pr = T(1);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          ev.center[d] = lane_read(mean, d);
          ev.ninv[d] = lane_read(ni, d);
          Pr *= lane_read(c, d);
        }
        ev.scale = Num<T>::rsqrt(Pr);
        return run(ev);
      }
      EvalFast<T, D, MASKED> ev;
      ev.tab = sExpTab;
      ev.act = act;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        ev.center[d] = lane_read(mean, d);
        ev.cov[d] = lane_read(cov, d);
      }
      return run(ev);
    } else {
      EvalGeneric<T, D> ev;
      ev.act = ds.mask_bits & ds.others_bits;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        ev.center[d] = lane_read(mean, d);
        ev.cov[d] = lane_read(cov, d);
      }
      return run(ev);
    }
  };

  // one (pass, density) step on a tile that is readable through one pointer: draw the label and adopt
  // it.  Updating the selected kernel right after the draw is equivalent to the reference's deferred
  // calcIndices! (:383): within the sampleIndices! pass nothing reads the selected kernels.
  auto step = [&](int j, const LevelDesc &ds, auto hdr, T mean, T cov, double u) {
    KSTAMP(ts0);
    auto rows = hdr + kTileHeader;
    using P = decltype(rows);
    const int pos = (vflags & 8) ? 0 : draw(ds, hdr, mean, cov, [&](const auto &ev) {
      return draw_label<T, P, kPrefetchRows, (WAVES <= 8)>(rows, ds, lane, ev, u, plan.levels KSTAMP_ARGS);
    });
    wave_sync();
    KSTAMP(ts1);
    if (!(vflags & 2)) set_particle(j, ds, hdr, pos);
    wave_sync();
    KSTAMP(ts2);
    KSTAMP_ADD(1, ts0, ts1);  // whole draw (setup + passes + scans)
    KSTAMP_ADD(5, ts1, ts2);  // set_particle
  };

  // Chunked step for tiles larger than half the LDS pool: pass 1 streams the rows through the two
  // pool halves `rc` rows at a time (copy of chunk g+1 overlaps the evaluation of chunk g, one barrier
  // per chunk); the short second pass and the new kernel are read from global memory.  `gchunk` is the
  // workgroup-wide running chunk counter that selects the pool half; the chunk after this tile's last
  // one is the first chunk of `dn` (the next step's tile), if there is a next step.
  int gchunk = 0;
  auto chunk_rows = [&](const LevelDesc &ds) -> int { return ds.chunk_rows; };
  auto stage_chunk = [&](const LevelDesc &ds, int r0, int half) {
    const int RS = ds.F * 64 + 1, rc = chunk_rows(ds);
    const int nrows = (ds.B - r0 < rc) ? (ds.B - r0) : rc;
    const int bytes = (nrows * RS * int(sizeof(T)) + 1023) & ~1023;
    stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + ds.hdr_off + kTileHeader + static_cast<int64_t>(r0) * RS),
                      pool + half * (kLdsPoolBytes / 2), bytes, wave, lane);
  };
  auto step_chunked = [&](int j, const LevelDesc &ds, const LevelDesc &dn, bool has_next, T mean, T cov, double u) {
    const T *hdr = data + ds.hdr_off;
    const int pos = draw(ds, hdr, mean, cov, [&](const auto &ev) {
      const int RS = ds.F * 64 + 1, rc = chunk_rows(ds);
      T S = T(0);
      for (int r0 = 0; r0 < ds.B; r0 += rc, ++gchunk) {
        __syncthreads();  // this chunk has landed for every wavefront; the other half is free again
        if (r0 + rc < ds.B) stage_chunk(ds, r0 + rc, (gchunk + 1) & 1);
        else if (has_next) stage_chunk(dn, 0, (gchunk + 1) & 1);
        const int nrows = (ds.B - r0 < rc) ? (ds.B - r0) : rc;
        S += lane_sum_rows<T, LdsPtr<T>, std::decay_t<decltype(ev)>, kPrefetchRows>(
            (LdsPtr<T>)(pool + (gchunk & 1) * (kLdsPoolBytes / 2)), nrows, RS, lane, ev);
      }
      // (a raised repeat of the evaluation reads the tile from global memory: no staging, no barriers)
      return select_or_raise<T, const T *>(S, hdr + kTileHeader, ds, lane, ev, u, plan.levels KSTAMP_ARGS);
    });
    wave_sync();
    set_particle(j, ds, hdr, pos);
    wave_sync();
  };

  // ---- conditional tables ------------------------------------------------------------------------
  // While a level's frontiers have at most 64 nodes, the conditional distribution a Gibbs step draws
  // from depends on the chain only through the OTHER densities' current labels -- finitely many
  // configurations, the same for every chain.  For the levels the plan selected (cfg counts within its
  // memory budget) the inclusive scans of all these conditionals are computed ONCE per plan by this
  // same kernel code (a.table_build: one wavefront per (level, density, configuration)), so a sweep
  // step on such a level is one table row load + the unchanged selection: bit-identical results
  // without the kernel evaluation, scan and state update on the per-step critical path.
  const TabTable tabs{(const __attribute__((address_space(4))) kdehip_v8i *)(plan.tabdesc)};
  T *tables = static_cast<T *>(const_cast<void *>(plan.tables));
  if (a.table_build) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * WAVES + wave;
    if (g >= plan.tab_rows_total) return;
    int tl = 1, tj = 0;
    TabDesc td = tabs[1];
    for (int l = 1; l <= plan.Lt; ++l)
      for (int j = 0; j < M; ++j) {
        const TabDesc c = tabs[j * (L + 1) + l];
        if (g >= c.row_base && g < c.row_base + c.ncfg) { td = c; tl = l; tj = j; }
      }
    const int cfg = static_cast<int>(g - td.row_base);
    // row index -> packed label word (the digit of density tj is the hole between the two parts)
    const uint32_t word = (static_cast<uint32_t>(cfg) & ((1u << td.shift) - 1u)) |
                          ((static_cast<uint32_t>(cfg) >> td.shift) << (td.shift + td.bits));
    for (int k = 0; k < M; ++k) {
      if (k == tj) continue;
      const TabDesc tk = tabs[k * (L + 1) + tl];
      const LevelDesc dk = levels[k * (L + 1) + tl];
      set_particle(k, dk, data + dk.hdr_off, static_cast<int>((word >> tk.shift) & ((1u << tk.bits) - 1u)));
    }
    wave_sync();
    const LevelDesc ds = levels[tj * (L + 1) + tl];
    T mean, cov;
    product_dim(tj, ds.others_bits, mean, cov);
    const T *hdr = data + ds.hdr_off;
    T *row = tables + td.off + static_cast<int64_t>(cfg) * (td.n + 1);
    draw(ds, hdr, mean, cov, [&](const auto &ev) {
      T incl = wave_inclusive_scan(lane_sum_rows<T, const T *, std::decay_t<decltype(ev)>, false>(
          hdr + kTileHeader, 1, ds.F * 64 + 1, lane, ev));
      // fp32: the scan of the first exponent offset at which the sum is large enough (see Num<float>::tiny_total);
      // an underflow in the reference's sense is stored as an all-zero row, which the sweep step takes as the
      // uniform fallback
      if constexpr (Num<T>::kOffsetSteps > 0) {
        bool ok = lane_read(incl, 63) >= Num<T>::tiny_total();
        for (int k = 1; k <= Num<T>::kOffsetSteps && !ok; ++k) {
          const auto evo = ev.with_offset(T(Num<T>::kOffsetStep) * T(k));
          incl = wave_inclusive_scan(lane_sum_rows<T, const T *, std::decay_t<decltype(evo)>, false>(
              hdr + kTileHeader, 1, ds.F * 64 + 1, lane, evo));
          ok = lane_read(incl, 63) >= (k == Num<T>::kOffsetSteps ? Num<T>::final_total() : Num<T>::tiny_total());
        }
        if (!ok) incl = T(0);
      }
      if (lane < td.n) row[lane] = incl;
      if (lane == 63) row[td.n] = incl;  // the total the selection reads from lane 63
      return 0;
    });
    return;
  }
  const int Lt = (vlev == 1 || vlev == 4 || !a.use_tables) ? 0 : plan.Lt;  // variant 4: tables off (A/B, tests)

  // init: frontier = {root}, label = root (levelInit!/initIndices!/calcIndices!, :587-589)
  for (int j = 0; j < M; ++j) {
    const LevelDesc ds = levels[j * (L + 1)];
    set_particle(j, ds, data + ds.hdr_off, 0);
  }
  wave_sync();

  // Uniform draws: select call c of this chain uses uniform c of its stream (the M init calls read
  // nothing).  They are produced 128 at a time across the lanes -- lane ln holds uniforms
  // 128*batch + 2*ln and + 2*ln+1 (one Philox block, or two stream elements) -- and handed out with
  // v_readlane, so the 10-round Philox is off the per-step critical path.
  uint32_t c = static_cast<uint32_t>(M);
  uint32_t ubatch = 0xFFFFFFFFu;
  double u_even = 0.0, u_odd = 0.0;
  auto next_uniform = [&]() -> double {
    if (vflags & 1) { ++c; return 0.37; }
    const uint32_t b = c >> 7;
    if (b != ubatch) {
      ubatch = b;
      if (a.rng_philox) {
        const Philox4 r = philox_block(a.seed, gs, b * 64u + static_cast<uint32_t>(lane), 0u);
        u_even = bits_to_unit(r.v[0], r.v[1]);
        u_odd = bits_to_unit(r.v[2], r.v[3]);
      } else {  // element i of the sample's slice feeds call i+1 (philox.hpp / product.hip)
        const int64_t i0 = s * a.K + static_cast<int64_t>(b) * 128 + 2 * lane - 1;
        u_even = (i0 >= 0 && i0 < a.nU) ? a.randU[i0] : 0.5;
        u_odd = (i0 + 1 < a.nU) ? a.randU[i0 + 1] : 0.5;
      }
    }
    const double pick = (c & 1u) ? u_odd : u_even;
    const double u = lane_read(pick, static_cast<int>((c & 127u) >> 1));
    ++c;
    return u;
  };
  auto normal_for_lane = [&](int q) -> double {  // normal (q, dl) of this sample
    const uint32_t r = static_cast<uint32_t>(q * D + dl);
    if (a.rng_philox) return philox_normal(a.seed, gs, r);
    return a.randN[s * a.R + r];
  };

  const int nsteps = M * (a.Niter + 1);  // per level: M sampleIndices! draws, then Niter sweeps of M
#ifdef KDEHIP_EXPERIMENTS
  const int Lrun = (vlev >= 100 && vlev - 100 < L) ? vlev - 100 : L;
#else
  const int Lrun = L;
#endif
  for (int l = 1; l <= Lrun; ++l) {
    // samplePoint! (:440-463): x = mean + sqrt(cov) * randn, all densities included
    T x;
    {
      T mean, cov;
      product_dim(-1, any_bits, mean, cov);
      x = mean + Num<T>::sqrt(cov) * static_cast<T>(normal_for_lane(l - 1));
    }
    const int mode = vlev == 1 ? int(kStageGlobal) : levels[l].stage_mode;

    if (mode == kStageResident) {
      __syncthreads();  // every wavefront is done reading the previous level's images
      for (int j = 0; j < M; ++j) {
        const LevelDesc ds = levels[j * (L + 1) + l];
        stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + ds.hdr_off), pool + ds.lds_off,
                   ds.stage_bytes, wave, lane);
      }
      __syncthreads();  // (waits for this wavefront's copies, then for everyone's)
    } else if (mode == kStageStream) {
      __syncthreads();
      const LevelDesc ds0 = levels[l];
      stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + ds0.hdr_off), pool, ds0.stage_bytes, wave, lane);
    } else if (mode == kStageChunked) {
      __syncthreads();
      stage_chunk(levels[l], 0, gchunk & 1);
    }

    const bool tabulated = (l <= Lt);
    const int t_general = tabulated ? M : nsteps;  // tabulated levels: only the sampleIndices! pass runs here
    // The step loop exists once per staging mode (a compile-time tag): every copy holds only its own mode's
    // code, which keeps live ranges -- and with them the scalar-register spills of every step -- short.
    auto run_steps = [&](auto mode_tag) {
      constexpr int kMode = decltype(mode_tag)::value;
      int j = 0;
      LevelDesc ds_next = levels[l];  // descriptor of step 0 (streamed modes: each step fetches its successor's early)
      for (int t = 0; t < t_general; ++t) {
        const int jn = (j + 1 == M) ? 0 : j + 1;
        // the streamed modes need the successor's descriptor anyway (they copy its tile during this step);
        // the others load their own at the top of the step: 16 fewer scalar registers live across it
        constexpr bool kNeedsNext = (kMode == kStageStream || kMode == kStageChunked);
        const LevelDesc ds = kNeedsNext ? ds_next : levels[j * (L + 1) + l];
        if constexpr (kNeedsNext) ds_next = levels[jn * (L + 1) + l];
        T mean = x, cov = T(0);      // sampleIndices! (:364-385): against the point just drawn
#ifdef KDEHIP_STAMPS
        stamp_on = (l == (vflags >> 8)) ;  // stamp only the level selected by the experiment
#endif
        KSTAMP(tq0);
        if (t >= M && !(vflags & 4)) product_dim(j, ds.others_bits, mean, cov);  // sampleIndex (:404-429): leave j out
        const double u = next_uniform();
        KSTAMP(tq1);
        KSTAMP_ADD(0, tq0, tq1);
        if constexpr (kMode == kStageGlobal) {
          step(j, ds, data + ds.hdr_off, mean, cov, u);
        } else if constexpr (kMode == kStageResident) {
          step(j, ds, (LdsPtr<T>)(pool + ds.lds_off), mean, cov, u);
        } else if constexpr (kMode == kStageChunked) {
          step_chunked(j, ds, ds_next, t + 1 < nsteps, mean, cov, u);
        } else {
          // tile t has been copied by all wavefronts once everyone passes this barrier; buffer
          // (t+1)&1 was last read in step t-1, which everyone has left -> start the next copy
          KSTAMP(tb0);
          __syncthreads();
          KSTAMP(tb1);
          KSTAMP_ADD(6, tb0, tb1);
          if (t + 1 < nsteps)
            stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + ds_next.hdr_off),
                       pool + ((t + 1) & 1) * (kLdsPoolBytes / 2), ds_next.stage_bytes, wave, lane);
          step(j, ds, (LdsPtr<T>)(pool + (t & 1) * (kLdsPoolBytes / 2)), mean, cov, u);
        }
        j = jn;
      }
    };
    if (mode == kStageGlobal) run_steps(std::integral_constant<int, kStageGlobal>{});
    else if (mode == kStageResident) run_steps(std::integral_constant<int, kStageResident>{});
    else if (mode == kStageChunked) run_steps(std::integral_constant<int, kStageChunked>{});
    else run_steps(std::integral_constant<int, kStageStream>{});
    if (tabulated) {
      // ---- tabulated sweeps: a loop of their own (short live ranges, nothing of the general step in it) ----
      // The labels of all densities are kept packed in one scalar word (density k in bits [shift_k,
      // shift_k + bits_k): frontier sizes are powers of two here); the row of density j is addressed by the
      // word with j's digit squeezed out.  The row holds the inclusive scan the regular path would compute
      // (lanes >= n read the total), so the selection is select_from_scan's for a single-row frontier.
      uint32_t word = 0;
      for (int k = 0; k < M; ++k) word |= static_cast<uint32_t>(psel[k]) << tabs[k * (L + 1) + l].shift;
      word = __builtin_amdgcn_readfirstlane(word);
      int jt = 0;
      for (int t = M; t < nsteps; ++t) {
        const TabDesc td = tabs[jt * (L + 1) + l];
        const uint32_t cfg = (word & ((1u << td.shift) - 1u)) | ((word >> (td.shift + td.bits)) << td.shift);
        const T *row = tables + td.off + static_cast<int64_t>(cfg) * (td.n + 1);
        const T incl = row[lane < td.n ? lane : td.n];
        const double u = next_uniform();
        const int n = td.n;
        const T total = lane_read(incl, n < 64 ? n : 63);  // (a 64-node row: its last scan value IS the total)
        int pos;
        if (!(total >= Num<T>::tiny_total())) {  // uniform fallback (:311-315), rare: fetch the descriptor here
          count_fallback(plan.levels, lane);
          const LevelDesc dk = levels[jt * (L + 1) + l];
          const T wl = ((LdsPtr<T>)(pool + dk.lds_off) + kTileHeader)[(dk.F - 1) * 64 + (n - 1)];
          int z = n - 1;
          if (wl > T(0)) {
            z = static_cast<int>(ceil(u * static_cast<double>(n))) - 1;
            z = z < 0 ? 0 : (z > n - 1 ? n - 1 : z);
          }
          pos = z;
        } else {
          const T target = static_cast<T>(u) * total;
          unsigned long long hit = __ballot(target <= incl);
          if (n < 64) hit &= (1ull << n) - 1ull;
          pos = hit ? (__ffsll(hit) - 1) : (n - 1);
        }
        word = (word & ~(((1u << td.bits) - 1u) << td.shift)) | (static_cast<uint32_t>(pos) << td.shift);
        jt = (jt + 1 == M) ? 0 : jt + 1;
      }
      // the level's sweeps are over: unpack the labels and adopt the selected kernels for what follows
      wave_sync();
      for (int k = 0; k < M; ++k) {
        const TabDesc tk = tabs[k * (L + 1) + l];
        const LevelDesc dk = levels[k * (L + 1) + l];
        const int pk = static_cast<int>((word >> tk.shift) & ((1u << tk.bits) - 1u));
        if (lane == 0) psel[k] = pk;
        set_particle(k, dk, (LdsPtr<T>)(pool + dk.lds_off), pk);
      }
      wave_sync();
    }
    if (a.labels && live && lane == 0) {
      for (int k = 0; k < M; ++k) {
        const LevelDesc ds = levels[k * (L + 1) + l];
        a.labels[(s * M + k) * L + (l - 1)] = plan.perm[ds.perm_off + psel[k]];
      }
    }
  }

#ifdef KDEHIP_STAMPS
  if (blockIdx.x == 3 && wave == 1 && lane == 0)
    for (int k = 0; k < 8; ++k) g_stamp_acc[k] = stamp_acc[k];
#endif
  // final labels (:612-616) and final point (:625)
  if (live && lane == 0) {
    for (int k = 0; k < M; ++k) {
      const LevelDesc ds = levels[k * (L + 1) + L];
      a.indices[s * M + k] = static_cast<int64_t>(plan.perm[ds.perm_off + psel[k]]) + 1;
    }
  }
  {
    T mean, cov;
    product_dim(-1, any_bits, mean, cov);
    T xf = mean;
    if (a.addEntropy) xf = mean + Num<T>::sqrt(cov) * static_cast<T>(normal_for_lane(L));
    if (live && lane < D) a.points[s * D + lane] = static_cast<double>(xf);
  }
}

// ---- launcher --------------------------------------------------------------------------------------

// Workgroup size: 8 chains per workgroup fill all 256 CUs from 2048 chains on; with >= 4096 chains
// 16 chains share each staged tile, which doubles the wavefronts per SIMD (2 -> 4) available to hide
// the per-step dependency chains (the LDS pool admits one workgroup per CU either way).
template <typename T, int D, int MODE, int WAVES>
static int launch_waves(const PlanDev &plan, const RunArgs &args, hipStream_t stream) {
  const int64_t blocks = (args.Np + WAVES - 1) / WAVES;
  hipLaunchKernelGGL((gibbs_product_kernel<T, D, MODE, WAVES>), dim3(static_cast<unsigned>(blocks)),
                     dim3(WAVES * 64), 0, stream, plan, args);
  return 0;
}

template <typename T, int D, int MODE>
static int launch_one(const PlanDev &plan, const RunArgs &args, hipStream_t stream) {
  if (args.Np <= 0) return KDEHIP_OK;
  // Chains per workgroup (= wavefronts per CU, one workgroup per CU at a time): fewer wavefronts per SIMD run
  // each chain faster, more hide each other's latencies.  Pick the width with the smallest estimated time
  // rounds(width) * cost(width); the relative costs of one round are measured ones (config 3: 0.59, 0.79,
  // 1.10, 1.37 ms for 4, 8, 12, 16 chains per workgroup) and differ little between shapes.
  const int v = args.variant % 1000;
  const int64_t cus = device_cu_count();
  int waves = 16;
  if (v == 8 || v == 12 || v == 16) waves = v;
  else if (v == 2) waves = 4;
  else {
    static const int kWidth[4] = {4, 8, 12, 16};
    static const double kCost[4] = {1.0, 1.34, 1.86, 2.31};
    double best = 0.0;
    for (int i = 0; i < 4; ++i) {
      const int64_t wgs = (args.Np + kWidth[i] - 1) / kWidth[i];
      const double t = static_cast<double>((wgs + cus - 1) / cus) * kCost[i];
      if (i == 0 || t < best) { best = t; waves = kWidth[i]; }
    }
  }
  if (waves == 16) launch_waves<T, D, MODE, 16>(plan, args, stream);
  else if (waves == 12) launch_waves<T, D, MODE, 12>(plan, args, stream);
  else if (waves == 8) launch_waves<T, D, MODE, 8>(plan, args, stream);
  else launch_waves<T, D, MODE, 4>(plan, args, stream);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return set_error(KDEHIP_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

// This file is compiled once per dimension count (-DKDEHIP_DIM=1..8, see the Makefile) so the 24 kernel
// variants of each dimension (2 precisions x 3 arithmetic modes x 4 workgroup widths) build in parallel.
#ifndef KDEHIP_DIM
#error "compile gibbs_kernel.hip with -DKDEHIP_DIM=<1..8>"
#endif
#define KDEHIP_CAT2(a, b) a##b
#define KDEHIP_CAT(a, b) KDEHIP_CAT2(a, b)

int KDEHIP_CAT(launch_gibbs_d, KDEHIP_DIM)(int precision, int mode, const PlanDev &plan, const RunArgs &args,
                                           void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int D = KDEHIP_DIM;
  if (precision != 64 && precision != 32) return set_error(KDEHIP_ERR_ARG, "precision must be 64 or 32");
  switch (mode) {
    case kModeGeneric:
      return precision == 64 ? launch_one<double, D, kModeGeneric>(plan, args, st) : launch_one<float, D, kModeGeneric>(plan, args, st);
    case kModeFast:
      return precision == 64 ? launch_one<double, D, kModeFast>(plan, args, st) : launch_one<float, D, kModeFast>(plan, args, st);
    case kModeFastMasked:
      return precision == 64 ? launch_one<double, D, kModeFastMasked>(plan, args, st) : launch_one<float, D, kModeFastMasked>(plan, args, st);
    default: return set_error(KDEHIP_ERR_ARG, "unknown arithmetic mode");
  }
}

}  // namespace kdehip

#if defined(KDEHIP_STAMPS) && KDEHIP_DIM == 6
extern "C" int kdehip_debug_read_stamps(unsigned long long *out) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kdehip::g_stamp_acc), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -5;
}
#endif
