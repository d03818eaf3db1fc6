// gibbs_device.hpp -- device-side building blocks shared by the sampler kernels (gibbs_kernel.hip: any density
// count, every arithmetic mode; gibbs_lean.hip: the register-resident kernel for products of 2..4 densities):
// arithmetic helpers, the three evaluators of one frontier entry, the wavefront scan, the categorical label draw
// over a tile, LDS layout and the direct-to-LDS tile copy.  See gibbs_kernel.hip for the mapping of the
// algorithm onto the machine.
#pragma once
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
#ifdef KDEHIP_EXP256  // (gibbs_lean.hip: 256-entry table, see fastexp.hpp)
  static __device__ __forceinline__ double exp_fast(double x, const double *tab) { return exp256_nonpos(x, tab); }
  using ExpMid = ExpSplit256;  // exp_fast in two halves (table lookup issued / result formed)
  static __device__ __forceinline__ ExpMid exp_begin(double x, const double *tab) { return exp256_nonpos_begin(x, tab); }
  static __device__ __forceinline__ double exp_end(const ExpMid &m) { return exp256_nonpos_end(m); }
#else
  static __device__ __forceinline__ double exp_fast(double x, const double *tab) { return exp_nonpos(x, tab); }
  using ExpMid = ExpSplit;  // exp_fast in two halves (table lookup issued / result formed)
  static __device__ __forceinline__ ExpMid exp_begin(double x, const double *tab) { return exp_nonpos_begin(x, tab); }
  static __device__ __forceinline__ double exp_end(const ExpMid &m) { return exp_nonpos_end(m); }
#endif
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
template <typename T> constexpr bool kIsLdsPointer<const volatile __attribute__((address_space(3))) T *> = true;

// ---- kernel evaluation of one frontier entry -----------------------------------------------------
// `e` points at (row, field 0, lane) of the entry (LDS or global pointer); field f is at e[f * TileAddr<T>::kField].

// The two rows of a pair for one (field, lane), as ONE 8-byte load (fp32 tiles keep them adjacent: TileAddr): the
// register pair the packed first pass works on, from one ds_read_b64 (2 LDS cycles; two ds_read_b32 take 4).
// (a volatile load: the compiler would otherwise merge two of them into a ds_read2st64_b64, which takes 8 LDS-array cycles
// where two ds_read_b64 take 2 + 2 -- config 5: 16.46 -> 15.94 ms, profiles/r04_experiments.md)
template <typename P>
__device__ __forceinline__ kdehip_f2 load_pair(P p) {
  using VP = std::conditional_t<kIsLdsPointer<P>, const volatile __attribute__((address_space(3))) kdehip_f2 *, const kdehip_f2 *>;
  return *(VP)(p);
}

// UNIFORM: the level has one bandwidth vector; scale = rsqrt(prod_d c_d).
//   fp32: ninv[d] = -1/(2 c_d) (times log2 e), exponent = sum_d (m_d - center_d)^2 * ninv_d: three instructions per dimension.
//   fp64 (round 4): ninv[d] = s_d = sqrt(1/(2 c_d)), center[d] = mu_d * s_d, exponent = -sum_d t_d^2 with
//         t_d = fma(m_d, s_d, -mu_d s_d): TWO instructions per dimension.  t_d carries an absolute rounding error of
//         ~1e-16 |m_d| s_d instead of the subtract-first form's 1e-16 |t_d|; the packer gives a frontier this compact tile
//         only while |m_d| s_d <= |m_d| / sqrt(2 bandwidth_d) <= kMaxUniformRatio (pack_layout_shapes), so the exponent
//         of any entry that can matter (|t| < 40) is good to 1e-9 at the very worst and to ~1e-14 for data within a few
//         hundred bandwidths of the origin -- beyond what decides a label except in exact ties.
// OFF (fp32 retries only): every exponent is raised by `xoff` (base-2 units) and clamped below the overflow
// of exp2 -- see Num<float>::tiny_total.
// (fp32 keeps the subtract-first form: measured at config 5, the two-instruction form buys 0.9 % there -- 19.01 ->
// 18.84 ms, profiles/r04_experiments.md -- and would change the labels the fp32 gates were tuned on)
template <typename T> constexpr bool kUniformTwoOp = std::is_same<T, double>::value;
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
  // what the dimension lanes hand to the evaluator (lanes = dimensions: `mean`, `c` = bandwidth + leave-one-out variance
  // of this lane's dimension; on = the dimension takes part): the two per-dimension operands, before their broadcast
  static __device__ __forceinline__ void operands(T mean, T c, bool on, T &cen, T &nin) {
    if constexpr (kUniformTwoOp<T>) {
      const T sl = Num<T>::rsqrt(c * T(2.0 / double(Num<T>::kExpArg)));  // sqrt(kExpArg / (2 c)): fp32 exponents are base 2
      nin = on ? sl : T(0);
      cen = on ? mean * sl : T(0);
    } else {
      const T ni = (T(-0.5) * T(Num<T>::kExpArg)) * fast_rcp(c);
      nin = on ? ni : T(0);
      cen = mean;
    }
  }
  template <typename V> struct RowT { V m[D], w; };  // the fields of one entry (V = T) or of two (packed pair)
  using Row = RowT<T>;
  template <typename P>
  __device__ __forceinline__ Row load(P e) const {
    Row r;
#pragma unroll
    for (int d = 0; d < D; ++d) r.m[d] = e[d * TileAddr<T>::kField];
    r.w = e[D * TileAddr<T>::kField];
    return r;
  }
  // value = front * exp(exponent)
  template <typename V>
  __device__ __forceinline__ V exponent(const RowT<V> &r, V &front) const {
    V acc;
    if constexpr (kUniformTwoOp<T>) {
      acc = V(0);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const V t = Num<V>::fma(r.m[d], V(ninv[d]), -V(center[d]));
        acc = Num<V>::fma(t, t, acc);
      }
      if constexpr (OFF) acc = clamp_hi(V(xoff) - acc, V(T(126)));
      else acc = -acc;
    } else {
      acc = OFF ? V(xoff) : V(0);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const V dl = r.m[d] - center[d];
        acc = Num<V>::fma(dl * dl, V(ninv[d]), acc);
      }
      if constexpr (OFF) acc = clamp_hi(acc, V(T(126)));
    }
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
  __device__ __forceinline__ kdehip_f2 pair(P e) const {  // the entries of rows 2p and 2p+1; e = row 2p's
    RowT<kdehip_f2> r;
#pragma unroll
    for (int d = 0; d < D; ++d) r.m[d] = load_pair(e + d * TileAddr<T>::kField);
    r.w = load_pair(e + D * TileAddr<T>::kField);
    return eval<kdehip_f2>(r);
  }
};

// sum over dimensions [LO, HI) of d2[d] / c[d] as one fraction n / e (EvalFast)
template <typename V, int LO, int HI>
__device__ __forceinline__ void fraction_sum(const V *d2, const V *c, V &n, V &e) {
  if constexpr (HI - LO == 1) {
    n = d2[LO];
    e = c[LO];
  } else {
    constexpr int MID = LO + (HI - LO + 1) / 2;
    V na, ea, nb, eb;
    fraction_sum<V, LO, MID>(d2, c, na, ea);
    fraction_sum<V, MID, HI>(d2, c, nb, eb);
    n = Num<V>::fma(na, eb, nb * ea);
    e = ea * eb;
  }
}

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
    for (int d = 0; d < D; ++d) { r.m[d] = e[d * TileAddr<T>::kField]; r.v[d] = e[(D + d) * TileAddr<T>::kField]; }
    r.w = e[2 * D * TileAddr<T>::kField];
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
#ifndef KDEHIP_X_PREFIXPROD
    // sum_d d2[d] / c[d] as ONE fraction num / prod, by pairwise addition of fractions n_a/e_a + n_b/e_b =
    // (n_a e_b + n_b e_a) / (e_a e_b) over a balanced tree: 3 (D - 1) instructions (round 4; D = 6: 15, where the
    // prefix/suffix products below take 19 -- c3 -1.5 %, c4 -1.3 %, c5 -1.4 %, profiles/r04_experiments.md)
    V num, prod;
    fraction_sum<V, 0, D>(d2, c, num, prod);
#else
    // (rounds 1-3) pre[d]*suf[d] = prod_{k != d} c[k]; P = prod_k c[k]
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
#endif
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
  __device__ __forceinline__ kdehip_f2 pair(P e) const {  // the entries of rows 2p and 2p+1; e = row 2p's
    RowT<kdehip_f2> r;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      r.m[d] = load_pair(e + d * TileAddr<T>::kField);
      r.v[d] = load_pair(e + (D + d) * TileAddr<T>::kField);
    }
    r.w = load_pair(e + 2 * D * TileAddr<T>::kField);
    return eval<kdehip_f2>(r);
  }
};

// The circular (2 pi) member of the enumerated manifolds (include/kdehip.h "manifolds"; no reference counterpart -- the
// reference takes its operators as callbacks, src/MSGibbs01.jl:650-653): wrap to [-pi, pi).  Same expression, same
// constants as oracle/kde_oracle.c circ_wrap (fp64: bit for bit).
template <typename T>
__device__ __forceinline__ T circ_wrap(T t) {
  constexpr double kTwoPi = 6.283185307179586476925286766559, kPi = 3.141592653589793238462643383279;
  return t - T(kTwoPi) * floor((t + T(kPi)) / T(kTwoPi));
}

// GENERIC: literally the reference's accumulation (:280-303) incl. inactive dimensions; `circ` bit d = the difference of
// dimension d is the circular diffop (:290).
template <typename T, int D, bool OFF = false>
struct EvalGeneric {
  T center[D], cov[D];
  T xoff;
  uint32_t act;
  uint32_t circ = 0;
  __device__ __forceinline__ EvalGeneric<T, D, true> with_offset(T o) const {
    EvalGeneric<T, D, true> e;
#pragma unroll
    for (int d = 0; d < D; ++d) { e.center[d] = center[d]; e.cov[d] = cov[d]; }
    e.act = act; e.circ = circ; e.xoff = o;
    return e;
  }
  struct Row { T m[D], v[D], w; };
  template <typename P>
  __device__ __forceinline__ Row load(P e) const {
    Row r;
#pragma unroll
    for (int d = 0; d < D; ++d) { r.m[d] = e[d * TileAddr<T>::kField]; r.v[d] = e[(D + d) * TileAddr<T>::kField]; }
    r.w = e[2 * D * TileAddr<T>::kField];
    return r;
  }
  __device__ __forceinline__ T operator()(const Row &row) const {
    T acc = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if ((act >> d) & 1u) {
        const T c = row.v[d] + cov[d];
        T dl = row.m[d] - center[d];
        if ((circ >> d) & 1u) dl = circ_wrap(dl);
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

// Issue priority of this wavefront on its SIMD (s_setprio): the latency-bound phases of a step (set-up, scan, selection,
// adoption) run at high priority, the rows -- throughput work -- at low, so that a wavefront in a dependent chain is
// served at once while its partner streams rows: oldest-first arbitration alone lets the OLDER wavefront's rows hold up the
// younger one's chain.  Config 3, 2048 chains: 0.6047 -> 0.5840 ms (interleaved A/B, profiles/r04_experiments.md); config 5
// -1 %; config 4 and 16-chain workgroups unchanged.  (-DKDEHIP_X_NO_SETPRIO: A/B builds without it.)
#ifndef KDEHIP_X_NO_SETPRIO
#define KDEHIP_PRIO_ROWS() __builtin_amdgcn_s_setprio(0)
#define KDEHIP_PRIO_CHAIN() __builtin_amdgcn_s_setprio(3)
#else
#define KDEHIP_PRIO_ROWS() do {} while (0)
#define KDEHIP_PRIO_CHAIN() do {} while (0)
#endif

// ---- one categorical label draw over a frontier -------------------------------------------------
// Evaluates every node of the frontier with `ev`, and returns the tile position (row*64 + lane) of
// the entry selected by the uniform draw `u`: the first z with u <= cdf[z], else the last
// (selectLabelOnLevel :330-351 applied to the CDF of makeFasterSampleIndex! :318-325).
// `rows` points at row 0, field 0, lane 0 of the tile (LDS or global pointer type P).
// pass 1 over rows held at `rows` (LDS or global): the lane's private sum over its contiguous entries
template <typename P> constexpr bool kIsLdsPtr = false;
template <typename T> constexpr bool kIsLdsPtr<const __attribute__((address_space(3))) T *> = true;
template <typename T> constexpr bool kIsLdsPtr<const volatile __attribute__((address_space(3))) T *> = true;

// ---- the lane's sum over its rows: ONE association for every kernel, width and staging mode ----------------------
// S = (a0 + a1) + (a2 + a3), where a_k is the sum, in increasing row order, of the lane's rows r = k (mod 4): every
// form of the first pass (two rows per trip, four, kept rows, chunks) adds the very same numbers in the very same order,
// so which label a uniform draw selects never depends on a scheduling choice.  Chunked tiles keep the sums across chunks
// (chunk starts are multiples of 4 rows, so a row's class is its class inside the chunk).
template <typename T>
struct LaneAcc {
  T a[4] = {T(0), T(0), T(0), T(0)};
  __device__ __forceinline__ T total() const { return (a[0] + a[1]) + (a[2] + a[3]); }
};

// every row of `rows` (row 0, field 0, lane 0; LDS or global), `nrows` of them, accumulated into acc
// (RS = TileAddr<T>::stride(F): elements per row in fp64, per PAIR of rows in fp32; row 0 of `rows` is an even row)
template <typename T, typename P, typename Eval, bool PREFETCH = true>
__device__ __forceinline__ void lane_rows_all(P rows, int nrows, int RS, int lane, const Eval &ev, LaneAcc<T> &acc) {
  using TA = TileAddr<T>;
  constexpr bool kUsePairs = sizeof(T) == 4 && Eval::kPairs;
  if constexpr (kUsePairs) {
    // fp32: two rows per trip through the packed-math pipe, each (field, lane) of a row pair one 8-byte load
    kdehip_f2 Sa = {acc.a[0], acc.a[1]}, Sb = {acc.a[2], acc.a[3]};
    P e2 = rows + lane * TA::kLane;
    int i2 = 0;
    for (; i2 + 4 <= nrows; i2 += 4, e2 += 2 * RS) {
      Sa += ev.pair(e2);
      Sb += ev.pair(e2 + RS);
    }
    if (i2 + 2 <= nrows) {
      Sa += ev.pair(e2);
      if (i2 + 2 < nrows) Sb.x += ev(e2 + RS);
    } else if (i2 < nrows) {
      Sa.x += ev(e2);
    }
    acc.a[0] = Sa.x; acc.a[1] = Sa.y; acc.a[2] = Sb.x; acc.a[3] = Sb.y;
    return;
  } else if constexpr (!PREFETCH) {
    P e0 = rows + lane * TA::kLane;
    int i = 0;
    for (; i + 4 <= nrows; i += 4, e0 += TA::rel(4, RS)) {
      acc.a[0] += ev(e0);
      acc.a[1] += ev(e0 + TA::rel(1, RS));
      __builtin_amdgcn_sched_barrier(0);  // two rows in flight, as before: the 128-register builds have no room for four
      acc.a[2] += ev(e0 + TA::rel(2, RS));
      acc.a[3] += ev(e0 + TA::rel(3, RS));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (i < nrows) acc.a[0] += ev(e0);
    if (i + 1 < nrows) acc.a[1] += ev(e0 + TA::rel(1, RS));
    if (i + 2 < nrows) acc.a[2] += ev(e0 + TA::rel(2, RS));
    return;
  } else {
    // software pipelined, two rows per trip with ping-pong register sets (no copies): the fields of the
    // next row are requested before the current row is evaluated, so the LDS (or L2) round trip overlaps
    // ~40-100 fp64 instructions instead of stalling in front of each of them.
    P e = rows + lane * TA::kLane;  // (always an EVEN row's entry: trips advance by two rows)
    typename Eval::Row ra = ev.load(e);
    // LDS tiles: have row 0 landed before the loop, otherwise the compiler's wait-count bookkeeping merges
    // "row 0 pending" into the loop head and waits for every prefetch right after issuing it
    if constexpr (kIsLdsPtr<P>) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
    int i = 0;
    auto trip = [&](T &x, T &y, int ii) {
      const typename Eval::Row rb = ev.load(e + TA::rel(1, RS));  // row ii+1
      __builtin_amdgcn_sched_barrier(0);              // keep the requests above the arithmetic
      const typename Eval::Mid ma = ev.arg(ra);       // ... ends by issuing row ii's exp table lookup
      __builtin_amdgcn_sched_barrier(0);
      const typename Eval::Mid mb = ev.arg(rb);       // hides the latency of row ii's lookup
      e += (ii + 2 < nrows) ? TA::rel(2, RS) : TA::rel(1, RS);  // row ii+2, or row ii+1 again (never past the tile; last use of e)
      ra = ev.load(e);
      __builtin_amdgcn_sched_barrier(0);
      x += ev.fin(ma);
      y += ev.fin(mb);
    };
    for (; i + 4 <= nrows; i += 4) {
      trip(acc.a[0], acc.a[1], i);
      trip(acc.a[2], acc.a[3], i + 2);
    }
    if (i + 2 <= nrows) {
      trip(acc.a[0], acc.a[1], i);
      if (i + 2 < nrows) acc.a[2] += ev(ra);
    } else if (i < nrows) {
      acc.a[0] += ev(ra);
    }
  }
}

// The same sums of a tile in GLOBAL memory with FOUR rows requested ahead (the fp64 repeat of a screened step reads its
// fp64 tile through the L2: with lane_rows_all's one row ahead every row waits out most of a ~2,000-cycle round trip, 60,000
// cycles for 64 rows; with the next trip's four rows in flight behind the current four the loads overlap ~200 fp64
// instructions).  Same values, same association: a[k] takes its rows r = k (mod 4) in increasing order.  Rows beyond the
// tile are re-reads of its last row whose values are never added.  Measured (interleaved A/B, profiles/r06_experiments.md):
// config 4 3.04 -> 2.99 ms, config 3 -0.1 %; EIGHT rows ahead spill (256 VGPRs + scratch) and lose: 3.01 ms.
template <typename T, typename Eval>
__device__ __forceinline__ void lane_rows_all_deep(const T *rows, int nrows, int RS, int lane, const Eval &ev, LaneAcc<T> &acc) {
  using TA = TileAddr<T>;
  using Row = typename Eval::Row;
  const T *e = rows + lane * TA::kLane;
  auto at = [&](int r) { return e + TA::row(r < nrows ? r : nrows - 1, RS); };
  Row a0 = ev.load(at(0)), a1 = ev.load(at(1)), a2 = ev.load(at(2)), a3 = ev.load(at(3));
  int i = 0;
  for (; i + 4 <= nrows; i += 4) {
    const Row b0 = ev.load(at(i + 4)), b1 = ev.load(at(i + 5)), b2 = ev.load(at(i + 6)), b3 = ev.load(at(i + 7));
    __builtin_amdgcn_sched_barrier(0);  // keep the requests above the arithmetic
    const typename Eval::Mid m0 = ev.arg(a0), m1 = ev.arg(a1);
    const typename Eval::Mid m2 = ev.arg(a2), m3 = ev.arg(a3);
    acc.a[0] += ev.fin(m0);
    acc.a[1] += ev.fin(m1);
    acc.a[2] += ev.fin(m2);
    acc.a[3] += ev.fin(m3);
    a0 = b0; a1 = b1; a2 = b2; a3 = b3;
  }
  if (i < nrows) acc.a[0] += ev(a0);
  if (i + 1 < nrows) acc.a[1] += ev(a1);
  if (i + 2 < nrows) acc.a[2] += ev(a2);
}

// the canonical lane sum of a whole tile
template <typename T, typename P, typename Eval, bool PREFETCH = true>
__device__ __forceinline__ T lane_sum_rows(P rows, int nrows, int RS, int lane, const Eval &ev) {
  LaneAcc<T> acc;
  lane_rows_all<T, P, Eval, PREFETCH>(rows, nrows, RS, lane, ev, acc);
  return acc.total();
}

// Uniform draws over the frontier that replaced an underflowed conditional (:311-315) are counted in the 8 bytes
// in front of the plan's level table (a layout contract with product.hip: the table's address is live in scalar
// registers anyway, so the counter costs the common path nothing): how tests compare the fp32 path's fallback
// behaviour with fp64's and the oracle's.  `fb` = the level table's address.
// Wavefronts whose draws must not be counted (the surplus wavefronts of the last workgroup, which replay the last chain)
// pass fb = nullptr.
__device__ __forceinline__ void count_fallback(const void *fb, int lane) {
  if (fb != nullptr && lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(const_cast<void *>(fb)) - 1, 1ull);
}

template <typename T, typename P, typename Eval, typename DS>
__device__ __forceinline__ int select_from_scan(T incl, T S, P rows, const DS &ds, int lane, const Eval &ev,
                                                double u, T thr, bool final, const void *fb
#ifdef KDEHIP_STAMPS
                                                , unsigned long long *stamp_acc, bool stamp_on
#endif
);

// Selection from the lane sums S: wavefront scan, winning lane, then pass 2 over the winning lane's
// block read through `rows` (row 0, field 0, lane 0 of the whole tile; LDS or global).
template <typename T, typename P, typename Eval, typename DS>
__device__ __forceinline__ int select_label(T S, P rows, const DS &ds, int lane, const Eval &ev, double u,
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
template <typename T, typename P, typename Eval, typename DS>
__device__ __forceinline__ int select_from_scan(T incl, T S, P rows, const DS &ds, int lane, const Eval &ev,
                                                double u, T thr, bool final, const void *fb
#ifdef KDEHIP_STAMPS
                                                , unsigned long long *stamp_acc, bool stamp_on
#endif
) {
  using TA = TileAddr<T>;
  const int n = ds.n, B = ds.B, F = ds.F;
  const int RS = TA::stride(F);
  KSTAMP(tp1);
  const T total = lane_read(incl, 63);

  if (!(total >= thr)) {  // also taken when every weight is NaN (:302 zeroes them all)
    if (!final) return -1;
    count_fallback(fb, lane);
    // "stick with selection of others": uniform over the frontier (:311-315); with a zero/NaN
    // last weight the reference's CDF is all-NaN and the last entry is taken.
    const int zl = n - 1;
    const T wl = rows[TA::row(zl % B, RS) + (F - 1) * TA::kField + (zl / B) * TA::kLane];
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
#ifdef KDEHIP_X_NOPASS2  // timing ablation only (wrong labels): what the second pass costs
  return lstar;
#endif

  // pass 2: narrow inside the winning lane's block until a single node is left
  T base = lane_read(incl - S, lstar);  // exclusive prefix of the block
  int r0 = 0;
  int len = n - lstar * B;
  if (len > B) len = B;
  P col = rows + lstar * TA::kLane;
  while (len > 64) {  // only for frontiers beyond 4096 nodes
    const int b2 = (len + 63) / 64;
    T S2 = T(0);
    for (int i = 0; i < b2; ++i) {
      const int r = lane * b2 + i;
      if (r < len) S2 += ev(col + TA::row(r0 + r, RS));
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
  if (lane < len) p2 = ev(col + TileAddr<T>::row(r0 + lane, RS));
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
// (`ra`: the lane's row 0, already requested by the caller -- a step's first row does not depend on the chain's state, so
// gibbs_lean.hip asks for it before the leave-one-out product and the broadcasts)
template <typename T, typename P, typename Eval, int BMAX, typename DS>
__device__ __forceinline__ int draw_label_kept(P rows, const DS &ds, int lane, const Eval &ev, double u,
                                               const void *fb, typename Eval::Row ra) {
  using TA = TileAddr<T>;
  const int n = ds.n, B = ds.B, F = ds.F;
  const int RS = TA::stride(F);
  T v[BMAX];
  P e = rows + lane * TA::kLane;
  // first pass: the two-rows-per-trip schedule of lane_sum_rows (next row requested early, the two rows
  // interleaved around their exp table lookups), fully unrolled so that the values stay in registers
  KDEHIP_PRIO_ROWS();
#pragma unroll
  for (int i = 0; i < BMAX; i += 2) {
    if (i + 2 <= B) {  // wave-uniform
      const typename Eval::Row rb = ev.load(e + TA::rel(i + 1, RS));
      __builtin_amdgcn_sched_barrier(0);
      const typename Eval::Mid ma = ev.arg(ra);
      if (i + 2 < B) ra = ev.load(e + TA::rel(i + 2, RS));
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
  KDEHIP_PRIO_CHAIN();
  // the canonical lane sum (LaneAcc): a_k over the rows = k (mod 4); the rows beyond B add an exact +0
  T ak[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ak[k] = v[k];
#pragma unroll
    for (int i = k + 4; i < BMAX; i += 4) ak[k] += v[i];
  }
  const T S = (ak[0] + ak[1]) + (ak[2] + ak[3]);
  const T incl = wave_inclusive_scan(S);
  const T total = lane_read(incl, 63);
  if (!(total >= Num<T>::tiny_total())) {  // uniform fallback (:311-315), as in select_from_scan
    count_fallback(fb, lane);
    const int zl = n - 1;
    const T wl = rows[TA::row(zl % B, RS) + (F - 1) * TA::kField + (zl / B) * TA::kLane];
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
  // second pass, in every lane on its own block: first row r with target <= exclusive prefix + (v[0] + .. + v[r]).
  // The partial sums are formed in the association order of the wavefront scan that select_from_scan applies to
  // the same values when they are spread over lanes (x[i] += x[i-1]; += x[i-2]; += x[i-4]), so both forms of the
  // second pass -- and with them every workgroup width, rank count and staging mode -- compare the very same
  // numbers: which label a uniform draw selects never depends on a scheduling choice.
  const T base = incl - S;
  T ps[BMAX];
#pragma unroll
  for (int r = 0; r < BMAX; ++r) ps[r] = v[r];
#pragma unroll
  for (int sh = 1; sh < BMAX; sh *= 2) {
#pragma unroll
    for (int r = BMAX - 1; r >= sh; --r) ps[r] = ps[r] + ps[r - sh];
  }
  int first = BMAX;
#pragma unroll
  for (int r = BMAX - 1; r >= 0; --r) first = (target <= base + ps[r]) ? r : first;
  int len = n - lstar * B;
  if (len > B) len = B;
  int istar = __builtin_amdgcn_readlane(first, lstar);
  if (istar > len - 1) istar = len - 1;  // no row reached the target (rounding), or only padding rows did
  return istar * 64 + lstar;
}
template <typename T, typename P, typename Eval, int BMAX, typename DS>
__device__ __forceinline__ int draw_label_kept(P rows, const DS &ds, int lane, const Eval &ev, double u, const void *fb) {
  return draw_label_kept<T, P, Eval, BMAX>(rows, ds, lane, ev, u, fb, ev.load(rows + lane * TileAddr<T>::kLane));
}

// fp32: the evaluation is repeated with every exponent raised by 110, 220, 330 binades while the sum stays below
// 2^-100 (Num<float>::tiny_total); the last attempt applies the reference's threshold.  Rare (densities far
// apart), so these passes are the plain ones: no prefetch, any readable pointer (LDS image or global memory).
template <typename T, typename P, typename Eval, typename DS>
__device__ __forceinline__ int draw_label_raised(P rows, const DS &ds, int lane, const Eval &ev, double u,
                                              const void *fb) {
  int pos = -1;
  for (int k = 1; k <= Num<T>::kOffsetSteps && pos < 0; ++k) {
    const auto evo = ev.with_offset(T(Num<T>::kOffsetStep) * T(k));
    const T S = lane_sum_rows<T, P, std::decay_t<decltype(evo)>, false>(rows, ds.B, TileAddr<T>::stride(ds.F), lane, evo);
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
template <typename T, typename P, typename Eval, typename DS>
__device__ __forceinline__ int select_or_raise(T S, P rows, const DS &ds, int lane, const Eval &ev, double u,
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

// ---- chunked tiles (rows streamed through LDS, the second pass reads global memory) ------------------------------
// The first pass notes the lane's running sum at segment boundaries (a segment = a whole number of chunks, at most 64
// rows): the second pass then finds the winning lane's segment from those few numbers and evaluates only its <= 64
// entries -- ONE round of global-memory loads instead of the two or three of the general narrowing (which costs a
// quarter of config 5's kernel time: every round is a dependent L2 round trip per step).
constexpr int kMaxSeg = 8;
template <typename T>
struct SegSums {
  T v[kMaxSeg] = {};  // v[k] = the lane's sum over rows [0, (k+1) * seg_rows)
  int k = 0;
  __device__ __forceinline__ void note(T S) {  // (k is wave-uniform; compile-time indices keep v in registers)
#pragma unroll
    for (int q = 0; q < kMaxSeg; ++q) v[q] = (q == k) ? S : v[q];
    ++k;
  }
};
// chunks per segment / segments per lane block of a chunked tile: from the descriptor (LevelDesc.seg, worked out by the
// packer: no integer division on the step path); 0 chunks per segment = the segment form does not apply
static_assert(kMaxSeg == kMaxSegDesc, "the packer's segment limit is the kernel's");
__device__ __forceinline__ int seg_chunks(int seg) { return seg & 0xFFFF; }
__device__ __forceinline__ int seg_count(int seg) { return seg >> 16; }

template <typename T, typename P, typename Eval, typename DS>
__device__ __forceinline__ int select_or_raise_seg(T S, const SegSums<T> &seg, int seg_rows, P rows, const DS &ds,
                                                   int lane, const Eval &ev, double u, const void *fb
#ifdef KDEHIP_STAMPS
                                                   , unsigned long long *stamp_acc, bool stamp_on
#endif
) {
  const int n = ds.n, B = ds.B;
  const int RS = TileAddr<T>::stride(ds.F);
  const T incl = wave_inclusive_scan(S);
  const T total = lane_read(incl, 63);
  // underflow (uniform fallback, fp32 raised repeats): the general path
  if (!(total >= Num<T>::tiny_total())) return select_or_raise<T, P>(S, rows, ds, lane, ev, u, fb KSTAMP_ARGS);
  const T target = static_cast<T>(u) * total;
  const unsigned long long hit = __ballot(target <= incl);
  const int last_lane = ds.last_lane;
  int lstar = hit ? (__ffsll(hit) - 1) : last_lane;
  if (lstar > last_lane) lstar = last_lane;
  const T base = lane_read(incl - S, lstar);  // exclusive prefix of the winning lane's block
  int lenl = n - lstar * B;                   // entries of that block
  if (lenl > B) lenl = B;
  const int nseg = seg_count(ds.seg);
  int sidx = nseg - 1;
  T before = T(0), run = T(0);
  bool found = false;
#pragma unroll
  for (int k = 0; k < kMaxSeg - 1; ++k) {
    if (k < nseg - 1) {  // wave-uniform
      const T pk = lane_read(seg.v[k], lstar);
      if (!found && target <= base + pk) { found = true; sidx = k; before = run; }
      run = pk;
    }
  }
  if (!found) before = run;
  int r0 = sidx * seg_rows;
  if (r0 >= lenl) return (lenl - 1) * 64 + lstar;  // (rounding pushed the target beyond the block's last entry)
  int len = lenl - r0;
  if (len > seg_rows) len = seg_rows;
  P col = rows + lstar * TileAddr<T>::kLane;
  T p2 = T(0);
  if (lane < len) p2 = ev(col + TileAddr<T>::row(r0 + lane, RS));
  const T inc3 = wave_inclusive_scan(p2);
  const unsigned long long h3 = __ballot((target <= (base + before) + inc3) && (lane < len));
  const int istar = h3 ? (__ffsll(h3) - 1) : (len - 1);
  return (r0 + istar) * 64 + lstar;
}

template <typename T, typename P, bool PREFETCH, bool kKeptRows, typename Eval, typename DS>
__device__ __forceinline__ int draw_label(P rows, const DS &ds, int lane, const Eval &ev, double u,
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
  KDEHIP_PRIO_ROWS();
  const T S = lane_sum_rows<T, P, Eval, PREFETCH>(rows, ds.B, TileAddr<T>::stride(ds.F), lane, ev);
  KDEHIP_PRIO_CHAIN();
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

// The geometry of one frontier tile, as the draw functions above need it (any type with these members works:
// LevelDesc, or this compact form whose fields are independent scalar registers).
struct TileGeom {
  int n, B, F, last_lane;
};

template <typename T> using LdsPtr = const __attribute__((address_space(3))) T *;
// The same with every field read a load of its own: through a plain pointer the compiler merges two 8-byte field reads into
// one ds_read2st64_b64 (8 LDS-array cycles for 1 KiB; two ds_read_b64: 2 + 2).  With four wavefronts per SIMD reading rows
// (16-chain workgroups) the LDS array is busy enough for that to show -- config 3 with 16,384 chains 3.91 -> 3.81 ms --
// with two per SIMD it is not (config 3: 0.5750 / 0.5749 ms, config 4: 4.37 -> 4.45 ms): gibbs_lean.hip picks by width.
template <typename T> using LdsPtrSplit = const volatile __attribute__((address_space(3))) T *;
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
// step descriptors (StepDesc, 32 B): the RAW eight dwords, so that a step can ask for its successor's descriptor and
// leave it in flight (unpacked a step later)
struct StepTable {
  const __attribute__((address_space(4))) kdehip_v8i *p;
  __device__ __forceinline__ kdehip_v8i raw(int idx) const { return p[idx]; }  // one s_load_dwordx8
};
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

// The workgroup barrier of the staging protocols.  Direct-to-LDS copies are ordered for ANOTHER wavefront's ds_read
// only by the ISSUING wavefront's vmcnt wait followed by a barrier that the reader passes (MI355X_MICROARCH.md,
// co-residence item 7; cdna_hip_programming.md "read a staged buffer one phase after the wait that retires it").
// __syncthreads() alone does not give that: for the MUBUF `buffer_load ... lds` form the compiler's fence emitted
// `s_waitcnt lgkmcnt(0)` only at the loop-carried barriers of the streamed and chunked levels, so a copy that took
// longer than the evaluation of the previous tile or chunk (freshly uploaded plans: cold TLB / HBM) was read before it
// had landed -- one wrong workgroup in a few thousand one-shot calls (found by scripts/soak_multi.py).
__device__ __forceinline__ void staging_barrier() {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wavefront's own copies have landed
  __syncthreads();
}

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
  // (the piece offset goes through readfirstlane: strength reduction otherwise keeps it in a vector register, and the
  // scalar-offset operand of the load is then fed by a waterfall loop -- 18 instructions per 1-KiB piece instead of 6)
#ifdef KDEHIP_X_OLDSTAGE  // (A/B only: the round-3 form)
  for (int c = wave; c < pieces; c += WAVES)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsVoidPtr)(dst + (c << 10)), 16, lane << 4, c << 10, 0, 0);
#else
  for (int c = wave; c < pieces; c += WAVES) {
    const int off = __builtin_amdgcn_readfirstlane(c << 10);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsVoidPtr)(dst + off), 16, lane << 4, off, 0, 0);
  }
#endif
}

// What a workgroup works on.  A plain launch: the kernel arguments, block = blockIdx.x.  A BATCHED launch
// (kdehip_prod_philox_batch): workgroup b belongs to product batch_map[b]; its plan and run parameters come from that
// product's BatchEntry -- three 64-byte scalar loads through the constant address space -- and its chains are numbered
// from the product's first workgroup.
template <bool BATCH>
struct LaunchView {
  const PlanDev &plan;
  const RunArgs &a;
  unsigned block;
  __device__ __forceinline__ LaunchView(const PlanDev &p, const RunArgs &a_) : plan(p), a(a_), block(blockIdx.x) {}
};
// The run parameters of a batched workgroup: RunArgs' member names, scalars only (a copy of RunArgs itself -- with its
// run-time-indexed peer arrays -- would have to live in scratch memory, and everything read from it would count as
// divergent); no caller streams, no peers in a batched launch.
struct BatchArgs {
  int64_t Np;
  int32_t Niter, addEntropy, variant, use_tables;
  static constexpr int32_t use_screen = 0;  // (batched launches run without the fp32 screen)
  uint64_t seed;
  int64_t sample_offset;
  double *points;
  int64_t *indices;
  int32_t *labels;
  static constexpr int32_t rng_philox = 1, npeers = 0, table_build = 0;
  static constexpr const double *randU = nullptr, *randN = nullptr;
  static constexpr int64_t K = 0, R = 0, nU = 0, nN = 0;
  double *peer_points[1];
  int64_t *peer_indices[1];
};
template <>
struct LaunchView<true> {
  PlanDev plan;
  BatchArgs a;
  unsigned block;
  __device__ __forceinline__ LaunchView(const PlanDev &, const RunArgs &a_) {
    const int e = ((const __attribute__((address_space(4))) int *)(a_.batch_map))[blockIdx.x];
    const auto *src = (const __attribute__((address_space(4))) kdehip_v16i *)(a_.batch + e);
    const kdehip_v16i r0 = src[0], r1 = src[1], r2 = src[2];  // three s_load_dwordx16
    BatchPlanHead h;
    BatchRun be;
    BatchFlags fl;
    __builtin_memcpy(&h, &r0, sizeof(h));
    __builtin_memcpy(&be, &r1, sizeof(be));
    __builtin_memcpy(&fl, &r2, sizeof(fl));
    plan.data = h.data; plan.perm = h.perm; plan.levels = h.levels; plan.tables = h.tables; plan.tabdesc = h.tabdesc;
    plan.tab_rows_total = h.tab_rows_total;
    plan.M = h.M; plan.L = h.L; plan.D = h.D; plan.Lt = h.Lt;
    plan.screened = 0;
    a.Np = be.Np; a.Niter = fl.Niter; a.addEntropy = fl.addEntropy; a.use_tables = fl.use_tables;
    a.variant = a_.variant;
    a.seed = be.seed; a.sample_offset = be.sample_offset;
    a.points = be.points; a.indices = be.indices; a.labels = be.labels;
    a.peer_points[0] = nullptr; a.peer_indices[0] = nullptr;
    block = blockIdx.x - static_cast<unsigned>(fl.first_block);
  }
};


}  // namespace kdehip
