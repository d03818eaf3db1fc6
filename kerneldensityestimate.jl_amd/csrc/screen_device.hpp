// screen_device.hpp -- fp32 SCREENING of a label draw with fp64 certification (gibbs_lean.hip, fp64 instantiations).
//
// makeFasterSampleIndex! + selectLabelOnLevel (reference src/MSGibbs01.jl:250-351) draw entry z of a frontier when the
// uniform u satisfies b[z-1] < u * b[n-1] <= b[z], b = cumulative sums of the kernel values.  Which z that is depends on
// the values only through those comparisons, so the values may be formed in ANY arithmetic as long as the comparisons are
// known to come out as in fp64.  On the deep levels the frontier is therefore evaluated in packed fp32 (two rows per
// v_pk_* instruction, hardware exp2 / rsq: ~2.5x fewer vector instructions per row than fp64) from an fp32 copy of the
// tile that is RESIDENT in LDS (half the bytes: what is streamed in fp64 fits at once in fp32, so the per-step workgroup
// barriers go too), together with a rigorous bound on the error of every cumulative sum; the fp32 decision is accepted
// when the target is farther than that bound from the boundaries on both sides of it, and the step is repeated in fp64
// (the unchanged path, tile read from global memory) otherwise.  Labels are then those of the fp64 path by construction;
// the adopted kernel (mean, variance) is always read from the fp64 tile, so points are too.
//
// Error model (DESIGN.md "fp32 screening" has the derivation).  Entry i has the value v_i = front_i * 2^(x_i),
// x_i = -c0 sum_d (m_id - centre_d)^2 / c_id <= 0, c0 = log2(e) / 2.  In fp32, on centred data (m' = m - mu0 and
// centre' = centre - mu0 are formed in fp64 and rounded once, u = 2^-24):
//   * the difference (m' - centre') carries, from the two roundings, |delta_d| <= u (|m'_d| + |centre'_d|), which is at most
//     u (mmax_d + |centre'_d|) and also at most u (|m'_d - centre'_d| + 2 |centre'_d|); the part proportional to the
//     difference itself is a relative perturbation (counted in kx below), the rest, u g_d with g_d = min(mmax_d +
//     |centre'_d|, 2 |centre'_d|), moves the exponent by at most 2 c0 sum_d |t_d| u g_d / sigma_d (t_d = the scaled
//     difference, sigma_d^2 = c_id >= cmin_d) <= 2 sqrt(c0 |x_i|) |a'| by Cauchy-Schwarz, a'_d = u g_d / sqrt(cmin_d), and
//     2 sqrt(y) <= 1 + y gives the linear form  na (1 + |x_i|),  na = sqrt(c0) |a'|;
//   * every other operation is a relative perturbation of x_i by at most kx u (ScreenConst below, each with 2 u of slack).
//     Shared bandwidths: the difference 2 u (its rounding, the relative part of the operands'), its square 5 u; the
//     coefficient -c0 / c_d 5 u (c_d rounded, v_rcp_f32 within 1 ulp = 2 u, the constant, the product); one rounding per
//     fma of the D same-signed terms: kx = 10 + D.  Per-node bandwidths: the square 5 u, c_d = variance + leave-one-out
//     variance 2 u, two roundings per level of the fraction tree (depth = ceil(log2 D)), D - 1 for the product of the
//     variances, 4 u for the square of v_rsq_f32 (1 ulp), the two products with it and the one with -c0 (a rounded
//     constant) 4 u: kx = 14 + 2 depth + D;
//   * front_i, exp2 and the final product add vc u relative to v_i: the weight u, the scale rsq(prod c_d) (D + 1.5) u with
//     shared and (1.5 D + 1.5) u with per-node bandwidths, their product u, v_exp_f32 (1 ulp) 2 u, the last product u:
//     vc = D + 6.5 or 1.5 D + 6.5; the fp32 sums (two half sums of ceil(B / 2) rows, their sum, six scan steps; the second
//     pass's scan is shorter) are within (ceil(B / 2) + 7) u of the exact sums of the computed values.
// So |v~_i - v_i| <= v_i (A + Bc |x_i|) with the wave-uniform A = ln2 na + (ceil(B / 2) + 9 + vc) u, Bc = ln2 (na + kx u)
// (each with a 1 % allowance for the second-order terms), every cumulative sum is within E = sum_i v~_i (A + Bc |x~_i|)
// of its fp64 value, and so is the target u * total.  (Rounds 5a-5o ran with the cruder kx = 32, vc = 24, B + 16 for the
// sums: 0.75 % of config 3's and 1.9 % of config 4's draws repeated; profiles/r05_experiments.md section 11.)  A decision is certified when the boundaries on both sides of the target are
// more than 2.1 E away (2 E would do).  Terms that fp32 flushes to zero or holds as denormals are below 2^-126 times a
// front that the range checks bound by 2^28 each: with total >= 2^-40 required, their sum is below 2^-40 of the margin.
// Range checks (else the step runs in fp64): tile values (screen_build_kernel: |m'| <= 2^16, variances in [2^-7, 2^8],
// weights in [0, 2]) and per step |centre'_d| <= 2^16, leave-one-out variance <= 2^8, and na <= 2^-11 (kScreenMaxNa: the
// linear form na (1 + |x|) stands for exp(d) - 1 with d = ln2 na (1 + |x|); the 1 % allowance and the margin's 5 % cover it
// up to d ~ 0.11, and |x| < 127 wherever the value is not flushed -- clusters thousands of bandwidths apart exceed it).
//
// THE HARDWARE PREMISE, MEASURED (csrc/selftest.hip, tests/test_gpu_ulp.py: every fp32 input swept on the device against
// fp64; MI355X, round 6): largest relative error of v_rcp_f32 1.535 u (all inputs with a normal reciprocal, both signs),
// v_rsq_f32 1.577 u (all positive normals), v_exp_f32 1.423 u on [-126, -0] and 1.395 u on [0, 127]; for x < -126 the
// result is flushed to zero (off by < 2^-126).  The constants above charge each of them 2 u.
#pragma once
#include "gibbs_device.hpp"

namespace kdehip {

// phase stamps of a screened step (diagnostic builds only: -DKDEHIP_SCREEN_STAMPS, scripts/screen_stamps.py)
#ifdef KDEHIP_SCREEN_STAMPS
static __device__ unsigned long long g_screen_stamps[32];
#define SSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define SSTAMP_ADD(slot, t0, t1) do { if (sstamp_on) sstamp[slot] += (t1) - (t0); } while (0)
#define SSTAMP_ARGS , sstamp, sstamp_on
#define SSTAMP_PARAMS , unsigned long long *sstamp, bool sstamp_on
#else
#define SSTAMP_ARGS
#define SSTAMP_PARAMS
#define SSTAMP(var) do {} while (0)
#define SSTAMP_ADD(slot, t0, t1) do {} while (0)
#endif

constexpr float kScreenU = 5.9604645e-8f;       // 2^-24
constexpr float kScreenC0 = 0.72134752f;        // log2(e) / 2
constexpr float kScreenSqrtC0 = 0.84932180f;    // sqrt(c0)
constexpr float kScreenLn2 = 0.69314718f;
// the model's constants in units of u for D dimensions (header comment; 2 u of slack each)
template <int D>
struct ScreenConst {
  static constexpr int depth = D <= 1 ? 0 : D <= 2 ? 1 : D <= 4 ? 2 : 3;  // levels of the fraction tree (fraction_sum)
  static constexpr float kx_uni = 12 + D, kx_node = 16 + 2 * depth + D;
  static constexpr float vc_uni = D + 9, vc_node = (3 * D + 18) / 2;
};
#ifdef KDEHIP_X_NO_REPEAT  // (timing experiment only: every fp32 decision accepted -- NOT the fp64 labels)
constexpr double kScreenMargin = 0.0;
#else
constexpr double kScreenMargin = 2.1;  // boundaries farther than this many error sums from the target are decided (2 suffices)
#endif

template <int D, bool UNI>
struct ScreenEval {
  float cen[D];  // centre'_d
  float b[D];    // UNI: -c0 / c_d (the level's shared variance + leave-one-out variance); else: the leave-one-out variance
  float scale;   // UNI: rsqrt(prod_d c_d)
  float A, Bc;
  static constexpr bool kUni = UNI;
  using TA = TileAddr<float>;
  template <typename V, typename LD>
  __device__ __forceinline__ V value(LD &&ld, V &x) const {
    V front;
    if constexpr (UNI) {
      V acc = V(0.0f);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const V dl = ld(d) - cen[d];
        acc = Num<V>::fma(dl * dl, V(b[d]), acc);
      }
      front = ld(D) * scale;
      x = acc;
    } else {
      V c[D], d2[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        c[d] = ld(D + d) + b[d];
        const V dl = ld(d) - cen[d];
        d2[d] = dl * dl;
      }
      V num, prod;
      fraction_sum<V, 0, D>(d2, c, num, prod);
      const V r = Num<V>::rsqrt(prod);
      const V q = num * r * r;
      front = ld(2 * D) * r;
      x = q * V(-kScreenC0);
    }
    return front * Num<V>::exp_fast(x, nullptr);
  }
  // one PAIR of rows; e = (row 2p, field 0, this lane).  The fields are requested first (one ds_read_b64 per field and
  // lane: load_pair), the arithmetic follows a trip later: value sums S and error-bound sums E (both halves)
  static constexpr int kFields = UNI ? D + 1 : 2 * D + 1;
  struct Pair { kdehip_f2 f[kFields]; };
  static __device__ __forceinline__ Pair load(LdsPtr<float> e) {
    Pair r;
#pragma unroll
    for (int f = 0; f < kFields; ++f) r.f[f] = load_pair(e + f * TA::kField);
    return r;
  }
  __device__ __forceinline__ kdehip_f2 pair(const Pair &r, kdehip_f2 &S, kdehip_f2 &E) const {
    kdehip_f2 x;
    const kdehip_f2 v = value<kdehip_f2>([&](int f) { return r.f[f]; }, x);
    S += v;
    const kdehip_f2 g = Num<kdehip_f2>::fma(-x, kdehip_f2(Bc), kdehip_f2(A));
    E = Num<kdehip_f2>::fma(v, g, E);
    return v;
  }
  // one entry (second pass); e = (its row, field 0, its lane)
  template <typename P>
  __device__ __forceinline__ float one(P e) const {
    float x;
    return value<float>([&](int f) { return e[f * TA::kField]; }, x);
  }
};

// First pass over `npairs` row pairs from `e` (= pair 0, field 0, this lane; LDS): value sums S and error-bound sums E.
// Two pairs per trip, the next pair's fields requested before the current pair is evaluated (ping-pong registers).
// The sums of a lane: the even rows' and the odd rows' halves of one packed accumulator each (values, error bounds): a
// chain of ceil(B / 2) additions each over all chunks, which is what the bound's summation term counts.  (Two accumulators
// each -- even and odd PAIRS, ceil(B / 4) + 1 additions, the term 15 u smaller at 64 rows -- measured: config 4 -0.7 %,
// config 3 +0.3 %; not kept.)
struct ScreenSums {
  kdehip_f2 S = {0.0f, 0.0f}, E = {0.0f, 0.0f};
  __device__ __forceinline__ float values() const { return S.x + S.y; }
  __device__ __forceinline__ float errors() const { return E.x + E.y; }
};
template <int D, bool UNI>
__device__ __forceinline__ void screen_rows(LdsPtr<float> e, int npairs, int RS, const ScreenEval<D, UNI> &ev, ScreenSums &q) {
  using Ev = ScreenEval<D, UNI>;
  typename Ev::Pair ra = Ev::load(e);
  int p = 0;
  for (; p + 2 <= npairs; p += 2) {
    const typename Ev::Pair rb = Ev::load(e + RS);  // pair p + 1
    __builtin_amdgcn_sched_barrier(0);
    ev.pair(ra, q.S, q.E);
    e += (p + 2 < npairs) ? 2 * RS : RS;  // pair p + 2, or pair p + 1 again (never past the tile)
    ra = Ev::load(e);
    __builtin_amdgcn_sched_barrier(0);
    ev.pair(rb, q.S, q.E);
  }
  if (p < npairs) ev.pair(ra, q.S, q.E);
}

// The decision from the lanes' sums: the tile position of the entry u selects, or -1 when the fp32 decision cannot be
// certified (the caller repeats the step in fp64).  `rows` = the screen tile's row 0, field 0, lane 0 -- in LDS (TWO = false:
// at most 64 rows per lane, one second-pass round) or in global memory (TWO: at most 128 rows, two rounds).
template <int D, bool UNI, bool TWO, typename P>
__device__ __forceinline__ int screen_decide(P rows, int n, int B, int RS, int lane, const ScreenEval<D, UNI> &ev, double u,
                                             float s1, float e1 SSTAMP_PARAMS) {
  using TA = TileAddr<float>;
  SSTAMP(tq1);
  const float incl = wave_inclusive_scan(s1);
  const float einc = wave_inclusive_scan(e1);
  const float total = lane_read(incl, 63), etot = lane_read(einc, 63);
  if (!(total >= 0x1p-40f && total < 0x1p100f)) return -1;  // (also a NaN)
  const double td = u * static_cast<double>(total), md = kScreenMargin * static_cast<double>(etot);
  // first lane that certainly reaches the target = first lane that possibly does
  const double id = static_cast<double>(incl);
  const unsigned long long hitA = __ballot(td + md <= id), hitB = __ballot(td - md <= id);
  if (hitA == 0ull) return -1;
  const int lstar = __ffsll(hitA) - 1;
  if (__ffsll(hitB) - 1 != lstar) return -1;
  SSTAMP(tq2);
  SSTAMP_ADD(5, tq1, tq2);
  // second pass: the winning lane's block, lanes = rows
  const float base = lstar > 0 ? lane_read(incl, lstar - 1) : 0.0f;
  int len = n - lstar * B;
  if (len > B) len = B;
  const bool in = lane < len;
  float p2 = 0.0f;
  if (in) p2 = ev.one(rows + lstar * TA::kLane + TA::row(lane, RS));
  const float inc3 = wave_inclusive_scan(p2);
  const double cum = static_cast<double>(base) + static_cast<double>(inc3);
  const unsigned long long hitA2 = __ballot(in && td + md <= cum), hitB2 = __ballot(in && td - md <= cum);
  if constexpr (TWO) {
    if (len > 64) {  // rows 64 .. len - 1 (wave-uniform branch)
      const bool in2 = lane + 64 < len;
      float p3 = 0.0f;
      if (in2) p3 = ev.one(rows + lstar * TA::kLane + TA::row(lane + 64, RS));
      const float inc4 = wave_inclusive_scan(p3);
      const double cum2 = static_cast<double>(base) + static_cast<double>(lane_read(inc3, 63)) + static_cast<double>(inc4);
      const unsigned long long hitA3 = __ballot(in2 && td + md <= cum2), hitB3 = __ballot(in2 && td - md <= cum2);
      const int ia = hitA2 ? __ffsll(hitA2) - 1 : (hitA3 ? 64 + __ffsll(hitA3) - 1 : -1);
      const int ib = hitB2 ? __ffsll(hitB2) - 1 : (hitB3 ? 64 + __ffsll(hitB3) - 1 : -1);
      if (ia < 0 || ia != ib) return -1;
      return ia * 64 + lstar;
    }
  }
  if (hitA2 == 0ull) return -1;
  const int istar = __ffsll(hitA2) - 1;
  if (__ffsll(hitB2) - 1 != istar) return -1;
  SSTAMP(tq3);
  SSTAMP_ADD(6, tq2, tq3);
  return istar * 64 + lstar;
}

// The draw on a screen tile that is whole in LDS (`rows` = row 0, field 0, lane 0).
template <int D, bool UNI>
__device__ __forceinline__ int screen_draw(LdsPtr<float> rows, int n, int B, int F, int lane, const ScreenEval<D, UNI> &ev,
                                           double u SSTAMP_PARAMS) {
  using TA = TileAddr<float>;
  const int RS = TA::stride(F);
  SSTAMP(tq0);
  ScreenSums q;
  KDEHIP_PRIO_ROWS();
  screen_rows<D, UNI>(rows + lane * TA::kLane, (B + 1) >> 1, RS, ev, q);  // (the missing second row of the last pair is padding: weight 0)
  KDEHIP_PRIO_CHAIN();
  const float s1 = q.values(), e1 = q.errors();
#ifdef KDEHIP_SCREEN_STAMPS
  asm volatile("" ::"v"(s1), "v"(e1));
#endif
  SSTAMP(tq1);
  SSTAMP_ADD(4, tq0, tq1);
  return screen_decide<D, UNI, false>(rows, n, B, RS, lane, ev, u, s1, e1 SSTAMP_ARGS);
}

}  // namespace kdehip
