// gibbs_lean.hip -- the multiscale-Gibbs product sampler for products of 2..4 (fp64: 2..8) densities with every dimension
// active (the common case: BASELINE configs 1, 2, 3 and 5), gfx950 only.
//
// Same algorithm, tiles, staging modes, conditional tables and random streams as gibbs_kernel.hip (the kernel
// for any density count / arithmetic mode; read its header first) -- what differs is where a chain keeps its
// state and how much bookkeeping a draw step carries:
//   * the density count M is a template parameter and the sweep over densities is unrolled, so "the currently
//     selected kernel of density j" (1/variance and mean/variance per dimension, lanes = dimensions) lives in
//     REGISTERS indexed at compile time instead of in per-wavefront LDS: the leave-one-out Gaussian product
//     (gaussianProductMeanCov!, src/MSGibbs01.jl:176-216) is M-1 register adds, adopting a drawn kernel
//     (updateGlbParticlesVariance!, :89-115) is one LDS gather + reciprocal, and no LDS round trip or
//     wavefront fence separates consecutive steps;
//   * a step's few descriptor fields are detached from the 16-dword scalar load into registers of their own (LeanTile);
//   * the D*(L+1) normal deviates of a chain are produced (Philox) or fetched (caller's randN) once, lane-parallel,
//     into a per-wavefront LDS strip, instead of D at a time at every level.
// Results: labels identical to gibbs_kernel.hip and to the oracle, points bit-identical to gibbs_kernel.hip
// (same reciprocal / product forms) -- tested (tests/test_gpu_lean.py).
#define KDEHIP_EXP256 1
#ifndef KDEHIP_PRELOAD_MAXB
#define KDEHIP_PRELOAD_MAXB 8   // rows per lane up to which a resident step requests its first row ahead (see `step`)
#endif
#include "gibbs_device.hpp"
#include "screen_device.hpp"

namespace kdehip {

template <int I> using IC = std::integral_constant<int, I>;
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(IC<I>{});
    static_for<N, I + 1>(f);
  }
}

// Three translation units per dimension count (build time: the kernel is instantiated per density count, precision
// and width, and the units compile in parallel): fp64 products of 2..4 densities at every workgroup width, the same
// in fp32 (-DKDEHIP_LEAN_F32), and fp64 products of 5..8 densities at 8 and 16 chains per workgroup (-DKDEHIP_LEAN_HI).
// (Products of 5, 6 and 7 densities run the general kernel since round 3: their instantiations were a third of the
// library's build time; a run-time density count inside a capacity-8 kernel was tried instead and made the register
// allocator spill -- config 4, 2048 chains: 4.74 -> 6.28 ms -- so the density count stays a compile-time constant.)
#if defined(KDEHIP_LEAN_HI) || (defined(KDEHIP_LEAN_DEV_M) && KDEHIP_LEAN_DEV_M > 4)
constexpr int kLeanMinDens = 8, kLeanMaxDens = 8;
#else
constexpr int kLeanMinDens = 2, kLeanMaxDens = 4;
#endif
constexpr int kLeanMaxNormals = 128;  // D*(L+1) normals of a chain kept in LDS (1 KiB per wavefront)

// A scalar copied through an opaque move: the copy is a register of its own.  The level descriptors arrive as one
// 16-dword scalar load; used directly, the register allocator treats the 16 registers as one unit and, under
// pressure, spills and reloads ALL of them around every use of one field (16 v_readlane per reload, several
// reloads per step).  The few fields a step needs are detached from that unit here, once per level.
__device__ __forceinline__ int scalar_copy(int x) {
  int y;
  asm("s_mov_b32 %0, %1" : "=s"(y) : "s"(x));
  return y;
}

// One density's tile on the current level: what the steps of a level need, as independent scalars.
template <int D>
struct LeanTile {
  int n;            // frontier size
  int flags;        // last_lane | uniform_bw << 8
  int lds_off;      // byte offset of the image in the LDS pool (resident mode)
  int stage_bytes;  // bytes of the image (streamed mode)
  int chunk_rows;   // rows per chunk (chunked mode)
  int seg;          // segment geometry of a chunked tile (LevelDesc.seg)
  int hdr_lo, hdr_hi;  // element offset of the tile header in the plan's data
  // the members the draw functions of gibbs_device.hpp read
  int B, F, last_lane;
  int uniform_bw;
  __device__ __forceinline__ int64_t hdr_off() const {
    return (static_cast<int64_t>(hdr_hi) << 32) | static_cast<uint32_t>(hdr_lo);
  }
  __device__ __forceinline__ void load(const kdehip_v8i &d) {  // (a StepDesc's eight dwords)
    n = scalar_copy(d[0]);
    flags = scalar_copy(d[1]);
    lds_off = scalar_copy(d[2]);
    stage_bytes = scalar_copy(d[3]);
    chunk_rows = scalar_copy(d[4]);
    seg = scalar_copy(d[5]);
    hdr_lo = scalar_copy(d[6]);
    hdr_hi = scalar_copy(d[7]);
    B = (n + 63) >> 6;
    last_lane = flags & 0xFF;
    uniform_bw = flags >> 8;
    F = uniform_bw ? D + 1 : 2 * D + 1;
  }
};

// (see `step`: the builds with registers to spare and the kept-rows second pass, on tiles that sit in LDS)
template <typename P, typename T, bool OK> constexpr bool kCanPreloadImpl = OK && kIsLdsPtr<P> && sizeof(T) == 8;

// SCHUNK: the build that also knows CHUNKED screen tiles (kStageScreenChunked).  A build of its own, launched only for plans
// that have such a level: with that code in every build, config 3 -- which never runs it -- was 3.3 % slower (32 more
// vector registers, 200 more scalar spills; profiles/r05_experiments.md section 10).
template <typename T, int D, int M, int WAVES, bool BATCH = false, bool SCHUNK = false>
__global__ __launch_bounds__(WAVES * 64) void gibbs_lean_kernel(PlanDev plan_, RunArgs a_) {
  const LaunchView<BATCH> view(plan_, a_);
  const PlanDev &plan = view.plan;
  const auto &a = view.a;
  constexpr bool kPrefetchRows = (WAVES <= 12) || sizeof(T) == 4 || D <= 4;  // as in gibbs_kernel.hip
  // how the rows of an LDS tile are read: see LdsPtrSplit (fp32 reads its row pairs as single loads either way: load_pair)
  using RowPtr = std::conditional_t<(WAVES == 16 && sizeof(T) == 8), LdsPtrSplit<T>, LdsPtr<T>>;
  // the wavefronts that issue the copies of streamed tiles and chunks: the OLDEST wavefront of every SIMD (wavefronts w,
  // w + 4, w + 8, w + 12 share a SIMD, the lowest is the oldest: scripts/micro/simd_map.hip); everyone when each SIMD has one
#ifdef KDEHIP_X_ALLCOPY  // (A/B only: rounds 1-3, every wavefront issues its share at the start of the step)
  constexpr int kCopyWaves = WAVES;
#else
  constexpr int kCopyWaves = WAVES > 4 ? 4 : WAVES;  // (16 wavefronts: the oldest of the four on a SIMD; its half: c5 +0.8 %)
#endif
  constexpr bool kKeptRows = (WAVES <= 8);
  // fp32 screening of the deep levels (screen_device.hpp): the fp64 instantiations of plain launches
  constexpr bool kScreen = sizeof(T) == 8 && !BATCH;
  constexpr bool kPreloadBuild = kKeptRows && kPrefetchRows;
  // LDS: [exp table 2 KiB][normals: 1 KiB per chain][uniforms: WAVES x 1 KiB][tile pool]
  constexpr int kNormOff = 2048;
  constexpr int kUnifOff = kNormOff + WAVES * kLeanMaxNormals * 8;
  constexpr int kPoolOff = kUnifOff + WAVES * 1024;
  // COOPERATIVE fp64 REPEAT (round 6; chunked screen levels of the 8-chain builds).  A wavefront whose fp32 decision is not
  // certified repeats the draw in fp64 -- 60-130 rows read through the L2, ~11 us -- while its workgroup waits at the next
  // chunk barrier (config 4: 13 % of the kernel).  With 8 chains per workgroup 22 KB of the CU's LDS are unused: an
  // exchange area behind the pool lets FOUR wavefronts take the four row classes of the canonical lane sums (LaneAcc: a_k =
  // the rows r = k mod 4 in increasing order -- each class is one sequential sum, so the split changes no bit), two requests
  // at a time.  Per request slot: mean and variance of the draw per dimension (what the helpers build the evaluator from)
  // and the 4 x 64 class sums.  Flags are double-buffered by step parity: a wavefront may be one step ahead of another.
#ifndef KDEHIP_X_NO_COOP
  constexpr bool kCoop = kScreen && SCHUNK && WAVES == 8;
#else
  constexpr bool kCoop = false;
#endif
  constexpr int kXchgOff = kPoolOff + kLdsPoolBytes;
  constexpr int kXchgSlotDoubles = 32 + 4 * 64;  // [0..7] mean, [8..15] variance, [32 + 64 k + lane] class sums
  constexpr int kXchgBytes = kCoop ? 64 + WAVES * kXchgSlotDoubles * 8 : 0;
  static_assert(kPoolOff + kLdsPoolBytes + kXchgBytes <= 160 * 1024, "LDS budget of one CU exceeded");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kPoolOff + kLdsPoolBytes + kXchgBytes];

  double *sExpTab = reinterpret_cast<double *>(smem);
  if (threadIdx.x < 256) sExpTab[threadIdx.x] = kExp2Tab256[threadIdx.x];
  if constexpr (kCoop) {
    if (threadIdx.x < 16) reinterpret_cast<int *>(smem + kXchgOff)[threadIdx.x] = 0;  // request flags [2 parities][WAVES]
  }

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  KDEHIP_PRIO_CHAIN();
  const int chain = wave;  // one wavefront = one chain
  int64_t s = static_cast<int64_t>(view.block) * WAVES + chain;
  const bool live = s < a.Np;  // surplus wavefronts of the last workgroup replay the last chain and store nothing
  if (!live) s = a.Np - 1;
  const uint64_t gs = static_cast<uint64_t>(a.sample_offset + s);
  const void *fb = live ? static_cast<const void *>(plan.levels) : nullptr;  // who counts uniform fallbacks

  const int L = plan.L;
  const T *__restrict__ data = static_cast<const T *>(plan.data);
  const LevelTable levels{(const __attribute__((address_space(4))) kdehip_v16i *)(plan.levels)};
  const StepTable steps{(const __attribute__((address_space(4))) kdehip_v8i *)(plan.levels + 2 * plan.M * (plan.L + 1))};
  unsigned char *pool = smem + kPoolOff;
  const int dl = lane < D ? lane : D - 1;  // this lane's dimension in the "lanes = dimensions" phases
  const int vlev = a.variant % 1000;

  // ---- the chain's normal deviates, once (samplePoint! consumes D per level + D at the end, :440-463) ----
  double *sNorm = reinterpret_cast<double *>(smem + kNormOff) + chain * kLeanMaxNormals;
  const int R = D * (L + 1);
#ifdef KDEHIP_X_OLDNORMALS
  for (int r = lane; r < R; r += 64)
    sNorm[r] = a.rng_philox ? philox_normal(a.seed, gs, static_cast<uint32_t>(r)) : a.randN[s * a.R + r];
#else
  if (a.rng_philox) {  // a lane makes BOTH normals of a Philox block (one logarithm, one sine/cosine pair for two)
    for (int b = lane; 2 * b < R; b += 64) {
      double n0, n1;
      philox_normal_pair(a.seed, gs, static_cast<uint32_t>(b), n0, n1);
      sNorm[2 * b] = n0;
      if (2 * b + 1 < R) sNorm[2 * b + 1] = n1;
    }
  } else {
    for (int r = lane; r < R; r += 64) sNorm[r] = a.randN[s * a.R + r];
  }
#endif
  __syncthreads();  // (exp table; the strip is only read by its own chain's wavefronts)

  // ---- chain state: selected kernel of every density, lanes = dimensions ----
  T lam[M], lmu[M];  // 1/variance and mean/variance
  int psel[M];       // tile position of the selected entry (wave-uniform)

  // adopt entry `pos` of the tile whose header is at `hdr` (LDS or global) as density j's kernel
  auto adopt = [&](auto jc, const auto &ds, auto hdr, int pos) {
    constexpr int j = decltype(jc)::value;
    using TA = TileAddr<T>;
    auto e = hdr + kTileHeader + TA::row(pos >> 6, TA::stride(ds.F)) + (pos & 63) * TA::kLane;
    const T mu = e[dl * TA::kField];
    const T var = ds.uniform_bw ? hdr[dl] : e[(D + dl) * TA::kField];
    const T l = fast_rcp(var);
    lam[j] = l;
    lmu[j] = mu * l;
    psel[j] = pos;
  };
  // Gaussian product of the selected kernels without density `skip` (-1: all) for this lane's dimension, the
  // reference's summation order (:199-213)
  auto product = [&](auto skipc, T &mean, T &cov) {
    constexpr int skip = decltype(skipc)::value;
    T ls = T(0), ms = T(0);
    static_for<M>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k != skip) { ls += lam[k]; ms += lmu[k]; }
    });
    cov = fast_rcp(ls);
    mean = cov * ms;
  };

  // One label draw on level descriptor ds against the per-dimension (mean, cov) held by the dimension lanes
  // (makeFasterSampleIndex! + selectLabelOnLevel, :250-351); `run(ev)` gets the evaluator.
  auto draw = [&](const auto &ds, auto hdr, T mean, T cov, auto &&run) -> int {
    if (ds.uniform_bw) {
      EvalUniform<T, D> ev;
      ev.tab = sExpTab;
      const T c = hdr[dl] + cov;
      T cen, nin;
      EvalUniform<T, D>::operands(mean, c, true, cen, nin);
      T Pr = T(1);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        ev.center[d] = lane_read(cen, d);
        ev.ninv[d] = lane_read(nin, d);
        Pr *= lane_read(c, d);
      }
      ev.scale = Num<T>::rsqrt(Pr);
      return run(ev);
    }
    EvalFast<T, D, false> ev;
    ev.tab = sExpTab;
    ev.act = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      ev.center[d] = lane_read(mean, d);
      ev.cov[d] = lane_read(cov, d);
    }
    return run(ev);
  };

  // ---- uniform draws: 128 at a time across the lanes, handed out with v_readlane (as gibbs_kernel.hip) ----
  uint32_t c = static_cast<uint32_t>(M);  // select call index; the M init calls read nothing (:477-497)
  uint32_t ubatch = 0xFFFFFFFFu;
  double *sUnif = reinterpret_cast<double *>(smem + kUnifOff) + wave * 128;  // this wavefront's current 128 uniforms
  auto next_uniform = [&]() -> double {
    const uint32_t b = c >> 7;
    if (b != ubatch) {  // lane ln produces uniforms 128*b + 2*ln and + 2*ln+1 (one Philox block, or two stream elements)
      ubatch = b;
      double u_even, u_odd;
      if (a.rng_philox) {
        const Philox4 r = philox_block(a.seed, gs, b * 64u + static_cast<uint32_t>(lane), 0u);
        u_even = bits_to_unit(r.v[0], r.v[1]);
        u_odd = bits_to_unit(r.v[2], r.v[3]);
      } else {  // element i of the sample's slice feeds call i+1
        const int64_t i0 = s * a.K + static_cast<int64_t>(b) * 128 + 2 * lane - 1;
        u_even = (i0 >= 0 && i0 < a.nU) ? a.randU[i0] : 0.5;
        u_odd = (i0 + 1 < a.nU) ? a.randU[i0 + 1] : 0.5;
      }
      wave_sync();
      sUnif[2 * lane] = u_even;
      sUnif[2 * lane + 1] = u_odd;
      wave_sync();
    }
    const double u = sUnif[c & 127u];  // one broadcast read: the value is the same in every lane
    ++c;
    return u;
  };

  // ---- init: frontier = {root}, label = root (levelInit!/initIndices!/calcIndices!, :587-589) ----
  static_for<M>([&](auto jc) {
    const LevelDesc ds = levels[decltype(jc)::value * (L + 1)];
    adopt(jc, ds, data + ds.hdr_off, 0);
  });

  const TabTable tabs{(const __attribute__((address_space(4))) kdehip_v8i *)(plan.tabdesc)};
  const T *tables = static_cast<const T *>(plan.tables);
  const int Lt = (vlev == 1 || vlev == 4 || !a.use_tables) ? 0 : plan.Lt;

  // the draw on a tile readable through one pointer
  auto draw_rows = [&](const auto &ds, auto rows, const auto &ev, double u) -> int {
    using P = decltype(rows);
    return draw_label<T, P, kPrefetchRows, kKeptRows>(rows, ds, lane, ev, u, fb);
  };

  // the fp64 REPEAT of a screened step on the plan's fp64 tile in global memory: the canonical lane sums with four rows
  // requested ahead where the registers allow it (up to 7 fields per row: every shared-bandwidth tile, per-node tiles up to
  // D = 3), then the unchanged selection.  ONLY in the chunked-screen builds (the plans whose workgroups wait for a repeating
  // wavefront: config 4 3.04 -> 2.99 ms): in the plain build the eight rows in flight cost 16 vector registers (209 -> 225,
  // i.e. 232 allocated), and with 2 x 232 of a SIMD's 512 taken the NEXT call's table-build kernel (64 registers per
  // wavefront) no longer fits beside the sampler -- config 3's complete call went from 0.481 to 0.494 ms per step with an
  // unchanged 0.471 ms kernel (profiles/r06_experiments.md section 10).
  auto draw_rows_repeat = [&](const auto &ds, const T *rows, const auto &ev, double u) -> int {
    using Ev = std::decay_t<decltype(ev)>;
#ifndef KDEHIP_X_NO_DEEP
    if constexpr (kScreen && SCHUNK && sizeof(typename Ev::Row) <= 7 * sizeof(T) && WAVES <= 8) {
      LaneAcc<T> acc;
      KDEHIP_PRIO_ROWS();
      lane_rows_all_deep<T, Ev>(rows, ds.B, TileAddr<T>::stride(ds.F), lane, ev, acc);
      KDEHIP_PRIO_CHAIN();
      return select_or_raise<T, const T *>(acc.total(), rows, ds, lane, ev, u, fb);
    }
#endif
    return draw_rows(ds, rows, ev, u);
  };

  // a step on an LDS tile with per-node bandwidths and at most 8 rows per lane whose first row `row` has been requested
  // already: broadcasts, the kept-rows draw (or the single-row one), adoption
#ifdef KDEHIP_SCREEN_STAMPS
  unsigned long long rstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r_prev_end = 0;
  bool rstamp_on = false;
#define RSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RSTAMP_ADD(slot, t0, t1) do { if (rstamp_on) rstamp[slot] += (t1) - (t0); } while (0)
#else
#define RSTAMP(var) do {} while (0)
#define RSTAMP_ADD(slot, t0, t1) do {} while (0)
#endif
  auto step_kept = [&](auto jc, const auto &ds, auto hdr, const auto &row, T mean, T cov, double u) {
    RSTAMP(tr1);
    auto rows1 = hdr + kTileHeader;
    using P1 = decltype(rows1);
    using Ev = EvalFast<T, D, false>;
    Ev ev;
    ev.tab = sExpTab;
    ev.act = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      ev.center[d] = lane_read(mean, d);
      ev.cov[d] = lane_read(cov, d);
    }
#ifdef KDEHIP_SCREEN_STAMPS
    asm volatile("" ::"s"(ev.center[D - 1]), "s"(ev.cov[D - 1]));
#endif
    RSTAMP(tr2);
    RSTAMP_ADD(1, tr1, tr2);
    int pos1;
    if (ds.B == 1) {
      const T S = ev(row);  // (= LaneAcc's total of a one-row lane: (v + 0) + (0 + 0))
      pos1 = select_or_raise<T, P1>(S, rows1, ds, lane, ev, u, fb);
    } else if (ds.B <= 4) {
      pos1 = draw_label_kept<T, P1, Ev, 4>(rows1, ds, lane, ev, u, fb, row);
    } else {
      pos1 = draw_label_kept<T, P1, Ev, 8>(rows1, ds, lane, ev, u, fb, row);
    }
    pos1 = __builtin_amdgcn_readfirstlane(pos1);
    RSTAMP(tr3);
    RSTAMP_ADD(2, tr2, tr3);
    adopt(jc, ds, hdr, pos1);
#ifdef KDEHIP_SCREEN_STAMPS
    {
      const double sink = lam[decltype(jc)::value] + lmu[decltype(jc)::value];
      asm volatile("" ::"v"(sink));
    }
#endif
    RSTAMP(tr4);
    RSTAMP_ADD(3, tr3, tr4);
#ifdef KDEHIP_SCREEN_STAMPS
    r_prev_end = tr4;
#endif
  };

  // one (pass, density) step on a tile readable through one pointer: leave-one-out product (sweeps) or the point
  // just drawn (sampleIndices! pass, :364-385), the draw, and the new kernel
  auto step = [&](auto jc, const auto &ds, auto hdr, bool first, T x) {
#ifndef KDEHIP_X_NO_PRELOAD
    // Frontiers with per-node bandwidths whose tile sits in LDS for the whole level, up to KDEHIP_PRELOAD_MAXB rows per
    // lane (config 3: levels 5..8 and the first pass of 1..4): the fields of the lane's first row do not depend on the
    // chain's state, so they are requested FIRST and land while the leave-one-out product and the broadcasts run -- a
    // single-row step is one dependent chain, and this takes an LDS round trip out of it (profiles/r04_experiments.md).
    if constexpr (kCanPreloadImpl<decltype(hdr + kTileHeader), T, kPreloadBuild>) {
      if (!ds.uniform_bw && ds.B <= KDEHIP_PRELOAD_MAXB) {
        RSTAMP(tr0);
#ifdef KDEHIP_SCREEN_STAMPS
        if (r_prev_end) RSTAMP_ADD(4, r_prev_end, tr0);
#endif
        const auto row = EvalFast<T, D, false>().load(hdr + kTileHeader + lane * TileAddr<T>::kLane);
        T mean1 = x, cov1 = T(0);
        if (!first) product(jc, mean1, cov1);
        const double u1 = next_uniform();
#ifdef KDEHIP_SCREEN_STAMPS
        asm volatile("" ::"v"(mean1), "v"(cov1), "v"(u1));
        {
          RSTAMP(trp);
          RSTAMP_ADD(0, tr0, trp);
        }
#endif
        step_kept(jc, ds, hdr, row, mean1, cov1, u1);
        return;
      }
    }
#endif
    T mean = x, cov = T(0);
    if (!first) product(jc, mean, cov);
    const double u = next_uniform();
    auto rows = hdr + kTileHeader;
    const int pos = __builtin_amdgcn_readfirstlane(draw(ds, hdr, mean, cov, [&](const auto &ev) {
      return draw_rows(ds, rows, ev, u);
    }));
    adopt(jc, ds, hdr, pos);
  };

  // ---- a step on a SCREENED level (screen_device.hpp): the draw in packed fp32 from the level's screen tile in LDS,
  // certified against the fp64 decision -- or repeated in fp64 from the plan's fp64 tile in global memory; the new kernel
  // always from the fp64 tile ----
  uint32_t n_screened = 0, n_repeated = 0;  // (diagnostic counters: kdehip_product_screen_stats)
#ifdef KDEHIP_SCREEN_STAMPS
  unsigned long long sstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool sstamp_on = false;
#endif
  // (chunked screen tiles, step_screen_chunked below) The fp32 evaluator and the error-bound coefficients of a screened
  // step, from the screen tile's header values (lanes = dimensions) and the chain's leave-one-out product; body(ev, ok):
  // ok = the step is inside the ranges the bound assumes (wave-uniform; with !ok the evaluator must not be used for a
  // decision -- every wavefront still has to walk the chunks: their barriers).  step_screen spells the same set-up out
  // itself: written through these two helpers it was 1.2 % slower on config 3 (A/B of development libraries, round 5).
  auto screen_eval = [&](const auto &ds, double mu0, double cmin, float mmax, float valid, T mean, T cov, auto &&body) -> int {
    const float cen = static_cast<float>(mean - mu0);
    const float cf = static_cast<float>(cmin + cov), covf = static_cast<float>(cov);
    const float acen = fabsf(cen);
    const float t = fminf(mmax + acen, 2.0f * acen);
    float a2 = lane < D ? t * t * __builtin_amdgcn_rcpf(cf) : 0.0f;
    // (a2 <= kScreenMaxA2 in every dimension keeps na <= 2^-11, the regime the bound is linearised for: screen_device.hpp)
    const bool inr = (acen <= kScreenMaxAbsMean) && (covf <= static_cast<float>(kScreenMaxVar)) && (a2 <= kScreenMaxA2);  // (false for a NaN)
    const bool ok = valid != 0.0f && __ballot(lane < D && !inr) == 0ull;
    a2 += dpp_fetch<0x111, 0xF>(a2);  // row_shr:1, 2, 4: lane 7 holds the sum over the (at most 8) dimension lanes
    a2 += dpp_fetch<0x112, 0xF>(a2);
    a2 += dpp_fetch<0x114, 0xF>(a2);
    const float na = __builtin_sqrtf(lane_read(a2, 7)) * (kScreenU * kScreenSqrtC0 * 1.01f);
    using SC = ScreenConst<D>;
    const float Bc = (kScreenLn2 * 1.01f) * (na + (ds.uniform_bw ? SC::kx_uni : SC::kx_node) * kScreenU);
    const float A = (kScreenLn2 * 1.01f) * na +
                    (static_cast<float>(((ds.B + 1) >> 1) + 9) + (ds.uniform_bw ? SC::vc_uni : SC::vc_node)) * kScreenU;
    if (ds.uniform_bw) {
      ScreenEval<D, true> ev;
      ev.A = A; ev.Bc = Bc;
      const float ninv = -kScreenC0 * __builtin_amdgcn_rcpf(cf);
      float pr = 1.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        ev.cen[d] = lane_read(cen, d);
        ev.b[d] = lane_read(ninv, d);
        pr *= lane_read(cf, d);
      }
      ev.scale = __builtin_amdgcn_rsqf(pr);
      return body(ev, ok);
    } else {
      ScreenEval<D, false> ev;
      ev.A = A; ev.Bc = Bc; ev.scale = 1.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        ev.cen[d] = lane_read(cen, d);
        ev.b[d] = lane_read(covf, d);
      }
      return body(ev, ok);
    }
  };
  // what follows the screen's verdict: the fp64 draw on the fp64 tile in global memory when it is not certified (or out of
  // the screen's range), and the new kernel from the fp64 tile
  auto screen_finish = [&](auto jc, const auto &ds, int pos, T mean, T cov, double u) {
    const T *hdrg = data + ds.hdr_off();
    ++n_screened;
    if (pos < 0) {
      ++n_repeated;
      pos = __builtin_amdgcn_readfirstlane(draw(ds, hdrg, mean, cov, [&](const auto &ev) {
        return draw_rows_repeat(ds, hdrg + kTileHeader, ev, u);
      }));
    }
    adopt(jc, ds, hdrg, pos);
  };
  // ---- the cooperative repeat's protocol (kCoop) ----
  // post: at the end of a chunked step whose decision was not certified -- the draw's operands and a flag (parity of the
  // step); the wavefront goes on to the next step's first barrier WITHOUT a label.  resolve: right behind that barrier every
  // wavefront looks at the flags of the step before; if any is set, wavefronts 0-3 take the four row classes of the first
  // request, 4-7 those of the second (then the third and fourth, ...), one more barrier, and every requester forms the
  // canonical total from the four class sums, selects (second pass from global memory, as the solo repeat does) and adopts.
  bool pend = false;
  T pend_mean = T(0), pend_cov = T(0);
  double pend_u = 0.0;
  int cstep = 0;  // chunked steps so far (wave-uniform; its parity picks the flag buffer)
  auto coop_post = [&](T mean, T cov, double u) {
    if constexpr (kCoop) {
      pend = true; pend_mean = mean; pend_cov = cov; pend_u = u;
      double *xs = reinterpret_cast<double *>(smem + kXchgOff + 64) + wave * kXchgSlotDoubles;
      if (lane < D) { xs[dl] = static_cast<double>(mean); xs[8 + dl] = static_cast<double>(cov); }
      if (lane == 0) reinterpret_cast<int *>(smem + kXchgOff)[(cstep & 1) * WAVES + wave] = 1;
    }
  };
  // (`par`: the parity of the step whose requests are looked at; jp / dsp: that step's density and tile)
  auto coop_resolve = [&](auto jp, const auto &dsp, int par) {
    if constexpr (kCoop) {
      int *flags = reinterpret_cast<int *>(smem + kXchgOff) + par * WAVES;
      const int f = flags[lane & (WAVES - 1)];
      const unsigned mask = static_cast<unsigned>(__ballot(f != 0 && lane < WAVES));
      if (mask == 0u) return;  // (wave-uniform, and the same in every wavefront: read behind a barrier, buffered by parity)
      using TA = TileAddr<T>;
      const T *hdrg = data + dsp.hdr_off();
      const T *rows = hdrg + kTileHeader;
      const int RS = TA::stride(dsp.F), B = dsp.B;
      const int group = wave >> 2, k = wave & 3;
      int idx = 0;
      for (unsigned m = mask; m != 0u; m &= m - 1u, ++idx) {
        if ((idx & 1) != group) continue;
        const int r = __builtin_ctz(m);
        double *xs = reinterpret_cast<double *>(smem + kXchgOff + 64) + r * kXchgSlotDoubles;
        const T mean_r = static_cast<T>(xs[dl]), cov_r = static_cast<T>(xs[8 + dl]);
        T acc = T(0);
        KDEHIP_PRIO_ROWS();
        draw(dsp, hdrg, mean_r, cov_r, [&](const auto &ev) {
          // class k: rows k, k + 4, k + 8, ... in increasing order, four of them requested ahead
          using Ev = std::decay_t<decltype(ev)>;
          using Row = typename Ev::Row;
          const T *e = rows + lane * TA::kLane;
          auto at = [&](int rr) { return e + TA::row(rr < B ? rr : B - 1, RS); };
          if constexpr (sizeof(Row) <= 4 * sizeof(T)) {  // (shared-bandwidth tiles up to D = 3: eight rows fit the registers)
            Row a0 = ev.load(at(k)), a1 = ev.load(at(k + 4)), a2 = ev.load(at(k + 8)), a3 = ev.load(at(k + 12));
            int i = k;
            for (; i + 12 < B; i += 16) {
              const Row b0 = ev.load(at(i + 16)), b1 = ev.load(at(i + 20)), b2 = ev.load(at(i + 24)), b3 = ev.load(at(i + 28));
              __builtin_amdgcn_sched_barrier(0);
              acc += ev(a0); acc += ev(a1); acc += ev(a2); acc += ev(a3);
              a0 = b0; a1 = b1; a2 = b2; a3 = b3;
            }
            if (i < B) acc += ev(a0);
            if (i + 4 < B) acc += ev(a1);
            if (i + 8 < B) acc += ev(a2);
          } else {
            Row a0 = ev.load(at(k)), a1 = ev.load(at(k + 4));
            int i = k;
            for (; i + 4 < B; i += 8) {
              const Row b0 = ev.load(at(i + 8)), b1 = ev.load(at(i + 12));
              __builtin_amdgcn_sched_barrier(0);
              acc += ev(a0); acc += ev(a1);
              a0 = b0; a1 = b1;
            }
            if (i < B) acc += ev(a0);
          }
          return 0;
        });
        KDEHIP_PRIO_CHAIN();
        xs[32 + 64 * k + lane] = static_cast<double>(acc);
      }
      __syncthreads();  // the class sums of every request are in place
      if (pend) {
        double *xs = reinterpret_cast<double *>(smem + kXchgOff + 64) + wave * kXchgSlotDoubles;
        const T a0 = static_cast<T>(xs[32 + lane]), a1 = static_cast<T>(xs[32 + 64 + lane]);
        const T a2 = static_cast<T>(xs[32 + 128 + lane]), a3 = static_cast<T>(xs[32 + 192 + lane]);
        const T S = (a0 + a1) + (a2 + a3);  // LaneAcc::total
        const int pos = __builtin_amdgcn_readfirstlane(draw(dsp, hdrg, pend_mean, pend_cov, [&](const auto &ev) {
          return select_or_raise<T, const T *>(S, rows, dsp, lane, ev, pend_u, fb);
        }));
        adopt(jp, dsp, hdrg, pos);
        pend = false;
        if (lane == 0) flags[wave] = 0;
      }
    }
  };

  auto step_screen = [&](auto jc, const auto &ds, int sc_lds_off, bool first, T x) {
    if constexpr (kScreen) {
      SSTAMP(ts0);
      // lanes = dimensions: what the centred operands and the error bound need from the tile's header (requested first:
      // none of it depends on the chain)
      const LdsPtr<float> h32 = (LdsPtr<float>)(pool + sc_lds_off);
      const LdsPtr<double> h64 = (LdsPtr<double>)(pool + sc_lds_off);
      const double mu0 = h64[dl], cmin = h64[8 + dl];
      const float mmax = h32[32 + dl], valid = h32[40];
      T mean = x, cov = T(0);
      if (!first) product(jc, mean, cov);
      const double u = next_uniform();
      SSTAMP(ts1);
      SSTAMP_ADD(0, ts0, ts1);
      const float cen = static_cast<float>(mean - mu0);
      const float cf = static_cast<float>(cmin + cov), covf = static_cast<float>(cov);
      const float acen = fabsf(cen);
      const float t = fminf(mmax + acen, 2.0f * acen);
      float a2 = lane < D ? t * t * __builtin_amdgcn_rcpf(cf) : 0.0f;
      // (a2 <= kScreenMaxA2 in every dimension keeps na <= 2^-11, the regime the bound is linearised for: screen_device.hpp)
#ifndef KDEHIP_X_NO_NA_GUARD
      const bool inr = (acen <= kScreenMaxAbsMean) && (covf <= static_cast<float>(kScreenMaxVar)) && (a2 <= kScreenMaxA2);  // (false for a NaN)
#else
      const bool inr = (acen <= kScreenMaxAbsMean) && (covf <= static_cast<float>(kScreenMaxVar));
#endif
      int pos = -1;
      const T *hdrg = data + ds.hdr_off();
      if (valid != 0.0f && __ballot(lane < D && !inr) == 0ull) {
        a2 += dpp_fetch<0x111, 0xF>(a2);  // row_shr:1, 2, 4: lane 7 holds the sum over the (at most 8) dimension lanes
        a2 += dpp_fetch<0x112, 0xF>(a2);
        a2 += dpp_fetch<0x114, 0xF>(a2);
        const float na = __builtin_sqrtf(lane_read(a2, 7)) * (kScreenU * kScreenSqrtC0 * 1.01f);
        using SC = ScreenConst<D>;
        const float Bc = (kScreenLn2 * 1.01f) * (na + (ds.uniform_bw ? SC::kx_uni : SC::kx_node) * kScreenU);
        const float A = (kScreenLn2 * 1.01f) * na +
                        (static_cast<float>(((ds.B + 1) >> 1) + 9) + (ds.uniform_bw ? SC::vc_uni : SC::vc_node)) * kScreenU;
        const LdsPtr<float> rows32 = h32 + kScreenHeaderFloats;
        if (ds.uniform_bw) {
          ScreenEval<D, true> ev;
          ev.A = A; ev.Bc = Bc;
          const float ninv = -kScreenC0 * __builtin_amdgcn_rcpf(cf);
          float pr = 1.0f;
#pragma unroll
          for (int d = 0; d < D; ++d) {
            ev.cen[d] = lane_read(cen, d);
            ev.b[d] = lane_read(ninv, d);
            pr *= lane_read(cf, d);
          }
          ev.scale = __builtin_amdgcn_rsqf(pr);
          pos = screen_draw<D, true>(rows32, ds.n, ds.B, ds.F, lane, ev, u SSTAMP_ARGS);
        } else {
          ScreenEval<D, false> ev;
          ev.A = A; ev.Bc = Bc; ev.scale = 1.0f;
#pragma unroll
          for (int d = 0; d < D; ++d) {
            ev.cen[d] = lane_read(cen, d);
            ev.b[d] = lane_read(covf, d);
          }
          pos = screen_draw<D, false>(rows32, ds.n, ds.B, ds.F, lane, ev, u SSTAMP_ARGS);
        }
        pos = __builtin_amdgcn_readfirstlane(pos);
      }
      SSTAMP(ts2);
      SSTAMP_ADD(1, ts1, ts2);
      ++n_screened;
      if (pos < 0) {  // not certified (or out of the screen's range): the fp64 draw on the fp64 tile
        ++n_repeated;
        pos = __builtin_amdgcn_readfirstlane(draw(ds, hdrg, mean, cov, [&](const auto &ev) {
          return draw_rows_repeat(ds, hdrg + kTileHeader, ev, u);
        }));
      }
      SSTAMP(ts3);
      SSTAMP_ADD(2, ts2, ts3);
      adopt(jc, ds, hdrg, pos);
#ifdef KDEHIP_SCREEN_STAMPS
      {  // (the adopted kernel has arrived: make the stamp wait for it)
        const double sink = lam[decltype(jc)::value] + lmu[decltype(jc)::value];
        asm volatile("" ::"v"(sink));
      }
#endif
      SSTAMP(ts4);
      SSTAMP_ADD(3, ts3, ts4);
    }
  };

  int gchunk = 0;  // workgroup-wide running chunk counter of the chunked mode (selects the pool half)
  auto stage_chunk = [&](const auto &ds, int r0, int half) {
    using TA = TileAddr<T>;
    const int RS = TA::stride(ds.F), rc = ds.chunk_rows;
    const int nrows = (ds.B - r0 < rc) ? (ds.B - r0) : rc;
    const int bytes = (static_cast<int>(TA::span(nrows, RS)) * int(sizeof(T)) + 1023) & ~1023;  // (r0: a multiple of 4 rows)
    const unsigned char *src = reinterpret_cast<const unsigned char *>(data + ds.hdr_off() + kTileHeader + TA::row(r0, RS));
    // Issued by the OLDER wavefront(s) of every SIMD only (kCopyWaves): a `buffer_load ... lds` costs its issuer 60-180
    // cycles, the hardware serves the older wavefront of a SIMD first, so the younger one is the step's critical path
    // and the older one waits for it at the next barrier anyway (config 4: 4.36 -> 4.16 ms, profiles/r04_experiments.md)
    if (wave < kCopyWaves) stage_tile<kCopyWaves>(src, pool + half * (kLdsPoolBytes / 2), bytes, wave, lane);
  };

  // ---- a step on a screened level whose screen tile comes through the pool halves in CHUNKS of whole row pairs (chunk 0
  // carries the header); one barrier per chunk, the copy of the next chunk -- or of the next step's chunk 0 -- overlaps the
  // evaluation of this one; the second pass reads the screen tile in global memory ----
  auto stage_schunk = [&](const LevelDesc &sc, int p0, int half) {
    using TA = TileAddr<float>;
    const int RS = TA::stride(sc.F), cp = sc.chunk_rows >> 1, npairs = (sc.B + 1) >> 1;
    const int np = (npairs - p0 < cp) ? (npairs - p0) : cp;
    const int head = p0 == 0 ? kScreenHeaderFloats : 0;
    const int bytes = ((head + np * RS) * 4 + 1023) & ~1023;
    const unsigned char *src = reinterpret_cast<const unsigned char *>(reinterpret_cast<const float *>(plan.data) + sc.hdr_off +
                                                                       (p0 == 0 ? 0 : kScreenHeaderFloats + p0 * RS));
    if (wave < kCopyWaves) stage_tile<kCopyWaves>(src, pool + half * (kLdsPoolBytes / 2), bytes, wave, lane);
  };
  auto step_screen_chunked = [&](auto jc, const auto &ds, const auto &dsp, const LevelDesc &sc, const LevelDesc &scn, bool more,
                                 bool first, bool level_start, T x) {
    if constexpr (kScreen && SCHUNK) {
      using TA = TileAddr<float>;
      T mean = x, cov = T(0);
      const bool was_pend = kCoop && pend;  // (its label of the step before is still out: the product has to wait)
      if (!first && !was_pend) product(jc, mean, cov);
      const double u = next_uniform();
      const int RS = TA::stride(sc.F), cp = sc.chunk_rows >> 1, npairs = (sc.B + 1) >> 1;
      staging_barrier();  // chunk 0 has landed for every wavefront; the other half is free again
      if (cp < npairs) stage_schunk(sc, cp, (gchunk + 1) & 1);
      else if (more) stage_schunk(scn, 0, (gchunk + 1) & 1);
      if constexpr (kCoop) {
        constexpr int j = decltype(jc)::value;
        if (!level_start) coop_resolve(IC<(j + M - 1) % M>{}, dsp, (cstep + 1) & 1);  // the requests of the step before
        if (was_pend && !first) product(jc, mean, cov);
      }
      const LdsPtr<float> h32 = (LdsPtr<float>)(pool + (gchunk & 1) * (kLdsPoolBytes / 2));
      const LdsPtr<double> h64 = (LdsPtr<double>)(pool + (gchunk & 1) * (kLdsPoolBytes / 2));
      const double mu0 = h64[dl], cmin = h64[8 + dl];
      const float mmax = h32[32 + dl], valid = h32[40];
      const float *grows = reinterpret_cast<const float *>(plan.data) + sc.hdr_off + kScreenHeaderFloats;
      const int pos = __builtin_amdgcn_readfirstlane(screen_eval(ds, mu0, cmin, mmax, valid, mean, cov, [&](const auto &ev, bool ok) {
        using Ev = std::decay_t<decltype(ev)>;
        ScreenSums q;
        KDEHIP_PRIO_ROWS();
        if (ok) screen_rows<D, Ev::kUni>(h32 + kScreenHeaderFloats + lane * TA::kLane, npairs < cp ? npairs : cp, RS, ev, q);
        ++gchunk;
        for (int p0 = cp; p0 < npairs; p0 += cp, ++gchunk) {
          staging_barrier();
          if (p0 + cp < npairs) stage_schunk(sc, p0 + cp, (gchunk + 1) & 1);
          else if (more) stage_schunk(scn, 0, (gchunk + 1) & 1);
          const int np = (npairs - p0 < cp) ? (npairs - p0) : cp;
          if (ok) screen_rows<D, Ev::kUni>((LdsPtr<float>)(pool + (gchunk & 1) * (kLdsPoolBytes / 2)) + lane * TA::kLane, np, RS, ev, q);
        }
        KDEHIP_PRIO_CHAIN();
        if (!ok) return -1;
        return screen_decide<D, Ev::kUni, true>(grows, ds.n, ds.B, RS, lane, ev, u, q.values(), q.errors() SSTAMP_ARGS);
      }));
      if constexpr (kCoop) {
        ++n_screened;
        if (pos < 0) {
          ++n_repeated;
          coop_post(mean, cov, u);
        } else {
          adopt(jc, ds, data + ds.hdr_off(), pos);
        }
        ++cstep;
      } else {
        screen_finish(jc, ds, pos, mean, cov, u);
      }
    }
  };

#ifdef KDEHIP_EXPERIMENTS  // diagnostic builds: variant 100+k stops after level k (scripts/level_profile.sh)
  const int Lrun = (vlev >= 100 && vlev - 100 < L) ? vlev - 100 : L;
#else
  const int Lrun = L;
#endif
  for (int l = 1; l <= Lrun; ++l) {
    // samplePoint! (:440-463): x = mean + sqrt(cov) * randn, all densities included
    T x;
    {
      T mean, cov;
      product(IC<-1>{}, mean, cov);
      x = mean + Num<T>::sqrt(cov) * static_cast<T>(sNorm[(l - 1) * D + dl]);
    }
    // Every step fetches its own tile descriptor -- and its successor's, whose tile it stages -- from the scalar cache
    // (one s_load_dwordx16 each, issued a step's worth of work before use).  Keeping the level's M descriptors in
    // scalar registers instead (round 2) crowds the register file: M x 7 live fields pushed the 16-wavefront builds into
    // scratch in their per-step code (config 4 with 16,384 chains: 39 ms against 31 ms; config 3: 4.32 against 4.22 ms)
    // and bought the 8-wavefront builds nothing (0.6255 against 0.6222 ms).
    const int level_mode = scalar_copy(levels[l].stage_mode);
    auto tile_raw = [&](int j) -> kdehip_v8i { return steps.raw(j * (L + 1) + l); };  // (j: a compile-time constant)
    auto tile = [&](int j) -> LeanTile<D> {
      LeanTile<D> t;
      t.load(tile_raw(j));
      return t;
    };
    const int mode = vlev == 1 ? int(kStageGlobal) : level_mode;
    const bool tabulated = (l <= Lt);
    const int npass = tabulated ? 1 : a.Niter + 1;  // tabulated levels: only the sampleIndices! pass runs here

    bool screened = false, screen_streamed = false, screen_chunked = false;
    if constexpr (kScreen) {
      if (a.use_screen && vlev != 1) {
        const int smode = scalar_copy(levels[M * (L + 1) + l].stage_mode);
        screened = smode == kStageScreen;
        screen_streamed = smode == kStageScreenStream;
        if constexpr (SCHUNK) screen_chunked = smode == kStageScreenChunked;
      }
    }
#ifdef KDEHIP_SCREEN_STAMPS
    rstamp_on = (vlev >= 300 && l == vlev - 300) && view.block == 3 && wave == 5;
    r_prev_end = 0;
#endif
    if (screened) {
#ifdef KDEHIP_SCREEN_STAMPS
      sstamp_on = (l == ((vlev >= 200) ? vlev - 200 : L)) && view.block == 3 && wave == 5;
#endif
      // the level's M screen tiles are resident in LDS for the whole level: no barrier between steps
      auto screen = [&](int j) -> LevelDesc { return levels[(M + j) * (L + 1) + l]; };  // (j: a compile-time constant)
      staging_barrier();
      static_for<M>([&](auto jc) {
        const LevelDesc sc = screen(decltype(jc)::value);
        stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(reinterpret_cast<const float *>(plan.data) + sc.hdr_off),
                          pool + sc.lds_off, sc.stage_bytes, wave, lane);
      });
      staging_barrier();
      for (int p = 0; p < npass; ++p)
        static_for<M>([&](auto jc) {
          const LeanTile<D> ds = tile(decltype(jc)::value);
          step_screen(jc, ds, ds.lds_off, p == 0, x);  // (on a screened level lds_off is the screen tile's)
        });
    } else if (screen_streamed) {
      // the screen tiles of the level one per step through the two pool halves, like the fp64 tiles of a streamed level:
      // the copy of step t + 1's tile overlaps step t; one barrier per step (a wavefront that has to repeat its step in
      // fp64 is waited for there)
      auto screen = [&](int j) -> LevelDesc { return levels[(M + j) * (L + 1) + l]; };
      auto stage_screen = [&](const LevelDesc &sc, int half, auto wc) {
        constexpr int W = decltype(wc)::value;
        if (wave < W)
          stage_tile<W>(reinterpret_cast<const unsigned char *>(reinterpret_cast<const float *>(plan.data) + sc.hdr_off),
                        pool + half * (kLdsPoolBytes / 2), sc.stage_bytes, wave, lane);
      };
      // (A barrier split into "arrive" and "wait" -- an LDS counter; a wavefront that must repeat its step arrives at the next
      // barrier BEFORE the repeat, so the others go on with the next tile meanwhile -- was built and measured: config 4
      // 3.04 -> 3.07 ms.  A late wavefront stays late, the workgroup waits for it one barrier later; only repeats that
      // overlap in time would be hidden.  profiles/r05_experiments.md section 11.)
      staging_barrier();
      stage_screen(screen(0), 0, IC<WAVES>{});
      int t = 0;
      const int nsteps = npass * M;
      for (int p = 0; p < npass; ++p)
        static_for<M>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int jn = (j + 1 == M) ? 0 : j + 1;
          const LeanTile<D> ds = tile(j);
          const LevelDesc scn = screen(jn);
          staging_barrier();  // tile t has landed for everyone; the other half was last read in step t - 1
          constexpr bool kLateCopy = kCopyWaves < WAVES;
          if (!kLateCopy && t + 1 < nsteps) stage_screen(scn, (t + 1) & 1, IC<WAVES>{});
          // (the cooperative repeat of the chunked levels was tried here too -- posting instead of repeating, resolving behind
          // this barrier: config 4 2.876 -> 2.909 ms: these tiles are 16-32 rows per lane, a solo repeat is short, and the
          // extra copies of the protocol cost registers -- 13 spilled against 2; profiles/r06_experiments.md section 12)
          step_screen(jc, ds, (t & 1) * (kLdsPoolBytes / 2), p == 0, x);
          if (kLateCopy && t + 1 < nsteps) stage_screen(scn, (t + 1) & 1, IC<kCopyWaves>{});
          ++t;
        });
    } else if (screen_chunked) {
      auto screen = [&](int j) -> LevelDesc { return levels[(M + j) * (L + 1) + l]; };
      staging_barrier();
      stage_schunk(screen(0), 0, gchunk & 1);
      int t = 0;
      const int nsteps = npass * M;
      for (int p = 0; p < npass; ++p)
        static_for<M>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int jn = (j + 1 == M) ? 0 : j + 1;
          constexpr int jp = (j + M - 1) % M;
          const LeanTile<D> ds = tile(j);
          step_screen_chunked(jc, ds, tile(jp), screen(j), screen(jn), t + 1 < nsteps, p == 0, t == 0, x);
          ++t;
        });
      if constexpr (kCoop) {  // the requests of the level's last step
        __syncthreads();
        coop_resolve(IC<M - 1>{}, tile(M - 1), (cstep + 1) & 1);
      }
    } else if (mode == kStageResident) {
      staging_barrier();  // every wavefront is done reading the previous level's images
      static_for<M>([&](auto jc) {
        const LeanTile<D> ds = tile(decltype(jc)::value);
        stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + ds.hdr_off()), pool + ds.lds_off, ds.stage_bytes, wave, lane);
      });
      staging_barrier();
      for (int p = 0; p < npass; ++p)
        static_for<M>([&](auto jc) {
          const LeanTile<D> ds = tile(decltype(jc)::value);
          step(jc, ds, (RowPtr)(pool + ds.lds_off), p == 0, x);
        });
    } else if (mode == kStageStream) {
      staging_barrier();
      {
        const LeanTile<D> d0 = tile(0);
        stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + d0.hdr_off()), pool, d0.stage_bytes, wave, lane);
      }
      int t = 0;
      const int nsteps = npass * M;
      for (int p = 0; p < npass; ++p)
        static_for<M>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int jn = (j + 1 == M) ? 0 : j + 1;
          const LeanTile<D> ds = tile(j);
          const LeanTile<D> dn = tile(jn);  // the next step's tile (staged during this step)
          T mean = x, cov = T(0);
          if (p != 0) product(jc, mean, cov);
          const double u = next_uniform();
          // tile t has been copied by all wavefronts once everyone passes this barrier; buffer (t+1)&1 was last
          // read in step t-1, which everyone has left -> start the next copy
          staging_barrier();
          // The copy of the next tile: at the END of this step by the older wavefronts (kCopyWaves) -- they finish first and
          // would idle at the next barrier -- instead of by everyone at its start, where it delayed every wavefront's rows
          // (config 3: 0.5726 -> 0.5569 ms).  With one wavefront per SIMD: at the start, as before.
          constexpr bool kLateCopy = kCopyWaves < WAVES;
          if (!kLateCopy && t + 1 < nsteps)
            stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + dn.hdr_off()),
                              pool + ((t + 1) & 1) * (kLdsPoolBytes / 2), dn.stage_bytes, wave, lane);
          {
            auto hdr = (RowPtr)(pool + (t & 1) * (kLdsPoolBytes / 2));
            auto rows = hdr + kTileHeader;
            // (requesting the first row ahead of the broadcasts here, as the resident steps do, was measured: no gain --
            // 0.5825 vs 0.5811 ms)
            const int pos = __builtin_amdgcn_readfirstlane(draw(ds, hdr, mean, cov, [&](const auto &ev) {
              return draw_rows(ds, rows, ev, u);
            }));
            adopt(jc, ds, hdr, pos);
          }
          if (kLateCopy && t + 1 < nsteps && wave < kCopyWaves)
            stage_tile<kCopyWaves>(reinterpret_cast<const unsigned char *>(data + dn.hdr_off()),
                                   pool + ((t + 1) & 1) * (kLdsPoolBytes / 2), dn.stage_bytes, wave, lane);
          ++t;
        });
    } else if (mode == kStageChunked) {
      // tiles larger than half the pool: pass 1 streams the rows through the two pool halves (one barrier per
      // chunk, the copy of chunk g+1 overlaps the evaluation of chunk g); the second pass and the new kernel are
      // read from global memory.
      staging_barrier();
      stage_chunk(tile(0), 0, gchunk & 1);
      int t = 0;
      const int nsteps = npass * M;
      for (int p = 0; p < npass; ++p)
        static_for<M>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int jn = (j + 1 == M) ? 0 : j + 1;
          const LeanTile<D> ds = tile(j);
          const LeanTile<D> dn = tile(jn);  // the next step's tile (its first chunk is staged during this step's last)
          T mean = x, cov = T(0);
          if (p != 0) product(jc, mean, cov);
          const double u = next_uniform();
          const T *hdr = data + ds.hdr_off();
          const int pos = draw(ds, hdr, mean, cov, [&](const auto &ev) {
            using Ev = std::decay_t<decltype(ev)>;
            const int RS = TileAddr<T>::stride(ds.F), rc = ds.chunk_rows;
            LaneAcc<T> acc;
            SegSums<T> seg;
            const int cps = seg_chunks(ds.seg);
            const bool use_seg = cps != 0;
            int cin = 0;
            KDEHIP_PRIO_ROWS();
            for (int r0 = 0; r0 < ds.B; r0 += rc, ++gchunk) {
              staging_barrier();  // this chunk has landed for every wavefront; the other half is free again
              if (r0 + rc < ds.B) stage_chunk(ds, r0 + rc, (gchunk + 1) & 1);
              else if (t + 1 < nsteps) stage_chunk(dn, 0, (gchunk + 1) & 1);
              const int nrows = (ds.B - r0 < rc) ? (ds.B - r0) : rc;
              const auto crows = (RowPtr)(pool + (gchunk & 1) * (kLdsPoolBytes / 2));
              lane_rows_all<T, RowPtr, Ev, kPrefetchRows>(crows, nrows, RS, lane, ev, acc);
              // the lane's running sum at a segment boundary
              if (use_seg && ++cin == cps) { seg.note(acc.total()); cin = 0; }
            }
            KDEHIP_PRIO_CHAIN();
            const T S = acc.total();
            if (use_seg)
              return __builtin_amdgcn_readfirstlane(
                  select_or_raise_seg<T, const T *>(S, seg, cps * rc, hdr + kTileHeader, ds, lane, ev, u, fb));
            return __builtin_amdgcn_readfirstlane(select_or_raise<T, const T *>(S, hdr + kTileHeader, ds, lane, ev, u, fb));
          });
          adopt(jc, ds, hdr, pos);
          ++t;
        });
    } else {  // kStageGlobal
      for (int p = 0; p < npass; ++p)
        static_for<M>([&](auto jc) {
          const LeanTile<D> ds = tile(decltype(jc)::value);
          step(jc, ds, data + ds.hdr_off(), p == 0, x);
        });
    }

    if (tabulated) {
      // ---- tabulated sweeps (see gibbs_kernel.hip "conditional tables"): the labels of all densities packed in
      // one scalar word, one table row load per step, the unchanged selection ----
      TabDesc td[M];
      static_for<M>([&](auto jc) { td[decltype(jc)::value] = tabs[decltype(jc)::value * (L + 1) + l]; });
      uint32_t word = 0;
      static_for<M>([&](auto jc) { word |= static_cast<uint32_t>(psel[decltype(jc)::value]) << td[decltype(jc)::value].shift; });
      word = __builtin_amdgcn_readfirstlane(word);
      for (int p = 1; p <= a.Niter; ++p)
        static_for<M>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const TabDesc &tj = td[j];
          const uint32_t cfg = (word & ((1u << tj.shift) - 1u)) | ((word >> (tj.shift + tj.bits)) << tj.shift);
          const T *row = tables + tj.off + static_cast<int64_t>(cfg) * (tj.n + 1);
          const int n = tj.n;
          const T incl = row[lane < n ? lane : n];
          const double u = next_uniform();
          const T total = lane_read(incl, n < 64 ? n : 63);  // (a 64-node row: its last scan value IS the total)
          int pos;
          if (!(total >= Num<T>::tiny_total())) {  // uniform fallback (:311-315), rare
            count_fallback(fb, lane);
            const LeanTile<D> dk = tile(j);
            const T wl = ((LdsPtr<T>)(pool + dk.lds_off) + kTileHeader)[(dk.F - 1) * TileAddr<T>::kField + (n - 1) * TileAddr<T>::kLane];
            int z = n - 1;
            if (wl > T(0)) {
              z = static_cast<int>(ceil(u * static_cast<double>(n))) - 1;
              z = z < 0 ? 0 : (z > n - 1 ? n - 1 : z);
            }
            pos = __builtin_amdgcn_readfirstlane(z);
          } else {
            const T target = static_cast<T>(u) * total;
            unsigned long long hit = __ballot(target <= incl);
            if (n < 64) hit &= (1ull << n) - 1ull;
            pos = hit ? (__ffsll(hit) - 1) : (n - 1);
          }
          word = (word & ~(((1u << tj.bits) - 1u) << tj.shift)) | (static_cast<uint32_t>(pos) << tj.shift);
        });
      // the level's sweeps are over: unpack the labels and adopt the selected kernels for what follows
      static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int pk = static_cast<int>((word >> td[j].shift) & ((1u << td[j].bits) - 1u));
        const LeanTile<D> dk = tile(j);
        adopt(jc, dk, (LdsPtr<T>)(pool + dk.lds_off), pk);
      });
    }
    if (a.labels && live && lane == 0) {
      static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        a.labels[(s * M + j) * L + (l - 1)] = plan.perm[levels[j * (L + 1) + l].perm_off + psel[j]];
      });
    }
    if (l == L && live && lane == 0) {  // final labels (:612-616)
      static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int64_t label = static_cast<int64_t>(plan.perm[levels[j * (L + 1) + l].perm_off + psel[j]]) + 1;
        a.indices[s * M + j] = label;
        for (int k = 0; k < a.npeers; ++k) a.peer_indices[k][s * M + j] = label;  // (multi-GPU: the all-gather)
      });
    }
  }

#ifdef KDEHIP_SCREEN_STAMPS
  if (view.block == 3 && wave == 5 && lane == 0)
    for (int k = 0; k < 8; ++k) { g_screen_stamps[k] = sstamp[k]; g_screen_stamps[8 + k] = rstamp[k]; }
#endif
  if constexpr (kScreen) {  // diagnostic counters in front of the level table: [-3] screened steps, [-2] repeated in fp64
    if (live && lane == 0 && n_screened != 0u) {
      unsigned long long *cnt = reinterpret_cast<unsigned long long *>(const_cast<LevelDesc *>(plan.levels));
      atomicAdd(cnt - 3, static_cast<unsigned long long>(n_screened));
      if (n_repeated != 0u) atomicAdd(cnt - 2, static_cast<unsigned long long>(n_repeated));
    }
  }
  {  // final point (:625)
    T mean, cov;
    product(IC<-1>{}, mean, cov);
    T xf = mean;
    if (a.addEntropy) xf = mean + Num<T>::sqrt(cov) * static_cast<T>(sNorm[L * D + dl]);
    if (live && lane < D) {
      a.points[s * D + lane] = static_cast<double>(xf);
      for (int k = 0; k < a.npeers; ++k) a.peer_points[k][s * D + lane] = static_cast<double>(xf);
    }
  }
}

// ---- launcher --------------------------------------------------------------------------------------

template <typename T, int D, int M, int WAVES>
static void launch_lean_waves(const PlanDev &plan, const RunArgs &args, hipStream_t stream) {
  const int64_t blocks = (args.Np + WAVES - 1) / WAVES;
  if constexpr (sizeof(T) == 8 && WAVES >= 8) {  // (4 chains per workgroup: small runs; chunked screen levels run in fp64 there)
    // 8 densities: the chunked-screen build is the only one (BASELINE config 4 needs it; a second build of these, the largest
    // kernels of the library, would be another 1.6 minutes of build time for plans of 8 small densities)
    if (M == 8 || (plan.screened == 2 && args.use_screen)) {
      hipLaunchKernelGGL((gibbs_lean_kernel<T, D, M, WAVES, false, true>), dim3(static_cast<unsigned>(blocks)), dim3(WAVES * 64),
                         0, stream, plan, args);
      return;
    }
  }
  if constexpr (!(sizeof(T) == 8 && WAVES >= 8 && M == 8))
    hipLaunchKernelGGL((gibbs_lean_kernel<T, D, M, WAVES>), dim3(static_cast<unsigned>(blocks)), dim3(WAVES * 64), 0,
                       stream, plan, args);
}

// wavefronts (= chains) per workgroup of this run
static int set_geometry(const PlanDev &, RunArgs &args, int) { return lean_waves(args.Np, args.variant); }

template <typename T, int D, int M>
static int launch_lean_m(const PlanDev &plan, const RunArgs &args_in, hipStream_t stream) {
  RunArgs args = args_in;
  const int waves = set_geometry(plan, args, sizeof(T) == 8 ? 64 : 32);
  if (waves == 16) launch_lean_waves<T, D, M, 16>(plan, args, stream);
  else if (waves == 8) launch_lean_waves<T, D, M, 8>(plan, args, stream);
  else launch_lean_waves<T, D, M, 4>(plan, args, stream);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return set_error(KDEHIP_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

#ifndef KDEHIP_DIM
#error "compile gibbs_lean.hip with -DKDEHIP_DIM=<1..8>"
#endif
#define KDEHIP_CAT2(a, b) a##b
#define KDEHIP_CAT(a, b) KDEHIP_CAT2(a, b)

template <typename T, int D, int M>
static int launch_lean_m_hi(const PlanDev &plan, const RunArgs &args_in, hipStream_t stream) {
  RunArgs args = args_in;
  const int waves = set_geometry(plan, args, sizeof(T) == 8 ? 64 : 32);
  if (waves == 16) launch_lean_waves<T, D, M, 16>(plan, args, stream);
  else if (waves == 8) launch_lean_waves<T, D, M, 8>(plan, args, stream);
  else return kLeanNotCovered;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return set_error(KDEHIP_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

#if !defined(KDEHIP_LEAN_HI) && !defined(KDEHIP_LEAN_F32) && !defined(KDEHIP_LEAN_DEV)
// kdehip_prod_philox_batch: one launch for a group of fp64 products of M densities (gibbs_dispatch.cpp launch_gibbs_batch)
template <int D, int M>
static void launch_lean_batch_m(const PlanDev &plan, const RunArgs &args, hipStream_t stream) {
  constexpr int W = 16;
  hipLaunchKernelGGL((gibbs_lean_kernel<double, D, M, W, true>), dim3(static_cast<unsigned>(args.Np / W)), dim3(W * 64), 0,
                     stream, plan, args);
}
int KDEHIP_CAT(launch_lean_batch_d, KDEHIP_DIM)(int M, const PlanDev &plan, const RunArgs &args, void *stream) {
  constexpr int D = KDEHIP_DIM;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (args.Np <= 0) return KDEHIP_OK;
  switch (M) {
    case 2: launch_lean_batch_m<D, 2>(plan, args, st); break;
    case 3: launch_lean_batch_m<D, 3>(plan, args, st); break;
    case 4: launch_lean_batch_m<D, 4>(plan, args, st); break;
    default: return set_error(KDEHIP_ERR_UNSUPPORTED, "batched launch: 2..4 densities");
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("batched kernel launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}
#endif

#if defined(KDEHIP_LEAN_DEV) && !defined(KDEHIP_LEAN_HI) && !defined(KDEHIP_LEAN_F32)
// (a development build replaces this translation unit with ONE instantiation: no batched kernels)
int KDEHIP_CAT(launch_lean_batch_d, KDEHIP_DIM)(int, const PlanDev &, const RunArgs &, void *) {
  return set_error(KDEHIP_ERR_UNSUPPORTED, "development library: no batched kernels");
}
#endif

#if defined(KDEHIP_SCREEN_STAMPS)
extern "C" int kdehip_debug_read_screen_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kdehip::g_screen_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -5;
}
#endif

#if defined(KDEHIP_LEAN_HI)
#define KDEHIP_LEAN_ENTRY launch_lean_hi_d
#elif defined(KDEHIP_LEAN_F32)
#define KDEHIP_LEAN_ENTRY launch_lean_f32_d
#else
#define KDEHIP_LEAN_ENTRY launch_lean_d
#endif

// Returns kLeanNotCovered when the run is outside this kernel's domain (the caller then uses gibbs_kernel.hip).
int KDEHIP_CAT(KDEHIP_LEAN_ENTRY, KDEHIP_DIM)(int precision, int mode, const PlanDev &plan, const RunArgs &args,
                                              void *stream) {
  constexpr int D = KDEHIP_DIM;
  if (mode != kModeFast || args.table_build || plan.M < kLeanMinDens || plan.M > kLeanMaxDens ||
      D * (plan.L + 1) > kLeanMaxNormals)
    return kLeanNotCovered;
  if (args.Np <= 0) return KDEHIP_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool f64 = (precision == 64);
#ifdef KDEHIP_LEAN_DEV  // development builds (scripts/dev_lean.sh): ONE instantiation, compiles in seconds
#ifndef KDEHIP_LEAN_DEV_M
#define KDEHIP_LEAN_DEV_M 4
#endif
#ifndef KDEHIP_LEAN_DEV_W
#define KDEHIP_LEAN_DEV_W 8
#endif
#ifdef KDEHIP_LEAN_DEV_F32
  using DevT = float;
#else
  using DevT = double;
#endif
  RunArgs dargs = args;
  if (f64 != (sizeof(DevT) == 8) || plan.M != KDEHIP_LEAN_DEV_M ||
      set_geometry(plan, dargs, precision) != KDEHIP_LEAN_DEV_W)
    return kLeanNotCovered;
  launch_lean_waves<DevT, D, KDEHIP_LEAN_DEV_M, KDEHIP_LEAN_DEV_W>(plan, dargs, st);
  return KDEHIP_OK;
#elif defined(KDEHIP_LEAN_HI)
  if (!f64) return kLeanNotCovered;  // (fp32 products of more than 4 densities run the general kernel: build time)
  return launch_lean_m_hi<double, D, 8>(plan, args, st);
#elif defined(KDEHIP_LEAN_F32)
  if (f64) return kLeanNotCovered;
  switch (plan.M) {
    case 2: return launch_lean_m<float, D, 2>(plan, args, st);
    case 3: return launch_lean_m<float, D, 3>(plan, args, st);
    default: return launch_lean_m<float, D, 4>(plan, args, st);
  }
#else
  if (!f64) return kLeanNotCovered;
  switch (plan.M) {
    case 2: return launch_lean_m<double, D, 2>(plan, args, st);
    case 3: return launch_lean_m<double, D, 3>(plan, args, st);
    default: return launch_lean_m<double, D, 4>(plan, args, st);
  }
#endif
}

}  // namespace kdehip
