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
//     4, 8 or 16 chains (picked per launch, see launch_one) share a workgroup, one workgroup per CU,
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
//   FAST     p = w * rsqrt(prod_d c_d) * exp(-1/2 * sum_d delta_d^2 / c_d), the sum formed as ONE fraction
//            over prod_d c_d (pairwise addition of fractions) and ONE rsqrt -- no divide, no log;
//   GENERIC  the reference's own per-dimension divide + log with its NaN rules (:287-303), used for
//            inputs whose variance products could leave the range of T or are not finite/positive.
// partialDimMask products (and one-density "products") run UNIFORM/FAST with the inactive dimensions
// contributing c = 1, delta = 0 (MODE = fast-masked).
// fp64 exp on the fast forms is a 32-entry-table (256 bytes of LDS) + degree-6
// polynomial, ~1 ulp; the GENERIC form calls the library exp/log.
//
// Compiled with -ffp-contract=off; fused multiply-adds are written explicitly where wanted.
#include "gibbs_device.hpp"

namespace kdehip {

// TBL: the instantiation that only fills the conditional tables (a.table_build) -- the same code, but without the
// 120 KiB tile pool in its LDS footprint, so that a CU holds many more of its wavefronts (one wavefront per table row,
// ~19,000 rows at config 3: 45 us with the sampler's one-workgroup-per-CU footprint, every one-shot call pays it).
// BATCH (with TBL only): the tables of MANY products in one launch (kdehip_prod_philox_batch) -- workgroup b fills rows of
// product batch_map[b], whose plan it fetches through the scalar cache (LaunchView, gibbs_device.hpp).
template <typename T, int D, int MODE, int WAVES, bool TBL = false, bool BATCH = false>
__global__ __launch_bounds__(WAVES * 64) void gibbs_product_kernel(PlanDev plan_, RunArgs a_) {
  static_assert(!BATCH || TBL, "batched launches of this kernel fill tables only");
  const LaunchView<BATCH> view(plan_, a_);
  const PlanDev &plan = view.plan;
  const auto &a = view.a;
  constexpr bool FAST = (MODE != kModeGeneric);      // product/rsqrt + uniform-bandwidth forms
  constexpr bool MASKED = (MODE == kModeFastMasked);  // ... with inactive dimensions
  constexpr bool kAllDimsOn = (MODE == kModeFast);    // the plan checked it: no mask tests in this build
  // pass 1 prefetches the next row's fields while it evaluates the current one; the 16-wavefront fp64
  // builds have 128 VGPRs and would spill from D = 6 on
  constexpr bool kPrefetchRows = (WAVES <= 12) || sizeof(T) == 4 || D <= 4;
#ifdef KDEHIP_X_ALLCOPY  // (A/B only: rounds 1-3)
  constexpr int kCopyWaves = WAVES;
#else
  constexpr int kCopyWaves = WAVES > 4 ? 4 : WAVES;  // who issues the copies of streamed tiles and chunks (gibbs_lean.hip)
#endif
  using Lay = LdsLayout<T, D, WAVES>;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[TBL ? Lay::kPoolOff : Lay::kBytes];

  double *sExpTab = reinterpret_cast<double *>(smem);
  if (threadIdx.x < 32) sExpTab[threadIdx.x] = kExp2Tab[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  // readfirstlane makes the wave id (and everything derived from it: sample index, RNG counters,
  // descriptor addresses) provably wave-uniform, so it lives in SGPRs / runs on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  int64_t s = static_cast<int64_t>(view.block) * WAVES + wave;
  // surplus wavefronts of the last workgroup keep taking part in staging and barriers: they replay
  // the last chain and store nothing
  const bool live = s < a.Np;
  if (!live) s = a.Np - 1;
  const void *fb = live ? static_cast<const void *>(plan.levels) : nullptr;  // who counts uniform fallbacks (not the replays)
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

  // variant: 0 default; 1 = read every tile from global memory (no LDS staging); 2 / 8 / 16 = 4 / 8 / 16 chains per workgroup.
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
    using TA = TileAddr<T>;
    auto e = hdr + kTileHeader + TA::row(pos >> 6, TA::stride(ds.F)) + (pos & 63) * TA::kLane;
    const T mu = e[dl * TA::kField];
    const T var = ds.uniform_bw ? hdr[dl] : e[(D + dl) * TA::kField];
    const bool on = kAllDimsOn || ((ds.mask_bits >> dl) & 1u);
    const T l = on ? (FAST ? fast_rcp(var) : T(1) / var) : T(0);
    if (lane < D) {
      lam[j * D + dl] = l;
      // (a circular dimension keeps the angle itself: its getMu works on the angles, product_dim below)
      lmu[j * D + dl] = on ? ((!FAST && ((plan.circ_bits >> dl) & 1u)) ? mu : mu * l) : T(0);
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
    if constexpr (!FAST) {
      // The enumerated circular operators (include/kdehip.h "manifolds") at the reference's hooks getLambda / getMu
      // (:183-184, applied :210-213): getLambda = the same sum; getMu = the information-weighted mean in the tangent space at
      // the FIRST contributing kernel's angle, mapped back -- the sums in density order, as oracle/kde_oracle.c forms them.
      if (plan.circ_bits != 0u) {  // (wave-uniform; plans without a circular dimension never come here)
        const bool circ = (plan.circ_bits >> dl) & 1u;
        T ls = T(0), ref = T(0);
        bool have = false;
        for (int k = 0; k < M; ++k) {
          const T l = lam[k * D + dl];
          ls += l;
          if (!have && l > T(0)) { ref = lmu[k * D + dl]; have = true; }
        }
        T acc = T(0);
        for (int k = 0; k < M; ++k) {
          const T l = lam[k * D + dl], m = lmu[k * D + dl];
          acc += circ ? l * circ_wrap(m - ref) : m;  // (Euclidean slots hold mean * lambda already)
        }
        const bool on = kAllDimsOn || ((info_bits >> dl) & 1u);
        cov = on ? T(1) / ls : T(0);
        mean = on ? (circ ? circ_wrap(ref + cov * acc) : cov * acc) : T(0);
        return;
      }
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
        bool on = true;
        if constexpr (MASKED) {  // an inactive dimension contributes nothing: c = 1, weight 0
          on = (act >> dl) & 1u;
          c = on ? c : T(1);
        }
        T cen, nin;
        EvalUniform<T, D>::operands(mean, c, on, cen, nin);
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
      ev.circ = plan.circ_bits;  // (diffop of the circular dimensions, :290)
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
      return draw_label<T, P, kPrefetchRows, (WAVES <= 8)>(rows, ds, lane, ev, u, fb KSTAMP_ARGS);
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
    using TA = TileAddr<T>;
    const int RS = TA::stride(ds.F), rc = chunk_rows(ds);
    const int nrows = (ds.B - r0 < rc) ? (ds.B - r0) : rc;
    const int bytes = (static_cast<int>(TA::span(nrows, RS)) * int(sizeof(T)) + 1023) & ~1023;  // (r0: a multiple of 4 rows)
    // (issued by the older wavefronts of every SIMD only, as in gibbs_lean.hip: kCopyWaves)
    if (wave < kCopyWaves)
      stage_tile<kCopyWaves>(reinterpret_cast<const unsigned char *>(data + ds.hdr_off + kTileHeader + TA::row(r0, RS)),
                             pool + half * (kLdsPoolBytes / 2), bytes, wave, lane);
  };
  auto step_chunked = [&](int j, const LevelDesc &ds, const LevelDesc &dn, bool has_next, T mean, T cov, double u) {
    const T *hdr = data + ds.hdr_off;
    const int pos = draw(ds, hdr, mean, cov, [&](const auto &ev) {
      const int RS = TileAddr<T>::stride(ds.F), rc = chunk_rows(ds);
      LaneAcc<T> acc;  // the lane's sums over its rows, kept across the chunks (gibbs_device.hpp: one association everywhere)
      SegSums<T> seg;
      const int cps = seg_chunks(ds.seg);
      const bool use_seg = cps != 0;
      int cin = 0;
      KDEHIP_PRIO_ROWS();
      for (int r0 = 0; r0 < ds.B; r0 += rc, ++gchunk) {
        staging_barrier();  // this chunk has landed for every wavefront; the other half is free again
        if (r0 + rc < ds.B) stage_chunk(ds, r0 + rc, (gchunk + 1) & 1);
        else if (has_next) stage_chunk(dn, 0, (gchunk + 1) & 1);
        const int nrows = (ds.B - r0 < rc) ? (ds.B - r0) : rc;
        lane_rows_all<T, LdsPtr<T>, std::decay_t<decltype(ev)>, kPrefetchRows>(
            (LdsPtr<T>)(pool + (gchunk & 1) * (kLdsPoolBytes / 2)), nrows, RS, lane, ev, acc);
        if (use_seg && ++cin == cps) { seg.note(acc.total()); cin = 0; }  // the lane's running sum at a segment boundary
      }
      KDEHIP_PRIO_CHAIN();
      const T S = acc.total();
      // (a raised repeat of the evaluation reads the tile from global memory: no staging, no barriers)
      if (use_seg)
        return select_or_raise_seg<T, const T *>(S, seg, cps * rc, hdr + kTileHeader, ds, lane, ev, u, fb KSTAMP_ARGS);
      return select_or_raise<T, const T *>(S, hdr + kTileHeader, ds, lane, ev, u, fb KSTAMP_ARGS);
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
  if (TBL || a.table_build) {
    const int64_t g = static_cast<int64_t>(view.block) * WAVES + wave;
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
          hdr + kTileHeader, 1, TileAddr<T>::stride(ds.F), lane, ev));
      // fp32: the scan of the first exponent offset at which the sum is large enough (see Num<float>::tiny_total);
      // an underflow in the reference's sense is stored as an all-zero row, which the sweep step takes as the
      // uniform fallback
      if constexpr (Num<T>::kOffsetSteps > 0) {
        bool ok = lane_read(incl, 63) >= Num<T>::tiny_total();
        for (int k = 1; k <= Num<T>::kOffsetSteps && !ok; ++k) {
          const auto evo = ev.with_offset(T(Num<T>::kOffsetStep) * T(k));
          incl = wave_inclusive_scan(lane_sum_rows<T, const T *, std::decay_t<decltype(evo)>, false>(
              hdr + kTileHeader, 1, TileAddr<T>::stride(ds.F), lane, evo));
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
  if constexpr (!TBL) {
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
      if constexpr (!FAST) {
        if ((plan.circ_bits >> dl) & 1u) x = circ_wrap(x);  // addop of a circular dimension (:456)
      }
    }
    const int mode = vlev == 1 ? int(kStageGlobal) : levels[l].stage_mode;

    if (mode == kStageResident) {
      staging_barrier();  // every wavefront is done reading the previous level's images
      for (int j = 0; j < M; ++j) {
        const LevelDesc ds = levels[j * (L + 1) + l];
        stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + ds.hdr_off), pool + ds.lds_off,
                   ds.stage_bytes, wave, lane);
      }
      staging_barrier();  // (waits for this wavefront's copies, then for everyone's)
    } else if (mode == kStageStream) {
      staging_barrier();
      const LevelDesc ds0 = levels[l];
      stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + ds0.hdr_off), pool, ds0.stage_bytes, wave, lane);
    } else if (mode == kStageChunked) {
      staging_barrier();
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
          staging_barrier();
          KSTAMP(tb1);
          KSTAMP_ADD(6, tb0, tb1);
          // (the next tile's copy: at the end of the step by the older wavefronts, gibbs_lean.hip "kLateCopy"; with one
          // wavefront per SIMD at its start by everyone)
          constexpr bool kLateCopy = kCopyWaves < WAVES;
          if (!kLateCopy && t + 1 < nsteps)
            stage_tile<WAVES>(reinterpret_cast<const unsigned char *>(data + ds_next.hdr_off),
                       pool + ((t + 1) & 1) * (kLdsPoolBytes / 2), ds_next.stage_bytes, wave, lane);
          step(j, ds, (LdsPtr<T>)(pool + (t & 1) * (kLdsPoolBytes / 2)), mean, cov, u);
          if (kLateCopy && t + 1 < nsteps && wave < kCopyWaves)
            stage_tile<kCopyWaves>(reinterpret_cast<const unsigned char *>(data + ds_next.hdr_off),
                                   pool + ((t + 1) & 1) * (kLdsPoolBytes / 2), ds_next.stage_bytes, wave, lane);
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
          count_fallback(fb, lane);
          const LevelDesc dk = levels[jt * (L + 1) + l];
          const T wl = ((LdsPtr<T>)(pool + dk.lds_off) + kTileHeader)[(dk.F - 1) * TileAddr<T>::kField + (n - 1) * TileAddr<T>::kLane];
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
      const int64_t label = static_cast<int64_t>(plan.perm[ds.perm_off + psel[k]]) + 1;
      a.indices[s * M + k] = label;
      for (int q = 0; q < a.npeers; ++q) a.peer_indices[q][s * M + k] = label;  // (multi-GPU: the all-gather)
    }
  }
  {
    T mean, cov;
    product_dim(-1, any_bits, mean, cov);
    T xf = mean;
    if (a.addEntropy) {
      xf = mean + Num<T>::sqrt(cov) * static_cast<T>(normal_for_lane(L));
      if constexpr (!FAST) {
        if ((plan.circ_bits >> dl) & 1u) xf = circ_wrap(xf);
      }
    }
    if (live && lane < D) {
      a.points[s * D + lane] = static_cast<double>(xf);
      for (int q = 0; q < a.npeers; ++q) a.peer_points[q][s * D + lane] = static_cast<double>(xf);
    }
  }
  }  // !TBL
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
  if (args.table_build) {  // the table-only instantiation: 4 wavefronts per workgroup, no tile pool
    if constexpr (MODE != kModeGeneric) {
      constexpr int TW = 4;
      const int64_t blocks = (args.Np + TW - 1) / TW;
      hipLaunchKernelGGL((gibbs_product_kernel<T, D, MODE, TW, true>), dim3(static_cast<unsigned>(blocks)), dim3(TW * 64), 0,
                         stream, plan, args);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess)
        return set_error(KDEHIP_ERR_HIP, std::string("table build launch failed: ") + hipGetErrorString(e));
      return KDEHIP_OK;
    }
  }
  const int waves = chains_per_workgroup(args.Np, args.variant);
  if (waves == 16) launch_waves<T, D, MODE, 16>(plan, args, stream);
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

// kdehip_prod_philox_batch: the conditional tables of a group of fp64 products in one launch; args.Np = workgroups x 4
int KDEHIP_CAT(launch_tables_batch_d, KDEHIP_DIM)(const PlanDev &plan, const RunArgs &args, void *stream) {
  constexpr int TW = 4;
  if (args.Np <= 0) return KDEHIP_OK;
  hipLaunchKernelGGL((gibbs_product_kernel<double, KDEHIP_DIM, kModeFast, TW, true, true>), dim3(static_cast<unsigned>(args.Np / TW)),
                     dim3(TW * 64), 0, static_cast<hipStream_t>(stream), plan, args);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("batched table build launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

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
