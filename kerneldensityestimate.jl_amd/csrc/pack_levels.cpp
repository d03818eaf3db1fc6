// pack_levels.cpp -- host re-layout of BallTreeDensity inputs into per-level tiles (product code).
//
// The reference builds the frontier lists on the fly for every output sample (levelInit!,
// levelDown!, src/MSGibbs01.jl:467-475,500-523).  They depend only on the trees, so here they are
// expanded once per product and stored level by level in the lane-blocked row/field layout
// described in kdehip_internal.hpp; the kernels never touch the tree topology.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "device_density.hpp"
#include "host_pool.hpp"
#include "kdehip_internal.hpp"

namespace kdehip {

namespace {
thread_local std::string g_err;
}

int set_error(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
const char *last_error_cstr() { return g_err.c_str(); }

int nlevels_for(int64_t maxNp) {
  // floor(Int, log(maxNp)/log(2) + 1), src/MSGibbs01.jl:568 (log/log, not log2, on purpose)
  return static_cast<int>(std::floor(std::log(static_cast<double>(maxNp)) / std::log(2.0) + 1.0));
}

// The product/rsqrt form multiplies up to D variances c_d in [bw_lo, 2*bw_hi] (bandwidth plus a
// leave-one-out product variance that is never larger than the largest bandwidth).  It is used
// only when no partial product can leave the comfortable range of T; otherwise the per-dimension
// divide+log form (the reference's own arithmetic) runs.
bool variances_in_range(const double *bw_lo, const double *bw_hi, int D, int precision) {
  double up = 1.0, dn = 1.0;
  for (int d = 0; d < D; ++d) {
    const double hi = 2.0 * bw_hi[d], lo = bw_lo[d];
    if (hi > 1.0) up *= hi;
    if (lo < 1.0) dn *= lo;
  }
  return (precision == 64) ? (up < 1e120 && dn > 1e-120) : (up < 1e15 && dn > 1e-15);
}

// The frontiers of one density, levels 0..L (levelInit! / levelDown!, src/MSGibbs01.jl:467-475, 503-511): the node ids
// only -- they depend on the child arrays alone -- with `fresh` marking the level at which a node ENTERS the frontier
// (a leaf re-enters the next frontier as its own left child: not new).
int expand_frontier_ids(const kdehip_density &t, int D, int L, Frontiers &out) {
  (void)D;
  const int64_t N = t.npts;
  out.ids.clear();
  out.fresh.clear();
  out.ids.reserve(static_cast<size_t>(N) * (L + 1) / 2 + 64);
  out.fresh.reserve(static_cast<size_t>(N) * (L + 1) / 2 + 64);
  out.off.assign(static_cast<size_t>(L) + 2, 0);
  out.nodes = 0;
  for (int l = 0; l <= L; ++l) {
    const size_t begin = out.ids.size();
    out.off[l] = static_cast<int64_t>(begin);
    if (l == 0) {
      out.ids.push_back(1);  // levelInit!: frontier = {root()}
      out.fresh.push_back(1);
    } else {
      const size_t pb = static_cast<size_t>(out.off[l - 1]), pe = begin;
      out.ids.resize(begin + 2 * (pe - pb));  // (at most two children per node; trimmed below)
      out.fresh.resize(begin + 2 * (pe - pb));
      int32_t *dst = out.ids.data() + begin;
      uint8_t *fr = out.fresh.data() + begin;
      const int32_t *src = out.ids.data() + pb;
      size_t cnt = 0;
      for (size_t z = 0; z < pe - pb; ++z) {
        const int64_t node = src[z];
        const int64_t a = t.left_child[node - 1], b = t.right_child[node - 1];
        if (a > 0 && a <= 2 * N) {  // validIndex, BallTree01.jl:83
          fr[cnt] = a != node;
          dst[cnt++] = static_cast<int32_t>(a);
        }
        if (b > 0 && b <= 2 * N) {
          fr[cnt] = b != node;
          dst[cnt++] = static_cast<int32_t>(b);
        }
      }
      out.ids.resize(begin + cnt);
      out.fresh.resize(begin + cnt);
      const int64_t n = static_cast<int64_t>(cnt);
      if (n == 0 || n > N) return set_error(KDEHIP_ERR_ARG, "malformed tree: frontier empty or larger than Npts");
      out.nodes += n;
    }
  }
  out.off[static_cast<size_t>(L) + 1] = static_cast<int64_t>(out.ids.size());
  return KDEHIP_OK;
}

// What the VALUES of the frontier nodes say (needs the ids above): whether a level shares one bandwidth vector, and with
// `look` every node examined ONCE, at the level where it enters the frontier -- finiteness and the bandwidth range of the
// arithmetic-form decision.
void examine_frontiers(const kdehip_density &t, int D, int L, bool look, Frontiers &out) {
  out.uniform.assign(static_cast<size_t>(L) + 1, 1);
  out.uratio.assign(static_cast<size_t>(L) + 1, 0.0);
  bool bad = false;
  double lo[KDEHIP_MAX_DIMS], hi[KDEHIP_MAX_DIMS];
  for (int d = 0; d < KDEHIP_MAX_DIMS; ++d) { lo[d] = INFINITY; hi[d] = 0.0; }
  auto look_at = [&](int64_t node) {  // (branch-free: minima / maxima / one sticky flag in locals)
    const double *mu = t.means + (node - 1) * D, *v = t.bandwidth + (node - 1) * D;
    for (int d = 0; d < D; ++d) {
      // (means beyond 1e100 would overflow the squared distances of the product/rsqrt forms)
      bad |= !(std::fabs(mu[d]) < 1e100) | !(v[d] > 0.0) | !(v[d] < INFINITY);
      lo[d] = v[d] < lo[d] ? v[d] : lo[d];
      hi[d] = v[d] > hi[d] ? v[d] : hi[d];
    }
    const double w = t.weights[node - 1];
    bad |= !(w >= 0.0) | !(w < INFINITY);
  };
  for (int l = 0; l <= L; ++l) {
    const size_t begin = static_cast<size_t>(out.off[l]), end = static_cast<size_t>(out.off[l + 1]);
    if (look)
      for (size_t z = begin; z < end; ++z)
        if (out.fresh[z]) look_at(out.ids[z]);
    // one bandwidth vector shared by the whole frontier?  (stops at the first node that differs: frontiers with
    // internal nodes are decided after a node or two, only the all-leaf frontiers are scanned in full)
    const double *bw0 = t.bandwidth + (static_cast<int64_t>(out.ids[begin]) - 1) * D;
    bool uni = true;
    for (size_t z = begin + 1; z < end && uni; ++z) {
      const double *v = t.bandwidth + (static_cast<int64_t>(out.ids[z]) - 1) * D;
      for (int d = 0; d < D; ++d)
        if (v[d] != bw0[d]) uni = false;
    }
    out.uniform[l] = uni ? 1 : 0;
    if (uni) {  // (all-leaf frontiers: scanned in full above anyway)
      // max over nodes and dimensions of |mean_d| / sqrt(2 bandwidth_d): the division by a positive constant is monotone,
      // so the largest quotient is the quotient of the largest |mean_d| -- one square root and one division per
      // DIMENSION instead of per node (they were 100 us of a 2048-point density's 110)
      double amax[KDEHIP_MAX_DIMS];
      for (int d = 0; d < D; ++d) amax[d] = 0.0;
      for (size_t z = begin; z < end; ++z) {
        const double *mu = t.means + (static_cast<int64_t>(out.ids[z]) - 1) * D;
        for (int d = 0; d < D; ++d) {
          const double a = std::fabs(mu[d]);
          amax[d] = a > amax[d] ? a : amax[d];   // (a NaN mean or a non-positive bandwidth: caught by `look` / pack_fill)
        }
      }
      double ratio = 0.0;
      for (int d = 0; d < D; ++d) {
        const double r = amax[d] / std::sqrt(2.0 * bw0[d]);
        ratio = r > ratio ? r : ratio;
      }
      out.uratio[l] = ratio;
    }
  }
  out.bad = bad;
  for (int d = 0; d < KDEHIP_MAX_DIMS; ++d) { out.lo[d] = lo[d]; out.hi[d] = hi[d]; }
}

int expand_frontiers(const kdehip_density &t, int D, int L, bool look, Frontiers &out) {
  const int rc = expand_frontier_ids(t, D, L, out);
  if (rc != KDEHIP_OK) return rc;
  examine_frontiers(t, D, L, look, out);
  return KDEHIP_OK;
}

// mode: kPackChecked looks at every node first (finiteness, bandwidth range) and decides between the fast and the
// generic arithmetic form; kPackOptimistic assumes the fast form and leaves those checks to pack_fill, which reads
// every value anyway (the checks are random accesses into the tree arrays: as expensive as the fill itself);
// kPackGeneric is the layout of the generic form without looking.
int pack_layout(int Ndens, const kdehip_density *trees, int ndims, const uint8_t *mask, int precision,
                PackedProduct &out, PackMode pmode) {
  if (Ndens < 1 || !trees) return set_error(KDEHIP_ERR_ARG, "need at least one density");
  if (Ndens > KDEHIP_MAX_DENS)
    return set_error(KDEHIP_ERR_UNSUPPORTED, "more than KDEHIP_MAX_DENS densities in one product");
  if (ndims < 1 || ndims > KDEHIP_MAX_DIMS)
    return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  int64_t maxN = 0;
  for (int j = 0; j < Ndens; ++j) {
    const kdehip_density &t = trees[j];
    if (t.ndim != ndims)  // error("kdes must have same dimension"), src/MSGibbs01.jl:720-722
      return set_error(KDEHIP_ERR_DIM_MISMATCH, "kdes must have same dimension");
    if (t.npts < 1) return set_error(KDEHIP_ERR_ARG, "density with no points");
    if (t.npts > (int64_t(1) << 30)) return set_error(KDEHIP_ERR_UNSUPPORTED, "density too large");
    if (!t.means || !t.bandwidth || !t.weights || !t.left_child || !t.right_child || !t.permutation)
      return set_error(KDEHIP_ERR_ARG, "density with a null array");
    if (t.npts > maxN) maxN = t.npts;
  }
  const int D = ndims, M = Ndens;
  const int L = nlevels_for(maxN);
  out = PackedProduct();
  out.front_off.assign(static_cast<size_t>(M) * (L + 1) + 1, 0);

  // ---- phase 1: expand every frontier (levelDown!, src/MSGibbs01.jl:503-511) into one flat id array, and (checked
  // mode) look at every frontier node once: finiteness, bandwidth range; whether a level shares one bandwidth vector
  double bw_lo[KDEHIP_MAX_DIMS], bw_hi[KDEHIP_MAX_DIMS];
  for (int d = 0; d < D; ++d) { bw_lo[d] = INFINITY; bw_hi[d] = 0.0; }
  bool finite_ok = true;
  std::vector<TileShape> shapes(static_cast<size_t>(M) * (L + 1));
  out.front.clear();
  int64_t nodes = 0;
  // (one density's frontiers do not depend on another's: large products expand them on the host pool's threads)
  std::vector<Frontiers> frs(static_cast<size_t>(M));
  std::vector<int> rcs(static_cast<size_t>(M), KDEHIP_OK);
  int64_t points = 0;
  for (int j = 0; j < M; ++j) points += trees[j].npts;
  HostPool *pool = (M > 1 && points >= 2048) ? &HostPool::get() : nullptr;
  const bool look = pmode == kPackChecked;
  if (pool && pool->workers() > 0) {
    TaskGroup group(*pool);
    for (int j = 1; j < M; ++j)
      group.run([trees, j, D, L, look, &frs, &rcs] { rcs[j] = expand_frontiers(trees[j], D, L, look, frs[j]); });
    rcs[0] = expand_frontiers(trees[0], D, L, look, frs[0]);
    group.wait();
  } else {
    for (int j = 0; j < M; ++j) rcs[j] = expand_frontiers(trees[j], D, L, look, frs[j]);
  }
  for (int j = 0; j < M; ++j)
    if (rcs[j] != KDEHIP_OK)  // (again on THIS thread: the error message is thread-local)
      return expand_frontiers(trees[j], D, L, look, frs[j]);
  for (int j = 0; j < M; ++j) {
    const Frontiers &fr = frs[j];
    const int64_t base = static_cast<int64_t>(out.front.size());
    out.front.insert(out.front.end(), fr.ids.begin(), fr.ids.end());
    for (int l = 0; l <= L; ++l) {
      const size_t idx = static_cast<size_t>(j) * (L + 1) + l;
      out.front_off[idx] = base + fr.off[l];
      shapes[idx].n = fr.off[l + 1] - fr.off[l];
      shapes[idx].uniform = fr.uniform[l] != 0;
      shapes[idx].uratio = fr.uratio[l];
    }
    nodes += fr.nodes;
    if (fr.bad) finite_ok = false;
    for (int d = 0; d < D; ++d) {
      if (fr.lo[d] < bw_lo[d]) bw_lo[d] = fr.lo[d];
      if (fr.hi[d] > bw_hi[d]) bw_hi[d] = fr.hi[d];
    }
  }
  out.front_off.back() = static_cast<int64_t>(out.front.size());
  const bool in_range = (pmode == kPackChecked) ? variances_in_range(bw_lo, bw_hi, D, precision) : true;
  if (pmode == kPackGeneric) finite_ok = false;
  const std::vector<int64_t> front_off = out.front_off;
  std::vector<int32_t> front;
  front.swap(out.front);
  const int rc = pack_layout_shapes(M, D, L, shapes.data(), mask, precision, finite_ok && in_range, out);
  out.front.swap(front);
  out.front_off = front_off;
  out.nodes_per_sweep = nodes;
  return rc;
}

// Tile geometry, staging modes and conditional tables of a product from the shapes of its frontiers alone
// (`shapes[j * (L+1) + l]`): what the host packer derives from the trees, and all the device packer needs
// (pack_device.hip: the frontiers of a density in HBM were expanded when it was uploaded).
int pack_layout_shapes(int M, int D, int L, const TileShape *shapes, const uint8_t *mask, int precision, bool fast,
                       PackedProduct &out) {
  out = PackedProduct();
  out.M = M; out.D = D; out.L = L;
  out.precision = precision;
  out.levels.resize(static_cast<size_t>(M) * (L + 1));
  out.front_off.assign(static_cast<size_t>(M) * (L + 1) + 1, 0);
  const uint32_t all = (1u << D) - 1u;
  std::vector<uint32_t> mask_bits(M, all), others_bits(M, 0);
  for (int j = 0; j < M; ++j) {
    if (mask) {
      uint32_t b = 0;
      for (int d = 0; d < D; ++d) if (mask[j * D + d]) b |= (1u << d);
      mask_bits[j] = b;
      if (b != all) out.masked = true;
    }
  }
  for (int j = 0; j < M; ++j)
    for (int k = 0; k < M; ++k) if (k != j) others_bits[j] |= mask_bits[k];
  // the fast forms evaluate every dimension: they need every dimension of every density to be informed by
  // some OTHER density too (a one-density "product" or a partialDimMask leaves dimensions inactive: masked fast form)
  out.all_active = true;
  for (int j = 0; j < M; ++j) if ((mask_bits[j] & others_bits[j]) != all) out.all_active = false;
  out.fast = fast;
  for (int j = 0; j < M; ++j)
    for (int l = 1; l <= L; ++l) out.nodes_per_sweep += shapes[static_cast<size_t>(j) * (L + 1) + l].n;

  // ---- phase 2: tile geometry and offsets (the payload is written by pack_fill)
  const int64_t esz = (precision == 64) ? 8 : 4;
  int64_t nelem = 0, nperm = 0;
  for (int j = 0; j < M; ++j) {
    for (int l = 0; l <= L; ++l) {
      const size_t idx = static_cast<size_t>(j) * (L + 1) + l;
      const int64_t n = shapes[idx].n;
      const int64_t B = (n + 63) / 64;
      // compact tiles only on the fast path.  fp64: the shared-bandwidth evaluator forms (m - mu) * s as fma(m, s, -mu*s)
      // (two instructions per dimension instead of three, gibbs_device.hpp EvalUniform), whose rounding error grows
      // with |m| * s <= |m| / sqrt(2 bandwidth): 1e-16 * kMaxUniformRatio = 1e-11 in the scaled difference at the very
      // most; frontiers beyond that (data 10^5 bandwidths away from the origin) keep the per-node form, which subtracts first
      // (fp32 keeps the subtract-first form: no limit)
      const bool uni = out.fast && shapes[idx].uniform && (precision == 32 || shapes[idx].uratio <= kMaxUniformRatio);
      const int F = uni ? D + 1 : 2 * D + 1;
      // (TileAddr: fp64 rows of F*64+1 elements; fp32 row PAIRS of 2*F*64+2)
      const bool paired = (precision == 32);
      const int64_t RS = paired ? TileAddrBytes<4>::stride(F) : TileAddrBytes<8>::stride(F);
      const int64_t body = paired ? TileAddrBytes<4>::body(B, F) : TileAddrBytes<8>::body(B, F);
      LevelDesc &ds = out.levels[idx];
      std::memset(&ds, 0, sizeof(ds));
      ds.n = static_cast<int32_t>(n);
      ds.B = static_cast<int32_t>(B);
      ds.F = F;
      ds.uniform_bw = uni ? 1 : 0;
      ds.mask_bits = mask_bits[j];
      ds.others_bits = others_bits[j];
      nelem = (nelem + 7) & ~int64_t(7);  // tiles start 64-byte (fp32: 32-byte) aligned
      ds.hdr_off = nelem;
      nelem += kTileHeader + body;
      ds.perm_off = nperm;
      nperm += B * 64;
      const int64_t bytes = (kTileHeader + body) * esz;
      if (bytes > (int64_t(1) << 30)) return set_error(KDEHIP_ERR_UNSUPPORTED, "level tile too large");
      ds.stage_bytes = static_cast<int32_t>((bytes + 1023) / 1024 * 1024);
      ds.last_lane = static_cast<int32_t>((n - 1) / B);
      ds.chunk_rows = static_cast<int32_t>(((paired ? 2 : 1) * ((kLdsPoolBytes / 2 - 1024) / (RS * esz))) & ~int64_t(3));
    }
  }
  out.perm_elems = nperm;
  out.tile_elems = nelem;

  // ---- phase 3: where each level's tiles live while the kernel works on that level
  for (int l = 0; l <= L; ++l) {
    int64_t sum = 0, mx = 0;
    for (int j = 0; j < M; ++j) {
      const int64_t b = out.levels[static_cast<size_t>(j) * (L + 1) + l].stage_bytes;
      sum += b;
      if (b > mx) mx = b;
    }
    int32_t mode = kStageGlobal;
    if (l == 0) mode = kStageGlobal;  // the roots are read once, straight from memory
    else if (sum <= kLdsPoolBytes) mode = kStageResident;
    else if (mx <= kLdsPoolBytes / 2) mode = kStageStream;
    else mode = kStageChunked;
    int64_t off = 0;
    for (int j = 0; j < M; ++j) {
      LevelDesc &ds = out.levels[static_cast<size_t>(j) * (L + 1) + l];
      ds.stage_mode = mode;
      ds.seg = (mode == kStageChunked) ? seg_geometry(ds.B, ds.chunk_rows) : 0;
      ds.lds_off = (mode == kStageResident) ? static_cast<int32_t>(off) : 0;
      off += ds.stage_bytes;
    }
  }

  // ---- phase 3b: fp32 screening (kdehip_internal.hpp "fp32 screening"): fp64 plans of the register-resident sampler's
  // domain; a level is screened when its fp64 tiles are streamed or chunked, every tile has 2..kScreenMaxRows rows per
  // lane and the M fp32 images fit the pool together.  The screen tiles follow the fp64 tiles in the plan's data.
  static const bool screen_on = [] { const char *e = std::getenv("KDEHIP_SCREEN"); return !(e && e[0] == '0'); }();
  if (screen_on && precision == 64 && out.fast && out.all_active && !out.masked && ((M >= 2 && M <= 4) || M == 8) &&
      D * (L + 1) <= 128) {
    std::vector<LevelDesc> scr(static_cast<size_t>(M) * (L + 1));
    std::memset(scr.data(), 0, scr.size() * sizeof(LevelDesc));
    int64_t felem = 2 * nelem;  // in floats from the start of the plan's data
    for (int l = 1; l <= L; ++l) {
      const int mode = out.levels[l].stage_mode;
      if (mode != kStageStream && mode != kStageChunked) continue;
      bool ok = true;
      int64_t sum = 0, mx = 0;
      int maxB = 0;
      for (int j = 0; j < M && ok; ++j) {
        const LevelDesc &ds = out.levels[static_cast<size_t>(j) * (L + 1) + l];
        if (ds.B < 2 || ds.B > kScreenMaxRowsChunked) ok = false;
        maxB = ds.B > maxB ? ds.B : maxB;
        const int64_t bytes = ((kScreenHeaderFloats + TileAddrBytes<4>::body(ds.B, ds.F)) * 4 + 1023) / 1024 * 1024;
        sum += bytes;
        mx = bytes > mx ? bytes : mx;
      }
      // all M images in the pool together (no barrier between the level's steps), or one per pool half, streamed like the
      // fp64 tiles of a streamed level (a barrier per step: a wavefront that repeats a step in fp64 holds up its workgroup)
      // or in chunks of whole row pairs through the halves (a barrier per chunk; the second pass from the screen tile in
      // global memory)
      static const bool stream_on = [] { const char *e = std::getenv("KDEHIP_SCREEN_STREAM"); return !(e && e[0] == '0'); }();
      static const bool chunk_on = [] { const char *e = std::getenv("KDEHIP_SCREEN_CHUNK"); return !(e && e[0] == '0'); }();
      const bool together = sum <= kLdsPoolBytes && maxB <= kScreenMaxRows;
      const bool streamed = !together && stream_on && mx <= kLdsPoolBytes / 2 && maxB <= kScreenMaxRows;
      const bool chunked = !together && !streamed && stream_on && chunk_on;
      if (!ok || !(together || streamed || chunked)) continue;
      int64_t off = 0;
      for (int j = 0; j < M; ++j) {
        const LevelDesc &ds = out.levels[static_cast<size_t>(j) * (L + 1) + l];
        LevelDesc &sc = scr[static_cast<size_t>(j) * (L + 1) + l];
        sc.n = ds.n; sc.B = ds.B; sc.F = ds.F; sc.uniform_bw = ds.uniform_bw; sc.last_lane = ds.last_lane;
        sc.stage_mode = together ? kStageScreen : streamed ? kStageScreenStream : kStageScreenChunked;
        if (chunked) {  // whole row pairs per chunk, an even number of them (16-byte aligned chunk starts), the header with chunk 0
          const int64_t pair_bytes = TileAddrBytes<4>::stride(ds.F) * 4;
          const int64_t cp = ((kLdsPoolBytes / 2 - kScreenHeaderFloats * 4) / pair_bytes) & ~int64_t(1);
          sc.chunk_rows = static_cast<int32_t>(2 * cp);
        }
        felem = (felem + 63) & ~int64_t(63);  // 256-byte aligned images
        sc.hdr_off = felem;
        const int64_t elems = kScreenHeaderFloats + TileAddrBytes<4>::body(ds.B, ds.F);
        felem += elems;
        sc.stage_bytes = static_cast<int32_t>((elems * 4 + 1023) / 1024 * 1024);
        sc.lds_off = together ? static_cast<int32_t>(off) : 0;
        off += sc.stage_bytes;
      }
      ++out.nscreened;
    }
    if (out.nscreened > 0) {
      out.screens.swap(scr);
      nelem = (felem + 1) / 2;
    }
  }
  out.data_elems = nelem + 1024 / 4;  // staged copies are rounded up to whole KiB: keep the tail readable
  out.steps.resize(out.levels.size());
  for (size_t idx = 0; idx < out.levels.size(); ++idx) {
    const LevelDesc &ds = out.levels[idx];
    StepDesc &s = out.steps[idx];
    s.n = ds.n;
    s.flags = ds.last_lane | (ds.uniform_bw << 8);
    s.lds_off = ds.lds_off;
    if (!out.screens.empty() && out.screens[idx].stage_mode == kStageScreen) s.lds_off = out.screens[idx].lds_off;  // (streamed screens: the pool half of the step)
    s.stage_bytes = ds.stage_bytes;
    s.chunk_rows = ds.chunk_rows;
    s.seg = ds.seg;
    s.hdr_lo = static_cast<int32_t>(ds.hdr_off & 0xFFFFFFFF);
    s.hdr_hi = static_cast<int32_t>(ds.hdr_off >> 32);
  }

  // ---- phase 4: conditional tables (gibbs_kernel.hip): levels whose frontiers all fit one wavefront row
  // and have power-of-two sizes, as long as the rows of all densities stay within the entry budget
  out.tabdesc.assign(static_cast<size_t>(M) * (L + 1), TabDesc{});
  for (auto &t : out.tabdesc) t.off = -1;
  if (out.fast && M >= 2) {
    int64_t entries = 0, rows = 0;
    for (int l = 1; l <= L; ++l) {
      bool ok = true;
      int total_bits = 0;
      std::vector<int> bits(M, 0);
      for (int j = 0; j < M && ok; ++j) {
        const LevelDesc &dj = out.levels[static_cast<size_t>(j) * (L + 1) + l];
        if (dj.B != 1 || dj.stage_mode != kStageResident || (dj.n & (dj.n - 1)) != 0) { ok = false; break; }
        while ((1 << bits[j]) < dj.n) ++bits[j];
        total_bits += bits[j];
      }
      if (!ok || total_bits > 30) break;
      int64_t lvl_entries = 0;
      for (int j = 0; j < M; ++j)
        lvl_entries += (int64_t(1) << (total_bits - bits[j])) * ((int64_t(1) << bits[j]) + 1);
      if (entries + lvl_entries > kTabMaxEntries) break;
      int shift = 0;
      for (int j = 0; j < M; ++j) {
        TabDesc &td = out.tabdesc[static_cast<size_t>(j) * (L + 1) + l];
        td.n = 1 << bits[j];
        td.bits = bits[j];
        td.shift = shift;
        shift += bits[j];
        td.ncfg = 1 << (total_bits - bits[j]);
        td.off = entries;
        td.row_base = rows;
        entries += static_cast<int64_t>(td.ncfg) * (td.n + 1);
        rows += td.ncfg;
      }
      out.Lt = l;
    }
    out.tab_entries = entries;
    out.tab_rows = rows;
  }
  return KDEHIP_OK;
}


// Writes the payload of a layout: tiles (element type by `precision` of the layout) and permutation rows.
// Row by row, field by field, 64 contiguous lanes at a time (the sources are gathered through the frontier ids).
// Returns whether every value met the conditions of the fast arithmetic form (finite means below 1e100, positive
// finite variances whose products stay in range, finite non-negative weights): what pack_layout's kPackChecked mode
// establishes beforehand, found here on the values that are being copied anyway.
// what the values of one density's tiles say about the fast arithmetic form
struct FillFindings {
  bool bad = false;
  double lo[KDEHIP_MAX_DIMS], hi[KDEHIP_MAX_DIMS];
  FillFindings() { for (int d = 0; d < KDEHIP_MAX_DIMS; ++d) { lo[d] = INFINITY; hi[d] = 0.0; } }
};

// the tiles of density j, levels l0 .. l1 - 1: contiguous in `data` (and in `perm`), no other tile's elements touched
template <typename T>
static void fill_density(const PackedProduct &pp, const kdehip_density &t, int j, int l0, int l1, T *data, int32_t *perm,
                         FillFindings &f) {
  const int D = pp.D, L = pp.L;
  bool bad = false;
  double *lo = f.lo, *hi = f.hi;
  for (int l = l0; l < l1; ++l) {
    const size_t idx = static_cast<size_t>(j) * (L + 1) + l;
    const LevelDesc &ds = pp.levels[idx];
    const int32_t *cur = pp.front.data() + pp.front_off[idx];
    const int64_t n = ds.n, B = ds.B;
    const int F = ds.F;
    const bool uni = ds.uniform_bw != 0;
    using TA = TileAddr<T>;
    const int64_t RS = TA::stride(F);
    T *hdr = data + ds.hdr_off;
    for (int d = 0; d < kTileHeader; ++d)
      hdr[d] = d < D ? static_cast<T>(t.bandwidth[(static_cast<int64_t>(cur[0]) - 1) * D + d]) : T(0);
    for (int d = 0; d < D; ++d) {  // (the one bandwidth vector of a uniform tile; entry 0's otherwise)
      const double v = t.bandwidth[(static_cast<int64_t>(cur[0]) - 1) * D + d];
      bad |= !(v > 0.0) | !(v < INFINITY);
      lo[d] = v < lo[d] ? v : lo[d];
      hi[d] = v > hi[d] ? v : hi[d];
    }
    T *tile = hdr + kTileHeader;
    int32_t *prow = perm + ds.perm_off;
    for (int64_t i = 0; i < B; ++i) {
      int64_t src[64];  // source offset (node - 1) of lane ln's entry in this row, -1 = padding
      for (int ln = 0; ln < 64; ++ln) {
        const int64_t z = static_cast<int64_t>(ln) * B + i;
        src[ln] = z < n ? static_cast<int64_t>(cur[z]) - 1 : -1;
      }
      T *row = tile + TA::row(i, RS);  // (row i, field 0, lane 0); field f of lane ln at row[f * kField + ln * kLane]
      T dst[64];                        // one field of the row, checked here and then scattered into the tile
      auto put = [&](int f) {
        for (int ln = 0; ln < 64; ++ln) row[f * TA::kField + ln * TA::kLane] = dst[ln];
      };
      for (int d = 0; d < D; ++d) {
        for (int ln = 0; ln < 64; ++ln) dst[ln] = src[ln] >= 0 ? static_cast<T>(t.means[src[ln] * D + d]) : T(0);
        // (checked on the 64 contiguous values just written: vectorisable, unlike the gather above)
        T amax = T(0), nan_acc = T(0);
        for (int ln = 0; ln < 64; ++ln) {
          const T a = dst[ln] < T(0) ? -dst[ln] : dst[ln];
          amax = a > amax ? a : amax;
          nan_acc += dst[ln] * T(0);  // 0 for a finite value, NaN otherwise
        }
        bad |= !(static_cast<double>(amax) < 1e100) | !(nan_acc == T(0));
        put(d);
      }
      if (!uni)
        for (int d = 0; d < D; ++d) {
          for (int ln = 0; ln < 64; ++ln) dst[ln] = src[ln] >= 0 ? static_cast<T>(t.bandwidth[src[ln] * D + d]) : T(1);
          T l = dst[0], h = dst[0], nan_acc = T(0);  // (padding entries carry variance 1: neutral for the range test)
          for (int ln = 0; ln < 64; ++ln) {
            l = dst[ln] < l ? dst[ln] : l;
            h = dst[ln] > h ? dst[ln] : h;
            nan_acc += dst[ln] * T(0);
          }
          bad |= !(l > T(0)) | !(static_cast<double>(h) < INFINITY) | !(nan_acc == T(0));
          lo[d] = static_cast<double>(l) < lo[d] ? static_cast<double>(l) : lo[d];
          hi[d] = static_cast<double>(h) > hi[d] ? static_cast<double>(h) : hi[d];
          put(D + d);
        }
      T *wdst = dst;
      for (int ln = 0; ln < 64; ++ln) wdst[ln] = src[ln] >= 0 ? static_cast<T>(t.weights[src[ln]]) : T(0);
      {
        T wl = wdst[0], wh = wdst[0], nan_acc = T(0);
        for (int ln = 0; ln < 64; ++ln) {
          wl = wdst[ln] < wl ? wdst[ln] : wl;
          wh = wdst[ln] > wh ? wdst[ln] : wh;
          nan_acc += wdst[ln] * T(0);
        }
        bad |= !(wl >= T(0)) | !(static_cast<double>(wh) < INFINITY) | !(nan_acc == T(0));
      }
      put(F - 1);
      row[F * TA::kField] = T(0);  // the pad element (fp32: one of the pair's two)
      int32_t *pdst = prow + i * 64;
      for (int ln = 0; ln < 64; ++ln) pdst[ln] = src[ln] >= 0 ? static_cast<int32_t>(t.permutation[src[ln]]) : 0;
    }
    if (TA::kPaired && (B & 1)) {  // the missing second row of the last pair: padding entries (weight 0, variance 1, mean 0)
      T *row = tile + TA::row(B, RS);
      for (int f = 0; f < F; ++f) {
        const T v = (!uni && f >= D && f < 2 * D) ? T(1) : T(0);
        for (int ln = 0; ln < 64; ++ln) row[f * TA::kField + ln * TA::kLane] = v;
      }
      row[F * TA::kField] = T(0);
    }
    // gap up to the next tile's aligned start
    const int64_t end = ds.hdr_off + kTileHeader + TA::body(B, F);
    // (the last tile: its readable tail; screen tiles, which follow, are written by the GPU)
    const int64_t tail = pp.tile_elems + 1024 / 4 < pp.data_elems ? pp.tile_elems + 1024 / 4 : pp.data_elems;
    const int64_t next = (idx + 1 < pp.levels.size()) ? pp.levels[idx + 1].hdr_off : tail;
    for (int64_t e = end; e < next; ++e) data[e] = T(0);
  }
  f.bad = bad;
}

constexpr int64_t kParallelFillElems = 32 * 1024;  // smaller products are packed by the calling thread alone (< 20 us)

template <typename T>
static bool fill_tiles(const PackedProduct &pp, const kdehip_density *trees, T *data, int32_t *perm) {
  const int D = pp.D, M = pp.M;
  const int L = pp.L;
  std::vector<FillFindings> part(static_cast<size_t>(2 * M));
  // the tiles are independent: each density is packed by threads of the host pool (csrc/host_pool.hpp) in two tasks of
  // about the same size -- its deepest level (half its frontier nodes) and all the levels above it; the calling thread
  // takes the first task and whatever no worker has started by the time it is done
  HostPool *pool = (pp.data_elems >= kParallelFillElems) ? &HostPool::get() : nullptr;
  if (pool && pool->workers() > 0) {
    TaskGroup group(*pool);
    for (int j = 0; j < M; ++j) {
      if (j > 0) group.run([&pp, trees, j, L, data, perm, &part] { fill_density<T>(pp, trees[j], j, 0, L, data, perm, part[2 * j]); });
      group.run([&pp, trees, j, L, data, perm, &part] { fill_density<T>(pp, trees[j], j, L, L + 1, data, perm, part[2 * j + 1]); });
    }
    fill_density<T>(pp, trees[0], 0, 0, L, data, perm, part[0]);
    group.wait();
  } else {
    for (int j = 0; j < M; ++j) fill_density<T>(pp, trees[j], j, 0, L + 1, data, perm, part[2 * j]);
  }
  bool bad = false;
  double lo[KDEHIP_MAX_DIMS], hi[KDEHIP_MAX_DIMS];
  for (int d = 0; d < KDEHIP_MAX_DIMS; ++d) { lo[d] = INFINITY; hi[d] = 0.0; }
  for (int j = 0; j < 2 * M; ++j) {
    bad |= part[j].bad;
    for (int d = 0; d < D; ++d) {
      lo[d] = part[j].lo[d] < lo[d] ? part[j].lo[d] : lo[d];
      hi[d] = part[j].hi[d] > hi[d] ? part[j].hi[d] : hi[d];
    }
  }
  return !bad && variances_in_range(lo, hi, D, pp.precision);
}

bool pack_fill(const PackedProduct &pp, const kdehip_density *trees, void *data, int32_t *perm) {
  const bool ok = (pp.precision == 64) ? fill_tiles<double>(pp, trees, static_cast<double *>(data), perm)
                                       : fill_tiles<float>(pp, trees, static_cast<float *>(data), perm);
  return ok || !pp.fast;  // (a generic-form layout has no conditions to meet)
}

// Layout + payload in host vectors (fp64 payload whatever the precision of the layout): tests and tools.
int pack_levels(int Ndens, const kdehip_density *trees, int ndims, const uint8_t *mask, int precision,
                PackedProduct &out) {
  const int rc = pack_layout(Ndens, trees, ndims, mask, precision, out);
  if (rc != KDEHIP_OK) return rc;
  out.data.assign(static_cast<size_t>(out.data_elems), 0.0);
  out.perm.assign(static_cast<size_t>(out.perm_elems), 0);
  fill_tiles<double>(out, trees, out.data.data(), out.perm.data());
  return KDEHIP_OK;
}

}  // namespace kdehip
