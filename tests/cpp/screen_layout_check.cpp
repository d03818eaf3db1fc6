// Host check of the fp32 screen layout the packer lays out behind an fp64 plan (csrc/pack_levels.cpp phase 3b): which
// levels are screened and in which staging mode (resident / streamed / chunked), that the images fit where the sampler
// puts them, that chunk starts are aligned, and that nothing overlaps in the plan's data.  Compiled by
// tests/test_screen_layout.py with g++ together with balltree.cpp and pack_levels.cpp; no GPU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "kdehip.h"
#include "kdehip_internal.hpp"
using namespace kdehip;
struct Dens { int64_t D, N; std::vector<double> centers, ranges, w, means, bw, bmin, bmax; std::vector<int64_t> l, r, lo, hi, perm; };
static Dens make(int64_t D, int64_t N, std::mt19937_64 &g) {
  Dens d; d.D = D; d.N = N;
  std::normal_distribution<double> nd;
  std::vector<double> pts(D * N), ks(D), wi(N, 1.0);
  for (auto &x : pts) x = nd(g);
  for (auto &x : ks) x = 0.3;
  d.centers.resize(2*N*D); d.ranges.resize(2*N*D); d.means.resize(2*N*D); d.bw.resize(2*N*D); d.w.resize(2*N);
  d.bmin.resize(N*D); d.bmax.resize(N*D); d.l.resize(2*N); d.r.resize(2*N); d.lo.resize(2*N); d.hi.resize(2*N); d.perm.resize(2*N);
  int rc = kdehip_make_density(D, N, pts.data(), ks.data(), D, wi.data(), d.centers.data(), d.ranges.data(), d.w.data(), d.l.data(), d.r.data(), d.lo.data(), d.hi.data(), d.perm.data(), d.means.data(), d.bw.data(), d.bmin.data(), d.bmax.data());
  if (rc) { printf("make_density rc=%d\n", rc); exit(1); }
  return d;
}
#define CHECK(c, ...) do { if (!(c)) { printf("FAILED %s:%d: " #c "  ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); exit(1); } } while (0)

static std::vector<int> pack_and_check(int D, int M, const std::vector<int64_t> &Ns, std::mt19937_64 &g, int precision = 64) {
  std::vector<Dens> ds; std::vector<kdehip_density> cd;
  for (int j = 0; j < M; ++j) ds.push_back(make(D, Ns[j % Ns.size()], g));
  for (auto &d : ds) { kdehip_density c{}; c.npts = d.N; c.ndim = d.D; c.means = d.means.data(); c.bandwidth = d.bw.data(); c.weights = d.w.data(); c.left_child = d.l.data(); c.right_child = d.r.data(); c.permutation = d.perm.data(); cd.push_back(c); }
  PackedProduct out;
  const int rc = pack_levels(M, cd.data(), D, nullptr, precision, out);
  CHECK(rc == 0, "pack_levels rc=%d: %s", rc, last_error_cstr());
  const int L = out.L;
  std::vector<int> modes(L + 1, 0);
  if (out.screens.empty()) { CHECK(out.nscreened == 0, "nscreened %d without descriptors", out.nscreened); return modes; }
  CHECK(precision == 64, "an fp32 plan has screen tiles");
  CHECK(out.screens.size() == out.levels.size() && out.steps.size() == out.levels.size(), "table sizes");
  using TA = TileAddrBytes<4>;
  const int64_t half = kLdsPoolBytes / 2;
  std::vector<std::pair<int64_t, int64_t>> images;  // [first byte, end) of every screen image in the plan's data
  int nscreened = 0;
  for (int l = 1; l <= L; ++l) {
    const int mode = out.screens[l].stage_mode;
    modes[l] = mode;
    if (mode == 0) {
      for (int j = 0; j < M; ++j) CHECK(out.screens[static_cast<size_t>(j) * (L + 1) + l].stage_mode == 0, "level %d: mixed", l);
      continue;
    }
    ++nscreened;
    CHECK(mode == kStageScreen || mode == kStageScreenStream || mode == kStageScreenChunked, "level %d: mode %d", l, mode);
    const int fp64_mode = out.levels[l].stage_mode;
    CHECK(fp64_mode == kStageStream || fp64_mode == kStageChunked, "level %d screened but its fp64 mode is %d", l, fp64_mode);
    int64_t sum = 0;
    for (int j = 0; j < M; ++j) {
      const size_t idx = static_cast<size_t>(j) * (L + 1) + l;
      const LevelDesc &sc = out.screens[idx], &dl = out.levels[idx];
      CHECK(sc.stage_mode == mode, "level %d density %d: mode %d != %d", l, j, sc.stage_mode, mode);
      CHECK(sc.n == dl.n && sc.B == dl.B && sc.F == dl.F && sc.uniform_bw == dl.uniform_bw && sc.last_lane == dl.last_lane, "shape");
      CHECK(sc.B >= 2 && sc.B <= (mode == kStageScreenChunked ? kScreenMaxRowsChunked : kScreenMaxRows), "level %d: %d rows", l, sc.B);
      const int64_t bytes = (kScreenHeaderFloats + TA::body(sc.B, sc.F)) * 4;
      CHECK(sc.stage_bytes >= bytes && sc.stage_bytes % 1024 == 0 && sc.stage_bytes < bytes + 1024, "stage_bytes %d for %lld", sc.stage_bytes, (long long)bytes);
      CHECK(sc.hdr_off % 64 == 0, "image not 256-byte aligned");
      CHECK(sc.hdr_off * 4 >= 0 && sc.hdr_off * 4 + sc.stage_bytes <= out.data_elems * 8, "image beyond the plan's data");
      images.push_back({sc.hdr_off * 4, sc.hdr_off * 4 + bytes});
      if (mode == kStageScreen) {
        CHECK(sc.lds_off == sum && out.steps[idx].lds_off == sc.lds_off, "resident offsets");
        sum += sc.stage_bytes;
      } else if (mode == kStageScreenStream) {
        CHECK(sc.stage_bytes <= half, "streamed image %d > half the pool", sc.stage_bytes);
      } else {
        const int64_t cp = sc.chunk_rows / 2, pair_bytes = TA::stride(sc.F) * 4;
        CHECK(sc.chunk_rows >= 4 && sc.chunk_rows % 4 == 0, "chunk_rows %d", sc.chunk_rows);
        CHECK(kScreenHeaderFloats * 4 + cp * pair_bytes <= half, "chunk 0 does not fit half the pool");
        CHECK((cp * pair_bytes) % 16 == 0, "chunk starts not 16-byte aligned");
        // the last chunk's copy is rounded up to 1 KiB: it may read past the image, never past the plan's data
        const int64_t npairs = (sc.B + 1) / 2, last0 = (npairs - 1) / cp * cp;
        const int64_t end = sc.hdr_off * 4 + kScreenHeaderFloats * 4 + last0 * pair_bytes + (((npairs - last0) * pair_bytes + 1023) & ~1023LL);
        CHECK(end <= out.data_elems * 8, "last chunk's copy reads beyond the plan's data");
      }
    }
    if (mode == kStageScreen) CHECK(sum <= kLdsPoolBytes, "resident images %lld > pool", (long long)sum);
    if (mode != kStageScreen) {
      bool some_big = false;
      int64_t tot = 0;
      for (int j = 0; j < M; ++j) { const auto &sc = out.screens[static_cast<size_t>(j) * (L + 1) + l]; tot += sc.stage_bytes; some_big |= sc.stage_bytes > half || sc.B > kScreenMaxRows; }
      CHECK(tot > kLdsPoolBytes || some_big, "level %d would fit the pool resident", l);
      if (mode == kStageScreenStream) CHECK(!some_big, "streamed level with an image > half");
      else CHECK(some_big, "chunked level without a large image");
    }
  }
  CHECK(nscreened == out.nscreened, "nscreened %d != %d", out.nscreened, nscreened);
  // the fp64 tiles end where the screen images begin; no two images overlap
  std::sort(images.begin(), images.end());
  for (size_t i = 1; i < images.size(); ++i) CHECK(images[i - 1].second <= images[i].first, "screen images overlap");
  for (size_t idx = 0; idx < out.levels.size(); ++idx) {
    const LevelDesc &dl = out.levels[idx];
    if (dl.n > 0 && !images.empty()) CHECK(dl.hdr_off * 8 + dl.stage_bytes <= images.front().first + 1024, "an fp64 tile overlaps the screen images");
  }
  return modes;
}

int main() {
  std::mt19937_64 g(11);
  {  // BASELINE config 3: 6-D, 4 x 1000: levels 9 and 10, resident together
    const auto m = pack_and_check(6, 4, {1000}, g);
    for (int l = 1; l <= 8; ++l) CHECK(m[l] == 0, "c3 level %d screened", l);
    CHECK(m[9] == kStageScreen && m[10] == kStageScreen, "c3 levels 9, 10: %d %d", m[9], m[10]);
  }
  {  // BASELINE config 4: 3-D, 8 x 5000: 9 resident, 10-11 streamed, 12-13 chunked
    const auto m = pack_and_check(3, 8, {5000}, g);
    CHECK(m[9] == kStageScreen && m[10] == kStageScreenStream && m[11] == kStageScreenStream && m[12] == kStageScreenChunked &&
              m[13] == kStageScreenChunked, "c4 levels 9-13: %d %d %d %d %d", m[9], m[10], m[11], m[12], m[13]);
  }
  {  // 6-D, 4 x 2048 (a chained product's operands): the two deepest levels streamed
    const auto m = pack_and_check(6, 4, {2048}, g);
    CHECK(m[10] == kStageScreenStream && m[11] == kStageScreenStream, "4 x 2048 levels 10, 11: %d %d", m[10], m[11]);
  }
  {  // BASELINE config 5 is fp32: no screen
    const auto m = pack_and_check(6, 4, {10000}, g, 32);
    for (int v : m) CHECK(v == 0, "fp32 plan screened");
  }
  int cases = 4;
  for (int D : {1, 2, 3, 4, 6, 8}) for (int M : {2, 3, 4, 8}) for (int64_t N : {300, 700, 1500, 3000, 6000, 9000, 20000}) {
    pack_and_check(D, M, {N, N / 2 + 1, N - 37}, g);
    ++cases;
  }
  for (int M : {1, 5, 6, 16}) { const auto m = pack_and_check(3, M, {5000}, g); for (int v : m) CHECK(v == 0, "M = %d screened", M); ++cases; }
  printf("screen layout ok: %d products\n", cases);
  return 0;
}
