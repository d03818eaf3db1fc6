// Host-side sanitizer harness (CPU build only: GPU AddressSanitizer is not available on the pool).
// Compiled by tests/test_host_sanitizers.py with g++ -fsanitize=address,undefined together with the two
// host translation units of libkdehip (balltree.cpp, pack_levels.cpp) and run over ragged / masked / tiny /
// large density sets in both precisions.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
#include "kdehip.h"
#include "kdehip_internal.hpp"
using namespace kdehip;
struct Dens { int64_t D, N; std::vector<double> centers, ranges, w, means, bw, bmin, bmax; std::vector<int64_t> l, r, lo, hi, perm; };
static Dens make(int64_t D, int64_t N, std::mt19937_64 &g) {
  Dens d; d.D = D; d.N = N;
  std::normal_distribution<double> nd;
  std::vector<double> pts(D * N), ks(D), wi(N);
  for (auto &x : pts) x = nd(g);
  for (auto &x : ks) x = 0.1 + 0.5 * std::abs(nd(g));
  for (auto &x : wi) x = 0.5 + std::abs(nd(g));
  d.centers.resize(2*N*D); d.ranges.resize(2*N*D); d.means.resize(2*N*D); d.bw.resize(2*N*D); d.w.resize(2*N);
  d.bmin.resize(N*D); d.bmax.resize(N*D); d.l.resize(2*N); d.r.resize(2*N); d.lo.resize(2*N); d.hi.resize(2*N); d.perm.resize(2*N);
  int rc = kdehip_make_density(D, N, pts.data(), ks.data(), D, wi.data(), d.centers.data(), d.ranges.data(), d.w.data(), d.l.data(), d.r.data(), d.lo.data(), d.hi.data(), d.perm.data(), d.means.data(), d.bw.data(), d.bmin.data(), d.bmax.data());
  if (rc) { printf("make_density rc=%d\n", rc); exit(1); }
  return d;
}
int main() {
  std::mt19937_64 g(7);
  int cases = 0;
  for (int D : {1, 2, 3, 6, 8}) for (int M : {1, 2, 4, 16}) for (int64_t N : {1, 2, 3, 63, 64, 65, 1000, 5000}) {
    std::vector<Dens> ds; std::vector<kdehip_density> cd;
    for (int j = 0; j < M; ++j) ds.push_back(make(D, (j % 2) ? N : (N / 2 + 1), g));
    for (auto &d : ds) { kdehip_density c{}; c.npts = d.N; c.ndim = d.D; c.means = d.means.data(); c.bandwidth = d.bw.data(); c.weights = d.w.data(); c.left_child = d.l.data(); c.right_child = d.r.data(); c.permutation = d.perm.data(); cd.push_back(c); }
    std::vector<uint8_t> mask(M * D, 1);
    if (D > 1) mask[0] = 0;
    for (int prec : {64, 32}) for (int um = 0; um < 2; ++um) {
      PackedProduct out;
      int rc = pack_levels(M, cd.data(), D, um ? mask.data() : nullptr, prec, out);
      if (rc) { printf("pack rc=%d D=%d M=%d N=%lld: %s\n", rc, D, M, (long long)N, last_error_cstr()); return 1; }
      ++cases;
    }
  }
  // several callers at once: the tree builder and the packer share the library's worker threads (csrc/host_pool.hpp)
  std::vector<std::thread> callers;
  std::vector<int> failed(4, 0);
  for (int c = 0; c < 4; ++c)
    callers.emplace_back([c, &failed] {
      std::mt19937_64 gc(100 + c);
      for (int rep = 0; rep < 6; ++rep) {
        const int D = 1 + (c + rep) % 6, M = 2 + rep % 4;
        std::vector<Dens> ds; std::vector<kdehip_density> cd;
        for (int j = 0; j < M; ++j) ds.push_back(make(D, 600 + 900 * ((j + c) % 4), gc));
        for (auto &d : ds) { kdehip_density k{}; k.npts = d.N; k.ndim = d.D; k.means = d.means.data(); k.bandwidth = d.bw.data(); k.weights = d.w.data(); k.left_child = d.l.data(); k.right_child = d.r.data(); k.permutation = d.perm.data(); cd.push_back(k); }
        PackedProduct out;
        if (pack_levels(M, cd.data(), D, nullptr, 64, out)) failed[c] = 1;
      }
    });
  for (auto &t : callers) t.join();
  for (int f : failed) if (f) { printf("concurrent pack failed\n"); return 1; }
  printf("ok: %d pack_levels cases, 4 concurrent callers\n", cases);
  return 0;
}
