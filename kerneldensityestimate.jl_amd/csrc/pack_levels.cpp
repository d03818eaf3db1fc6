// pack_levels.cpp -- host re-layout of BallTreeDensity inputs into per-level tiles (product code).
//
// The reference builds the frontier lists on the fly for every output sample (levelInit!,
// levelDown!, src/MSGibbs01.jl:467-475,500-523).  They depend only on the trees, so here they are
// expanded once per product and stored level by level in the lane-blocked SoA layout described in
// kdehip_internal.hpp; the kernels never touch the tree topology.
#include <cmath>
#include <cstring>
#include <mutex>

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

int pack_levels(int Ndens, const kdehip_density *trees, int ndims, const uint8_t *mask,
                PackedProduct &out) {
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
  out.M = M; out.D = D; out.L = L;
  out.levels.resize(static_cast<size_t>(M) * (L + 1));

  const uint32_t all = (D >= 32) ? 0xFFFFFFFFu : ((1u << D) - 1u);
  for (int j = 0; j < M; ++j) {
    uint32_t b = all;
    if (mask) {
      b = 0;
      for (int d = 0; d < D; ++d) if (mask[j * D + d]) b |= (1u << d);
      if (b != all) out.masked = true;
    }
    out.mask_bits[j] = b;
  }
  for (int j = 0; j < M; ++j) {
    uint32_t o = 0;
    for (int k = 0; k < M; ++k) if (k != j) o |= out.mask_bits[k];
    out.others_bits[j] = o;
  }

  // range bookkeeping for the product/rsqrt evaluation (see eval_fast in gibbs_kernel.hip)
  std::vector<double> bw_lo(D, INFINITY), bw_hi(D, 0.0);
  bool finite_ok = true;

  const int F = 2 * D + 1;
  std::vector<int64_t> cur, nxt;
  for (int j = 0; j < M; ++j) {
    const kdehip_density &t = trees[j];
    const int64_t N = t.npts;
    auto valid = [N](int64_t id) { return id > 0 && id <= 2 * N; };  // BallTree01.jl:83
    cur.assign(1, 1);  // levelInit!: frontier = {root()}
    for (int l = 0; l <= L; ++l) {
      if (l > 0) {  // levelDown!, src/MSGibbs01.jl:503-511
        nxt.clear();
        for (int64_t node : cur) {
          const int64_t a = t.left_child[node - 1], b = t.right_child[node - 1];
          if (valid(a)) nxt.push_back(a);
          if (valid(b)) nxt.push_back(b);
        }
        if (nxt.empty() || static_cast<int64_t>(nxt.size()) > N)
          return set_error(KDEHIP_ERR_ARG, "malformed tree: frontier empty or larger than Npts");
        cur.swap(nxt);
        out.nodes_per_sweep += static_cast<int64_t>(cur.size());
      }
      const int64_t n = static_cast<int64_t>(cur.size());
      const int64_t B = (n + 63) / 64;
      const int64_t ld = B * 64;
      LevelDesc &ds = out.levels[static_cast<size_t>(j) * (L + 1) + l];
      ds.n = static_cast<int32_t>(n);
      ds.B = static_cast<int32_t>(B);
      ds.data_off = static_cast<int64_t>(out.data.size());
      ds.perm_off = static_cast<int64_t>(out.perm.size());
      ds.pad_ = 0;
      out.data.resize(out.data.size() + static_cast<size_t>(F * ld));
      out.perm.resize(out.perm.size() + static_cast<size_t>(ld), 0);
      double *tile = out.data.data() + ds.data_off;
      int32_t *prow = out.perm.data() + ds.perm_off;
      for (int64_t p = 0; p < ld; ++p) {  // padding: mean 0, variance 1, weight 0
        for (int d = 0; d < D; ++d) { tile[d * ld + p] = 0.0; tile[(D + d) * ld + p] = 1.0; }
        tile[2 * D * ld + p] = 0.0;
      }
      bool uni = true;
      for (int64_t z = 0; z < n; ++z) {
        const int64_t node = cur[static_cast<size_t>(z)];
        if (!valid(node)) return set_error(KDEHIP_ERR_ARG, "malformed tree: child id out of range");
        const int64_t p = (z % B) * 64 + z / B;
        for (int d = 0; d < D; ++d) {
          const double mu = t.means[(node - 1) * D + d];
          const double v = t.bandwidth[(node - 1) * D + d];
          tile[d * ld + p] = mu;
          tile[(D + d) * ld + p] = v;
          if (v != t.bandwidth[(cur[0] - 1) * D + d]) uni = false;
          if (!(std::isfinite(mu) && std::isfinite(v) && v > 0.0)) finite_ok = false;
          if (v < bw_lo[d]) bw_lo[d] = v;
          if (v > bw_hi[d]) bw_hi[d] = v;
        }
        const double w = t.weights[node - 1];
        if (!(std::isfinite(w) && w >= 0.0)) finite_ok = false;
        tile[2 * D * ld + p] = w;
        prow[p] = static_cast<int32_t>(t.permutation[node - 1]);
      }
      ds.uniform_bw = uni ? 1 : 0;
    }
  }

  // The product/rsqrt form multiplies up to D variances c_d in [bw_lo, 2*bw_hi] (bandwidth plus a
  // leave-one-out product variance that is never larger than the largest bandwidth).  It is used
  // only when no partial product can leave the comfortable range of T; otherwise, and for masked
  // products, the per-dimension divide+log form (the reference's own arithmetic) runs.
  double up = 1.0, dn = 1.0;
  for (int d = 0; d < D; ++d) {
    const double hi = 2.0 * bw_hi[d], lo = bw_lo[d];
    if (hi > 1.0) up *= hi;
    if (lo < 1.0) dn *= lo;
  }
  const bool range64 = finite_ok && up < 1e120 && dn > 1e-120;
  const bool range32 = finite_ok && up < 1e15 && dn > 1e-15;
  out.fast_ok_f64 = range64 && !out.masked;
  out.fast_ok_f32 = range32 && !out.masked;
  return KDEHIP_OK;
}

}  // namespace kdehip
