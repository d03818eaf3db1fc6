// pack_device.hip -- densities that LIVE in HBM, and the per-product re-layout ("pack_levels") done by the GPU.
//
// The reference hands `gibbs1` host arrays (src/MSGibbs01.jl:527-537) and a drop-in caller does the same
// (kdehip_gibbs1 / kdehip_prod_philox: pack on the host, one upload).  A caller that keeps its densities on the device
// -- the inputs of the next product are the outputs of the previous ones -- does not want the 1 MB of tree arrays to
// cross PCIe and the host to spend 0.1 ms re-laying them out for every product.  A kdehip_device_density holds, in ONE
// device block, what the sampler's tiles are made of: means, bandwidth (variances), weights, permutation
// (src/BallTreeDensity01.jl:11-24, src/BallTree01.jl:10-28) and the frontier of every level (the node ids levelDown!
// visits, src/MSGibbs01.jl:500-523; they depend on the tree only, so they are expanded once per density, not once
// per product).  kdehip_prod_philox_device then lays a product out from shapes alone (pack_layout_shapes), uploads
// the few KB of descriptors, and one gather kernel writes the tiles straight into the plan's device image --
// the same bytes the host packer (pack_levels.cpp pack_fill) writes, tested bit for bit.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "device_density.hpp"
#include "host_pool.hpp"
#include "kdehip_internal.hpp"
#include "loocv_search.hpp"

using namespace kdehip;

#define KDEHIP_CHECK(expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return set_error(KDEHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

namespace kdehip {

// One wavefront per (tile, row): lane ln writes entry z = ln*B + row of the frontier (kdehip_internal.hpp "packed
// per-level layout"), field by field -- 64 contiguous elements per store (fp32: every other element of a row pair's
// 128) -- from the density's arrays in HBM.
template <typename T>
__global__ __launch_bounds__(64) void fill_tiles_kernel(const FillJob *__restrict__ jobs) {
  const FillJob job = jobs[blockIdx.y];
  const int row = blockIdx.x;
  if (row >= job.B) return;
  const int lane = threadIdx.x;
  const int D = job.D, F = job.F;
  using TA = TileAddr<T>;
  const int64_t RS = TA::stride(F);
  T *hdr = static_cast<T *>(job.hdr);
  const int32_t *front = job.front;
  const double *means = job.means, *bw = job.bandwidth;
  if (row == 0 && lane < kTileHeader) {  // the bandwidth vector of entry 0 (= of every entry of a uniform tile)
    const int64_t n0 = static_cast<int64_t>(front[0]) - 1;
    hdr[lane] = lane < D ? static_cast<T>(bw[n0 * D + lane]) : T(0);
  }
  const int64_t z = static_cast<int64_t>(lane) * job.B + row;
  const int64_t src = z < job.n ? static_cast<int64_t>(front[z]) - 1 : -1;
  T *r = hdr + kTileHeader + TA::row(static_cast<int64_t>(row), RS) + lane * TA::kLane;  // (row, field 0, this lane); field f at r[f * kField]
  for (int d = 0; d < D; ++d) r[d * TA::kField] = src >= 0 ? static_cast<T>(means[src * D + d]) : T(0);
  if (!job.uniform)
    for (int d = 0; d < D; ++d) r[(D + d) * TA::kField] = src >= 0 ? static_cast<T>(bw[src * D + d]) : T(1);
  r[(F - 1) * TA::kField] = src >= 0 ? static_cast<T>(job.weights[src]) : T(0);
  if (lane == 0) r[F * TA::kField] = T(0);  // the pad element
  if (TA::kPaired && row == job.B - 1 && (job.B & 1)) {  // fp32: the missing second row of the last pair = padding entries
    T *q = r + 1;
    for (int d = 0; d < D; ++d) q[d * TA::kField] = T(0);
    if (!job.uniform)
      for (int d = 0; d < D; ++d) q[(D + d) * TA::kField] = T(1);
    q[(F - 1) * TA::kField] = T(0);
    if (lane == 0) q[F * TA::kField] = T(0);
  }
  job.perm_out[static_cast<int64_t>(row) * 64 + lane] = src >= 0 ? static_cast<int32_t>(job.perm[src]) : 0;
}

int launch_fill_tiles(int precision, const FillJob *d_jobs, int ntiles, int maxB, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  for (int t0 = 0; t0 < ntiles; t0 += 65535) {  // (grid.y limit)
    const int nt = ntiles - t0 < 65535 ? ntiles - t0 : 65535;
    const dim3 grid(static_cast<unsigned>(maxB), static_cast<unsigned>(nt));
    if (precision == 64) hipLaunchKernelGGL(fill_tiles_kernel<double>, grid, dim3(64), 0, stream, d_jobs + t0);
    else hipLaunchKernelGGL(fill_tiles_kernel<float>, grid, dim3(64), 0, stream, d_jobs + t0);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("tile fill launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

// ---- fp32 screen tiles (kdehip_internal.hpp "fp32 screening") ------------------------------------------------------
// One workgroup per (density j, level l) of an fp64 plan; those whose level is screened re-write their fp64 tile as the
// fp32 screen tile: means centred at the density's root mean (subtracted in fp64, rounded to nearest once), variances,
// weights, in the fp32 row-pair layout; and the header the sampler's error bound is made of -- mu0, the smallest variance
// per dimension, the largest |m'_d| (rounded up), and whether every value lies inside the ranges that bound assumes.
__global__ __launch_bounds__(256) void screen_build_kernel(PlanDev plan) {
  const int M = plan.M, L = plan.L, D = plan.D;
  const int idx = blockIdx.x;  // j * (L + 1) + l
  const LevelDesc sc = plan.levels[M * (L + 1) + idx];
  if (sc.stage_mode < kStageScreen) return;
  const LevelDesc ds = plan.levels[idx];
  const int j = idx / (L + 1);
  const LevelDesc root = plan.levels[j * (L + 1)];
  const double *data = static_cast<const double *>(plan.data);
  const double *src = data + ds.hdr_off;                   // fp64 tile: header, then rows
  float *dst = reinterpret_cast<float *>(const_cast<double *>(data)) + sc.hdr_off;
  using TA8 = TileAddrBytes<8>;
  using TA4 = TileAddrBytes<4>;
  const int F = ds.F, B = ds.B, n = ds.n;
  const int RS8 = TA8::stride(F), RS4 = TA4::stride(F);
  const bool uni = ds.uniform_bw != 0;
  __shared__ double s_mu0[KDEHIP_MAX_DIMS];
  __shared__ float s_mmax[4][KDEHIP_MAX_DIMS], s_vmin[4][KDEHIP_MAX_DIMS], s_vmax[4][KDEHIP_MAX_DIMS];
  __shared__ int s_bad[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < KDEHIP_MAX_DIMS) s_mu0[t] = t < D ? (data + root.hdr_off + kTileHeader)[t * TA8::kField] : 0.0;  // root = entry 0 of the level-0 tile
  __syncthreads();
  float mmax[KDEHIP_MAX_DIMS], vmin[KDEHIP_MAX_DIMS], vmax[KDEHIP_MAX_DIMS];
  for (int d = 0; d < KDEHIP_MAX_DIMS; ++d) { mmax[d] = 0.0f; vmin[d] = INFINITY; vmax[d] = 0.0f; }
  int bad = 0;
  float *rows = dst + kScreenHeaderFloats;
  const int Bp = (B + 1) & ~1;  // rows of whole pairs: the missing second row of the last pair is padding
  for (int i = wave; i < Bp; i += 4) {
    const int z = lane * B + i;
    const bool real = i < B && z < n;
    const double *e8 = src + kTileHeader + TA8::row(i, RS8) + lane;
    float *e4 = rows + TA4::row(i, RS4) + lane * TA4::kLane;
    for (int d = 0; d < D; ++d) {
      float m = 0.0f;
      if (real) {
        const double mc = e8[d * TA8::kField] - s_mu0[d];
        m = static_cast<float>(mc);
        const float am = fabsf(m);
        bad |= !(am <= kScreenMaxAbsMean);
        mmax[d] = am > mmax[d] ? am : mmax[d];
      }
      e4[d * TA4::kField] = m;
    }
    if (!uni)
      for (int d = 0; d < D; ++d) {
        float v = 1.0f;
        if (real) {
          const double vd = e8[(D + d) * TA8::kField];
          v = static_cast<float>(vd);
          bad |= !(vd >= kScreenMinVar && vd <= kScreenMaxVar);
          vmin[d] = v < vmin[d] ? v : vmin[d];
          vmax[d] = v > vmax[d] ? v : vmax[d];
        }
        e4[(D + d) * TA4::kField] = v;
      }
    float w = 0.0f;
    if (real) {
      const double wd = e8[(F - 1) * TA8::kField];
      w = static_cast<float>(wd);
      bad |= !(wd >= 0.0 && wd <= 2.0);
    }
    e4[(F - 1) * TA4::kField] = w;
    if (lane == 0) e4[F * TA4::kField] = 0.0f;  // the pad element of this row's half of the pair
  }
  // reduce over the workgroup: lanes by DPP-free shuffles (this kernel runs once per plan)
  for (int d = 0; d < D; ++d)
    for (int o = 32; o > 0; o >>= 1) {
      mmax[d] = fmaxf(mmax[d], __shfl_xor(mmax[d], o));
      vmin[d] = fminf(vmin[d], __shfl_xor(vmin[d], o));
      vmax[d] = fmaxf(vmax[d], __shfl_xor(vmax[d], o));
    }
  bad = __any(bad) ? 1 : 0;
  if (lane == 0) {
    for (int d = 0; d < KDEHIP_MAX_DIMS; ++d) { s_mmax[wave][d] = mmax[d]; s_vmin[wave][d] = vmin[d]; s_vmax[wave][d] = vmax[d]; }
    s_bad[wave] = bad;
  }
  __syncthreads();
  if (t < KDEHIP_MAX_DIMS) {
    const int d = t;
    float mm = 0.0f, lo = INFINITY;
    for (int w = 0; w < 4; ++w) { mm = fmaxf(mm, s_mmax[w][d]); lo = fminf(lo, s_vmin[w][d]); }
    double cmin = static_cast<double>(lo);
    int b = s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3];
    if (uni && d < D) {  // the one bandwidth vector of a shared-bandwidth tile: its header
      cmin = src[d];
      b |= !(cmin >= kScreenMinVar && cmin <= kScreenMaxVar);
    }
    if (d >= D) { cmin = 1.0; mm = 0.0f; }
    // (a variance that was rounded to fp32 may sit one ulp ABOVE the fp64 value: the bound's smallest variance steps down)
    if (!uni && d < D) cmin = cmin * (1.0 - 1.2e-7);
    double *h8 = reinterpret_cast<double *>(dst);
    h8[d] = s_mu0[d];
    h8[8 + d] = cmin;
    dst[32 + d] = mm * (1.0f + 2.4e-7f);
    // flags[0]: 1 = every value inside the ranges the error analysis assumes (the other flags: 0)
    const unsigned long long anybad = __ballot(b != 0);
    dst[40 + d] = (d == 0 && anybad == 0ull) ? 1.0f : 0.0f;
  }
}

int launch_screen_build(const PlanDev &plan, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(screen_build_kernel, dim3(static_cast<unsigned>(plan.M * (plan.L + 1))), dim3(256), 0, stream, plan);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("screen tile build launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

}  // namespace kdehip

// ---- densities in HBM ----------------------------------------------------------------------------------------------

namespace {

// The one device block of a density: [means | bandwidth | weights | permutation | frontier ids].
struct BlockLayout {
  size_t o_mean = 0, o_bw = 0, o_w = 0, o_perm = 0, o_front = 0, total = 0, nd = 0, n2 = 0;
  BlockLayout(int64_t N, int64_t D, size_t nfront) {
    auto al = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
    nd = sizeof(double) * 2 * N * D;
    n2 = sizeof(double) * 2 * N;
    o_mean = 0; o_bw = al(o_mean + nd); o_w = al(o_bw + nd); o_perm = al(o_w + n2);
    o_front = al(o_perm + sizeof(int64_t) * 2 * N);
    total = al(o_front + sizeof(int32_t) * nfront);
  }
};
void bind_block(kdehip_device_density *h, const BlockLayout &bl) {
  unsigned char *db = static_cast<unsigned char *>(h->d_blob);
  h->means = reinterpret_cast<const double *>(db + bl.o_mean);
  h->bandwidth = reinterpret_cast<const double *>(db + bl.o_bw);
  h->weights = reinterpret_cast<const double *>(db + bl.o_w);
  h->perm = reinterpret_cast<const int64_t *>(db + bl.o_perm);
  h->front = reinterpret_cast<const int32_t *>(db + bl.o_front);
  std::vector<int32_t>().swap(h->fr.ids);  // (the ids live on the device now; sizes, offsets and flags stay)
  std::vector<uint8_t>().swap(h->fr.fresh);
}

// Frontiers, the arithmetic-form examination and the one device block of a density whose six arrays `host` describes
// (h->device, N, D, Lown are set; the device is current).  Everything travels on `st`, which is waited for.
int upload_common(kdehip_device_density *h, const kdehip_density &host, hipStream_t st) {
  const int64_t N = h->N, D = h->D;
  int rc = expand_frontiers(host, h->D, h->Lown, /*look=*/true, h->fr);
  if (rc != KDEHIP_OK) return rc;
  const BlockLayout bl(N, D, h->fr.ids.size());
  const size_t total = bl.total;
  void *pin = nullptr;
  hipError_t e = cached_host_malloc(&pin, total);
  if (e == hipSuccess) e = cached_malloc(&h->d_blob, total);
  if (e != hipSuccess) {
    if (pin) cached_host_free(pin, total);
    h->d_blob = nullptr;
    return set_error(KDEHIP_ERR_HIP, std::string("density upload: ") + hipGetErrorString(e));
  }
  h->blob_bytes = total;
  unsigned char *hp = static_cast<unsigned char *>(pin);
  std::memcpy(hp + bl.o_mean, host.means, bl.nd);
  std::memcpy(hp + bl.o_bw, host.bandwidth, bl.nd);
  std::memcpy(hp + bl.o_w, host.weights, bl.n2);
  std::memcpy(hp + bl.o_perm, host.permutation, sizeof(int64_t) * 2 * N);
  std::memcpy(hp + bl.o_front, h->fr.ids.data(), sizeof(int32_t) * h->fr.ids.size());
  e = hipMemcpyAsync(h->d_blob, pin, total, hipMemcpyHostToDevice, st);
  const hipError_t se = hipStreamSynchronize(st);
  cached_host_free(pin, total);
  if (e != hipSuccess || se != hipSuccess) {
    cached_free(h->d_blob, total);
    h->d_blob = nullptr;
    return set_error(KDEHIP_ERR_HIP, std::string("density upload: ") + hipGetErrorString(e != hipSuccess ? e : se));
  }
  bind_block(h, bl);
  return KDEHIP_OK;
}

// A stream of the calling thread's own beside hipStreamPerThread, per device (created on first use, never destroyed --
// like the per-thread stream itself): copies that must not queue behind the work already on the thread's stream.
hipStream_t side_stream(int device) {
  static thread_local hipStream_t streams[64] = {};
  static thread_local bool tried[64] = {};
  if (device < 0 || device >= 64) return nullptr;
  if (!tried[device]) {
    tried[device] = true;
    if (hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) {
      streams[device] = nullptr;
      (void)hipGetLastError();
    }
  }
  return streams[device];
}

int check_shape(int64_t N, int64_t D) {
  if (N < 1) return set_error(KDEHIP_ERR_ARG, "density with no points");
  if (N > (int64_t(1) << 30)) return set_error(KDEHIP_ERR_UNSUPPORTED, "density too large");
  if (D < 1 || D > KDEHIP_MAX_DIMS) return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  return KDEHIP_OK;
}

// getPoints (src/KDE01.jl:91-101) of a resident density: the leaves back in the caller's original column order
__global__ void unpermute_points_kernel(const double *__restrict__ means, const int64_t *__restrict__ perm, int64_t N, int D,
                                        double *__restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= N * D) return;
  const int64_t i = t / D;
  const int k = static_cast<int>(t - i * D);
  out[(perm[N + i] - 1) * D + k] = means[(N + i) * D + k];
}

}  // namespace

extern "C" int kdehip_density_upload(kdehip_device_density **out, const kdehip_density *host, int device) {
  if (!out) return set_error(KDEHIP_ERR_ARG, "null out pointer");
  *out = nullptr;
  if (!host) return set_error(KDEHIP_ERR_ARG, "null density");
  const int64_t N = host->npts, D = host->ndim;
  int rc = check_shape(N, D);
  if (rc != KDEHIP_OK) return rc;
  if (!host->means || !host->bandwidth || !host->weights || !host->left_child || !host->right_child || !host->permutation)
    return set_error(KDEHIP_ERR_ARG, "density with a null array");
  kdehip_device_density *h = new (std::nothrow) kdehip_device_density();
  if (!h) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
  h->device = device;
  h->N = N;
  h->D = static_cast<int>(D);
  h->Lown = nlevels_for(N);
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc == KDEHIP_OK) rc = upload_common(h, *host, hipStreamPerThread);
  if (rc != KDEHIP_OK) { delete h; return rc; }
  *out = h;
  return KDEHIP_OK;
}

// ---- the resident chain: kde!(pGM) of a product that is still in HBM, and `*` on handles --------------------------
// src/MSGibbs01.jl:724-725: `pGM, = prodAppxMSGibbsS(...); return kde!(pGM)` -- the output of one product is the input
// of the next ones (every message of a belief-propagation sweep).  With kdehip_prod_philox_device the sample matrix
// never leaves the device; this entry turns it into the next product's input there: the LOOCV bandwidth search
// (src/KDE01.jl:3-27) reads the device matrix as it is, the ball tree (src/BallTree01.jl:415-434: sequential
// quick-selects, faster on the host's pooled builder than on the GPU for ONE density -- DESIGN.md f-3) is built from one
// D x N copy that goes down while the search runs, and the density's block goes straight back up.  PCIe traffic per
// link: 8*D*N bytes down, the density's block up; the caller's host never sees either.
extern "C" int kdehip_density_from_device_points(kdehip_device_density **out, const double *d_points, int64_t D,
                                                 int64_t N, int device, void *stream, double *bw_out, int32_t *nevals) {
  if (!out) return set_error(KDEHIP_ERR_ARG, "null out pointer");
  *out = nullptr;
  if (!d_points) return set_error(KDEHIP_ERR_ARG, "null points");
  int rc = check_shape(N, D);
  if (rc != KDEHIP_OK) return rc;
  if (N < 2) return set_error(KDEHIP_ERR_ARG, "kde!(points) needs at least two points");
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;
  static const bool timing = std::getenv("KDEHIP_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
  hipStream_t cs = hipStreamPerThread, ps = static_cast<hipStream_t>(stream);
  if (ps != cs) KDEHIP_CHECK(hipStreamSynchronize(ps));  // the producer of d_points (blocking entry: the host waits anyway)
  kdehip_device_density *h = new (std::nothrow) kdehip_device_density();
  if (!h) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
  struct Cleanup {
    kdehip_device_density *h; void *pin = nullptr; size_t pin_bytes = 0;
    hipStream_t s1 = nullptr, s2 = nullptr;
    ~Cleanup() {  // (error paths: nothing may still be in flight into or out of the blocks that go back to the caches)
      if (s1) (void)hipStreamSynchronize(s1);
      if (s2) (void)hipStreamSynchronize(s2);
      if (pin) cached_host_free(pin, pin_bytes);
      if (h && h->d_blob) cached_free(h->d_blob, h->blob_bytes);
      if (h && h->mirror) cached_host_free(h->mirror, h->mirror_bytes);
      delete h;
    }
  } cl{h};
  h->device = device; h->N = N; h->D = static_cast<int>(D); h->Lown = nlevels_for(N);
  cl.pin_bytes = sizeof(double) * N * D;
  KDEHIP_CHECK(cached_host_malloc(&cl.pin, cl.pin_bytes));
  // The matrix comes down on a stream of its own, behind an event on the thread's stream (the product that made it may
  // still be running there): with marginals the device prepares itself (N <= kLoocvPrepMaxN) the bandwidth search is
  // enqueued right behind that event too and needs no host copy -- the copy, the tree, the frontiers and the upload of
  // everything that does not depend on the bandwidth then all run UNDER the search.
  hipStream_t xs = side_stream(device);
  const bool under = xs != nullptr && N <= kLoocvPrepMaxN;
  cl.s1 = cs; cl.s2 = xs;
  if (under) {
    hipEvent_t ev = nullptr;
    KDEHIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, cs);
    if (e == hipSuccess) e = hipStreamWaitEvent(xs, ev, 0);
    (void)hipEventDestroy(ev);  // (released by the runtime once the wait has been served)
    if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("kdehip_density_from_device_points: ") + hipGetErrorString(e));
    KDEHIP_CHECK(hipMemcpyAsync(cl.pin, d_points, cl.pin_bytes, hipMemcpyDeviceToHost, xs));
  } else {
    KDEHIP_CHECK(hipMemcpyAsync(cl.pin, d_points, cl.pin_bytes, hipMemcpyDeviceToHost, cs));
    KDEHIP_CHECK(hipStreamSynchronize(cs));
  }
  const double *pts = static_cast<const double *>(cl.pin);
  // The reference's twelve arrays, kept as the handle's host mirror (kdehip_density_download), in one pinned block whose
  // head is the image of the device block (device_density.hpp).  The frontier ids of a tree of N leaves number at most
  // (Lown + 1) * N; the layout reserves that much.
  const size_t nd = static_cast<size_t>(2 * N * D), n2 = static_cast<size_t>(2 * N);
  const BlockLayout bl(N, D, static_cast<size_t>(h->Lown + 1) * static_cast<size_t>(N) + 64);
  auto al = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t o_extra = bl.total;  // centers, ranges (nd each), bwmin, bwmax (nd / 2 each), left, right, lowest, highest (n2 each)
  h->mirror_bytes = al(o_extra + sizeof(double) * (2 * nd + nd) + sizeof(int64_t) * 4 * n2);
  KDEHIP_CHECK(cached_host_malloc(&h->mirror, h->mirror_bytes));
  unsigned char *mb = static_cast<unsigned char *>(h->mirror);
  double *means = reinterpret_cast<double *>(mb + bl.o_mean), *bandwidth = reinterpret_cast<double *>(mb + bl.o_bw);
  double *weights = reinterpret_cast<double *>(mb + bl.o_w);
  int64_t *perm = reinterpret_cast<int64_t *>(mb + bl.o_perm);
  double *centers = reinterpret_cast<double *>(mb + o_extra), *ranges = centers + nd, *bwmin = ranges + nd, *bwmax = bwmin + nd / 2;
  int64_t *left = reinterpret_cast<int64_t *>(bwmax + nd / 2), *right = left + n2, *lowest = right + n2, *highest = lowest + n2;
  h->m = {centers, ranges, means, bandwidth, bwmin, bwmax, weights, left, right, lowest, highest, perm};
  const kdehip_density host{N, D, means, bandwidth, weights, left, right, perm};
  double bw[KDEHIP_MAX_DIMS];
  int tree_rc = KDEHIP_OK, side_rc = KDEHIP_OK;
  std::string side_msg;
  double us_tree = 0.0, us_side = 0.0;
  const double one = 1.0;  // (placeholder: only `bandwidth`, bandwidthMin/Max depend on the bandwidth)
  // what needs the host but not the bandwidth: the tree (topology, bounding boxes, weights, means), the frontier ids,
  // the device block and the upload of everything in it but the variances
  std::vector<int64_t> order;
  auto host_side_body = [&]() {
    if (hipStreamSynchronize(xs) != hipSuccess) { side_rc = KDEHIP_ERR_HIP; side_msg = "the copy of the points failed"; return; }
    tree_rc = kdehip_make_density(D, N, pts, &one, 1, nullptr, centers, ranges, weights, left, right, lowest, highest, perm,
                                  means, bandwidth, bwmin, bwmax);
    us_tree = us();
    if (tree_rc != KDEHIP_OK) return;
    side_rc = expand_frontier_ids(host, h->D, h->Lown, h->fr);
    if (side_rc != KDEHIP_OK) { side_msg = kdehip_last_error(); return; }
    side_rc = children_first_order(N, left, right, order);  // (the order the variances are filled in: topology only)
    if (side_rc != KDEHIP_OK) { side_msg = kdehip_last_error(); return; }
    if (h->fr.ids.size() > static_cast<size_t>(h->Lown + 1) * static_cast<size_t>(N) + 64) {
      side_rc = KDEHIP_ERR_ARG; side_msg = "more frontier ids than a tree of N leaves has"; return;
    }
    hipError_t e = cached_malloc(&h->d_blob, bl.total);
    if (e == hipSuccess) h->blob_bytes = bl.total; else h->d_blob = nullptr;
    if (e == hipSuccess) {  // everything but the variances, straight from the mirror (pinned): no staging copy
      std::memcpy(mb + bl.o_front, h->fr.ids.data(), sizeof(int32_t) * h->fr.ids.size());
      unsigned char *db = static_cast<unsigned char *>(h->d_blob);
      e = hipMemcpyAsync(db + bl.o_mean, mb + bl.o_mean, bl.nd, hipMemcpyHostToDevice, xs);
      if (e == hipSuccess)
        e = hipMemcpyAsync(db + bl.o_w, mb + bl.o_w, bl.o_front + sizeof(int32_t) * h->fr.ids.size() - bl.o_w, hipMemcpyHostToDevice, xs);
    }
    if (e != hipSuccess) { side_rc = KDEHIP_ERR_HIP; side_msg = std::string("density block: ") + hipGetErrorString(e); }
    us_side = us();
  };
  // (nothing may throw out of an extern "C" entry point, nor through the search's stack: vector growth in the frontier
  // expansion is reported like any other failure of the host side -- ADVICE round 5)
  auto host_side = [&]() {
    try { host_side_body(); }
    catch (const std::exception &e) { side_rc = KDEHIP_ERR_ALLOC; side_msg = e.what(); }
  };
  if (under) {
    try {
      const std::function<void()> fn = host_side;
      rc = auto_bandwidth_run(static_cast<int>(D), N, nullptr, d_points, cs, bw, nevals, &fn);
    } catch (const std::exception &e) {
      return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_density_from_device_points: ") + e.what());
    }
    if (rc != KDEHIP_OK) return rc;
  } else {
    // (larger marginals are prepared on the host: the search needs the copy first; the tree still builds under it)
    try {
      TaskGroup group(HostPool::get());
      group.run([&] {
        tree_rc = kdehip_make_density(D, N, pts, &one, 1, nullptr, centers, ranges, weights, left, right, lowest, highest,
                                      perm, means, bandwidth, bwmin, bwmax);
      });
      rc = auto_bandwidth_run(static_cast<int>(D), N, pts, d_points, cs, bw, nevals);
      group.wait();
    } catch (const std::exception &e) {
      return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_density_from_device_points: ") + e.what());
    }
    if (rc != KDEHIP_OK) return rc;
  }
  const double us_search = us();
  if (tree_rc != KDEHIP_OK) return set_error(tree_rc, "kdehip_density_from_device_points: the tree build failed");
  if (side_rc != KDEHIP_OK) return set_error(side_rc, "kdehip_density_from_device_points: " + side_msg);
  double us_setbw = 0.0, us_exam = 0.0, us_copy = 0.0;
  for (int k = 0; k < D; ++k) { h->bw[k] = bw[k]; if (bw_out) bw_out[k] = bw[k]; }
  h->built = true;
  if (under) {
    // the bandwidth-dependent rest: the variances (and, on the values it walks through, what the packers' examination of
    // the nodes needs), then their upload from the pinned mirror with the per-level flags formed under the transfer
    NodeStats stats{};
    rc = set_bandwidth_examined(D, N, bw, D, weights, left, right, means, bandwidth, bwmin, bwmax, &stats, &order);
    if (rc != KDEHIP_OK) return rc;
    us_setbw = us();
    unsigned char *db = static_cast<unsigned char *>(h->d_blob);
    KDEHIP_CHECK(hipMemcpyAsync(db + bl.o_bw, mb + bl.o_bw, bl.nd, hipMemcpyHostToDevice, xs));  // (from the pinned mirror)
    us_copy = us();
    examine_frontiers(host, h->D, h->Lown, /*look=*/false, h->fr);
    h->fr.bad = stats.bad;
    for (int k = 0; k < KDEHIP_MAX_DIMS; ++k) { h->fr.lo[k] = stats.lo[k]; h->fr.hi[k] = stats.hi[k]; }
    us_exam = us();
    KDEHIP_CHECK(hipStreamSynchronize(xs));
    bind_block(h, bl);
  } else {
    rc = kdehip_density_set_bandwidth(D, N, bw, D, weights, left, right, means, bandwidth, bwmin, bwmax);
    if (rc != KDEHIP_OK) return rc;
    rc = upload_common(h, host, cs);
    if (rc != KDEHIP_OK) return rc;
  }
  if (timing)
    std::fprintf(stderr, "kdehip_density_from_device_points D=%lld N=%lld: tree built at %.0f us | block prepared at %.0f us | search over at %.0f us | variances %.0f | flags %.0f | staged %.0f | done at %.0f us\n",
                 static_cast<long long>(D), static_cast<long long>(N), us_tree, us_side, us_search, us_setbw, us_exam, us_copy, us());
  cl.s1 = cl.s2 = nullptr;  // (everything has been waited for)
  cl.h = nullptr;
  *out = h;
  return KDEHIP_OK;
}

// `*(trees; addEntropy)` (src/MSGibbs01.jl:707-726) on handles: Np = round(mean Npts), Niter = 5, then kde!(pGM) -- and
// the "hack fix for #70" (:713-716: one density, no entropy -> kde! of its own points).
extern "C" int kdehip_mul_device(kdehip_device_density **out, int Ndens, kdehip_device_density *const *trees, uint64_t seed,
                                 int addEntropy, double *bw_out, int32_t *nevals) {
  if (!out) return set_error(KDEHIP_ERR_ARG, "null out pointer");
  *out = nullptr;
  if (Ndens < 1 || !trees) return set_error(KDEHIP_ERR_ARG, "need at least one density");
  for (int j = 0; j < Ndens; ++j) {
    if (!trees[j]) return set_error(KDEHIP_ERR_ARG, "null density");
    if (trees[j]->D != trees[0]->D) return set_error(KDEHIP_ERR_DIM_MISMATCH, "kdes must have same dimension");
    if (trees[j]->device != trees[0]->device) return set_error(KDEHIP_ERR_ARG, "densities on different devices");
  }
  const int D = trees[0]->D, device = trees[0]->device;
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;
  hipStream_t cs = hipStreamPerThread;
  struct Scratch {
    void *p = nullptr; size_t n = 0; hipStream_t st;
    ~Scratch() { if (p) { (void)hipStreamSynchronize(st); cached_free(p, n); } }
  } sc;
  sc.st = cs;
  if (Ndens == 1 && !addEntropy) {
    const int64_t N = trees[0]->N;
    sc.n = sizeof(double) * N * D;
    KDEHIP_CHECK(cached_malloc(&sc.p, sc.n));
    const int64_t items = N * D;
    hipLaunchKernelGGL(unpermute_points_kernel, dim3(static_cast<unsigned>((items + 255) / 256)), dim3(256), 0, cs,
                       trees[0]->means, trees[0]->perm, N, D, static_cast<double *>(sc.p));
    KDEHIP_CHECK(hipGetLastError());
    return kdehip_density_from_device_points(out, static_cast<const double *>(sc.p), D, N, device, cs, bw_out, nevals);
  }
  double sum = 0.0;  // numpts = round(Int, mean(Npts.(trees))): Julia rounds halves to even, like nearbyint
  for (int j = 0; j < Ndens; ++j) sum += static_cast<double>(trees[j]->N);
  const int64_t Np = static_cast<int64_t>(std::nearbyint(sum / static_cast<double>(Ndens)));
  auto al = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t off_i = al(sizeof(double) * Np * D);
  sc.n = off_i + sizeof(int64_t) * Np * Ndens;
  KDEHIP_CHECK(cached_malloc(&sc.p, sc.n));
  double *d_pts = static_cast<double *>(sc.p);
  int64_t *d_ind = reinterpret_cast<int64_t *>(static_cast<unsigned char *>(sc.p) + off_i);
  rc = prod_philox_device_blocking_stream(Ndens, trees, Np, /*Niter=*/5, seed, 0, addEntropy, nullptr, 64, d_pts, d_ind, cs);
  if (rc != KDEHIP_OK) return rc;
  return kdehip_density_from_device_points(out, d_pts, D, Np, device, cs, bw_out, nevals);
}

// ---- `*` for MANY products in one call ----------------------------------------------------------------------------
// The reference's `*` (src/MSGibbs01.jl:707-726) is what a belief-propagation host calls dozens of times per sweep, on
// densities of 100-300 points (test/runtests.jl:189-201): product, then kde!(pGM) = LOOCV bandwidth per dimension
// (src/KDE01.jl:3-27, src/CrossValidation.jl:44-120) + ball tree (src/BallTree01.jl:415-434).  One at a time each is a
// blocking call of >= 10 dependent launches that fill a fraction of the device.  Here ALL products are sampled by the batched
// sampler (kdehip_prod_philox_batch: one launch per (dimension count, density count) group) into one scratch block; the
// bandwidth searches of ALL outputs of one size advance in the SAME launches (LoocvSearch: a launch indexes marginals, a
// batch of nb products of D dimensions is nb * D of them); the matrices come down in ONE copy and the nb trees are built by
// the pooled host builder UNDER the searches; the densities share one device block and one pinned mirror (two uploads for
// the batch).  Every result is, bit for bit, what kdehip_mul_device returns for the same item.
namespace {

struct MulPlan {   // what one item of the batch becomes
  int D = 0, M = 0;
  int64_t N = 0;          // points of the result (Np of the product, or the density's own count for the shortcut)
  bool shortcut = false, loose = false;  // loose: outside the batched path (fewer than 2 or more than 2048 points): a call of its own
  int group = -1;                        // its (D, N) group
  size_t pts_off = 0, ind_off = 0;       // in the scratch block (bytes)
  size_t a_off = 0, b_off = 0, x_off = 0;  // in the shared block: region A (means, weights, permutation, ids), B (variances); mirror extras
  size_t front_cap = 0;
  kdehip_device_density *h = nullptr;
  std::vector<int64_t> order;
  int rc = KDEHIP_OK;
  std::string msg;
};
struct MulGroup { int D; int64_t N; std::vector<int> members; size_t pts_off = 0; LoocvSearch *search = nullptr; };

}  // namespace

static int mul_device_batch_impl(int nprod, const kdehip_mul_item *items, kdehip_device_density **out, double *bw_out,
                                 int32_t *nevals);
extern "C" int kdehip_mul_device_batch(int nprod, const kdehip_mul_item *items, kdehip_device_density **out, double *bw_out,
                                       int32_t *nevals) {
  // (the bookkeeping below lives in std::vectors: nothing may throw out of an extern "C" entry point -- the unwinding runs the
  // clean-up that takes every block and handle back first)
  try {
    return mul_device_batch_impl(nprod, items, out, bw_out, nevals);
  } catch (const std::exception &e) {
    if (out) for (int i = 0; i < nprod; ++i) out[i] = nullptr;
    return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_mul_device_batch: ") + e.what());
  }
}
static int mul_device_batch_impl(int nprod, const kdehip_mul_item *items, kdehip_device_density **out, double *bw_out,
                                 int32_t *nevals) {
  if (nprod < 0 || (nprod > 0 && (!items || !out))) return set_error(KDEHIP_ERR_ARG, "kdehip_mul_device_batch: bad item list");
  for (int i = 0; i < nprod; ++i) out[i] = nullptr;
  if (nprod == 0) return KDEHIP_OK;
  std::vector<MulPlan> mp(static_cast<size_t>(nprod));
  int device = -1;
  for (int i = 0; i < nprod; ++i) {
    const kdehip_mul_item &it = items[i];
    if (it.Ndens < 1 || !it.trees) return set_error(KDEHIP_ERR_ARG, "need at least one density");
    for (int j = 0; j < it.Ndens; ++j) {
      if (!it.trees[j]) return set_error(KDEHIP_ERR_ARG, "null density");
      if (it.trees[j]->D != it.trees[0]->D) return set_error(KDEHIP_ERR_DIM_MISMATCH, "kdes must have same dimension");
      if (device < 0) device = it.trees[j]->device;
      if (it.trees[j]->device != device) return set_error(KDEHIP_ERR_ARG, "densities on different devices");
    }
    MulPlan &m = mp[i];
    m.D = it.trees[0]->D; m.M = it.Ndens;
    m.shortcut = it.Ndens == 1 && !it.addEntropy;  // the "hack fix for #70" (:713-716)
    if (m.shortcut) m.N = it.trees[0]->N;
    else {
      double sum = 0.0;  // numpts = round(Int, mean(Npts.(trees))): halves to even, like nearbyint
      for (int j = 0; j < it.Ndens; ++j) sum += static_cast<double>(it.trees[j]->N);
      m.N = static_cast<int64_t>(std::nearbyint(sum / static_cast<double>(it.Ndens)));
    }
  }
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;
  hipStream_t cs = hipStreamPerThread, xs = side_stream(device);
  auto al = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };

  // groups of equal (D, N): their searches share launches and their matrices sit one behind the other
  std::vector<MulGroup> groups;
  for (int i = 0; i < nprod; ++i) {
    MulPlan &m = mp[i];
    m.loose = xs == nullptr || m.N < 2 || m.N > kLoocvPrepMaxN;
    if (m.loose) continue;
    int g = -1;
    const size_t cap = static_cast<size_t>(kLoocvMaxMarginals / m.D);  // (products per search: its launches index marginals)
    for (size_t k = 0; k < groups.size(); ++k)
      if (groups[k].D == m.D && groups[k].N == m.N && groups[k].members.size() < cap) g = static_cast<int>(k);
    if (g < 0) { groups.push_back(MulGroup{m.D, m.N, {}}); g = static_cast<int>(groups.size()) - 1; }
    m.group = g;
    groups[g].members.push_back(i);
  }
  size_t pts_bytes = 0, ind_bytes = 0, a_bytes = 0, b_bytes = 0, x_bytes = 0;
  for (MulGroup &g : groups) {
    g.pts_off = pts_bytes;
    for (int i : g.members) {
      MulPlan &m = mp[i];
      m.pts_off = pts_bytes; pts_bytes += sizeof(double) * m.N * m.D;  // (no padding inside a group: LoocvSearch strides by N * D)
    }
    pts_bytes = al(pts_bytes);
  }
  int nbatched = 0;
  for (int i = 0; i < nprod; ++i) {
    MulPlan &m = mp[i];
    if (m.loose) continue;
    ++nbatched;
    if (!m.shortcut) { m.ind_off = ind_bytes; ind_bytes = al(ind_bytes + sizeof(int64_t) * m.N * m.M); }
    const size_t nd = sizeof(double) * 2 * m.N * m.D, n2 = sizeof(double) * 2 * m.N;
    m.front_cap = static_cast<size_t>(nlevels_for(m.N) + 1) * static_cast<size_t>(m.N) + 64;
    m.a_off = a_bytes; a_bytes = al(a_bytes + al(nd) + al(n2) + al(n2) + sizeof(int32_t) * m.front_cap);
    m.b_off = b_bytes; b_bytes = al(b_bytes + nd);
    m.x_off = x_bytes; x_bytes = al(x_bytes + nd * 3 + sizeof(int64_t) * 4 * 2 * m.N);  // centers, ranges, bwmin + bwmax, left .. highest
  }

  // everything the error paths have to take back
  struct Cleanup {
    std::vector<MulPlan> *mp; std::vector<MulGroup> *groups;
    hipStream_t cs, xs;
    void *scratch = nullptr; size_t scratch_bytes = 0;
    void *pin = nullptr; size_t pin_bytes = 0;
    SharedBlock *sb = nullptr;
    bool keep = false;
    ~Cleanup() {
      for (MulGroup &g : *groups) if (g.search) loocv_delete(g.search);  // (an abandoned search waits for its stream)
      (void)hipStreamSynchronize(cs);
      if (xs) (void)hipStreamSynchronize(xs);
      if (scratch) cached_free(scratch, scratch_bytes);
      if (pin) cached_host_free(pin, pin_bytes);
      if (keep) return;
      for (MulPlan &m : *mp) { delete m.h; m.h = nullptr; }
      if (sb) {
        if (sb->d_blob) cached_free(sb->d_blob, sb->blob_bytes);
        if (sb->mirror) cached_host_free(sb->mirror, sb->mirror_bytes);
        delete sb;
      }
    }
  } cl{&mp, &groups, cs, xs};

  if (nbatched > 0) {
    cl.scratch_bytes = al(pts_bytes) + ind_bytes + 256;
    KDEHIP_CHECK(cached_malloc(&cl.scratch, cl.scratch_bytes));
    unsigned char *sc = static_cast<unsigned char *>(cl.scratch);
    // (1) the products: one batched call (its own groups by (D, M)); the shortcut items un-permute their own leaves
    std::vector<kdehip_batch_item> prod;
    for (int i = 0; i < nprod; ++i) {
      MulPlan &m = mp[i];
      if (m.loose) continue;
      double *d_pts = reinterpret_cast<double *>(sc + m.pts_off);
      if (m.shortcut) {
        const kdehip_device_density *t = items[i].trees[0];
        const int64_t n = m.N * m.D;
        hipLaunchKernelGGL(unpermute_points_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, cs, t->means, t->perm,
                           m.N, m.D, d_pts);
        KDEHIP_CHECK(hipGetLastError());
        continue;
      }
      kdehip_batch_item b{};
      b.Ndens = m.M; b.Niter = 5; b.trees = items[i].trees; b.Np = m.N; b.seed = items[i].seed; b.sample_offset = 0;
      b.addEntropy = items[i].addEntropy; b.partialDimMask = nullptr; b.d_points = d_pts;
      b.d_indices = reinterpret_cast<int64_t *>(sc + al(pts_bytes) + m.ind_off); b.d_labels = nullptr;
      prod.push_back(b);
    }
    if (!prod.empty()) {
      rc = kdehip_prod_philox_batch(static_cast<int>(prod.size()), prod.data(), 64, cs);
      if (rc != KDEHIP_OK) return rc;
    }
    // (2) the matrices come down in one copy on the side stream, behind the products
    cl.pin_bytes = pts_bytes;
    KDEHIP_CHECK(cached_host_malloc(&cl.pin, cl.pin_bytes));
    {
      hipEvent_t ev = nullptr;
      KDEHIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      hipError_t e = hipEventRecord(ev, cs);
      if (e == hipSuccess) e = hipStreamWaitEvent(xs, ev, 0);
      (void)hipEventDestroy(ev);
      if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("kdehip_mul_device_batch: ") + hipGetErrorString(e));
      KDEHIP_CHECK(hipMemcpyAsync(cl.pin, sc, pts_bytes, hipMemcpyDeviceToHost, xs));
    }
    // (3) the bandwidth searches: one per group, all of its marginals in the same launches, nothing needs the host
    for (MulGroup &g : groups) {
      g.search = loocv_new();
      if (!g.search) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
      rc = loocv_begin(g.search, static_cast<int>(g.members.size()), g.D, g.N, reinterpret_cast<const double *>(sc + g.pts_off), cs);
      if (rc != KDEHIP_OK) return rc;
    }
    // (4) UNDER the searches: handles, the shared blocks, and -- once the copy is down -- the trees (one pool task each)
    cl.sb = new (std::nothrow) SharedBlock();
    if (!cl.sb) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
    cl.sb->blob_bytes = a_bytes + b_bytes;
    cl.sb->mirror_bytes = a_bytes + b_bytes + x_bytes;
    KDEHIP_CHECK(cached_malloc(&cl.sb->d_blob, cl.sb->blob_bytes));
    KDEHIP_CHECK(cached_host_malloc(&cl.sb->mirror, cl.sb->mirror_bytes));
    unsigned char *mb = static_cast<unsigned char *>(cl.sb->mirror), *db = static_cast<unsigned char *>(cl.sb->d_blob);
    for (int i = 0; i < nprod; ++i) {
      MulPlan &m = mp[i];
      if (m.loose) continue;
      kdehip_device_density *h = m.h = new (std::nothrow) kdehip_device_density();
      if (!h) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
      h->device = device; h->N = m.N; h->D = m.D; h->Lown = nlevels_for(m.N);
      const size_t nd = sizeof(double) * 2 * m.N * m.D, n2 = sizeof(double) * 2 * m.N;
      const size_t o_w = al(nd), o_perm = o_w + al(n2), o_front = o_perm + al(n2);
      unsigned char *a = mb + m.a_off, *x = mb + a_bytes + b_bytes + m.x_off;
      kdehip_device_density::Mirror &q = h->m;
      q.means = reinterpret_cast<double *>(a); q.weights = reinterpret_cast<double *>(a + o_w);
      q.perm = reinterpret_cast<int64_t *>(a + o_perm);
      q.bandwidth = reinterpret_cast<double *>(mb + a_bytes + m.b_off);
      q.centers = reinterpret_cast<double *>(x); q.ranges = q.centers + nd / 8; q.bwmin = q.ranges + nd / 8; q.bwmax = q.bwmin + nd / 16;
      q.left = reinterpret_cast<int64_t *>(q.bwmax + nd / 16); q.right = q.left + 2 * m.N; q.lowest = q.right + 2 * m.N;
      q.highest = q.lowest + 2 * m.N;
      h->means = reinterpret_cast<const double *>(db + m.a_off);
      h->weights = reinterpret_cast<const double *>(db + m.a_off + o_w);
      h->perm = reinterpret_cast<const int64_t *>(db + m.a_off + o_perm);
      h->front = reinterpret_cast<const int32_t *>(db + m.a_off + o_front);
      h->bandwidth = reinterpret_cast<const double *>(db + a_bytes + m.b_off);
    }
    KDEHIP_CHECK(hipStreamSynchronize(xs));  // the matrices are down
    const unsigned char *pin = static_cast<const unsigned char *>(cl.pin);
    const double one = 1.0;  // (placeholder: only `bandwidth`, bandwidthMin/Max depend on the bandwidth)
    try {
      TaskGroup trees(HostPool::get());
      for (int i = 0; i < nprod; ++i) {
        MulPlan *m = &mp[i];
        if (m->loose) continue;
        trees.run([m, pin, mb, al, &one] {
          kdehip_device_density *h = m->h;
          kdehip_device_density::Mirror &q = h->m;
          m->rc = kdehip_make_density(m->D, m->N, reinterpret_cast<const double *>(pin + m->pts_off), &one, 1, nullptr, q.centers, q.ranges,
                                      q.weights, q.left, q.right, q.lowest, q.highest, q.perm, q.means, q.bandwidth, q.bwmin, q.bwmax);
          const kdehip_density host{m->N, m->D, q.means, q.bandwidth, q.weights, q.left, q.right, q.perm};
          if (m->rc == KDEHIP_OK) m->rc = expand_frontier_ids(host, h->D, h->Lown, h->fr);
          if (m->rc == KDEHIP_OK) m->rc = children_first_order(m->N, q.left, q.right, m->order);
          if (m->rc != KDEHIP_OK) { m->msg = kdehip_last_error(); return; }
          if (h->fr.ids.size() > m->front_cap) { m->rc = KDEHIP_ERR_ARG; m->msg = "more frontier ids than a tree of N leaves has"; return; }
          const size_t nd = sizeof(double) * 2 * m->N * m->D, n2 = sizeof(double) * 2 * m->N;
          std::memcpy(mb + m->a_off + al(nd) + 2 * al(n2), h->fr.ids.data(), sizeof(int32_t) * h->fr.ids.size());
        });
      }
      trees.wait();
    } catch (const std::exception &e) {
      return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_mul_device_batch: ") + e.what());
    }
    for (const MulPlan &m : mp) if (!m.loose && m.rc != KDEHIP_OK) return set_error(m.rc, "kdehip_mul_device_batch: " + m.msg);
    KDEHIP_CHECK(hipMemcpyAsync(db, mb, a_bytes, hipMemcpyHostToDevice, xs));  // everything but the variances, ONE transfer
    // (5) the searches: wait, look, go on where one needs more rounds
    for (bool all = false; !all;) {
      KDEHIP_CHECK(hipStreamSynchronize(cs));
      all = true;
      for (MulGroup &g : groups) {
        if (!g.search) continue;
        bool done = false;
        rc = loocv_poll(g.search, &done);
        if (rc != KDEHIP_OK) return rc;
        if (!done) { all = false; continue; }
        std::vector<double> bw(g.members.size() * g.D);
        std::vector<int32_t> ne(g.members.size());
        rc = loocv_finish(g.search, bw.data(), ne.data());
        loocv_delete(g.search);
        g.search = nullptr;
        if (rc != KDEHIP_OK) return rc;
        for (size_t k = 0; k < g.members.size(); ++k) {
          const int i = g.members[k];
          for (int d = 0; d < g.D; ++d) {
            mp[i].h->bw[d] = bw[k * g.D + d];
            if (bw_out) bw_out[static_cast<size_t>(i) * KDEHIP_MAX_DIMS + d] = bw[k * g.D + d];
          }
          if (nevals) nevals[i] = ne[k];
        }
      }
    }
    // (6) the bandwidth-dependent rest: variances (+ the packers' examination of the nodes), one upload, per-level flags
    try {
      TaskGroup rest(HostPool::get());
      for (int i = 0; i < nprod; ++i) {
        MulPlan *m = &mp[i];
        if (m->loose) continue;
        rest.run([m] {
          kdehip_device_density *h = m->h;
          kdehip_device_density::Mirror &q = h->m;
          NodeStats stats{};
          m->rc = set_bandwidth_examined(m->D, m->N, h->bw, m->D, q.weights, q.left, q.right, q.means, q.bandwidth, q.bwmin, q.bwmax,
                                         &stats, &m->order);
          if (m->rc != KDEHIP_OK) { m->msg = kdehip_last_error(); return; }
          const kdehip_density host{m->N, m->D, q.means, q.bandwidth, q.weights, q.left, q.right, q.perm};
          examine_frontiers(host, h->D, h->Lown, /*look=*/false, h->fr);
          h->fr.bad = stats.bad;
          for (int k = 0; k < KDEHIP_MAX_DIMS; ++k) { h->fr.lo[k] = stats.lo[k]; h->fr.hi[k] = stats.hi[k]; }
          std::vector<int32_t>().swap(h->fr.ids);  // (the ids live on the device now; sizes, offsets and flags stay)
          std::vector<uint8_t>().swap(h->fr.fresh);
        });
      }
      rest.wait();
    } catch (const std::exception &e) {
      return set_error(KDEHIP_ERR_ALLOC, std::string("kdehip_mul_device_batch: ") + e.what());
    }
    for (const MulPlan &m : mp) if (!m.loose && m.rc != KDEHIP_OK) return set_error(m.rc, "kdehip_mul_device_batch: " + m.msg);
    KDEHIP_CHECK(hipMemcpyAsync(db + a_bytes, mb + a_bytes, b_bytes, hipMemcpyHostToDevice, xs));
    KDEHIP_CHECK(hipStreamSynchronize(xs));
  }
  // items outside the batched path: a call of their own each
  for (int i = 0; i < nprod; ++i) {
    if (!mp[i].loose) continue;
    rc = kdehip_mul_device(&mp[i].h, items[i].Ndens, items[i].trees, items[i].seed, items[i].addEntropy,
                           bw_out ? bw_out + static_cast<size_t>(i) * KDEHIP_MAX_DIMS : nullptr, nevals ? nevals + i : nullptr);
    if (rc != KDEHIP_OK) {
      for (MulPlan &m : mp) if (m.loose && m.h) { kdehip_density_free(m.h); m.h = nullptr; }
      return rc;
    }
  }
  for (int i = 0; i < nprod; ++i) {
    MulPlan &m = mp[i];
    if (!m.loose) { m.h->built = true; m.h->shared = cl.sb; cl.sb->refs.fetch_add(1, std::memory_order_relaxed); }
    out[i] = m.h;
  }
  if (cl.sb && nbatched == 0) { delete cl.sb; cl.sb = nullptr; }
  cl.keep = true;
  return KDEHIP_OK;
}

// The arrays of a density the library built (kdehip_density_from_device_points / kdehip_mul_device), shaped as in
// kdehip_make_density; any pointer may be NULL.  A density that came from kdehip_density_upload has no mirror: its
// arrays are the caller's.
extern "C" int kdehip_density_download(const kdehip_device_density *h, double *centers, double *ranges, double *weights,
                                       int64_t *left_child, int64_t *right_child, int64_t *lowest_leaf,
                                       int64_t *highest_leaf, int64_t *permutation, double *means, double *bandwidth,
                                       double *bandwidthMin, double *bandwidthMax, double *bw_out) {
  if (!h) return set_error(KDEHIP_ERR_ARG, "null density");
  if (!h->built) return set_error(KDEHIP_ERR_UNSUPPORTED, "this density was uploaded by the caller, who holds its arrays");
  const size_t nd = static_cast<size_t>(2 * h->N * h->D), n2 = static_cast<size_t>(2 * h->N);
  const kdehip_device_density::Mirror &m = h->m;
  auto cp = [](auto *dst, const auto *src, size_t n) { if (dst) std::memcpy(dst, src, n * sizeof(*src)); };
  cp(centers, m.centers, nd); cp(ranges, m.ranges, nd); cp(means, m.means, nd); cp(bandwidth, m.bandwidth, nd);
  cp(bandwidthMin, m.bwmin, nd / 2); cp(bandwidthMax, m.bwmax, nd / 2); cp(weights, m.weights, n2);
  cp(left_child, m.left, n2); cp(right_child, m.right, n2); cp(lowest_leaf, m.lowest, n2); cp(highest_leaf, m.highest, n2);
  cp(permutation, m.perm, n2);
  if (bw_out) for (int k = 0; k < h->D; ++k) bw_out[k] = h->bw[k];
  return KDEHIP_OK;
}

extern "C" void kdehip_density_free(kdehip_device_density *h) {
  if (!h) return;
  DeviceGuard guard;
  SharedBlock *sb = h->shared;
  const bool last = sb && sb->refs.fetch_sub(1, std::memory_order_acq_rel) == 1;  // (a batch's blocks go with its last density)
  if (guard.enter(h->device) == KDEHIP_OK) {
    (void)hipDeviceSynchronize();  // products enqueued on caller streams may still read the block
    if (h->d_blob) cached_free(h->d_blob, h->blob_bytes);
    if (h->mirror) cached_host_free(h->mirror, h->mirror_bytes);
    if (last) {
      if (sb->d_blob) cached_free(sb->d_blob, sb->blob_bytes);
      if (sb->mirror) cached_host_free(sb->mirror, sb->mirror_bytes);
    }
  } else {  // (no device to enter: hand the pinned block back to the driver)
    if (h->mirror) (void)hipHostFree(h->mirror);
    if (last && sb->mirror) (void)hipHostFree(sb->mirror);
  }
  if (last) delete sb;
  delete h;
}

extern "C" int64_t kdehip_density_npts(const kdehip_device_density *h) { return h ? h->N : -1; }
extern "C" int kdehip_density_ndim(const kdehip_device_density *h) { return h ? h->D : -1; }
