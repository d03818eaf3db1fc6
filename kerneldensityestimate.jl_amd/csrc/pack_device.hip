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

#include <cmath>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "device_density.hpp"
#include "kdehip_internal.hpp"

using namespace kdehip;

#define KDEHIP_CHECK(expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return set_error(KDEHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

namespace kdehip {

// One wavefront per (tile, row): lane ln writes entry z = ln*B + row of the frontier (kdehip_internal.hpp "packed
// per-level layout"), field by field -- 64 contiguous elements per store -- from the density's arrays in HBM.
template <typename T>
__global__ __launch_bounds__(64) void fill_tiles_kernel(FillArgs a) {
  const FillJob job = a.jobs[blockIdx.y];
  const int row = blockIdx.x;
  if (row >= job.B) return;
  const int lane = threadIdx.x;
  const int D = a.D, F = job.F, j = job.dens;
  const int64_t RS = static_cast<int64_t>(F) * 64 + 1;
  T *hdr = static_cast<T *>(a.data) + job.hdr_off;
  const int32_t *front = a.front[j] + job.front_off;
  const double *means = a.means[j], *bw = a.bandwidth[j];
  if (row == 0 && lane < kTileHeader) {  // the bandwidth vector of entry 0 (= of every entry of a uniform tile)
    const int64_t n0 = static_cast<int64_t>(front[0]) - 1;
    hdr[lane] = lane < D ? static_cast<T>(bw[n0 * D + lane]) : T(0);
  }
  const int64_t z = static_cast<int64_t>(lane) * job.B + row;
  const int64_t src = z < job.n ? static_cast<int64_t>(front[z]) - 1 : -1;
  T *r = hdr + kTileHeader + row * RS;
  for (int d = 0; d < D; ++d) r[d * 64 + lane] = src >= 0 ? static_cast<T>(means[src * D + d]) : T(0);
  if (!job.uniform)
    for (int d = 0; d < D; ++d) r[(D + d) * 64 + lane] = src >= 0 ? static_cast<T>(bw[src * D + d]) : T(1);
  r[(F - 1) * 64 + lane] = src >= 0 ? static_cast<T>(a.weights[j][src]) : T(0);
  if (lane == 0) r[F * 64] = T(0);  // the pad element
  a.perm_out[job.perm_off + static_cast<int64_t>(row) * 64 + lane] = src >= 0 ? static_cast<int32_t>(a.perm[j][src]) : 0;
}

int launch_fill_tiles(int precision, const FillArgs &a, int ntiles, int maxB, void *stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 grid(static_cast<unsigned>(maxB), static_cast<unsigned>(ntiles));
  if (precision == 64) hipLaunchKernelGGL(fill_tiles_kernel<double>, grid, dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(fill_tiles_kernel<float>, grid, dim3(64), 0, stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("tile fill launch failed: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

}  // namespace kdehip

// ---- densities in HBM ----------------------------------------------------------------------------------------------

extern "C" int kdehip_density_upload(kdehip_device_density **out, const kdehip_density *host, int device) {
  if (!out) return set_error(KDEHIP_ERR_ARG, "null out pointer");
  *out = nullptr;
  if (!host) return set_error(KDEHIP_ERR_ARG, "null density");
  const int64_t N = host->npts, D = host->ndim;
  if (N < 1) return set_error(KDEHIP_ERR_ARG, "density with no points");
  if (N > (int64_t(1) << 30)) return set_error(KDEHIP_ERR_UNSUPPORTED, "density too large");
  if (D < 1 || D > KDEHIP_MAX_DIMS) return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
  if (!host->means || !host->bandwidth || !host->weights || !host->left_child || !host->right_child || !host->permutation)
    return set_error(KDEHIP_ERR_ARG, "density with a null array");
  kdehip_device_density *h = new (std::nothrow) kdehip_device_density();
  if (!h) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
  h->device = device;
  h->N = N;
  h->D = static_cast<int>(D);
  h->Lown = nlevels_for(N);
  // the frontiers of the density's own levels and what the arithmetic-form decision needs (every node is looked at once)
  int rc = expand_frontiers(*host, h->D, h->Lown, /*look=*/true, h->fr);
  if (rc != KDEHIP_OK) { delete h; return rc; }
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != KDEHIP_OK) { delete h; return rc; }
  auto al = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t nd = sizeof(double) * 2 * N * D, n2 = sizeof(double) * 2 * N;
  const size_t o_mean = 0, o_bw = al(o_mean + nd), o_w = al(o_bw + nd), o_perm = al(o_w + n2);
  const size_t o_front = al(o_perm + sizeof(int64_t) * 2 * N);
  const size_t total = al(o_front + sizeof(int32_t) * h->fr.ids.size());
  void *pin = nullptr;
  hipError_t e = cached_host_malloc(&pin, total);
  if (e == hipSuccess) e = cached_malloc(&h->d_blob, total);
  if (e != hipSuccess) {
    if (pin) cached_host_free(pin, total);
    delete h;
    return set_error(KDEHIP_ERR_HIP, std::string("density upload: ") + hipGetErrorString(e));
  }
  h->blob_bytes = total;
  unsigned char *hp = static_cast<unsigned char *>(pin);
  std::memcpy(hp + o_mean, host->means, nd);
  std::memcpy(hp + o_bw, host->bandwidth, nd);
  std::memcpy(hp + o_w, host->weights, n2);
  std::memcpy(hp + o_perm, host->permutation, sizeof(int64_t) * 2 * N);
  std::memcpy(hp + o_front, h->fr.ids.data(), sizeof(int32_t) * h->fr.ids.size());
  e = hipMemcpyAsync(h->d_blob, pin, total, hipMemcpyHostToDevice, nullptr);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  cached_host_free(pin, total);
  if (e != hipSuccess) {
    cached_free(h->d_blob, total);
    delete h;
    return set_error(KDEHIP_ERR_HIP, std::string("density upload: ") + hipGetErrorString(e));
  }
  unsigned char *db = static_cast<unsigned char *>(h->d_blob);
  h->means = reinterpret_cast<const double *>(db + o_mean);
  h->bandwidth = reinterpret_cast<const double *>(db + o_bw);
  h->weights = reinterpret_cast<const double *>(db + o_w);
  h->perm = reinterpret_cast<const int64_t *>(db + o_perm);
  h->front = reinterpret_cast<const int32_t *>(db + o_front);
  std::vector<int32_t>().swap(h->fr.ids);  // (the ids live on the device now; sizes, offsets and flags stay)
  *out = h;
  return KDEHIP_OK;
}

extern "C" void kdehip_density_free(kdehip_device_density *h) {
  if (!h) return;
  DeviceGuard guard;
  if (guard.enter(h->device) == KDEHIP_OK) {
    (void)hipDeviceSynchronize();  // products enqueued on caller streams may still read the block
    if (h->d_blob) cached_free(h->d_blob, h->blob_bytes);
  }
  delete h;
}

extern "C" int64_t kdehip_density_npts(const kdehip_device_density *h) { return h ? h->N : -1; }
extern "C" int kdehip_density_ndim(const kdehip_device_density *h) { return h ? h->D : -1; }
