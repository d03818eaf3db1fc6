// product.hip -- C-ABI entry points of libkdehip.so for the Gibbs product: resident plans, runs,
// the gibbs1 drop-in and the host twin of the device RNG.  (Kernels: gibbs_kernel.hip; host
// re-layout: pack_levels.cpp; density construction: balltree.cpp.)
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "device_density.hpp"
#include "kdehip_internal.hpp"
#include "phase_timer.hpp"

namespace kdehip { extern std::atomic<unsigned> g_peer_epoch; }  // devmem.cpp: bumped by kdehip_clear_cache
#include "philox.hpp"

using namespace kdehip;

struct kdehip_product {
  int device = 0;
  int precision = 64;
  bool fast = true;
  int mode = kModeFast;
  int variant = 0;
  PackedProduct host;  // descriptors (payload vectors are released after upload)
  void *d_blob = nullptr;   // the one device allocation of the plan; the pointers below point into it
  size_t blob_bytes = 0;
  std::atomic<bool> async_pending{false};  // set (never cleared) by the device-pointer entry points: a run may be in flight on a caller stream
  void *d_data = nullptr;
  int32_t *d_perm = nullptr;
  LevelDesc *d_levels = nullptr;
  void *d_tables = nullptr;
  TabDesc *d_tabdesc = nullptr;
  bool tables_built = false;
  bool screens_built = false;  // the fp32 screen tiles (kdehip_internal.hpp "fp32 screening") are written
  std::mutex tables_mutex;  // concurrent first runs on one plan build the tables (and the screen tiles) once
  unsigned long long *d_fallbacks = nullptr;  // device counter of uniform-fallback draws (:311-315), in the blob
  void *d_work = nullptr;   // scratch of the host-buffer entry points (outputs / uploaded streams), grown on demand
  size_t work_cap = 0;
  std::mutex work_mutex;    // host-buffer calls on one plan are serialised
  int64_t packed_bytes = 0;
  PlanDev dev{};
};

namespace {

#define KDEHIP_CHECK(expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return set_error(KDEHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// The stream of the BLOCKING entry points (host buffers in and out): the calling thread's own stream.  On the legacy
// null stream the calls of concurrent host threads -- a multi-threaded belief-propagation host issues many small
// products at once -- would run one after the other on the device although each of them fills a fraction of it.
inline hipStream_t call_stream() { return hipStreamPerThread; }

int check_run(const kdehip_product *plan, int64_t Np, int Niter, const void *d_points,
              const void *d_indices) {
  if (!plan) return set_error(KDEHIP_ERR_ARG, "null plan");
  if (Np < 0) return set_error(KDEHIP_ERR_ARG, "Np must be >= 0");
  if (Niter < 0) return set_error(KDEHIP_ERR_ARG, "Niter must be >= 0");
  if (Np > 0 && (!d_points || !d_indices)) return set_error(KDEHIP_ERR_ARG, "null output pointer");
  if (Np > (int64_t(1) << 31) * 4 - 8) return set_error(KDEHIP_ERR_UNSUPPORTED, "Np too large for one launch");
  return KDEHIP_OK;
}

// Conditional tables are filled by the sampler kernel itself (table_build launch), on the caller's stream,
// the first time a run is large enough to pay for them.
int maybe_build_tables(kdehip_product *plan, int64_t Np, RunArgs &a, void *stream, bool private_plan = false) {
  a.table_build = 0;
  a.use_tables = 0;
  if (plan->dev.Lt <= 0 || plan->dev.tab_rows_total <= 0) return KDEHIP_OK;
  std::lock_guard<std::mutex> lock(plan->tables_mutex);
  if (!plan->tables_built) {
    if (Np < kTabMinChains) return KDEHIP_OK;
    RunArgs b = a;
    b.table_build = 1;
    b.Np = plan->dev.tab_rows_total;  // one wavefront per table row
    b.variant = 8;
    const int rc = launch_gibbs(plan->precision, plan->mode, plan->dev, b, stream);
    if (rc != KDEHIP_OK) return rc;
    // one-time: runs on other streams must not overtake the build (a plan private to one call has no other streams)
    if (!private_plan) KDEHIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    plan->tables_built = true;
  }
  a.use_tables = 1;
  return KDEHIP_OK;
}

// The fp32 screen tiles of an fp64 plan are written by the GPU from the plan's own fp64 tiles (pack_device.hip
// screen_build_kernel), on the stream that prepares the plan, before its first run.  Plan variant 5 runs without them.
int maybe_build_screens(kdehip_product *plan, RunArgs &a, void *stream, bool private_plan = false) {
  a.use_screen = 0;
  if (!plan->dev.screened || plan->variant % 1000 == 5) return KDEHIP_OK;
  std::lock_guard<std::mutex> lock(plan->tables_mutex);
  if (!plan->screens_built) {
    const int rc = launch_screen_build(plan->dev, stream);
    if (rc != KDEHIP_OK) return rc;
    if (!private_plan) KDEHIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    plan->screens_built = true;
  }
  a.use_screen = 1;
  return KDEHIP_OK;
}

// the plan's level table on the device: [M][L+1] level descriptors, as many screen descriptors (all zero when no level
// is screened), then [M][L+1] step descriptors
inline size_t level_table_bytes(const PackedProduct &pp) {
  return pp.levels.size() * (2 * sizeof(LevelDesc) + sizeof(StepDesc));
}
inline void copy_level_table(const PackedProduct &pp, unsigned char *dst) {
  const size_t nb = pp.levels.size() * sizeof(LevelDesc);
  std::memcpy(dst, pp.levels.data(), nb);
  if (!pp.screens.empty()) std::memcpy(dst + nb, pp.screens.data(), nb);
  else std::memset(dst + nb, 0, nb);
  std::memcpy(dst + 2 * nb, pp.steps.data(), pp.steps.size() * sizeof(StepDesc));
}

// Scratch of the host-buffer entry points: one device buffer per plan, grown on demand (no per-call
// hipMalloc / hipFree once it is large enough).  Callers hold plan->work_mutex.
int reserve_work(kdehip_product *plan, size_t bytes) {
  if (bytes <= plan->work_cap) return KDEHIP_OK;
  if (plan->d_work) { cached_free(plan->d_work, plan->work_cap); plan->d_work = nullptr; plan->work_cap = 0; }
  const size_t cap = (bytes + 4095) & ~static_cast<size_t>(4095);
  KDEHIP_CHECK(cached_malloc(&plan->d_work, cap));
  plan->work_cap = cap;
  return KDEHIP_OK;
}
inline size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

// Results of a host-buffer run: the output arrays sit back to back in the plan's scratch, so they come back with ONE
// DMA transfer into pinned memory (from the library's cache) and are handed out from there -- two or three blocking
// copies into pageable memory cost tens of microseconds each, which is visible against a ~0.6 ms product.
struct OutPiece { void *host; size_t dev_off, bytes; };
int copy_out(const void *d_base, size_t span, const OutPiece *pieces, int npieces) {
  void *h = nullptr;
  KDEHIP_CHECK(cached_host_malloc(&h, span));
  hipError_t e = hipMemcpyAsync(h, d_base, span, hipMemcpyDeviceToHost, call_stream());
  if (e == hipSuccess) e = hipStreamSynchronize(call_stream());
  if (e == hipSuccess)
    for (int i = 0; i < npieces; ++i)
      if (pieces[i].host && pieces[i].bytes) std::memcpy(pieces[i].host, static_cast<unsigned char *>(h) + pieces[i].dev_off, pieces[i].bytes);
  cached_host_free(h, span);
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("result copy: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

// The two run forms, enqueue only (no bookkeeping of who waits for the work: see the callers).
int enqueue_streams(kdehip_product *plan, int64_t Np, int Niter, const double *d_randU, int64_t nU,
                    const double *d_randN, int64_t nN, int addEntropy, double *d_points, int64_t *d_indices,
                    int32_t *d_labels, void *stream, bool private_plan = false) {
  int rc = check_run(plan, Np, Niter, d_points, d_indices);
  if (rc != KDEHIP_OK) return rc;
  if (Np == 0) return KDEHIP_OK;
  const int64_t K = kdehip_product_randu_per_sample(plan, Niter);
  const int64_t R = kdehip_product_randn_per_sample(plan);
  // last uniform read is 0-based element Np*K - 2; the reference raises BoundsError when short
  if (!d_randU || nU < Np * K - 1)
    return set_error(KDEHIP_ERR_RAND_SHORT, "randU shorter than Np*K-1 values (Julia: BoundsError)");
  if (!d_randN || nN < Np * R)
    return set_error(KDEHIP_ERR_RAND_SHORT, "randN shorter than Np*R values (Julia: BoundsError)");
  DeviceGuard guard;
  rc = guard.enter(plan->device);
  if (rc != KDEHIP_OK) return rc;
  RunArgs a{};
  a.Np = Np; a.Niter = Niter; a.addEntropy = addEntropy ? 1 : 0; a.rng_philox = 0;
  a.variant = plan->variant;
  a.randU = d_randU; a.randN = d_randN; a.K = K; a.R = R; a.nU = nU; a.nN = nN;
  a.seed = 0; a.sample_offset = 0;
  a.points = d_points; a.indices = d_indices; a.labels = d_labels;
  rc = maybe_build_screens(plan, a, stream, private_plan);
  if (rc != KDEHIP_OK) return rc;
  rc = maybe_build_tables(plan, Np, a, stream, private_plan);
  if (rc != KDEHIP_OK) return rc;
  return launch_gibbs(plan->precision, plan->mode, plan->dev, a, stream);
}

// the other devices' output arrays of a multi-GPU run (kernel epilogue stores, RunArgs.peer_*)
struct PeerOutputs {
  int n = 0;
  double *points[kMaxPeers] = {};
  int64_t *indices[kMaxPeers] = {};
};

int enqueue_philox(kdehip_product *plan, int64_t Np, int Niter, uint64_t seed, int64_t sample_offset,
                   int addEntropy, double *d_points, int64_t *d_indices, int32_t *d_labels, void *stream,
                   bool private_plan = false, const PeerOutputs *peers = nullptr, void *table_stream = nullptr,
                   hipEvent_t tables_done = nullptr, hipEvent_t t_begin = nullptr, hipEvent_t t_end = nullptr) {
  int rc = check_run(plan, Np, Niter, d_points, d_indices);
  if (rc != KDEHIP_OK) return rc;
  if (Np == 0) return KDEHIP_OK;
  if (sample_offset < 0) return set_error(KDEHIP_ERR_ARG, "sample_offset must be >= 0");
  DeviceGuard guard;
  rc = guard.enter(plan->device);
  if (rc != KDEHIP_OK) return rc;
  RunArgs a{};
  a.Np = Np; a.Niter = Niter; a.addEntropy = addEntropy ? 1 : 0; a.rng_philox = 1;
  a.variant = plan->variant;
  a.randU = nullptr; a.randN = nullptr;
  a.K = kdehip_product_randu_per_sample(plan, Niter);
  a.R = kdehip_product_randn_per_sample(plan);
  a.seed = seed; a.sample_offset = sample_offset;
  a.points = d_points; a.indices = d_indices; a.labels = d_labels;
  // (table_stream: a plan private to one call may fill its tables on another stream; the sampler waits for tables_done)
  rc = maybe_build_screens(plan, a, table_stream ? table_stream : stream, private_plan);
  if (rc != KDEHIP_OK) return rc;
  rc = maybe_build_tables(plan, Np, a, table_stream ? table_stream : stream, private_plan);
  if (rc != KDEHIP_OK) return rc;
  if (table_stream) {
    KDEHIP_CHECK(hipEventRecord(tables_done, static_cast<hipStream_t>(table_stream)));
    KDEHIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), tables_done, 0));
  }
  if (peers) {
    a.npeers = peers->n;
    for (int k = 0; k < peers->n; ++k) { a.peer_points[k] = peers->points[k]; a.peer_indices[k] = peers->indices[k]; }
  }
  if (t_begin) KDEHIP_CHECK(hipEventRecord(t_begin, static_cast<hipStream_t>(stream)));
  rc = launch_gibbs(plan->precision, plan->mode, plan->dev, a, stream);
  if (rc == KDEHIP_OK && t_end) KDEHIP_CHECK(hipEventRecord(t_end, static_cast<hipStream_t>(stream)));
  return rc;
}


// The device image of a product, assembled once in pinned host memory (from the library's cache) in the tiles'
// final precision: [.. fallback counter | levels | table descriptors | permutation | tiles] (+ room for the
// conditional tables on the device).  One image can be instantiated on several devices (multi-GPU entry points).
struct PlanImage {
  PackedProduct host;
  void *h_blob = nullptr;
  size_t off_lev = 0, off_count = 0, off_tab = 0, off_perm = 0, off_data = 0, off_tables = 0, total = 0;
  int precision = 64;
  size_t blob_bytes = 0;
  ~PlanImage() { if (h_blob) cached_host_free(h_blob, blob_bytes); }
};

int build_image(PlanImage &im, int Ndens, const kdehip_density *trees, int ndims, const uint8_t *partialDimMask,
                int precision, bool force_generic = false) {
  if (precision != 64 && precision != 32) return set_error(KDEHIP_ERR_ARG, "precision must be 64 or 32");
  static const bool timing = std::getenv("KDEHIP_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
  // First the layout of the fast arithmetic form WITHOUT looking at the node values: whether they allow that form
  // (finite, variances in range) is established by pack_fill on the values it copies anyway -- looking first costs
  // as much as the fill (random accesses into the tree arrays).  The rare density set that does not qualify is laid
  // out again in the generic form and refilled.
  for (PackMode pmode : {kPackOptimistic, kPackGeneric}) {
    if (force_generic && pmode != kPackGeneric) continue;  // (on-manifold operators run the reference's own accumulation)
    int rc = pack_layout(Ndens, trees, ndims, partialDimMask, precision, im.host, pmode);
    if (rc != KDEHIP_OK) return rc;
    const double us_layout = us();
    im.precision = precision;
    const size_t nelem = static_cast<size_t>(im.host.data_elems);
    const size_t esz = (precision == 64) ? sizeof(double) : sizeof(float);
    const size_t nperm = static_cast<size_t>(im.host.perm_elems);
    const size_t ntab = im.host.tabdesc.size();
    im.off_lev = 256;                                        // the fallback counter sits in the 8 bytes before it
    im.off_count = im.off_lev - sizeof(unsigned long long);
    im.off_tab = align256(im.off_lev + level_table_bytes(im.host));
    im.off_perm = align256(im.off_tab + ntab * sizeof(TabDesc));
    im.off_data = align256(im.off_perm + nperm * sizeof(int32_t));
    im.off_tables = align256(im.off_data + nelem * esz);
    im.total = im.off_tables + static_cast<size_t>(im.host.tab_entries) * esz;
    // (pinned memory needs a HIP runtime with a device: a host without one fails here, loudly, as it must)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return set_error(KDEHIP_ERR_NO_DEVICE, "no HIP device available (libkdehip has no CPU fallback by design)");
    KDEHIP_CHECK(cached_host_malloc(&im.h_blob, im.off_tables));
    im.blob_bytes = im.off_tables;
    const double us_alloc = us();
    unsigned char *hb = static_cast<unsigned char *>(im.h_blob);
    std::memset(hb, 0, im.off_lev);
    copy_level_table(im.host, hb + im.off_lev);
    std::memcpy(hb + im.off_tab, im.host.tabdesc.data(), ntab * sizeof(TabDesc));
    const bool ok = pack_fill(im.host, trees, hb + im.off_data, reinterpret_cast<int32_t *>(hb + im.off_perm));
    if (timing)
      std::fprintf(stderr, "kdehip image%s: layout %.0f us | pinned block %.0f us | tiles filled %.0f us\n",
                   ok ? "" : " (fast form refused: again in the generic form)", us_layout, us_alloc, us());
    if (ok) break;
    cached_host_free(im.h_blob, im.blob_bytes);
    im.h_blob = nullptr;
  }
  std::vector<int32_t>().swap(im.host.front);  // (only the descriptors are needed from here on)
  return KDEHIP_OK;
}

// The device pointers and the kernel-side description of a plan whose blob (layout of PlanImage) is allocated and whose
// host-side descriptors (p->host, p->precision, p->mode) are set.
void bind_plan(kdehip_product *p, size_t off_lev, size_t off_count, size_t off_tab, size_t off_perm, size_t off_data,
               size_t off_tables, size_t total) {
  unsigned char *base = static_cast<unsigned char *>(p->d_blob);
  p->d_levels = reinterpret_cast<LevelDesc *>(base + off_lev);
  p->d_tabdesc = reinterpret_cast<TabDesc *>(base + off_tab);
  p->d_perm = reinterpret_cast<int32_t *>(base + off_perm);
  p->d_data = base + off_data;
  p->d_fallbacks = reinterpret_cast<unsigned long long *>(base + off_count);
  p->d_tables = p->host.tab_entries ? base + off_tables : nullptr;
  p->packed_bytes = static_cast<int64_t>(total);
  p->dev.data = p->d_data;
  p->dev.perm = p->d_perm;
  p->dev.levels = p->d_levels;
  p->dev.tables = p->d_tables;
  p->dev.tabdesc = p->d_tabdesc;
  p->dev.tab_rows_total = p->host.tab_rows;
  p->dev.M = p->host.M;
  p->dev.L = p->host.L;
  p->dev.D = p->host.D;
  p->dev.Lt = (p->mode == kModeGeneric) ? 0 : p->host.Lt;
  p->dev.screened = (p->mode == kModeFast && !p->host.screens.empty()) ? 1 : 0;
  if (p->dev.screened)
    for (const LevelDesc &sc : p->host.screens)
      if (sc.stage_mode == kStageScreenChunked) p->dev.screened = 2;  // (selects the sampler build that knows chunked screens)
}

// A plan on `device` from an image: one device allocation, one DMA transfer (hipMalloc / hipFree cost tens of
// microseconds each and would dominate a one-shot small product; blocks come from the library's cache).
// wait = false (one-shot calls, whose image outlives the work they enqueue): the upload is left in flight on the
// calling thread's stream (call_stream); everything the caller enqueues there afterwards is ordered behind it.
int instantiate(const PlanImage &im, int device, kdehip_product **out, bool wait = true) {
  *out = nullptr;
  kdehip_product *p = new (std::nothrow) kdehip_product();
  if (!p) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != KDEHIP_OK) { delete p; return rc; }
  p->device = device;
  p->precision = im.precision;
  p->host = im.host;  // descriptors (the frontier ids were released by build_image)
  p->fast = p->host.fast;
  p->mode = !p->host.fast ? kModeGeneric : (p->host.all_active ? kModeFast : kModeFastMasked);
  hipError_t e = cached_malloc(&p->d_blob, im.total);
  if (e == hipSuccess) p->blob_bytes = im.total;
  if (e == hipSuccess) e = hipMemcpyAsync(p->d_blob, im.h_blob, im.off_tables, hipMemcpyHostToDevice, call_stream());
  if (e == hipSuccess && wait) e = hipStreamSynchronize(call_stream());  // (the pinned image may be recycled after this call)
  if (e != hipSuccess) {
    const std::string m = std::string("plan upload: ") + hipGetErrorString(e);
    kdehip_product_destroy(p);
    return set_error(KDEHIP_ERR_HIP, m);
  }
  bind_plan(p, im.off_lev, im.off_count, im.off_tab, im.off_perm, im.off_data, im.off_tables, im.total);
  *out = p;
  return KDEHIP_OK;
}

// Physical ordinal of logical device `d` of a multi-GPU call.  Normally the identity.  With KDEHIP_ALIAS_DEVICES=1
// (tests on single-GPU machines) logical devices wrap around the visible ones, so that the complete N > 1 code path
// -- slicing, per-device plans, peer copies, event waits -- runs on one GPU: several "devices" are then the same
// physical one, which every step of that path tolerates (a peer copy becomes a device-to-device copy).
inline bool alias_devices() {
  static const bool on = [] { const char *e = std::getenv("KDEHIP_ALIAS_DEVICES"); return e && e[0] == '1'; }();
  return on;
}
inline int phys(int d) {
  if (!alias_devices()) return d;
  int n = 1;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) n = 1;
  return d % n;
}

// contiguous share of `Np` chains for device g of G
inline int64_t share_begin(int64_t Np, int g, int G) { return Np * g / G; }

int check_devices(int device, int ngpus) {
  if (ngpus < 1) return set_error(KDEHIP_ERR_ARG, "ngpus must be >= 1");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return set_error(KDEHIP_ERR_NO_DEVICE, "no HIP device available (libkdehip has no CPU fallback by design)");
  if (alias_devices() ? (device < 0 || ngpus > KDEHIP_MAX_DENS) : (device < 0 || device + ngpus > n))
    return set_error(KDEHIP_ERR_ARG, "device range outside the visible devices");
  return KDEHIP_OK;
}

}  // namespace

extern "C" {

int kdehip_version(void) { return KDEHIP_VERSION; }
const char *kdehip_last_error(void) { return last_error_cstr(); }

int kdehip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int kdehip_product_create(kdehip_product **out, int Ndens, const kdehip_density *trees, int ndims,
                          const uint8_t *partialDimMask, int precision, int device) {
  if (!out) return set_error(KDEHIP_ERR_ARG, "null out pointer");
  *out = nullptr;
  PlanImage im;
  int rc = build_image(im, Ndens, trees, ndims, partialDimMask, precision);
  if (rc != KDEHIP_OK) return rc;
  return instantiate(im, device, out);
}

void kdehip_product_destroy(kdehip_product *plan) {
  if (!plan) return;
  DeviceGuard guard;  // (reached from garbage collectors: the caller's current device must survive this call)
  if (guard.enter(plan->device) == KDEHIP_OK) {
    // the blocks go back to the allocation cache and may be handed out again at once: work enqueued on the
    // caller's streams must be over (hipFree used to imply this)
    if (plan->async_pending.load()) (void)hipDeviceSynchronize();
    if (plan->d_blob) cached_free(plan->d_blob, plan->blob_bytes);
    if (plan->d_work) cached_free(plan->d_work, plan->work_cap);
  }
  delete plan;
}

int kdehip_product_info(const kdehip_product *plan, kdehip_product_info_t *info) {
  if (!plan || !info) return set_error(KDEHIP_ERR_ARG, "null argument");
  info->ndens = plan->host.M;
  info->ndims = plan->host.D;
  info->nlevels = plan->host.L;
  info->precision = plan->precision;
  info->nodes_per_sweep = plan->host.nodes_per_sweep;
  info->bytes_per_eval = (2 * plan->host.D + 1) * (plan->precision == 64 ? 8 : 4);
  info->packed_bytes = plan->packed_bytes;
  info->fast_math_path = plan->fast ? 1 : 0;
  info->device = plan->device;
  return KDEHIP_OK;
}

int64_t kdehip_product_randu_per_sample(const kdehip_product *plan, int Niter) {
  if (!plan || Niter < 0) return -1;
  // M init calls + per level: M (sampleIndices!) + Niter*M (sampleIndex)
  return static_cast<int64_t>(plan->host.M) * (1 + static_cast<int64_t>(plan->host.L) * (Niter + 1));
}
int64_t kdehip_product_randn_per_sample(const kdehip_product *plan) {
  if (!plan) return -1;
  return static_cast<int64_t>(plan->host.D) * (plan->host.L + 1);
}

int64_t kdehip_product_fallback_count(kdehip_product *plan) {
  if (!plan) return set_error(KDEHIP_ERR_ARG, "null plan");
  DeviceGuard guard;
  int rc = guard.enter(plan->device);
  if (rc != KDEHIP_OK) return rc;
  unsigned long long v = 0;
  KDEHIP_CHECK(hipDeviceSynchronize());
  KDEHIP_CHECK(hipMemcpy(&v, plan->d_fallbacks, sizeof(v), hipMemcpyDeviceToHost));
  return static_cast<int64_t>(v);
}

int kdehip_product_screen_stats(kdehip_product *plan, int32_t *levels, int64_t *steps, int64_t *repeats) {
  if (!plan) return set_error(KDEHIP_ERR_ARG, "null plan");
  DeviceGuard guard;
  const int rc = guard.enter(plan->device);
  if (rc != KDEHIP_OK) return rc;
  unsigned long long v[2] = {0, 0};  // [screened steps, repeated in fp64]: the 16 bytes in front of the fallback counter
  KDEHIP_CHECK(hipDeviceSynchronize());
  KDEHIP_CHECK(hipMemcpy(v, plan->d_fallbacks - 2, sizeof(v), hipMemcpyDeviceToHost));
  if (levels) *levels = plan->dev.screened ? plan->host.nscreened : 0;
  if (steps) *steps = static_cast<int64_t>(v[0]);
  if (repeats) *repeats = static_cast<int64_t>(v[1]);
  return KDEHIP_OK;
}

int kdehip_product_set_variant(kdehip_product *plan, int variant) {
  if (!plan) return set_error(KDEHIP_ERR_ARG, "null plan");
  plan->variant = variant;
  return KDEHIP_OK;
}

int kdehip_product_launch_geometry(const kdehip_product *plan, int64_t Np, int32_t *waves_per_workgroup,
                                   int32_t *waves_per_chain) {
  if (!plan || !waves_per_workgroup || !waves_per_chain) return set_error(KDEHIP_ERR_ARG, "null argument");
  DeviceGuard guard;
  const int rc = guard.enter(plan->device);  // (the width heuristic asks the plan's device for its CU count)
  if (rc != KDEHIP_OK) return rc;
  const bool lean = plan->mode == kModeFast && ((plan->host.M >= 2 && plan->host.M <= 4) || (plan->host.M == 8 && plan->precision == 64));
  const int v = plan->variant % 1000;
  const bool as_lean = lean && !(v >= kVariantGenericBase && v < kVariantGenericBase + 20);
  *waves_per_workgroup = as_lean ? lean_waves(Np, plan->variant) : chains_per_workgroup(Np, plan->variant);
  *waves_per_chain = 1;
  return KDEHIP_OK;
}

const char *kdehip_product_kernel_name(const kdehip_product *plan, int64_t Np) {
  if (!plan) return "";
  const int v = plan->variant % 1000;
  const bool forced_general = (v >= kVariantGenericBase && v < kVariantGenericBase + 20);
  const int M = plan->host.M, L = plan->host.L, D = plan->host.D;
  bool lean = !forced_general && plan->mode == kModeFast && D * (L + 1) <= 128 &&
              ((M >= 2 && M <= 4) || (M == 8 && plan->precision == 64));
  if (lean && M == 8) {  // (the 8-density instantiations exist for 8 and 16 chains per workgroup)
    DeviceGuard guard;
    if (guard.enter(plan->device) != KDEHIP_OK) return "";
    lean = lean_waves(Np, plan->variant) != 4;
  }
  return lean ? "gibbs_lean_kernel" : "gibbs_product_kernel";
}

int kdehip_product_sample_streams(kdehip_product *plan, int64_t Np, int Niter, const double *d_randU,
                                  int64_t nU, const double *d_randN, int64_t nN, int addEntropy,
                                  double *d_points, int64_t *d_indices, int32_t *d_labels,
                                  void *stream) {
  const int rc = enqueue_streams(plan, Np, Niter, d_randU, nU, d_randN, nN, addEntropy, d_points, d_indices,
                                 d_labels, stream);
  if (rc == KDEHIP_OK && Np > 0) plan->async_pending.store(true);  // the caller's stream may still be running it
  return rc;
}

int kdehip_product_sample_philox(kdehip_product *plan, int64_t Np, int Niter, uint64_t seed,
                                 int64_t sample_offset, int addEntropy, double *d_points,
                                 int64_t *d_indices, int32_t *d_labels, void *stream) {
  const int rc = enqueue_philox(plan, Np, Niter, seed, sample_offset, addEntropy, d_points, d_indices, d_labels,
                                stream);
  if (rc == KDEHIP_OK && Np > 0) plan->async_pending.store(true);
  return rc;
}

int kdehip_product_sample_philox_host(kdehip_product *plan, int64_t Np, int Niter, uint64_t seed,
                                      int64_t sample_offset, int addEntropy, double *points,
                                      int64_t *indices, int32_t *labels) {
  int rc = check_run(plan, Np, Niter, points, indices);
  if (rc != KDEHIP_OK) return rc;
  if (Np == 0) return KDEHIP_OK;
  DeviceGuard guard;
  rc = guard.enter(plan->device);
  if (rc != KDEHIP_OK) return rc;
  const size_t D = plan->host.D, M = plan->host.M, L = plan->host.L;
  std::lock_guard<std::mutex> lock(plan->work_mutex);
  const size_t off_i = align256(sizeof(double) * D * Np);
  const size_t off_l = align256(off_i + sizeof(int64_t) * M * Np);
  rc = reserve_work(plan, off_l + (labels ? sizeof(int32_t) * M * L * Np : 0));
  if (rc != KDEHIP_OK) return rc;
  unsigned char *w = static_cast<unsigned char *>(plan->d_work);
  double *dp = reinterpret_cast<double *>(w);
  int64_t *di = reinterpret_cast<int64_t *>(w + off_i);
  int32_t *dl = labels ? reinterpret_cast<int32_t *>(w + off_l) : nullptr;
  // (this call's run is waited for by the blocking copies below, so it never marks the plan "async pending";
  // the flag is only ever set, by the device-pointer entry points, and read by kdehip_product_destroy)
  rc = enqueue_philox(plan, Np, Niter, seed, sample_offset, addEntropy, dp, di, dl, call_stream());
  if (rc != KDEHIP_OK) return rc;
  const size_t lab_bytes = labels ? sizeof(int32_t) * M * L * Np : 0;
  const OutPiece out[3] = {{points, 0, sizeof(double) * D * Np}, {indices, off_i, sizeof(int64_t) * M * Np},
                           {labels, off_l, lab_bytes}};
  (void)dl;
  return copy_out(w, off_l + lab_bytes, out, 3);
}

// ---- one-shot entry points (host buffers in, host buffers out, blocking; one or several GPUs) -----------------
namespace {

// Per-device state of a one-shot call.  Chains are split into contiguous ranges; nothing crosses
// devices: every device returns its own slice to the host.
struct Shard {
  kdehip_product *plan = nullptr;
  int device = 0;
  int64_t lo = 0, hi = 0;
  size_t off_p = 0, off_i = 0, off_l = 0, span = 0;  // layout of the plan's scratch: [streams ..][points][indices][labels]
  void *h_out = nullptr;                              // pinned landing zone of the results
  ~Shard() {
    if (h_out) cached_host_free(h_out, span);
    kdehip_product_destroy(plan);
  }
};

// randU == nullptr: device Philox stream keyed by (seed, global sample index); otherwise the caller's streams.
int one_shot(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
             const double *randU, int64_t nU, const double *randN, int64_t nN, uint64_t seed, int addEntropy,
             int ndims, const uint8_t *partialDimMask, int precision, int device, int ngpus, int32_t *labels,
             const uint8_t *manifold = nullptr) {
  // the enumerated manifolds (kdehip.h "manifolds"): bit d of circ_bits = dimension d is circular
  uint32_t circ_bits = 0;
  if (manifold) {
    if (ndims < 1 || ndims > KDEHIP_MAX_DIMS) return set_error(KDEHIP_ERR_UNSUPPORTED, "ndims outside 1..KDEHIP_MAX_DIMS");
    for (int d = 0; d < ndims; ++d) {
      if (manifold[d] > KDEHIP_MANIFOLD_CIRCULAR) return set_error(KDEHIP_ERR_ARG, "manifold: 0 (Euclidean) or 1 (circular) per dimension");
      if (manifold[d] == KDEHIP_MANIFOLD_CIRCULAR) circ_bits |= 1u << d;
    }
  }
  int rc = check_devices(device, ngpus);
  if (rc != KDEHIP_OK) {
    // argument errors of the product itself take precedence over "no device" only when they are detectable
    // without one: validate the densities first so that hosts without a GPU still get the reference's messages
    PackedProduct probe;
    const int prc = pack_layout(Ndens, trees, ndims, partialDimMask, precision == 32 ? 32 : 64, probe);
    return prc != KDEHIP_OK ? prc : rc;
  }
  // KDEHIP_TIMING=1: host-side phase times of this call on stderr (scripts/call_breakdown.py)
  static const bool timing = std::getenv("KDEHIP_TIMING") != nullptr;
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  auto us_since = [&](clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); };
  PlanImage im;
  struct DrainOnExit {
    std::vector<int> devs;
    ~DrainOnExit() {
      DeviceGuard g;
      for (int d : devs) if (g.enter(d) == KDEHIP_OK) (void)hipStreamSynchronize(call_stream());
    }
  };
  rc = build_image(im, Ndens, trees, ndims, partialDimMask, precision, /*force_generic=*/circ_bits != 0u);
  if (rc != KDEHIP_OK) return rc;
  const double us_pack = us_since(t_begin);
  if (Np < 0) return set_error(KDEHIP_ERR_ARG, "Np must be >= 0");
  if (Niter < 0) return set_error(KDEHIP_ERR_ARG, "Niter must be >= 0");
  if (Np > 0 && (!pts || !ind)) return set_error(KDEHIP_ERR_ARG, "null output pointer");
  if (Np == 0) return KDEHIP_OK;
  const size_t D = ndims, M = Ndens, L = im.host.L;
  const int64_t K = static_cast<int64_t>(M) * (1 + static_cast<int64_t>(L) * (Niter + 1));
  const int64_t R = static_cast<int64_t>(D) * (L + 1);
  const bool streams = randU != nullptr || randN != nullptr;
  if (streams) {
    if (!randU || nU < Np * K - 1)
      return set_error(KDEHIP_ERR_RAND_SHORT, "randU shorter than Np*K-1 values (Julia: BoundsError)");
    if (!randN || nN < Np * R)
      return set_error(KDEHIP_ERR_RAND_SHORT, "randN shorter than Np*R values (Julia: BoundsError)");
  }
  // the reference records a label only inside sampleIndex (:426): nothing with Niter = 0
  const bool trace = labels != nullptr && Niter > 0;
  if (ngpus > Np) ngpus = static_cast<int>(Np);
  std::vector<Shard> sh(ngpus);
  // Declared AFTER the shards (destroyed before them) and after the image: on an early error return every device
  // touched is drained first; only then do the shards hand their device blocks and pinned landing zones back to the
  // caches (another thread may be given them at once), and last of all the pinned image is recycled.
  DrainOnExit drain;
  DeviceGuard guard;
  // launch everywhere first (uploads and kernels of different devices overlap), then collect
  for (int g = 0; g < ngpus; ++g) {
    Shard &S = sh[g];
    S.device = phys(device + g);
    S.lo = share_begin(Np, g, ngpus);
    S.hi = share_begin(Np, g + 1, ngpus);
    const int64_t n = S.hi - S.lo;
    drain.devs.push_back(S.device);
    rc = instantiate(im, S.device, &S.plan, /*wait=*/false);  // the upload overlaps the host side of the launches
    if (rc != KDEHIP_OK) return rc;
    S.plan->dev.circ_bits = circ_bits;
    rc = guard.enter(S.device);
    if (rc != KDEHIP_OK) return rc;
    const int64_t useU = streams ? ((nU - S.lo * K < n * K) ? nU - S.lo * K : n * K) : 0, useN = streams ? n * R : 0;
    const size_t off_n = align256(sizeof(double) * useU);
    S.off_p = align256(off_n + sizeof(double) * useN);
    S.off_i = align256(S.off_p + sizeof(double) * D * n);
    S.off_l = align256(S.off_i + sizeof(int64_t) * M * n);
    const size_t lab_bytes = trace ? sizeof(int32_t) * M * L * n : 0;
    S.span = S.off_l + lab_bytes - S.off_p;
    rc = reserve_work(S.plan, S.off_l + lab_bytes);
    if (rc != KDEHIP_OK) return rc;
    unsigned char *w = static_cast<unsigned char *>(S.plan->d_work);
    double *dp = reinterpret_cast<double *>(w + S.off_p);
    int64_t *di = reinterpret_cast<int64_t *>(w + S.off_i);
    int32_t *dl = trace ? reinterpret_cast<int32_t *>(w + S.off_l) : nullptr;
    if (streams) {
      double *du = reinterpret_cast<double *>(w), *dn = reinterpret_cast<double *>(w + off_n);
      // sample s of this shard reads element (s - lo)*K + c - 1 of its slice = element s*K + c - 1 of the caller's array
      KDEHIP_CHECK(hipMemcpyAsync(du, randU + S.lo * K, sizeof(double) * useU, hipMemcpyHostToDevice, call_stream()));
      KDEHIP_CHECK(hipMemcpyAsync(dn, randN + S.lo * R, sizeof(double) * useN, hipMemcpyHostToDevice, call_stream()));
      rc = enqueue_streams(S.plan, n, Niter, du, useU, dn, useN, addEntropy, dp, di, dl, call_stream(), /*private_plan=*/true);
    } else {
      rc = enqueue_philox(S.plan, n, Niter, seed, S.lo, addEntropy, dp, di, dl, call_stream(), /*private_plan=*/true);
    }
    if (rc != KDEHIP_OK) return rc;
    KDEHIP_CHECK(cached_host_malloc(&S.h_out, S.span));
    KDEHIP_CHECK(hipMemcpyAsync(S.h_out, w + S.off_p, S.span, hipMemcpyDeviceToHost, call_stream()));
  }
  const double us_enqueue = us_since(t_begin);
  double us_wait = 0.0;
  for (int g = 0; g < ngpus; ++g) {
    Shard &S = sh[g];
    rc = guard.enter(S.device);
    if (rc != KDEHIP_OK) return rc;
    KDEHIP_CHECK(hipStreamSynchronize(call_stream()));
    if (g == ngpus - 1) us_wait = us_since(t_begin);
    const int64_t n = S.hi - S.lo;
    const unsigned char *h = static_cast<const unsigned char *>(S.h_out);
    std::memcpy(pts + S.lo * D, h, sizeof(double) * D * n);
    std::memcpy(ind + S.lo * M, h + (S.off_i - S.off_p), sizeof(int64_t) * M * n);
    if (trace) std::memcpy(labels + S.lo * M * L, h + (S.off_l - S.off_p), sizeof(int32_t) * M * L * n);
  }
  drain.devs.clear();  // every device has been waited for
  if (timing)
    std::fprintf(stderr, "kdehip one-shot: pack %.0f us | upload + launches enqueued %.0f us | device done %.0f us | results copied %.0f us\n",
                 us_pack, us_enqueue, us_wait, us_since(t_begin));
  return KDEHIP_OK;
}

}  // namespace

int kdehip_gibbs1(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                  int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                  int addEntropy, int ndims, const uint8_t *partialDimMask, int device) {
  return kdehip_gibbs1_multi(Ndens, trees, Np, Niter, pts, ind, randU, nU, randN, nN, addEntropy, ndims,
                             partialDimMask, device, 1, nullptr);
}

int kdehip_gibbs1_trace(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                        int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                        int addEntropy, int ndims, const uint8_t *partialDimMask, int device,
                        int32_t *labels) {
  return kdehip_gibbs1_multi(Ndens, trees, Np, Niter, pts, ind, randU, nU, randN, nN, addEntropy, ndims,
                             partialDimMask, device, 1, labels);
}

int kdehip_gibbs1_multi(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                        int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                        int addEntropy, int ndims, const uint8_t *partialDimMask, int device, int ngpus,
                        int32_t *labels) {
  static const double kNone = 0.0;  // (a null stream pointer must mean "too short", not "use Philox")
  return one_shot(Ndens, trees, Np, Niter, pts, ind, randU ? randU : &kNone, randU ? nU : 0, randN ? randN : &kNone,
                  randN ? nN : 0, 0, addEntropy, ndims, partialDimMask, 64, device, ngpus, labels);
}

int kdehip_gibbs1_manifold(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
                           const double *randU, int64_t nU, const double *randN, int64_t nN, int addEntropy, int ndims,
                           const uint8_t *partialDimMask, const uint8_t *manifold, int device, int32_t *labels) {
  static const double kNone = 0.0;
  return one_shot(Ndens, trees, Np, Niter, pts, ind, randU ? randU : &kNone, randU ? nU : 0, randN ? randN : &kNone,
                  randN ? nN : 0, 0, addEntropy, ndims, partialDimMask, 64, device, 1, labels, manifold);
}

int kdehip_prod_philox(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
                       uint64_t seed, int addEntropy, int ndims, const uint8_t *partialDimMask, int precision,
                       int device, int ngpus, int32_t *labels) {
  return one_shot(Ndens, trees, Np, Niter, pts, ind, nullptr, 0, nullptr, 0, seed, addEntropy, ndims, partialDimMask,
                  precision, device, ngpus, labels);
}

// ---- products of densities that live in HBM (pack_device.hip) ------------------------------------------------------
namespace {

constexpr int kMaxDevices = 64;  // devices the bookkeeping below has slots for (a node has 8)

// Plans of enqueue-only calls live until the work that uses them has run: they wait here, with an event recorded behind
// their last launch, and are released by later calls (or kdehip_clear_cache) once the event has fired.  One entry = one
// call: the plan of kdehip_prod_philox_device, or all plans of a kdehip_prod_philox_batch (they share one device block).
struct PendingPlan {
  std::vector<kdehip_product *> plans;  // descriptors only when `d_blob` is set (the block below is theirs, shared)
  void *d_blob = nullptr;               // a batch's one device block
  size_t blob_bytes = 0;
  hipEvent_t done = nullptr;
  void *h_desc = nullptr;
  size_t h_bytes = 0;
  hipStream_t stream = nullptr;           // the caller's stream the work was enqueued on
  hipEvent_t prepared = nullptr;          // tiles and tables are in place (recorded on the device's preparation stream)
  hipEvent_t t_begin = nullptr, t_end = nullptr;  // kdehip_profile_sampler: around the sampling launch, on the caller's stream
  int device = 0;
};

// kdehip_profile_sampler: durations of the sampling launches of released plans, per device and caller stream
std::atomic<int> g_profile_sampler{0};
std::mutex g_profile_mu;
struct ProfileSum { hipStream_t stream; double ms; long long launches; };
std::vector<ProfileSum> g_profile[kMaxDevices];

// Preparing a product of resident densities (descriptor upload, tile gather, conditional tables: ~35 us of small,
// latency-bound launches) does not depend on anything the caller's stream holds -- the densities are immutable, the
// plan's block is its own -- so it runs on a stream of the library's and the sampler waits for it with an event.  A caller
// that enqueues products back to back gets product k+1 prepared WHILE product k samples (the sampler leaves 112
// registers per SIMD and 19 KB of LDS per CU free: the small kernels fit beside it).
std::mutex g_prep_mu;
hipStream_t g_prep_stream[kMaxDevices];
bool g_prep_tried[kMaxDevices];
hipStream_t prep_stream(int device) {  // (the device is current)
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_prep_mu);
  if (!g_prep_tried[device]) {
    g_prep_tried[device] = true;
    if (hipStreamCreateWithFlags(&g_prep_stream[device], hipStreamNonBlocking) != hipSuccess) {
      g_prep_stream[device] = nullptr;
      (void)hipGetLastError();
    }
  }
  return g_prep_stream[device];
}
std::mutex g_pending_mu;
std::deque<PendingPlan> g_pending[kMaxDevices];
constexpr size_t kMaxPending = 8;  // per (device, stream)

void release_pending(PendingPlan &pp) {
  if (pp.t_begin && pp.t_end) {  // (the work is over: `done` was recorded behind t_end)
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, pp.t_begin, pp.t_end) == hipSuccess && pp.device >= 0 && pp.device < kMaxDevices) {
      std::lock_guard<std::mutex> lock(g_profile_mu);
      ProfileSum *ps = nullptr;
      for (ProfileSum &c : g_profile[pp.device]) if (c.stream == pp.stream) ps = &c;
      if (!ps) { g_profile[pp.device].push_back(ProfileSum{pp.stream, 0.0, 0}); ps = &g_profile[pp.device].back(); }
      ps->ms += ms;
      ps->launches += 1;
    }
    (void)hipGetLastError();
  }
  if (pp.t_begin) (void)hipEventDestroy(pp.t_begin);
  if (pp.t_end) (void)hipEventDestroy(pp.t_end);
  if (pp.done) (void)hipEventDestroy(pp.done);
  if (pp.prepared) (void)hipEventDestroy(pp.prepared);
  if (pp.h_desc) cached_host_free(pp.h_desc, pp.h_bytes);
  for (kdehip_product *p : pp.plans) {
    if (!p) continue;
    if (!pp.d_blob && p->d_blob) cached_free(p->d_blob, p->blob_bytes);
    if (p->d_work) cached_free(p->d_work, p->work_cap);
    delete p;
  }
  if (pp.d_blob) cached_free(pp.d_blob, pp.blob_bytes);
}
// (current device = `device`)  Releases every queued call whose work is over, wherever it sits in the queue.  A caller
// with more than kMaxPending calls in flight ON ITS OWN STREAM then waits for the oldest of THOSE -- outside the lock,
// and never for another stream's work: a stalled stream (a long kernel, an event another host thread records later)
// holds up neither the other threads nor, through them, itself.  all = true (kdehip_clear_cache, the profile read-out):
// wait for everything.
void reap_pending(int device, bool all, hipStream_t mine = nullptr) {
  if (device < 0 || device >= kMaxDevices) return;
  for (;;) {
    std::vector<PendingPlan> finished;
    PendingPlan wait_for;
    bool have_wait = false;
    {
      std::lock_guard<std::mutex> lock(g_pending_mu);
      auto &q = g_pending[device];
      size_t of_mine = 0;
      for (auto it = q.begin(); it != q.end();) {
        if (hipEventQuery(it->done) == hipSuccess) { finished.push_back(std::move(*it)); it = q.erase(it); }
        else { if (!all && it->stream == mine) ++of_mine; ++it; }
      }
      (void)hipGetLastError();  // (hipEventQuery reports "not ready" as an error)
      if (all ? !q.empty() : of_mine > kMaxPending) {
        for (auto it = q.begin(); it != q.end(); ++it)
          if (all || it->stream == mine) { wait_for = std::move(*it); q.erase(it); have_wait = true; break; }
      }
    }
    for (PendingPlan &f : finished) release_pending(f);
    if (!have_wait) return;
    (void)hipEventSynchronize(wait_for.done);
    release_pending(wait_for);
  }
}

}  // namespace

}  // extern "C"
void kdehip::drain_pending() {  // kdehip_clear_cache: nothing may stay behind
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return;
  DeviceGuard guard;
  for (int d = 0; d < n && d < kMaxDevices; ++d)
    if (guard.enter(d) == KDEHIP_OK) reap_pending(d, true);
}

extern "C" {

namespace {

// The layout of one product of resident densities inside a device block: [counter | levels | table descriptors] in the
// block's head (what crosses PCIe), [permutation | tiles | tables] in its body (written by the GPU).
struct ResidentLayout {
  size_t off_lev = 0, off_count = 0, off_tab = 0, head_end = 0;  // head, relative to the head's start
  size_t off_perm = 0, off_data = 0, off_tables = 0, body_end = 0;  // body, relative to the body's start
};

// Validates one product of resident densities and lays it out from the shapes of its densities' frontiers alone
// (p->host, p->precision, p->mode, p->device are set; nothing is allocated).
int layout_resident(int Ndens, kdehip_device_density *const *trees, const uint8_t *partialDimMask, int precision,
                    kdehip_product *p, ResidentLayout &lay) {
  if (Ndens < 1 || !trees) return set_error(KDEHIP_ERR_ARG, "need at least one density");
  if (Ndens > KDEHIP_MAX_DENS) return set_error(KDEHIP_ERR_UNSUPPORTED, "more than KDEHIP_MAX_DENS densities in one product");
  for (int j = 0; j < Ndens; ++j) {
    if (!trees[j]) return set_error(KDEHIP_ERR_ARG, "null density");
    if (trees[j]->D != trees[0]->D) return set_error(KDEHIP_ERR_DIM_MISMATCH, "kdes must have same dimension");
    if (trees[j]->device != trees[0]->device) return set_error(KDEHIP_ERR_ARG, "densities on different devices");
  }
  const int M = Ndens, D = trees[0]->D;
  int64_t maxN = 0;
  for (int j = 0; j < M; ++j) if (trees[j]->N > maxN) maxN = trees[j]->N;
  const int L = nlevels_for(maxN);
  // the layout, from shapes alone: beyond a density's own depth its frontier stays what it was (all leaves)
  std::vector<TileShape> shapes(static_cast<size_t>(M) * (L + 1));
  double lo[KDEHIP_MAX_DIMS], hi[KDEHIP_MAX_DIMS];
  for (int d = 0; d < KDEHIP_MAX_DIMS; ++d) { lo[d] = INFINITY; hi[d] = 0.0; }
  bool finite_ok = true;
  for (int j = 0; j < M; ++j) {
    const kdehip_device_density &t = *trees[j];
    for (int l = 0; l <= L; ++l) {
      const int lj = l < t.Lown ? l : t.Lown;
      shapes[static_cast<size_t>(j) * (L + 1) + l] = {t.fr.off[lj + 1] - t.fr.off[lj], t.fr.uniform[lj] != 0, t.fr.uratio[lj]};
    }
    if (t.fr.bad) finite_ok = false;
    for (int d = 0; d < D; ++d) {
      if (t.fr.lo[d] < lo[d]) lo[d] = t.fr.lo[d];
      if (t.fr.hi[d] > hi[d]) hi[d] = t.fr.hi[d];
    }
  }
  const bool fast = finite_ok && variances_in_range(lo, hi, D, precision);
  const int rc = pack_layout_shapes(M, D, L, shapes.data(), partialDimMask, precision, fast, p->host);
  if (rc != KDEHIP_OK) return rc;
  p->device = trees[0]->device;
  p->precision = precision;
  p->fast = p->host.fast;
  p->mode = !p->host.fast ? kModeGeneric : (p->host.all_active ? kModeFast : kModeFastMasked);
  const size_t ntab = p->host.tabdesc.size();
  const size_t esz = (precision == 64) ? sizeof(double) : sizeof(float);
  lay.off_lev = 256;
  lay.off_count = lay.off_lev - sizeof(unsigned long long);
  lay.off_tab = align256(lay.off_lev + level_table_bytes(p->host));
  lay.head_end = align256(lay.off_tab + ntab * sizeof(TabDesc));
  lay.off_perm = 0;
  lay.off_data = align256(static_cast<size_t>(p->host.perm_elems) * sizeof(int32_t));
  lay.off_tables = align256(lay.off_data + static_cast<size_t>(p->host.data_elems) * esz);
  lay.body_end = align256(lay.off_tables + static_cast<size_t>(p->host.tab_entries) * esz);
  return KDEHIP_OK;
}

// Descriptors of product p into the pinned head at hb (its part starts at `head`), its fill jobs to `jobs`; binds the
// plan's device pointers (block d_blob, head at `head`, body at `body`).  Returns the largest tile's rows per lane.
int describe_resident(kdehip_product *p, kdehip_device_density *const *trees, const ResidentLayout &lay, void *d_blob,
                      unsigned char *hb, size_t head, size_t body, FillJob *jobs) {
  const int M = p->host.M, L = p->host.L, D = p->host.D;
  const size_t ntab = p->host.tabdesc.size();
  const size_t esz = (p->precision == 64) ? sizeof(double) : sizeof(float);
  std::memset(hb + head, 0, lay.off_lev);
  copy_level_table(p->host, hb + head + lay.off_lev);
  std::memcpy(hb + head + lay.off_tab, p->host.tabdesc.data(), ntab * sizeof(TabDesc));
  p->d_blob = d_blob;
  bind_plan(p, head + lay.off_lev, head + lay.off_count, head + lay.off_tab, body + lay.off_perm, body + lay.off_data,
            body + lay.off_tables, lay.head_end + lay.body_end);
  int maxB = 1;
  unsigned char *data = static_cast<unsigned char *>(p->d_data);
  for (int j = 0; j < M; ++j)
    for (int l = 0; l <= L; ++l) {
      const size_t idx = static_cast<size_t>(j) * (L + 1) + l;
      const LevelDesc &ds = p->host.levels[idx];
      const kdehip_device_density &t = *trees[j];
      const int lj = l < t.Lown ? l : t.Lown;
      jobs[idx] = FillJob{t.means, t.bandwidth, t.weights, t.perm, t.front + t.fr.off[lj],
                          data + static_cast<size_t>(ds.hdr_off) * esz, p->d_perm + ds.perm_off,
                          ds.n, ds.B, ds.F, ds.uniform_bw, D, 0};
      if (ds.B > maxB) maxB = ds.B;
    }
  return maxB;
}

// own_prep: prepare on the library's stream (an asynchronous caller: the next product's preparation overlaps this
// product's sampling); a blocking caller, who waits for every product, keeps everything on its own stream
int prod_philox_device(int Ndens, kdehip_device_density *const *trees, int64_t Np, int Niter, uint64_t seed,
                       int64_t sample_offset, int addEntropy, const uint8_t *partialDimMask, int precision,
                       double *d_points, int64_t *d_indices, int32_t *d_labels, void *stream, bool own_prep) {
  if (precision != 64 && precision != 32) return set_error(KDEHIP_ERR_ARG, "precision must be 64 or 32");
  if (Np < 0) return set_error(KDEHIP_ERR_ARG, "Np must be >= 0");
  if (Niter < 0) return set_error(KDEHIP_ERR_ARG, "Niter must be >= 0");
  if (Np > 0 && (!d_points || !d_indices)) return set_error(KDEHIP_ERR_ARG, "null output pointer");
  kdehip_product *p = new (std::nothrow) kdehip_product();
  if (!p) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
  ResidentLayout lay;
  int rc = layout_resident(Ndens, trees, partialDimMask, precision, p, lay);
  if (rc != KDEHIP_OK || Np == 0) { delete p; return rc; }
  const int device = p->device;
  if (device < 0 || device >= kMaxDevices) { delete p; return set_error(KDEHIP_ERR_UNSUPPORTED, "device ordinal beyond the library's bookkeeping (64)"); }
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != KDEHIP_OK) { delete p; return rc; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  reap_pending(device, false, st);
  // the block: head [counter | levels | table descriptors | fill jobs], body [permutation | tiles | tables]
  const size_t nlev = p->host.levels.size();
  const size_t off_jobs = lay.head_end, head_bytes = align256(off_jobs + nlev * sizeof(FillJob));
  const size_t total = head_bytes + lay.body_end;
  PendingPlan pend;
  pend.plans.push_back(p);
  pend.h_bytes = head_bytes;
  pend.stream = st;
  pend.device = device;
  hipStream_t prep = own_prep ? prep_stream(device) : nullptr;
  void *blob = nullptr;
  hipError_t e = cached_malloc(&blob, total);
  if (e == hipSuccess) { p->d_blob = blob; p->blob_bytes = total; }
  if (e == hipSuccess) e = cached_host_malloc(&pend.h_desc, pend.h_bytes);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&pend.done, hipEventDisableTiming);
  if (e == hipSuccess && prep) e = hipEventCreateWithFlags(&pend.prepared, hipEventDisableTiming);
  if (e == hipSuccess && g_profile_sampler.load(std::memory_order_relaxed)) {
    e = hipEventCreate(&pend.t_begin);
    if (e == hipSuccess) e = hipEventCreate(&pend.t_end);
  }
  if (e != hipSuccess) {
    release_pending(pend);
    return set_error(KDEHIP_ERR_HIP, std::string("device product: ") + hipGetErrorString(e));
  }
  // descriptors: a few KB from pinned memory, in front of the launches
  unsigned char *hb = static_cast<unsigned char *>(pend.h_desc);
  const int maxB = describe_resident(p, trees, lay, blob, hb, 0, head_bytes, reinterpret_cast<FillJob *>(hb + off_jobs));
  p->blob_bytes = total;
  hipStream_t ps = prep ? prep : st;  // where the plan is prepared
  auto fail = [&](int code) {  // (nothing of this plan has been handed to the queue yet)
    if (prep) (void)hipStreamSynchronize(prep);
    (void)hipStreamSynchronize(st);
    release_pending(pend);
    return code;
  };
  if (hipMemcpyAsync(blob, hb, head_bytes, hipMemcpyHostToDevice, ps) != hipSuccess)
    return fail(set_error(KDEHIP_ERR_HIP, "device product: descriptor upload failed"));
  rc = launch_fill_tiles(precision, reinterpret_cast<const FillJob *>(static_cast<unsigned char *>(blob) + off_jobs),
                         static_cast<int>(nlev), maxB, ps);
  if (rc != KDEHIP_OK) return fail(rc);
  rc = enqueue_philox(p, Np, Niter, seed, sample_offset, addEntropy, d_points, d_indices, d_labels, stream,
                      /*private_plan=*/true, nullptr, prep, pend.prepared, pend.t_begin, pend.t_end);
  if (rc != KDEHIP_OK) return fail(rc);
  if (hipEventRecord(pend.done, st) != hipSuccess) return fail(set_error(KDEHIP_ERR_HIP, "device product: hipEventRecord failed"));
  {
    std::lock_guard<std::mutex> lock(g_pending_mu);
    g_pending[device].push_back(std::move(pend));
  }
  return KDEHIP_OK;
}

// whether a product can ride in a batched launch of the register-resident sampler (gibbs_lean.hip, BATCH instantiations:
// fp64 products of 2..4 densities with every dimension active)
bool batchable(const kdehip_product *p) {
  const int M = p->host.M, L = p->host.L, D = p->host.D;
  return p->precision == 64 && p->mode == kModeFast && M >= 2 && M <= 4 && D * (L + 1) <= 128;
}

}  // namespace

// Many products in ONE call (the serving pattern: a belief-propagation sweep issues dozens of 100-300-chain products, each
// of which fills a fraction of the device and is latency bound): one device block, one descriptor upload, one gather launch
// for all tiles, and ONE sampling launch per (dimension count, density count) group -- workgroups indexed by (product,
// chain block), every workgroup fetching its product's plan through the scalar cache.  Each product's result is bit for
// bit what kdehip_prod_philox_device gives for it (same layout, same kernel code, same Philox keys).  Products outside
// the batched kernel's domain (fp32, masks, 1 or more than 4 densities) are enqueued one by one inside the same call.
int kdehip_prod_philox_batch(int nprod, const kdehip_batch_item *items, int precision, void *stream) {
  if (nprod < 0 || (nprod > 0 && !items)) return set_error(KDEHIP_ERR_ARG, "kdehip_prod_philox_batch: bad item list");
  if (precision != 64 && precision != 32) return set_error(KDEHIP_ERR_ARG, "precision must be 64 or 32");
  if (nprod == 0) return KDEHIP_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PendingPlan pend;
  pend.stream = st;
  struct Abort {  // on an error before the hand-over to the queue: nothing has been enqueued that uses the block
    PendingPlan *pp; bool armed = true;
    // (the blocks go back to the cache of the CURRENT device, and the function's own DeviceGuard -- declared later, destroyed
    // earlier -- has restored the caller's by now: enter the batch's device again for the release)
    ~Abort() {
      if (!armed) return;
      DeviceGuard g;
      (void)g.enter(pp->device);
      release_pending(*pp);
    }
  } abort_guard{&pend};
  std::vector<ResidentLayout> lays(nprod);
  std::vector<size_t> head_at(nprod), body_at(nprod), jobs_at(nprod);
  size_t head = 0, body = 0, njobs = 0;
  int device = -1;
  for (int i = 0; i < nprod; ++i) {
    const kdehip_batch_item &it = items[i];
    if (it.Np < 0) return set_error(KDEHIP_ERR_ARG, "Np must be >= 0");
    if (it.Niter < 0) return set_error(KDEHIP_ERR_ARG, "Niter must be >= 0");
    if (it.sample_offset < 0) return set_error(KDEHIP_ERR_ARG, "sample_offset must be >= 0");
    if (it.Np > 0 && (!it.d_points || !it.d_indices)) return set_error(KDEHIP_ERR_ARG, "null output pointer");
    kdehip_product *p = new (std::nothrow) kdehip_product();
    if (!p) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
    pend.plans.push_back(p);
    const int rc = layout_resident(it.Ndens, it.trees, it.partialDimMask, precision, p, lays[i]);
    if (rc != KDEHIP_OK) return rc;
    if (device < 0) device = p->device;
    if (p->device != device) return set_error(KDEHIP_ERR_ARG, "kdehip_prod_philox_batch: products on different devices");
    head_at[i] = head; head += lays[i].head_end;
    body_at[i] = body; body += lays[i].body_end;
    jobs_at[i] = njobs; njobs += p->host.levels.size();
  }
  if (device < 0 || device >= kMaxDevices) return set_error(KDEHIP_ERR_UNSUPPORTED, "device ordinal beyond the library's bookkeeping (64)");
  pend.device = device;
  // groups of batchable products by (D, M); everything else runs one by one
  struct Group {
    int D, M;
    std::vector<int> members;
    int64_t blocks = 0;
    size_t ent_at = 0, map_at = 0;
    // the members whose conditional tables are worth building (gibbs_kernel.hip "conditional tables"), their table rows in
    // workgroups of kTabWaves wavefronts (one wavefront per row)
    std::vector<int> tabbed;
    int64_t tab_blocks = 0;
    size_t tent_at = 0, tmap_at = 0;
  };
  constexpr int kTabWaves = 4;
  std::vector<Group> groups;
  std::vector<int> singles;
  constexpr int kBatchWaves = 16;  // chains per workgroup of the batched instantiations
  for (int i = 0; i < nprod; ++i) {
    if (items[i].Np == 0) continue;
    kdehip_product *p = pend.plans[i];
    if (!batchable(p)) { singles.push_back(i); continue; }
    Group *g = nullptr;
    for (Group &c : groups) if (c.D == p->host.D && c.M == p->host.M) g = &c;
    if (!g) { groups.emplace_back(); g = &groups.back(); g->D = p->host.D; g->M = p->host.M; }
    g->members.push_back(i);
    g->blocks += (items[i].Np + kBatchWaves - 1) / kBatchWaves;
  }
  for (size_t k = 0; k < groups.size();)  // a group of one gains nothing from the batched kernel
    if (groups[k].members.size() == 1) { singles.push_back(groups[k].members[0]); groups.erase(groups.begin() + k); } else ++k;
  // Whether a member's conditional tables pay for themselves INSIDE the batch: a table row costs about as much as a full
  // step of one chain, a tabulated sweep step saves ~60 % of one -- and the many small products a batch is made of have few
  // chains per table row (config 2's shape: 4,092 rows for 256 chains).  Measured on MI355X (profiles/r04_experiments.md section 6):
  // tables when chains x tabulated levels x sweeps x densities >= 6.5 x rows.  KDEHIP_BATCH_TABLES=0/1 forces never/always.
  static const int force_tables = [] { const char *e = std::getenv("KDEHIP_BATCH_TABLES"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  auto has_tables = [&](int i) {
    const kdehip_product *p = pend.plans[i];
    if (p->mode == kModeGeneric || p->host.Lt <= 0 || p->host.tab_rows <= 0 || items[i].Np < kTabMinChains) return false;
    if (force_tables >= 0) return force_tables == 1;
    const double saved = static_cast<double>(items[i].Np) * p->host.Lt * items[i].Niter * p->host.M;
    return saved >= 6.5 * static_cast<double>(p->host.tab_rows);
  };
  for (Group &g : groups)
    for (int i : g.members)
      if (has_tables(i)) {
        g.tabbed.push_back(i);
        g.tab_blocks += (pend.plans[i]->host.tab_rows + kTabWaves - 1) / kTabWaves;
      }
  const size_t off_jobs = head;
  size_t at = align256(off_jobs + njobs * sizeof(FillJob));
  for (Group &g : groups) {
    if (g.blocks > (int64_t(1) << 31) - 1) return set_error(KDEHIP_ERR_UNSUPPORTED, "kdehip_prod_philox_batch: too many chains for one launch");
    if (g.tab_blocks > (int64_t(1) << 31) - 1) return set_error(KDEHIP_ERR_UNSUPPORTED, "kdehip_prod_philox_batch: too many table rows for one launch");
    g.ent_at = at; at = align256(at + g.members.size() * sizeof(BatchEntry));
    g.map_at = at; at = align256(at + static_cast<size_t>(g.blocks) * sizeof(int32_t));
    g.tent_at = at; at = align256(at + g.tabbed.size() * sizeof(BatchEntry));
    g.tmap_at = at; at = align256(at + static_cast<size_t>(g.tab_blocks) * sizeof(int32_t));
  }
  const size_t head_bytes = at, total = head_bytes + body;
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;
  reap_pending(device, false, st);
  pend.h_bytes = head_bytes;
  pend.blob_bytes = total;
  hipError_t e = cached_malloc(&pend.d_blob, total);
  if (e != hipSuccess) pend.d_blob = nullptr;
  if (e == hipSuccess) e = cached_host_malloc(&pend.h_desc, pend.h_bytes);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&pend.done, hipEventDisableTiming);
  if (e != hipSuccess) return set_error(KDEHIP_ERR_HIP, std::string("batched products: ") + hipGetErrorString(e));
  unsigned char *hb = static_cast<unsigned char *>(pend.h_desc), *db = static_cast<unsigned char *>(pend.d_blob);
  int maxB = 1;
  for (int i = 0; i < nprod; ++i) {
    const int b = describe_resident(pend.plans[i], items[i].trees, lays[i], pend.d_blob, hb, head_at[i], head_bytes + body_at[i],
                                    reinterpret_cast<FillJob *>(hb + off_jobs) + jobs_at[i]);
    if (b > maxB) maxB = b;
  }
  for (Group &g : groups) {
    BatchEntry *ent = reinterpret_cast<BatchEntry *>(hb + g.ent_at);
    int32_t *map = reinterpret_cast<int32_t *>(hb + g.map_at);
    int32_t block = 0;
    for (size_t k = 0; k < g.members.size(); ++k) {
      const int i = g.members[k];
      const kdehip_batch_item &it = items[i];
      BatchEntry be{};
      const PlanDev &pd = pend.plans[i]->dev;
      be.head = BatchPlanHead{pd.data, pd.perm, pd.levels, pd.tables, pd.tabdesc, pd.tab_rows_total, pd.M, pd.L, pd.D, pd.Lt};
      be.run.Np = it.Np; be.run.seed = it.seed; be.run.sample_offset = it.sample_offset;
      be.run.points = it.d_points; be.run.indices = it.d_indices; be.run.labels = it.d_labels;
      be.flags.Niter = it.Niter; be.flags.addEntropy = it.addEntropy ? 1 : 0; be.flags.use_tables = has_tables(i) ? 1 : 0;
      be.flags.first_block = block;
      ent[k] = be;
      const int32_t nb = static_cast<int32_t>((it.Np + kBatchWaves - 1) / kBatchWaves);
      for (int32_t q = 0; q < nb; ++q) map[block + q] = static_cast<int32_t>(k);
      block += nb;
    }
    // the table launch of the group: the same entries, numbered from the product's first TABLE workgroup
    BatchEntry *tent = reinterpret_cast<BatchEntry *>(hb + g.tent_at);
    int32_t *tmap = reinterpret_cast<int32_t *>(hb + g.tmap_at);
    int32_t tblock = 0, kt = 0;
    for (size_t k = 0; k < g.members.size(); ++k) {
      if (!ent[k].flags.use_tables) continue;
      tent[kt] = ent[k];
      tent[kt].flags.first_block = tblock;
      const int32_t nb = static_cast<int32_t>((pend.plans[g.members[k]]->host.tab_rows + kTabWaves - 1) / kTabWaves);
      for (int32_t q = 0; q < nb; ++q) tmap[tblock + q] = kt;
      tblock += nb;
      ++kt;
    }
  }
  auto fail = [&](int code) {  // work may have been enqueued: wait for it before the block goes back to the cache
    (void)hipStreamSynchronize(st);
    return code;
  };
  if (hipMemcpyAsync(pend.d_blob, hb, head_bytes, hipMemcpyHostToDevice, st) != hipSuccess)
    return fail(set_error(KDEHIP_ERR_HIP, "batched products: descriptor upload failed"));
  rc = launch_fill_tiles(precision, reinterpret_cast<const FillJob *>(db + off_jobs), static_cast<int>(njobs), maxB, st);
  if (rc != KDEHIP_OK) return fail(rc);
  for (const Group &g : groups) {
    if (g.tab_blocks > 0) {  // one launch fills the conditional tables of every member that has them
      RunArgs t{};
      t.table_build = 1;
      t.batch = reinterpret_cast<const BatchEntry *>(db + g.tent_at);
      t.batch_map = reinterpret_cast<const int32_t *>(db + g.tmap_at);
      t.Np = g.tab_blocks * kTabWaves;
      rc = launch_tables_batch(g.D, pend.plans[g.tabbed[0]]->dev, t, st);
      if (rc != KDEHIP_OK) return fail(rc);
    }
    RunArgs a{};
    a.rng_philox = 1;
    a.batch = reinterpret_cast<const BatchEntry *>(db + g.ent_at);
    a.batch_map = reinterpret_cast<const int32_t *>(db + g.map_at);
    a.Np = g.blocks * kBatchWaves;  // (the launcher's grid: g.blocks workgroups)
    rc = launch_gibbs_batch(g.D, g.M, pend.plans[g.members[0]]->dev, a, st);
    if (rc != KDEHIP_OK) return fail(rc);
  }
  for (int i : singles) {
    const kdehip_batch_item &it = items[i];
    rc = enqueue_philox(pend.plans[i], it.Np, it.Niter, it.seed, it.sample_offset, it.addEntropy, it.d_points, it.d_indices,
                        it.d_labels, stream, /*private_plan=*/true);
    if (rc != KDEHIP_OK) return fail(rc);
  }
  if (hipEventRecord(pend.done, st) != hipSuccess) return fail(set_error(KDEHIP_ERR_HIP, "batched products: hipEventRecord failed"));
  abort_guard.armed = false;
  {
    std::lock_guard<std::mutex> lock(g_pending_mu);
    g_pending[device].push_back(std::move(pend));
  }
  return KDEHIP_OK;
}

}  // extern "C"
int kdehip::prod_philox_device_blocking_stream(int Ndens, kdehip_device_density *const *trees, int64_t Np, int Niter,
                                               uint64_t seed, int64_t sample_offset, int addEntropy,
                                               const uint8_t *partialDimMask, int precision, double *d_points,
                                               int64_t *d_indices, void *stream) {
  return prod_philox_device(Ndens, trees, Np, Niter, seed, sample_offset, addEntropy, partialDimMask, precision, d_points,
                            d_indices, nullptr, stream, /*own_prep=*/false);
}
extern "C" {

int kdehip_prod_philox_device(int Ndens, kdehip_device_density *const *trees, int64_t Np, int Niter, uint64_t seed,
                              int64_t sample_offset, int addEntropy, const uint8_t *partialDimMask, int precision,
                              double *d_points, int64_t *d_indices, int32_t *d_labels, void *stream) {
  return prod_philox_device(Ndens, trees, Np, Niter, seed, sample_offset, addEntropy, partialDimMask, precision, d_points,
                            d_indices, d_labels, stream, /*own_prep=*/true);
}

// Diagnostic: time the sampling launch of every kdehip_prod_philox_device call with a pair of events on the caller's
// stream (what bench.py reports as the kernel's duration INSIDE its timed region).
void kdehip_profile_sampler(int enable) {
  std::lock_guard<std::mutex> lock(g_profile_mu);
  for (int d = 0; d < kMaxDevices; ++d) g_profile[d].clear();
  g_profile_sampler.store(enable ? 1 : 0, std::memory_order_relaxed);
  profile_phases_set(enable != 0);  // (phase_timer.hpp: LOOCV search, evaluation, GPU tree build)
}
int kdehip_profile_sampler_read(int device, void *stream, double *total_ms, int64_t *launches) {
  if (device < 0 || device >= kMaxDevices) return set_error(KDEHIP_ERR_ARG, "device ordinal outside 0..63");
  DeviceGuard guard;
  const int rc = guard.enter(device);
  if (rc != KDEHIP_OK) return rc;
  reap_pending(device, true);  // (waits for the plans still in flight: their launches count too)
  std::lock_guard<std::mutex> lock(g_profile_mu);
  double ms = 0.0;
  long long n = 0;
  for (const ProfileSum &c : g_profile[device])
    if (c.stream == static_cast<hipStream_t>(stream)) { ms += c.ms; n += c.launches; }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  return KDEHIP_OK;
}

// The same with HOST output buffers, blocking: what a host without device arrays of its own (a Julia caller without
// AMDGPU.jl) uses once its densities are uploaded -- no host re-layout, no upload of tiles, one copy back.
int kdehip_prod_philox_resident(int Ndens, kdehip_device_density *const *trees, int64_t Np, int Niter, uint64_t seed,
                                int addEntropy, const uint8_t *partialDimMask, int precision, double *pts, int64_t *ind) {
  if (Ndens < 1 || !trees || !trees[0]) return set_error(KDEHIP_ERR_ARG, "need at least one density");
  if (Np < 0) return set_error(KDEHIP_ERR_ARG, "Np must be >= 0");
  if (Np > 0 && (!pts || !ind)) return set_error(KDEHIP_ERR_ARG, "null output pointer");
  if (Np == 0) return KDEHIP_OK;
  const size_t D = trees[0]->D, M = Ndens;
  DeviceGuard guard;
  int rc = guard.enter(trees[0]->device);
  if (rc != KDEHIP_OK) return rc;
  const size_t off_i = align256(sizeof(double) * D * Np), span = off_i + sizeof(int64_t) * M * Np;
  void *d_out = nullptr, *h_out = nullptr;
  KDEHIP_CHECK(cached_malloc(&d_out, span));
  hipError_t e = cached_host_malloc(&h_out, span);
  if (e != hipSuccess) { cached_free(d_out, span); return set_error(KDEHIP_ERR_HIP, "pinned result block"); }
  unsigned char *w = static_cast<unsigned char *>(d_out);
  rc = prod_philox_device(Ndens, trees, Np, Niter, seed, 0, addEntropy, partialDimMask, precision,
                          reinterpret_cast<double *>(w), reinterpret_cast<int64_t *>(w + off_i), nullptr, call_stream(),
                          /*own_prep=*/false);
  if (rc == KDEHIP_OK) e = hipMemcpyAsync(h_out, d_out, span, hipMemcpyDeviceToHost, call_stream());
  const hipError_t se = hipStreamSynchronize(call_stream());  // (also before the blocks go back to the caches on an error)
  if (rc == KDEHIP_OK && e == hipSuccess && se == hipSuccess) {
    std::memcpy(pts, h_out, sizeof(double) * D * Np);
    std::memcpy(ind, static_cast<unsigned char *>(h_out) + off_i, sizeof(int64_t) * M * Np);
  }
  cached_host_free(h_out, span);
  cached_free(d_out, span);
  if (rc != KDEHIP_OK) return rc;
  if (e != hipSuccess || se != hipSuccess) return set_error(KDEHIP_ERR_HIP, "device product: result copy failed");
  return KDEHIP_OK;
}

// ---- resident multi-GPU plans: one plan per device, chains in contiguous ranges, one all-gather ----------------
struct kdehip_product_multi {
  int first_device = 0, ngpus = 0;
  std::vector<kdehip_product *> plans;
  std::vector<hipEvent_t> done;   // per device: its slice has been written to every device
  std::vector<hipEvent_t> ready;  // per device: the work queued on its stream before this call is over (its arrays may be overwritten)
  bool peer_stores = true;        // every device can store into every other device's memory (else: peer copies)
  int last_transfers = -1;        // copy-engine transfers per device of the last product (-1: none yet)
  std::vector<hipEvent_t> t_begin, t_end;  // kdehip_profile_sampler: around device g's sampling launch (created on demand)
  bool timed = false;             // the last product was bracketed
  std::vector<char> timed_dev;    // ... and device g had chains in it (an empty slice records no events)
  // verdicts of peer_can_store per (array, writing device): a caller passes the same arrays product after product, and
  // a look-up is up to three driver queries (112 look-ups per product at 8 GPUs)
  // The verdicts are keyed by raw pointer: an array freed and re-allocated at the same address from ANOTHER allocator would
  // keep a stale one.  kdehip_clear_cache() invalidates them in every plan (peer_epoch against g_peer_epoch); a caller that
  // changes allocators without it sets KDEHIP_PEER_STORES (include/kdehip.h section 2b).
  struct PeerVerdict { const void *p; int writer; bool ok; };
  std::vector<PeerVerdict> peer_cache;
  unsigned peer_epoch = 0;
};

namespace {
// Whether a kernel on `writer` may store through `p`, an array on another device, once peer access is enabled.
// KDEHIP_PEER_STORES=0 forces the copy path (a caller whose allocator this check cannot see through), =1 skips the check.
bool peer_can_store(const void *p, int writer) {
  static const int forced = [] { const char *e = std::getenv("KDEHIP_PEER_STORES"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  if (forced >= 0) return forced == 1;
  hipMemLocation loc{};
  loc.type = hipMemLocationTypeDevice;
  loc.id = writer;
  // a stream-ordered allocation: its pool must grant the writer read-write access
  hipMemPool_t pool = nullptr;
  if (hipPointerGetAttribute(&pool, HIP_POINTER_ATTRIBUTE_MEMPOOL_HANDLE, const_cast<void *>(p)) == hipSuccess && pool) {
    hipMemAccessFlags fl = hipMemAccessFlagsProtNone;
    const bool ok = hipMemPoolGetAccess(&fl, pool, &loc) == hipSuccess && fl == hipMemAccessFlagsProtReadWrite;
    (void)hipGetLastError();
    return ok;
  }
  (void)hipGetLastError();
  // a virtual-memory mapping (hipMemMap): hipMemGetAccess knows it; plain hipMalloc memory makes the query fail
  unsigned long long vf = 0;
  if (hipMemGetAccess(&vf, &loc, const_cast<void *>(p)) == hipSuccess) {
    (void)hipGetLastError();
    return vf == static_cast<unsigned long long>(hipMemAccessFlagsProtReadWrite);
  }
  (void)hipGetLastError();
  // plain device memory of another device of this process: covered by hipDeviceEnablePeerAccess
  hipPointerAttribute_t at{};
  const bool ok = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice;
  (void)hipGetLastError();
  return ok;
}
}  // namespace

int kdehip_product_multi_create(kdehip_product_multi **out, int Ndens, const kdehip_density *trees, int ndims,
                                const uint8_t *partialDimMask, int precision, int first_device, int ngpus) {
  if (!out) return set_error(KDEHIP_ERR_ARG, "null out pointer");
  *out = nullptr;
  int rc = check_devices(first_device, ngpus);
  if (rc != KDEHIP_OK) return rc;
  PlanImage im;
  rc = build_image(im, Ndens, trees, ndims, partialDimMask, precision);
  if (rc != KDEHIP_OK) return rc;
  kdehip_product_multi *mp = new (std::nothrow) kdehip_product_multi();
  if (!mp) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
  mp->first_device = first_device;
  mp->ngpus = ngpus;
  if (ngpus - 1 > kMaxPeers) mp->peer_stores = false;  // (more "devices" than a node has: aliased tests)
  DeviceGuard guard;
  for (int g = 0; g < ngpus && rc == KDEHIP_OK; ++g) {
    kdehip_product *p = nullptr;
    rc = instantiate(im, phys(first_device + g), &p);
    if (rc != KDEHIP_OK) break;
    mp->plans.push_back(p);
    rc = guard.enter(phys(first_device + g));
    if (rc != KDEHIP_OK) break;
    hipEvent_t ev, ev2;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { rc = set_error(KDEHIP_ERR_HIP, "hipEventCreate failed"); break; }
    mp->done.push_back(ev);
    if (hipEventCreateWithFlags(&ev2, hipEventDisableTiming) != hipSuccess) { rc = set_error(KDEHIP_ERR_HIP, "hipEventCreate failed"); break; }
    mp->ready.push_back(ev2);
    // direct peer stores over xGMI where the topology allows them (else hipMemcpyPeerAsync, staged through the host)
    for (int h = 0; h < ngpus; ++h) {
      if (phys(first_device + h) == phys(first_device + g)) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, phys(first_device + g), phys(first_device + h)) == hipSuccess && can) {
        const hipError_t pe = hipDeviceEnablePeerAccess(phys(first_device + h), 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) mp->peer_stores = false;
      } else {
        mp->peer_stores = false;
      }
    }
    (void)hipGetLastError();
  }
  if (rc != KDEHIP_OK) { kdehip_product_multi_destroy(mp); return rc; }
  *out = mp;
  return KDEHIP_OK;
}

void kdehip_product_multi_destroy(kdehip_product_multi *mp) {
  if (!mp) return;
  DeviceGuard guard;
  for (size_t g = 0; g < mp->done.size(); ++g)
    if (guard.enter(phys(mp->first_device + static_cast<int>(g))) == KDEHIP_OK) {
      (void)hipDeviceSynchronize();
      (void)hipEventDestroy(mp->done[g]);
      if (g < mp->ready.size()) (void)hipEventDestroy(mp->ready[g]);
      if (g < mp->t_begin.size()) (void)hipEventDestroy(mp->t_begin[g]);
      if (g < mp->t_end.size()) (void)hipEventDestroy(mp->t_end[g]);
    }
  for (kdehip_product *p : mp->plans) kdehip_product_destroy(p);
  delete mp;
}

int kdehip_product_multi_ngpus(const kdehip_product_multi *mp) { return mp ? mp->ngpus : 0; }
kdehip_product *kdehip_product_multi_plan(kdehip_product_multi *mp, int g) {
  return (mp && g >= 0 && g < mp->ngpus) ? mp->plans[g] : nullptr;
}

int kdehip_product_multi_sample_philox(kdehip_product_multi *mp, int64_t Np, int Niter, uint64_t seed,
                                       int64_t sample_offset, int addEntropy, double *const *d_points,
                                       int64_t *const *d_indices, void *const *streams) {
  if (!mp || !d_points || !d_indices) return set_error(KDEHIP_ERR_ARG, "null argument");
  if (Np < 0) return set_error(KDEHIP_ERR_ARG, "Np must be >= 0");
  if (Np == 0) return KDEHIP_OK;
  const int G = mp->ngpus;
  const size_t D = mp->plans[0]->host.D, M = mp->plans[0]->host.M;
  for (int g = 0; g < G; ++g)
    if (!d_points[g] || !d_indices[g]) return set_error(KDEHIP_ERR_ARG, "null output pointer");
  DeviceGuard guard;
  auto stream_of = [&](int g) { return static_cast<hipStream_t>(streams ? streams[g] : nullptr); };
  // The fused gather stores into the CALLER's arrays on the other devices.  hipDeviceEnablePeerAccess maps plain hipMalloc
  // memory only: an array from a stream-ordered pool (hipMallocAsync) is reachable from a peer only if its pool grants
  // that device access, one made of virtual-memory mappings only if hipMemSetAccess did.  Looked at per call (the
  // caller may pass other arrays every time); anything that cannot be shown reachable takes the copy path.
  bool peer_stores = mp->peer_stores;
  if (const unsigned ep = g_peer_epoch.load(std::memory_order_relaxed); ep != mp->peer_epoch) {
    mp->peer_cache.clear();  // kdehip_clear_cache() has run since the verdicts were formed
    mp->peer_epoch = ep;
  }
  for (int h = 0; h < G && peer_stores && G > 1; ++h)
    for (int g = 0; g < G && peer_stores; ++g) {
      // (KDEHIP_PEER_CHECK_ALIASED=1: tests on one GPU run the look-up on aliased devices too)
      static const bool check_aliased = [] { const char *e = std::getenv("KDEHIP_PEER_CHECK_ALIASED"); return e && e[0] == '1'; }();
      if (g == h || (phys(mp->first_device + g) == phys(mp->first_device + h) && !check_aliased)) continue;
      auto can = [&](const void *p, int writer) {
        for (const auto &c : mp->peer_cache) if (c.p == p && c.writer == writer) return c.ok;
        const bool ok = peer_can_store(p, writer);
        if (mp->peer_cache.size() >= 256) mp->peer_cache.clear();  // (a caller that passes new arrays every time)
        mp->peer_cache.push_back({p, writer, ok});
        return ok;
      };
      if (!can(d_points[h], phys(mp->first_device + g)) || !can(d_indices[h], phys(mp->first_device + g)))
        peer_stores = false;
    }
  mp->last_transfers = (peer_stores || G == 1) ? 0 : 2 * (G - 1);
  // kdehip_profile_sampler: every device's sampling launch between a pair of timing events of its own (kdehip_product_multi_timing)
  const bool timed = g_profile_sampler.load(std::memory_order_relaxed) != 0;
  if (timed && mp->t_begin.empty())
    for (int g = 0; g < G; ++g) {
      int rc = guard.enter(phys(mp->first_device + g));
      if (rc != KDEHIP_OK) return rc;
      hipEvent_t a = nullptr, b = nullptr;
      KDEHIP_CHECK(hipEventCreate(&a));
      mp->t_begin.push_back(a);
      KDEHIP_CHECK(hipEventCreate(&b));
      mp->t_end.push_back(b);
    }
  mp->timed = timed && static_cast<int>(mp->t_end.size()) == G;
  mp->timed_dev.assign(static_cast<size_t>(G), 0);
  // (1) Device g is about to write into EVERY device's arrays: whatever is queued on the other devices' streams --
  // consumers of the previous product, typically -- must be over first (write after read).
  if (G > 1) {
    for (int h = 0; h < G; ++h) {
      int rc = guard.enter(phys(mp->first_device + h));
      if (rc != KDEHIP_OK) return rc;
      KDEHIP_CHECK(hipEventRecord(mp->ready[h], stream_of(h)));
    }
  }
  // (2) One launch per device.  The all-gather of [pGM | indices] is part of the kernel: its epilogue stores each
  // final point and label into the arrays of all G devices (peer-mapped pointers, xGMI) -- no copies, no extra
  // launches.  Topologies without peer access fall back to one peer copy per array and destination.
  for (int g = 0; g < G; ++g) {
    const int64_t lo = share_begin(Np, g, G), hi = share_begin(Np, g + 1, G);
    int rc = guard.enter(phys(mp->first_device + g));
    if (rc != KDEHIP_OK) return rc;
    hipStream_t st = stream_of(g);
    for (int h = 0; h < G; ++h)
      if (h != g) KDEHIP_CHECK(hipStreamWaitEvent(st, mp->ready[h], 0));
    if (hi > lo) {
      PeerOutputs peers;
      if (peer_stores)
        for (int h = 0; h < G; ++h) {
          if (h == g) continue;
          peers.points[peers.n] = d_points[h] + lo * D;
          peers.indices[peers.n] = d_indices[h] + lo * M;
          ++peers.n;
        }
      // global sample index = sample_offset + lo + s: the result does not depend on the number of devices
      rc = enqueue_philox(mp->plans[g], hi - lo, Niter, seed, sample_offset + lo, addEntropy, d_points[g] + lo * D,
                          d_indices[g] + lo * M, nullptr, st, /*private_plan=*/false, &peers, nullptr, nullptr,
                          mp->timed ? mp->t_begin[g] : nullptr, mp->timed ? mp->t_end[g] : nullptr);
      if (rc != KDEHIP_OK) return rc;
      if (mp->timed) mp->timed_dev[g] = 1;
      mp->plans[g]->async_pending.store(true);
      if (!peer_stores)
        for (int h = 0; h < G; ++h) {
          if (h == g) continue;
          KDEHIP_CHECK(hipMemcpyPeerAsync(d_points[h] + lo * D, phys(mp->first_device + h), d_points[g] + lo * D,
                                          phys(mp->first_device + g), sizeof(double) * D * (hi - lo), st));
          KDEHIP_CHECK(hipMemcpyPeerAsync(d_indices[h] + lo * M, phys(mp->first_device + h), d_indices[g] + lo * M,
                                          phys(mp->first_device + g), sizeof(int64_t) * M * (hi - lo), st));
        }
    }
    KDEHIP_CHECK(hipEventRecord(mp->done[g], st));
  }
  // (3) every device's stream continues only once all slices have arrived in its arrays
  for (int h = 0; h < G; ++h) {
    int rc = guard.enter(phys(mp->first_device + h));
    if (rc != KDEHIP_OK) return rc;
    hipStream_t st = stream_of(h);
    for (int g = 0; g < G; ++g)
      if (g != h) KDEHIP_CHECK(hipStreamWaitEvent(st, mp->done[g], 0));
  }
  return KDEHIP_OK;
}

// Diagnostic (kdehip_profile_sampler on): of the LAST product, per device, the duration of its sampling launch and the
// host time at which its slice had arrived everywhere (`done` event seen complete), relative to the first device to get
// there -- what tells a straggling device or link from a slow kernel when the N > 1 path is first run on hardware.
int kdehip_product_multi_timing(kdehip_product_multi *mp, double *kernel_ms, double *done_ms) {
  if (!mp || !kernel_ms || !done_ms) return set_error(KDEHIP_ERR_ARG, "null argument");
  if (!mp->timed) return set_error(KDEHIP_ERR_ARG, "kdehip_product_multi_timing: the last product was not timed (kdehip_profile_sampler(1) first)");
  const int G = mp->ngpus;
  DeviceGuard guard;
  std::vector<char> seen(G, 0);
  std::vector<double> at(G, 0.0);
  const auto t0 = std::chrono::steady_clock::now();
  for (int left = G; left > 0;)
    for (int g = 0; g < G; ++g) {
      if (seen[g]) continue;
      const int rc = guard.enter(phys(mp->first_device + g));
      if (rc != KDEHIP_OK) return rc;
      const hipError_t q = hipEventQuery(mp->done[g]);
      if (q == hipSuccess) {
        at[g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        seen[g] = 1;
        --left;
      } else if (q != hipErrorNotReady) {
        return set_error(KDEHIP_ERR_HIP, std::string("kdehip_product_multi_timing: ") + hipGetErrorString(q));
      }
    }
  (void)hipGetLastError();
  double first = at[0];
  for (int g = 1; g < G; ++g) if (at[g] < first) first = at[g];
  for (int g = 0; g < G; ++g) {
    const int rc = guard.enter(phys(mp->first_device + g));
    if (rc != KDEHIP_OK) return rc;
    float ms = 0.0f;  // (a device whose slice was empty launched nothing and recorded no events: 0)
    if (mp->timed_dev[g]) KDEHIP_CHECK(hipEventElapsedTime(&ms, mp->t_begin[g], mp->t_end[g]));
    kernel_ms[g] = ms;
    done_ms[g] = at[g] - first;
  }
  return KDEHIP_OK;
}

int kdehip_product_multi_transfers_per_product(const kdehip_product_multi *mp) {
  if (!mp) return -1;
  if (mp->last_transfers >= 0) return mp->last_transfers;  // what the last product did (its arrays were looked at)
  return mp->peer_stores ? 0 : 2 * (mp->ngpus - 1);
}

// ---- host twin of the device RNG ------------------------------------------------------------------
// Element i of a sample's uniform slice is consumed by select call c = i + 1 (the reference reads
// randU[ruptr] BEFORE incrementing a cursor that starts at 0; src/MSGibbs01.jl:337,348).
void kdehip_philox_fill_uniform(uint64_t seed, int64_t sample_begin, int64_t nsamples, int64_t K,
                                double *out_u) {
  for (int64_t s = 0; s < nsamples; ++s)
    for (int64_t i = 0; i < K; ++i)
      out_u[s * K + i] = philox_uniform(seed, static_cast<uint64_t>(sample_begin + s),
                                        static_cast<uint32_t>(i + 1));
}
void kdehip_philox_fill_normal(uint64_t seed, int64_t sample_begin, int64_t nsamples, int64_t R,
                               double *out_n) {
  for (int64_t s = 0; s < nsamples; ++s)
    for (int64_t r = 0; r < R; ++r)
      out_n[s * R + r] = philox_normal(seed, static_cast<uint64_t>(sample_begin + s),
                                       static_cast<uint32_t>(r));
}

}  // extern "C"
