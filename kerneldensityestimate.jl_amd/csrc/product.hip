// product.hip -- C-ABI entry points of libkdehip.so for the Gibbs product: resident plans, runs,
// the gibbs1 drop-in and the host twin of the device RNG.  (Kernels: gibbs_kernel.hip; host
// re-layout: pack_levels.cpp; density construction: balltree.cpp.)
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "kdehip_internal.hpp"
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
  std::mutex tables_mutex;  // concurrent first runs on one plan build the tables once
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
int maybe_build_tables(kdehip_product *plan, int64_t Np, RunArgs &a, void *stream) {
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
    // one-time: runs on other streams must not overtake the build
    KDEHIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    plan->tables_built = true;
  }
  a.use_tables = 1;
  return KDEHIP_OK;
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

// The two run forms, enqueue only (no bookkeeping of who waits for the work: see the callers).
int enqueue_streams(kdehip_product *plan, int64_t Np, int Niter, const double *d_randU, int64_t nU,
                    const double *d_randN, int64_t nN, int addEntropy, double *d_points, int64_t *d_indices,
                    int32_t *d_labels, void *stream) {
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
  rc = maybe_build_tables(plan, Np, a, stream);
  if (rc != KDEHIP_OK) return rc;
  return launch_gibbs(plan->precision, plan->mode, plan->dev, a, stream);
}

int enqueue_philox(kdehip_product *plan, int64_t Np, int Niter, uint64_t seed, int64_t sample_offset,
                   int addEntropy, double *d_points, int64_t *d_indices, int32_t *d_labels, void *stream) {
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
  rc = maybe_build_tables(plan, Np, a, stream);
  if (rc != KDEHIP_OK) return rc;
  return launch_gibbs(plan->precision, plan->mode, plan->dev, a, stream);
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
  if (precision != 64 && precision != 32) return set_error(KDEHIP_ERR_ARG, "precision must be 64 or 32");
  kdehip_product *p = new (std::nothrow) kdehip_product();
  if (!p) return set_error(KDEHIP_ERR_ALLOC, "out of host memory");
  DeviceGuard guard;
  int rc = pack_levels(Ndens, trees, ndims, partialDimMask, precision, p->host);
  if (rc == KDEHIP_OK) rc = guard.enter(device);
  if (rc != KDEHIP_OK) { delete p; return rc; }
  p->device = device;
  p->precision = precision;
  p->fast = p->host.fast;
  p->mode = !p->host.fast ? kModeGeneric : (p->host.all_active ? kModeFast : kModeFastMasked);

  // One device allocation and one upload per plan (hipMalloc / hipFree cost tens of microseconds each and
  // dominate a one-shot small product): [levels | table descriptors | permutation | tiles | tables], the
  // table region is filled later by the table-build launch.
  const size_t nelem = p->host.data.size();
  const size_t esz = (precision == 64) ? sizeof(double) : sizeof(float);
  const size_t nperm = p->host.perm.size();
  const size_t nlev = p->host.levels.size();
  const size_t ntab = p->host.tabdesc.size();
  const size_t tab_bytes = static_cast<size_t>(p->host.tab_entries) * esz;
  auto align = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  const size_t off_lev = 256;                                  // the fallback counter sits in the 8 bytes before it
  const size_t off_count = off_lev - sizeof(unsigned long long); // (zeroed with the rest of the upload)
  const size_t off_tab = align(off_lev + nlev * sizeof(LevelDesc));
  const size_t off_perm = align(off_tab + ntab * sizeof(TabDesc));
  const size_t off_data = align(off_perm + nperm * sizeof(int32_t));
  const size_t off_tables = align(off_data + nelem * esz);
  const size_t total = off_tables + tab_bytes;
  std::vector<unsigned char> blob(off_tables, 0);
  std::memcpy(blob.data() + off_lev, p->host.levels.data(), nlev * sizeof(LevelDesc));
  std::memcpy(blob.data() + off_tab, p->host.tabdesc.data(), ntab * sizeof(TabDesc));
  std::memcpy(blob.data() + off_perm, p->host.perm.data(), nperm * sizeof(int32_t));
  if (precision == 64) {
    std::memcpy(blob.data() + off_data, p->host.data.data(), nelem * esz);
  } else {
    float *f = reinterpret_cast<float *>(blob.data() + off_data);
    for (size_t i = 0; i < nelem; ++i) f[i] = static_cast<float>(p->host.data[i]);
  }
  hipError_t e = cached_malloc(&p->d_blob, total);
  if (e == hipSuccess) p->blob_bytes = total;
  if (e == hipSuccess) e = hipMemcpy(p->d_blob, blob.data(), off_tables, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    const std::string m = std::string("plan upload: ") + hipGetErrorString(e);
    kdehip_product_destroy(p);
    return set_error(KDEHIP_ERR_HIP, m);
  }
  unsigned char *base = static_cast<unsigned char *>(p->d_blob);
  p->d_levels = reinterpret_cast<LevelDesc *>(base + off_lev);
  p->d_tabdesc = reinterpret_cast<TabDesc *>(base + off_tab);
  p->d_perm = reinterpret_cast<int32_t *>(base + off_perm);
  p->d_data = base + off_data;
  p->d_fallbacks = reinterpret_cast<unsigned long long *>(base + off_count);
  p->d_tables = tab_bytes ? base + off_tables : nullptr;
  p->packed_bytes = static_cast<int64_t>(total);
  std::vector<double>().swap(p->host.data);
  std::vector<int32_t>().swap(p->host.perm);

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
  *out = p;
  return KDEHIP_OK;
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

int kdehip_product_set_variant(kdehip_product *plan, int variant) {
  if (!plan) return set_error(KDEHIP_ERR_ARG, "null plan");
  plan->variant = variant;
  return KDEHIP_OK;
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
  rc = enqueue_philox(plan, Np, Niter, seed, sample_offset, addEntropy, dp, di, dl, nullptr);
  if (rc != KDEHIP_OK) return rc;
  KDEHIP_CHECK(hipMemcpy(points, dp, sizeof(double) * D * Np, hipMemcpyDeviceToHost));
  KDEHIP_CHECK(hipMemcpy(indices, di, sizeof(int64_t) * M * Np, hipMemcpyDeviceToHost));
  if (labels) KDEHIP_CHECK(hipMemcpy(labels, dl, sizeof(int32_t) * M * L * Np, hipMemcpyDeviceToHost));
  return KDEHIP_OK;
}

int kdehip_gibbs1(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                  int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                  int addEntropy, int ndims, const uint8_t *partialDimMask, int device) {
  return kdehip_gibbs1_trace(Ndens, trees, Np, Niter, pts, ind, randU, nU, randN, nN, addEntropy, ndims,
                             partialDimMask, device, nullptr);
}

int kdehip_gibbs1_trace(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                        int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                        int addEntropy, int ndims, const uint8_t *partialDimMask, int device,
                        int32_t *labels) {
  kdehip_product *plan = nullptr;
  int rc = kdehip_product_create(&plan, Ndens, trees, ndims, partialDimMask, 64, device);
  if (rc != KDEHIP_OK) return rc;
  struct Guard { kdehip_product *p; ~Guard() { kdehip_product_destroy(p); } } guard{plan};
  rc = check_run(plan, Np, Niter, pts, ind);
  if (rc != KDEHIP_OK) return rc;
  if (Np == 0) return KDEHIP_OK;
  const int64_t K = kdehip_product_randu_per_sample(plan, Niter);
  const int64_t R = kdehip_product_randn_per_sample(plan);
  if (!randU || nU < Np * K - 1)
    return set_error(KDEHIP_ERR_RAND_SHORT, "randU shorter than Np*K-1 values (Julia: BoundsError)");
  if (!randN || nN < Np * R)
    return set_error(KDEHIP_ERR_RAND_SHORT, "randN shorter than Np*R values (Julia: BoundsError)");
  const size_t D = ndims, M = Ndens;
  const int64_t useU = (nU < Np * K) ? nU : Np * K, useN = Np * R;
  DeviceGuard dguard;
  rc = dguard.enter(device);
  if (rc != KDEHIP_OK) return rc;
  std::lock_guard<std::mutex> lock(plan->work_mutex);
  const size_t off_n = align256(sizeof(double) * useU);
  const size_t off_p = align256(off_n + sizeof(double) * useN);
  const size_t off_i = align256(off_p + sizeof(double) * D * Np);
  const size_t off_l = align256(off_i + sizeof(int64_t) * M * Np);
  // the reference records a label only inside sampleIndex (:426): nothing with Niter = 0
  const bool trace = labels != nullptr && Niter > 0;
  const size_t lab_bytes = trace ? sizeof(int32_t) * M * static_cast<size_t>(plan->host.L) * Np : 0;
  rc = reserve_work(plan, off_l + lab_bytes);
  if (rc != KDEHIP_OK) return rc;
  unsigned char *w = static_cast<unsigned char *>(plan->d_work);
  double *du = reinterpret_cast<double *>(w), *dn = reinterpret_cast<double *>(w + off_n);
  double *dp = reinterpret_cast<double *>(w + off_p);
  int64_t *di = reinterpret_cast<int64_t *>(w + off_i);
  int32_t *dl = trace ? reinterpret_cast<int32_t *>(w + off_l) : nullptr;
  KDEHIP_CHECK(hipMemcpy(du, randU, sizeof(double) * useU, hipMemcpyHostToDevice));
  KDEHIP_CHECK(hipMemcpy(dn, randN, sizeof(double) * useN, hipMemcpyHostToDevice));
  rc = enqueue_streams(plan, Np, Niter, du, useU, dn, useN, addEntropy, dp, di, dl, nullptr);
  if (rc != KDEHIP_OK) return rc;
  KDEHIP_CHECK(hipMemcpy(pts, dp, sizeof(double) * D * Np, hipMemcpyDeviceToHost));
  KDEHIP_CHECK(hipMemcpy(ind, di, sizeof(int64_t) * M * Np, hipMemcpyDeviceToHost));
  if (trace) KDEHIP_CHECK(hipMemcpy(labels, dl, lab_bytes, hipMemcpyDeviceToHost));
  return KDEHIP_OK;
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
