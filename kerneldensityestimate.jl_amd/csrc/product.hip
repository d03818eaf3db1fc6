// product.hip -- C-ABI entry points of libkdehip.so for the Gibbs product: resident plans, runs,
// the gibbs1 drop-in and the host twin of the device RNG.  (Kernels: gibbs_kernel.hip; host
// re-layout: pack_levels.cpp; density construction: balltree.cpp.)
#include <hip/hip_runtime.h>

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
  void *d_data = nullptr;
  int32_t *d_perm = nullptr;
  LevelDesc *d_levels = nullptr;
  void *d_tables = nullptr;
  TabDesc *d_tabdesc = nullptr;
  bool tables_built = false;
  std::mutex tables_mutex;  // concurrent first runs on one plan build the tables once
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

int use_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(KDEHIP_ERR_NO_DEVICE,
                     "no HIP device available (libkdehip has no CPU fallback by design)");
  if (device < 0 || device >= n) return set_error(KDEHIP_ERR_ARG, "device ordinal out of range");
  e = hipSetDevice(device);
  if (e != hipSuccess)
    return set_error(KDEHIP_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return KDEHIP_OK;
}

// RAII device buffer for the host-pointer convenience paths
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

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
  int rc = pack_levels(Ndens, trees, ndims, partialDimMask, precision, p->host);
  if (rc == KDEHIP_OK) rc = use_device(device);
  if (rc != KDEHIP_OK) { delete p; return rc; }
  p->device = device;
  p->precision = precision;
  p->fast = p->host.fast;
  p->mode = !p->host.fast ? kModeGeneric : (p->host.all_active ? kModeFast : kModeFastMasked);

  const size_t nelem = p->host.data.size();
  const size_t esz = (precision == 64) ? sizeof(double) : sizeof(float);
  const size_t nperm = p->host.perm.size();
  const size_t nlev = p->host.levels.size();
  auto fail = [&](hipError_t e, const char *what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    kdehip_product_destroy(p);
    return set_error(KDEHIP_ERR_HIP, m);
  };
  hipError_t e;
  if ((e = hipMalloc(&p->d_data, nelem * esz)) != hipSuccess) return fail(e, "hipMalloc(data)");
  if ((e = hipMalloc(reinterpret_cast<void **>(&p->d_perm), nperm * sizeof(int32_t))) != hipSuccess)
    return fail(e, "hipMalloc(perm)");
  if ((e = hipMalloc(reinterpret_cast<void **>(&p->d_levels), nlev * sizeof(LevelDesc))) != hipSuccess)
    return fail(e, "hipMalloc(levels)");
  if (precision == 64) {
    e = hipMemcpy(p->d_data, p->host.data.data(), nelem * esz, hipMemcpyHostToDevice);
  } else {
    std::vector<float> f(nelem);
    for (size_t i = 0; i < nelem; ++i) f[i] = static_cast<float>(p->host.data[i]);
    e = hipMemcpy(p->d_data, f.data(), nelem * esz, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) return fail(e, "hipMemcpy(data)");
  if ((e = hipMemcpy(p->d_perm, p->host.perm.data(), nperm * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess)
    return fail(e, "hipMemcpy(perm)");
  if ((e = hipMemcpy(p->d_levels, p->host.levels.data(), nlev * sizeof(LevelDesc), hipMemcpyHostToDevice)) != hipSuccess)
    return fail(e, "hipMemcpy(levels)");
  const size_t ntab = p->host.tabdesc.size();
  const size_t tab_bytes = static_cast<size_t>(p->host.tab_entries) * esz;
  if ((e = hipMalloc(reinterpret_cast<void **>(&p->d_tabdesc), ntab * sizeof(TabDesc))) != hipSuccess)
    return fail(e, "hipMalloc(tabdesc)");
  if ((e = hipMemcpy(p->d_tabdesc, p->host.tabdesc.data(), ntab * sizeof(TabDesc), hipMemcpyHostToDevice)) != hipSuccess)
    return fail(e, "hipMemcpy(tabdesc)");
  if (tab_bytes && (e = hipMalloc(&p->d_tables, tab_bytes)) != hipSuccess) return fail(e, "hipMalloc(tables)");
  p->packed_bytes = static_cast<int64_t>(nelem * esz + nperm * sizeof(int32_t) + nlev * sizeof(LevelDesc) +
                                         ntab * sizeof(TabDesc) + tab_bytes);
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
  if (hipSetDevice(plan->device) == hipSuccess) {
    if (plan->d_data) (void)hipFree(plan->d_data);
    if (plan->d_perm) (void)hipFree(plan->d_perm);
    if (plan->d_levels) (void)hipFree(plan->d_levels);
    if (plan->d_tables) (void)hipFree(plan->d_tables);
    if (plan->d_tabdesc) (void)hipFree(plan->d_tabdesc);
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

int kdehip_product_set_variant(kdehip_product *plan, int variant) {
  if (!plan) return set_error(KDEHIP_ERR_ARG, "null plan");
  plan->variant = variant;
  return KDEHIP_OK;
}

int kdehip_product_sample_streams(kdehip_product *plan, int64_t Np, int Niter, const double *d_randU,
                                  int64_t nU, const double *d_randN, int64_t nN, int addEntropy,
                                  double *d_points, int64_t *d_indices, int32_t *d_labels,
                                  void *stream) {
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
  KDEHIP_CHECK(hipSetDevice(plan->device));
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

int kdehip_product_sample_philox(kdehip_product *plan, int64_t Np, int Niter, uint64_t seed,
                                 int64_t sample_offset, int addEntropy, double *d_points,
                                 int64_t *d_indices, int32_t *d_labels, void *stream) {
  int rc = check_run(plan, Np, Niter, d_points, d_indices);
  if (rc != KDEHIP_OK) return rc;
  if (Np == 0) return KDEHIP_OK;
  if (sample_offset < 0) return set_error(KDEHIP_ERR_ARG, "sample_offset must be >= 0");
  KDEHIP_CHECK(hipSetDevice(plan->device));
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

int kdehip_product_sample_philox_host(kdehip_product *plan, int64_t Np, int Niter, uint64_t seed,
                                      int64_t sample_offset, int addEntropy, double *points,
                                      int64_t *indices, int32_t *labels) {
  int rc = check_run(plan, Np, Niter, points, indices);
  if (rc != KDEHIP_OK) return rc;
  if (Np == 0) return KDEHIP_OK;
  KDEHIP_CHECK(hipSetDevice(plan->device));
  const size_t D = plan->host.D, M = plan->host.M, L = plan->host.L;
  DevBuf dp, di, dl;
  KDEHIP_CHECK(dp.alloc(sizeof(double) * D * Np));
  KDEHIP_CHECK(di.alloc(sizeof(int64_t) * M * Np));
  if (labels) KDEHIP_CHECK(dl.alloc(sizeof(int32_t) * M * L * Np));
  rc = kdehip_product_sample_philox(plan, Np, Niter, seed, sample_offset, addEntropy,
                                    static_cast<double *>(dp.p), static_cast<int64_t *>(di.p),
                                    labels ? static_cast<int32_t *>(dl.p) : nullptr, nullptr);
  if (rc != KDEHIP_OK) return rc;
  KDEHIP_CHECK(hipDeviceSynchronize());
  KDEHIP_CHECK(hipMemcpy(points, dp.p, sizeof(double) * D * Np, hipMemcpyDeviceToHost));
  KDEHIP_CHECK(hipMemcpy(indices, di.p, sizeof(int64_t) * M * Np, hipMemcpyDeviceToHost));
  if (labels) KDEHIP_CHECK(hipMemcpy(labels, dl.p, sizeof(int32_t) * M * L * Np, hipMemcpyDeviceToHost));
  return KDEHIP_OK;
}

int kdehip_gibbs1(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                  int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                  int addEntropy, int ndims, const uint8_t *partialDimMask, int device) {
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
  DevBuf du, dn, dp, di;
  KDEHIP_CHECK(du.alloc(sizeof(double) * useU));
  KDEHIP_CHECK(dn.alloc(sizeof(double) * useN));
  KDEHIP_CHECK(dp.alloc(sizeof(double) * D * Np));
  KDEHIP_CHECK(di.alloc(sizeof(int64_t) * M * Np));
  KDEHIP_CHECK(hipMemcpy(du.p, randU, sizeof(double) * useU, hipMemcpyHostToDevice));
  KDEHIP_CHECK(hipMemcpy(dn.p, randN, sizeof(double) * useN, hipMemcpyHostToDevice));
  rc = kdehip_product_sample_streams(plan, Np, Niter, static_cast<const double *>(du.p), useU,
                                     static_cast<const double *>(dn.p), useN, addEntropy,
                                     static_cast<double *>(dp.p), static_cast<int64_t *>(di.p),
                                     nullptr, nullptr);
  if (rc != KDEHIP_OK) return rc;
  KDEHIP_CHECK(hipDeviceSynchronize());
  KDEHIP_CHECK(hipMemcpy(pts, dp.p, sizeof(double) * D * Np, hipMemcpyDeviceToHost));
  KDEHIP_CHECK(hipMemcpy(ind, di.p, sizeof(int64_t) * M * Np, hipMemcpyDeviceToHost));
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
