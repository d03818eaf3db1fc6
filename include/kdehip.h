/*
 * kdehip.h -- C ABI of libkdehip.so: the MI355X (gfx950) implementation of the multiscale-Gibbs
 * KDE product hot path of JuliaRobotics/KernelDensityEstimate.jl.
 *
 * The reference is pure Julia and has no FFI layer; the seam this library replaces is the Julia
 * method `gibbs1` (reference src/MSGibbs01.jl:527-629), called only from `prodAppxMSGibbsS`
 * (src/MSGibbs01.jl:645-703).  A Julia caller binds these entry points with `ccall`
 * (see INTEGRATION.md); the Python mirror in kerneldensityestimate.jl_amd/ binds them with ctypes.
 *
 * Conventions: plain pointers and sizes only.  All matrices are column-major as in Julia
 * (points: ndims x Np, indices: Ndens x Np).  Node ids inside a density are the reference's
 * 1-based ids with NO_CHILD = -1 (src/BallTree01.jl:5).  Every function returning `int` returns
 * KDEHIP_OK (0) or a negative error code; kdehip_last_error() gives the message (thread-local).
 * The library never keeps a caller's host pointer after a call returns.
 * There is no CPU fallback: without a usable HIP device every compute entry point fails with
 * KDEHIP_ERR_NO_DEVICE.
 */
#ifndef KDEHIP_H
#define KDEHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KDEHIP_VERSION 600 /* 0.6.0 */

enum {
  KDEHIP_OK = 0,
  KDEHIP_ERR_ARG = -1,         /* invalid argument (message says which)                            */
  KDEHIP_ERR_DIM_MISMATCH = -2,/* "kdes must have same dimension"  (src/MSGibbs01.jl:720-722)     */
  KDEHIP_ERR_RAND_SHORT = -3,  /* randU / randN shorter than the run consumes (Julia BoundsError) */
  KDEHIP_ERR_NO_DEVICE = -4,   /* no HIP device / HIP runtime failure at init                      */
  KDEHIP_ERR_HIP = -5,         /* a HIP call failed                                                */
  KDEHIP_ERR_ALLOC = -6,
  KDEHIP_ERR_UNSUPPORTED = -7  /* e.g. ndims or Ndens above the compiled limits                    */
};

#define KDEHIP_MAX_DIMS 8    /* kernels are instantiated for ndims = 1..8                          */
#define KDEHIP_MAX_DENS 16   /* densities per product                                              */

/* The six flat arrays of a BallTreeDensity that gibbs1 reads through its accessors
 * (reference src/BallTreeDensity01.jl:11-24,86-93; src/BallTree01.jl:10-28,78-94).
 * Lengths: means/bandwidth = ndim*2*npts (node i, dim k at (i-1)*ndim + k-1; bandwidth holds
 * VARIANCES, src/KDE01.jl:45); the others = 2*npts. */
typedef struct kdehip_density {
  int64_t npts;               /* bt.num_points */
  int64_t ndim;               /* bt.dims       */
  const double *means;
  const double *bandwidth;
  const double *weights;      /* bt.weights      */
  const int64_t *left_child;  /* bt.left_child   */
  const int64_t *right_child; /* bt.right_child  */
  const int64_t *permutation; /* bt.permutation  */
} kdehip_density;

/* ---- library ---------------------------------------------------------------------------------- */
int kdehip_version(void);
const char *kdehip_last_error(void);
int kdehip_device_count(void); /* 0 when no device is usable */
/* The library keeps freed device / pinned-host blocks (up to 1 GiB per device) for the next call instead of
 * returning them to the driver: a one-shot product would otherwise spend most of its time in hipMalloc/hipFree.
 * This hands everything back (no reference counterpart: Julia's GC owns the reference's scratch). */
void kdehip_clear_cache(void);

/* ---- (1) drop-in for gibbs1 (reference src/MSGibbs01.jl:527-537) ------------------------------
 * Host buffers in, host buffers out, blocking.  The blocking entry points of this header are thread-safe and enqueue
 * their work on the calling thread's own stream (hipStreamPerThread): calls of concurrent host threads overlap on the
 * device instead of queueing behind each other on the null stream.  Arguments in the reference's order:
 *   Ndens, trees, Np, Niter, pts (out, ndims*Np), ind (out, Ndens*Np, = permutation+1, :615),
 *   randU (nU values), randN (nN values), then the keywords addEntropy, ndims, partialDimMask
 *   (Ndens*ndims bytes, density-major, 1 = active; NULL = all active).  `device` = HIP ordinal.
 * RNG consumption is the reference's: 0-based sample s, select call c reads randU[s*K + c - 1],
 * normal q*ndims+d of sample s is randN[s*R + q*ndims + d] (K, R from kdehip_product_info).
 * Only the Euclidean manifold operators (the reference defaults addop=+, diffop=-, getEuclidMu,
 * getEuclidLambda) exist behind this ABI. */
int kdehip_gibbs1(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                  int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                  int addEntropy, int ndims, const uint8_t *partialDimMask, int device);

/* gibbs1 called with glbs.recordChoosen = true (reference src/MSGibbs01.jl:29-31, :109-112, :426, :575-583):
 * additionally returns the label trace labels[(s*Ndens + j)*nlevels + (l-1)] = bt.permutation[ind_j] as it
 * stands after the last sampleIndex of level l = 1..nlevels (what labelsChoosen[s+1][j+1][l] ends up holding;
 * with Niter = 0 the reference records nothing and the buffer is left as it was).  nlevels =
 * floor(log(max_j Npts_j)/log 2 + 1) (:568).  labels == NULL makes this kdehip_gibbs1. */
int kdehip_gibbs1_trace(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                        int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                        int addEntropy, int ndims, const uint8_t *partialDimMask, int device,
                        int32_t *labels);

/* The same on `ngpus` GPUs of one node (devices device .. device+ngpus-1), one process: the chains are split into
 * contiguous ranges (sample s of the call depends only on the densities and on its own slices randU[s*K ..],
 * randN[s*R ..], so the result is identical for every ngpus), each device gets the packed densities and its slice
 * of the streams, and returns its slice of pts / ind (/ labels, as kdehip_gibbs1_trace; may be NULL) to the host.
 * ngpus = 1 is kdehip_gibbs1_trace. */
int kdehip_gibbs1_multi(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts,
                        int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                        int addEntropy, int ndims, const uint8_t *partialDimMask, int device, int ngpus,
                        int32_t *labels);

/* ---- manifolds: the operator tuples addop / diffop / getMu / getLambda as an ENUM ------------------------------------
 * The reference takes its on-manifold operators as per-dimension Julia FUNCTIONS (src/MSGibbs01.jl:650-653, broadcast at
 * :672-675) and applies them at three hook points: diffop inside every kernel evaluation (:290), getLambda / getMu in the
 * Gaussian product (:183-184, applied :210-213), addop where a sample is composed with its noise (:456).  Functions
 * cannot cross a C ABI, and the reference itself defines only the Euclidean set (its callers bring the others): this
 * entry takes, per dimension, KDEHIP_MANIFOLD_EUCLIDEAN (the reference's defaults) or KDEHIP_MANIFOLD_CIRCULAR with THIS
 * library's stated semantic -- nothing in the reference pins it:
 *     wrap(t)        = t - 2 pi floor((t + pi) / (2 pi))            in [-pi, pi)
 *     diffop(a, b)   = wrap(a - b)         addop(a, b) = wrap(a + b)         getLambda(lambdas) = sum(lambdas)
 *     getMu(mus, lambdas, scale) = addop(ref, scale * sum_j lambdas_j diffop(mus_j, ref)),  ref = mus of the first j with
 *                      lambdas_j > 0   (the information-weighted mean in the tangent space at ref; the Euclidean formula
 *                      whenever no difference wraps)
 * The densities are the caller's arrays as built (the reference's tree construction has hooks of its own,
 * src/BallTree01.jl:315, which are the caller's business).  Runs the general sampler's generic arithmetic (the
 * reference's divide + log accumulation); caller streams in the reference's order, as kdehip_gibbs1_trace.  manifold == NULL
 * = all Euclidean = kdehip_gibbs1_trace.  oracle/kde_oracle.c okde_gibbs1_manifold is the same enum on the CPU; the Julia
 * shim maps NOTHING to it automatically (a caller's circular functions need not be these). */
#define KDEHIP_MANIFOLD_EUCLIDEAN 0
#define KDEHIP_MANIFOLD_CIRCULAR 1
int kdehip_gibbs1_manifold(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
                           const double *randU, int64_t nU, const double *randN, int64_t nN, int addEntropy, int ndims,
                           const uint8_t *partialDimMask, const uint8_t *manifold /* ndims bytes or NULL */, int device,
                           int32_t *labels /* optional */);

/* prodAppxMSGibbsS when the caller passes no randU / randN (reference src/MSGibbs01.jl:645-703; its
 * `rand(...)` / `randn(...)` defaults, :661-662, are replaced by the on-device Philox4x32-10 stream keyed by
 * (seed, sample index, draw index): kdehip_philox_fill_* reproduce the numbers).  One-shot: pack, upload, run,
 * copy back.  precision 64 or 32; ngpus and labels as in kdehip_gibbs1_multi. */
int kdehip_prod_philox(int Ndens, const kdehip_density *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
                       uint64_t seed, int addEntropy, int ndims, const uint8_t *partialDimMask, int precision,
                       int device, int ngpus, int32_t *labels);

/* ---- (2) resident product plan ----------------------------------------------------------------
 * The densities are re-laid-out once (per-level tiles, "pack_levels") and kept in HBM so repeated
 * products -- and bench.py's timed region -- start with inputs resident on the device.  The first run
 * of a plan additionally fills its conditional tables (tens of microseconds) and synchronises its
 * stream once; later runs only enqueue work.  A plan may be used from several threads/streams. */
typedef struct kdehip_product kdehip_product;

typedef struct kdehip_product_info_t {
  int32_t ndens, ndims, nlevels;     /* nlevels = floor(log(max Npts)/log 2 + 1), :568            */
  int32_t precision;                 /* 64 or 32                                                   */
  int64_t nodes_per_sweep;           /* sum_j sum_{l=1..L} n_{j,l}: kernel evals of one sweep      */
  int64_t bytes_per_eval;            /* (2*ndims+1)*sizeof(T)                           */
  int64_t packed_bytes;              /* device bytes held by the plan                              */
  int32_t fast_math_path;            /* 1: product/rsqrt forms; 0: the reference's divide+log form  */
  int32_t device;
} kdehip_product_info_t;

/* precision: 64 (reference arithmetic) or 32.  mask as in kdehip_gibbs1. */
int kdehip_product_create(kdehip_product **out, int Ndens, const kdehip_density *trees, int ndims,
                          const uint8_t *partialDimMask, int precision, int device);
/* (waits for the device first if runs were enqueued through the device-pointer entry points below) */
void kdehip_product_destroy(kdehip_product *plan);
int kdehip_product_info(const kdehip_product *plan, kdehip_product_info_t *info);
/* Per-sample RNG consumption for a given Niter: K uniforms (first M slots never read), R normals. */
int64_t kdehip_product_randu_per_sample(const kdehip_product *plan, int Niter);
int64_t kdehip_product_randn_per_sample(const kdehip_product *plan);

/* Run Np chains.  All pointers are DEVICE pointers on the plan's device; `stream` is a hipStream_t
 * (NULL = default stream); the call only enqueues work.  d_points: double[ndims*Np];
 * d_indices: int64[Ndens*Np]; d_labels: optional int32[Np*Ndens*nlevels] (permutation of the label
 * kept at the end of every level, the recordChoosen trace of src/MSGibbs01.jl:109-112) or NULL. */
int kdehip_product_sample_streams(kdehip_product *plan, int64_t Np, int Niter,
                                  const double *d_randU, int64_t nU, const double *d_randN,
                                  int64_t nN, int addEntropy, double *d_points, int64_t *d_indices,
                                  int32_t *d_labels, void *stream);
/* Same, random numbers from the on-device Philox4x32-10 stream keyed by (seed, global sample
 * index = sample_offset + s, draw index): results do not depend on how samples are split over
 * calls or GPUs. */
int kdehip_product_sample_philox(kdehip_product *plan, int64_t Np, int Niter, uint64_t seed,
                                 int64_t sample_offset, int addEntropy, double *d_points,
                                 int64_t *d_indices, int32_t *d_labels, void *stream);
/* Host-buffer convenience wrapper around the Philox run (allocates, runs, copies back, blocking). */
int kdehip_product_sample_philox_host(kdehip_product *plan, int64_t Np, int Niter, uint64_t seed,
                                      int64_t sample_offset, int addEntropy, double *points,
                                      int64_t *indices, int32_t *labels);
/* Diagnostic: how many label draws of this plan's runs so far took the reference's underflow branch
 * (`pT < 1e-99` -> uniform draw over the frontier, src/MSGibbs01.jl:311-315).  Waits for the device.
 * Negative = error code. */
int64_t kdehip_product_fallback_count(kdehip_product *plan);
/* Diagnostic: fp32 screening of the deep levels (fp64 plans of 2..4 or 8 densities with every dimension active): on a
 * level whose fp64 tiles are streamed or chunked through LDS, the plan also holds the level's tiles in fp32 -- resident in
 * LDS together, streamed one per step, or in chunks, whichever fits; up to 128 entries per lane --; a draw step evaluates
 * the frontier in packed fp32 with a rigorous bound on the error of its cumulative
 * sums and keeps the fp32 decision only when the uniform draw is farther than that bound from every boundary it could
 * cross -- otherwise the step is repeated in fp64.  Labels and points are those of the fp64 arithmetic bit for bit
 * (csrc/screen_device.hpp).  levels: how many levels of the plan are screened; steps / repeats: label draws
 * taken on screened levels so far, and how many of them were repeated in fp64.  Waits for the device.  Any pointer may be
 * NULL. */
int kdehip_product_screen_stats(kdehip_product *plan, int32_t *levels, int64_t *steps, int64_t *repeats);
/* Scheduling knob for experiments/benchmarks; results never depend on it.  0 = library default,
 * 1 = read every tile from global memory (no LDS staging), 4 = no conditional tables, 5 = no fp32 screening,
 * 2 / 8 / 16 = 4 / 8 / 16 chains per workgroup (one wavefront per chain).  Default: chosen from the number of chains. */
int kdehip_product_set_variant(kdehip_product *plan, int variant);
/* Diagnostic: the launch geometry a run of Np chains of this plan gets under its current variant -- wavefronts per
 * workgroup and wavefronts per chain (always 1 since round 5: a chain is one wavefront). */
int kdehip_product_launch_geometry(const kdehip_product *plan, int64_t Np, int32_t *waves_per_workgroup,
                                   int32_t *waves_per_chain);
/* Diagnostic: the sampling kernel a run of Np chains of this plan launches under its current variant --
 * "gibbs_lean_kernel" (chain state in registers: products of 2..4 densities, fp64 products of 8, every dimension
 * active) or "gibbs_product_kernel" (any density count, masks, the reference's divide + log arithmetic).  A static
 * string. */
const char *kdehip_product_kernel_name(const kdehip_product *plan, int64_t Np);

/* ---- (2b) resident plans on several GPUs of one node (one process) --------------------------------
 * One plan per device (the packed densities are replicated), chains in contiguous ranges, Philox counters keyed by
 * the GLOBAL sample index (results identical for every number of devices), and ONE all-gather of [pGM | indices]
 * fused into the sampling kernel: its epilogue stores every final point and label straight into the arrays of ALL
 * devices (peer-mapped pointers, the stores travel over xGMI; no copy engine, no extra launch).  After the call's
 * work has run, EVERY device holds the complete d_points[g] (double[ndims*Np]) and d_indices[g] (int64[Ndens*Np]);
 * these are device pointers on device first_device+g, streams[g] (hipStream_t, or streams == NULL for the null
 * streams) is where device g's work is enqueued.  The call only enqueues.  Ordering, both ways: device g's kernel
 * starts only after the work already queued on EVERY streams[h] is over (it overwrites their arrays: consumers of
 * the previous product on those streams are safe), and each stream continues only once the slices of all other
 * devices have arrived.  Topologies without peer access fall back to hipMemcpyPeerAsync, one copy per array and
 * destination (kdehip_product_multi_transfers_per_product tells: 0 = fused).
 * VISIBILITY of the peer-written slices: the stores are plain global stores of the producing kernel; they are complete
 * and visible to device h when work on streams[h] that was enqueued AFTER this call starts (the call makes streams[h]
 * wait for every device's `done` event, and a kernel that starts behind that wait begins with a system-scope acquire
 * of its caches).  A consumer that is ALREADY RUNNING on device h while the product is sampled -- a persistent kernel
 * polling the arrays -- has no such acquire and may read stale lines from its L2: consume the result from work
 * enqueued behind the call.
 * ALLOCATION of d_points[h] / d_indices[h]: plain hipMalloc memory is peer-mapped by hipDeviceEnablePeerAccess (done at
 * create).  Arrays from a stream-ordered pool (hipMallocAsync) or from virtual-memory mappings are reachable from a
 * peer only if the pool / mapping grants that device access (hipMemPoolSetAccess / hipMemSetAccess); the call looks at
 * every destination once per product and takes the copy path for anything it cannot show reachable
 * (KDEHIP_PEER_STORES=0 forces the copy path, =1 skips the look-up).  The verdicts are remembered per plan, keyed by the
 * array's ADDRESS and the writing device: an array that is freed and re-allocated at the same address from another kind
 * of allocator keeps its old verdict until kdehip_clear_cache() is called (which forgets all of them) -- a caller that
 * switches allocators under a live plan calls it, or sets KDEHIP_PEER_STORES. */
typedef struct kdehip_product_multi kdehip_product_multi;
int kdehip_product_multi_create(kdehip_product_multi **out, int Ndens, const kdehip_density *trees, int ndims,
                                const uint8_t *partialDimMask, int precision, int first_device, int ngpus);
void kdehip_product_multi_destroy(kdehip_product_multi *mp);
int kdehip_product_multi_ngpus(const kdehip_product_multi *mp);
kdehip_product *kdehip_product_multi_plan(kdehip_product_multi *mp, int g); /* the plan on device g (owned by mp) */
int kdehip_product_multi_sample_philox(kdehip_product_multi *mp, int64_t Np, int Niter, uint64_t seed,
                                       int64_t sample_offset, int addEntropy, double *const *d_points,
                                       int64_t *const *d_indices, void *const *streams);
/* copy-engine transfers each device issues per product: 0 when the all-gather is fused into the kernel */
int kdehip_product_multi_transfers_per_product(const kdehip_product_multi *mp);

/* ---- (2c) densities that live in HBM, and products of them ------------------------------------------------
 * The reference hands gibbs1 host arrays (src/MSGibbs01.jl:527-537) and sections (1) and (2) do the same.  A caller
 * whose densities stay on the device between products -- the inputs of the next product are the outputs of the
 * previous ones -- uploads each BallTreeDensity ONCE (its means, bandwidth, weights, permutation and the frontier of
 * every level, expanded from its child arrays at upload; src/BallTreeDensity01.jl:11-24, src/MSGibbs01.jl:500-523)
 * and then runs prodAppxMSGibbsS (src/MSGibbs01.jl:645-703) on handles: the per-product re-layout into tiles is a
 * gather kernel on the GPU, only a few KB of descriptors cross PCIe, nothing comes back.  Same results as
 * kdehip_prod_philox, bit for bit (same layout, same kernels, same Philox stream). */
typedef struct kdehip_device_density kdehip_device_density;
int kdehip_density_upload(kdehip_device_density **out, const kdehip_density *host, int device);
void kdehip_density_free(kdehip_device_density *d); /* waits for the device first */
int64_t kdehip_density_npts(const kdehip_device_density *d);
int kdehip_density_ndim(const kdehip_device_density *d);
/* prodAppxMSGibbsS on device-resident densities, enqueue only: pack (GPU) + conditional tables + sampling.  The sampling
 * -- everything that touches the outputs -- runs on `stream` (hipStream_t, NULL = default stream); the preparation of
 * the product (descriptor upload, tile gather, tables: it reads only the immutable densities and writes only the
 * plan's own block) runs on a stream of the library's, and `stream` waits for it: of products enqueued back to back,
 * number k+1 is prepared while number k samples.  (Because of that second stream a call cannot be recorded by a stream
 * capture on `stream`; graphs are made of the runs of a resident plan, kdehip_product_sample_*.)  d_points (double[ndims*Np]), d_indices (int64[Ndens*Np]) and the optional
 * d_labels (as kdehip_product_sample_philox) are device pointers on the densities' device.  Random numbers: the
 * device Philox stream keyed by (seed, sample_offset + s, draw).  The plan built for the call is released by a later
 * call (or kdehip_clear_cache) once its work has run; at most 8 such calls are in flight per device and stream (a
 * caller beyond that waits for the oldest call of ITS OWN stream, never for another stream's work).  Devices 0..63. */
int kdehip_prod_philox_device(int Ndens, kdehip_device_density *const *trees, int64_t Np, int Niter, uint64_t seed,
                              int64_t sample_offset, int addEntropy, const uint8_t *partialDimMask, int precision,
                              double *d_points, int64_t *d_indices, int32_t *d_labels, void *stream);

/* The same with host output buffers (pts: ndims*Np, ind: Ndens*Np), blocking: for hosts that keep no device arrays of
 * their own -- a Julia caller without AMDGPU.jl uploads its densities once and then pays neither the host re-layout
 * nor the upload of the tiles per product (sample offset 0). */
int kdehip_prod_philox_resident(int Ndens, kdehip_device_density *const *trees, int64_t Np, int Niter, uint64_t seed,
                                int addEntropy, const uint8_t *partialDimMask, int precision, double *pts, int64_t *ind);

/* ---- (2d) the resident chain: products feed products without leaving HBM -----------------------------------
 * The reference's `*` is `pGM, = prodAppxMSGibbsS(...); kde!(pGM)` (src/MSGibbs01.jl:707-726), and in a belief-propagation
 * sweep its result is an input of the next products.  kdehip_prod_philox_device leaves pGM in HBM; these entries turn it
 * into the next product's input there.
 *
 * kdehip_density_from_device_points = `kde!(points)` (src/KDE01.jl:3-27) on a D x N column-major matrix that lives on
 * `device` (d_points; `stream` = the hipStream_t that produced it, waited for): the LOOCV bandwidth search reads the device
 * matrix as it is, the ball tree (src/BallTree01.jl:415-434) is built by the library's pooled host builder from ONE 8*D*N
 * byte copy that comes down while the search runs, and the density's block goes straight back up.  Blocking; the result
 * is bit for bit the density kdehip_make_density_auto builds from the same points.  bw_out (D standard deviations) and
 * nevals are optional.  N >= 2. */
int kdehip_density_from_device_points(kdehip_device_density **out, const double *d_points, int64_t D, int64_t N,
                                      int device, void *stream, double *bw_out, int32_t *nevals);
/* `*(trees; addEntropy)` (src/MSGibbs01.jl:707-726) on handles: Np = round(mean Npts), Niter = 5, device Philox keyed by
 * `seed`, then kde!(pGM) -- the product matrix never leaves the device; one density with addEntropy = 0 is the reference's
 * shortcut (:713-716: kde! of its own points).  Blocking, on the calling thread's stream.  Same numbers as
 * kdehip_prod_philox(seed) followed by kdehip_make_density_auto on the host. */
int kdehip_mul_device(kdehip_device_density **out, int Ndens, kdehip_device_density *const *trees, uint64_t seed,
                      int addEntropy, double *bw_out, int32_t *nevals);
/* `*` for MANY products in one call -- the reference's serving shape: a belief-propagation sweep calls `*`
 * (src/MSGibbs01.jl:707-726) dozens of times on densities of 100-300 points (test/runtests.jl:189-201), and one at a time
 * each is a blocking call of >= 10 dependent small launches.  Here every product is sampled by the batched sampler
 * (kdehip_prod_philox_batch), the LOOCV bandwidth searches (src/KDE01.jl:3-27, src/CrossValidation.jl:44-120) of all
 * outputs of one size advance in the SAME launches, the ball trees (src/BallTree01.jl:415-434) are built by the pooled
 * host builder under the searches from ONE copy of all matrices, and the nprod resulting densities share one device block.
 * out[i] (nprod handles, each freed with kdehip_density_free; the shared block goes with the last of them) is bit for
 * bit the density kdehip_mul_device(items[i]) returns: same product (Philox keyed by items[i].seed), same bandwidth
 * search evaluations, same tree.  bw_out (optional): nprod rows of KDEHIP_MAX_DIMS doubles, row i = the D bandwidths of
 * result i; nevals (optional): nprod counts.  An item with one density and addEntropy = 0 is the reference's shortcut
 * (:713-716); results of fewer than 2 or more than 2048 points are built by a call of their own inside this one.
 * All densities on one device.  Blocking, on the calling thread's stream.  On an error no handle is returned. */
typedef struct kdehip_mul_item {
  int32_t Ndens;
  int32_t addEntropy;
  kdehip_device_density *const *trees; /* Ndens handles */
  uint64_t seed;
} kdehip_mul_item;
int kdehip_mul_device_batch(int nprod, const kdehip_mul_item *items, kdehip_device_density **out, double *bw_out,
                            int32_t *nevals);
/* The reference's arrays of a density the library built (the two entries above), shaped as in kdehip_make_density; any
 * pointer may be NULL; bw_out: its D LOOCV bandwidths (standard deviations).  A density that came from
 * kdehip_density_upload has no such mirror (KDEHIP_ERR_UNSUPPORTED): its arrays are the caller's. */
int kdehip_density_download(const kdehip_device_density *d, double *centers, double *ranges, double *weights,
                            int64_t *left_child, int64_t *right_child, int64_t *lowest_leaf, int64_t *highest_leaf,
                            int64_t *permutation, double *means, double *bandwidth, double *bandwidthMin,
                            double *bandwidthMax, double *bw_out);

/* ---- (2e) many products in one call ----------------------------------------------------------------------------
 * The serving pattern of a belief-propagation host: dozens of independent 100-300-chain products per sweep, each of which
 * fills a fraction of the device and is latency bound.  One call lays all of them out in one device block (one descriptor
 * upload, one gather launch for every tile) and samples each (dimension count, density count) group of fp64 products of
 * 2..4 densities with every dimension active in ONE launch -- workgroups indexed by (product, chain block), each fetching
 * its product's plan through the scalar cache.  Every product's result is bit for bit that of kdehip_prod_philox_device
 * with the same arguments.  Products outside that domain (fp32, masks, 1 or more than 4 densities) are enqueued one by one
 * inside the same call.  Enqueue only, everything on `stream`; all densities on one device. */
typedef struct kdehip_batch_item {
  int32_t Ndens;
  int32_t Niter;
  kdehip_device_density *const *trees;  /* Ndens handles */
  int64_t Np;
  uint64_t seed;
  int64_t sample_offset;
  int32_t addEntropy;
  int32_t reserved_;
  const uint8_t *partialDimMask;        /* Ndens*ndims bytes or NULL */
  double *d_points;                     /* device, double[ndims*Np]  */
  int64_t *d_indices;                   /* device, int64[Ndens*Np]   */
  int32_t *d_labels;                    /* device, optional          */
} kdehip_batch_item;
int kdehip_prod_philox_batch(int nprod, const kdehip_batch_item *items, int precision, void *stream);

/* ---- diagnostics (not part of the drop-in surface; bench.py and the tests use them) -----------------------------
 * While enabled, every kdehip_prod_philox_device call brackets its sampling launch with a pair of timing events on the
 * caller's stream; kdehip_profile_sampler_read waits for the device's calls in flight and returns the sum of those
 * durations and their count for the calls that were enqueued on `stream` since the switch was last set (which also resets
 * the sums).  The switch is process-wide; the sums are kept per device and caller stream, so concurrent callers on
 * streams of their own do not mix. */
void kdehip_profile_sampler(int enable);
int kdehip_profile_sampler_read(int device, void *stream, double *total_ms, int64_t *launches);
/* With the switch on, kdehip_product_multi_sample_philox brackets every device's sampling launch too; this returns, for
 * the LAST product of `mp` (waiting for it), kernel_ms[g] = the duration of device g's launch and done_ms[g] = the host
 * time at which g's slice had arrived on every device, relative to the first device to get there (both arrays: ngpus
 * entries): a straggling device or link shows up as skew, a slow kernel as duration. */
int kdehip_product_multi_timing(kdehip_product_multi *mp, double *kernel_ms, double *done_ms);

/* The callers either side of the product (section 5, 2d) are blocking entries on the calling thread's own stream: their
 * device work cannot be bracketed from outside.  With kdehip_profile_sampler(1), the LOOCV bandwidth search (which = 0:
 * preparation + every round of a search, per batch of rounds), kdehip_evaluate (which = 1: the partial and finish
 * kernels) and the GPU tree builder (which = 2: kdehip_make_densities_device's kernel) bracket their launches with a pair
 * of timing events; this returns the sum of those durations and the number of bracketed phases since the last read, and
 * resets both (process-wide sums).  bench.py --frow reports them as kernel time. */
int kdehip_profile_phase_read(int which, double *total_ms, int64_t *count);
/* The fp32 screen of the deep levels (csrc/screen_device.hpp) certifies its decisions with an error bound whose premise is
 * that the hardware's v_rcp_f32, v_rsq_f32 and v_exp_f32 are within 1 ulp (a relative error of at most 2 u, u = 2^-24).
 * This entry MEASURES that on `device`: over the `count` fp32 bit patterns from `first_bits` on it returns the largest
 * error of instruction `which` against fp64 formed on the device -- 0: v_rcp_f32, 1: v_rsq_f32, 2: v_exp_f32 (2^x), each
 * relative, in units of u; 3: |v_exp_f32(x) - 2^x| in units of 2^-126 (for x < -126, where the bound only needs "off by
 * less than the smallest normal").  worst_bits / worst_result_bits (optional): the input with the largest error and the
 * hardware's result for it.  tests/test_gpu_ulp.py sweeps the screen's whole input ranges with it.  Blocking. */
int kdehip_selftest_fp32(int which, uint32_t first_bits, uint64_t count, int device, double *max_err,
                         uint32_t *worst_bits, uint32_t *worst_result_bits);

/* ---- (3) host twin of the device RNG ----------------------------------------------------------
 * Fills the arrays a caller would pass as randU / randN so that a streams-run (or the Julia
 * reference, via its randU=/randN= keywords, src/MSGibbs01.jl:661-662) consumes exactly the
 * numbers the Philox run draws.  out_u has nsamples*K, out_n has nsamples*R entries. */
void kdehip_philox_fill_uniform(uint64_t seed, int64_t sample_begin, int64_t nsamples, int64_t K,
                                double *out_u);
void kdehip_philox_fill_normal(uint64_t seed, int64_t sample_begin, int64_t nsamples, int64_t R,
                               double *out_n);

/* ---- (4) density construction (host; reference kde!(points, ks, weights), src/KDE01.jl:34-57 ->
 * makeBallTreeDensity, src/BallTreeDensity01.jl:192-231 -> buildTree!, src/BallTree01.jl:415-434).
 * A Julia caller keeps using its own kde!; this entry serves hosts without the reference.
 * points: D x N column-major; ks: nks = 1 or D standard deviations (squared inside);
 * weights_in: N or NULL (= ones).  Outputs are caller-allocated: centers, ranges, means,
 * bandwidth: D*2N; weights and the five index arrays: 2N; bandwidthMin/Max: D*N.
 * Thread-safe.  From 512 points up the top levels hand their left subtree to a process-wide pool of at most 15
 * worker threads (started on first use, asleep in between, never joined; csrc/host_pool.hpp): the arrays do not
 * depend on the number of threads, and a process that fork()ed away from the workers builds serially. */
int kdehip_make_density(int64_t D, int64_t N, const double *points, const double *ks, int64_t nks,
                        const double *weights_in, double *centers, double *ranges, double *weights,
                        int64_t *left_child, int64_t *right_child, int64_t *lowest_leaf,
                        int64_t *highest_leaf, int64_t *permutation, double *means,
                        double *bandwidth, double *bandwidthMin, double *bandwidthMax);

/* The bandwidth-dependent half of the construction on an existing tree: topology, bounding boxes, weights and means
 * do not depend on ks, only `bandwidth` (leaves ks^2, internal nodes by moment matching,
 * src/BallTreeDensity01.jl:141-187) and bandwidthMin/Max do.  Lets kde!(points) build its tree while the GPU searches
 * the LOOCV bandwidth; the result is bit-identical to kdehip_make_density called with this ks. */
int kdehip_density_set_bandwidth(int64_t D, int64_t N, const double *ks, int64_t nks, const double *weights,
                                 const int64_t *left_child, const int64_t *right_child, const double *means,
                                 double *bandwidth, double *bandwidthMin, double *bandwidthMax);

/* The same construction on the GPU, for a batch of `nb` densities of one dimension count (one workgroup per
 * density, level-synchronous; csrc/treebuild.hip): bit-identical arrays -- same node numbering, leaf order and
 * statistics as kdehip_make_density and the reference.  Every pointer argument is an array of nb pointers to
 * caller-allocated host arrays shaped as in kdehip_make_density (ks[j]: nks values; weights_in may be NULL, or hold
 * NULL entries, for unit weights).  Densities the device builder cannot hold in LDS are refused with
 * KDEHIP_ERR_UNSUPPORTED (kdehip_make_density_device_supported tells beforehand; callers then use
 * kdehip_make_density). */
int kdehip_make_density_device_supported(int64_t D, int64_t N);
int kdehip_make_densities_device(int nb, int64_t D, const int64_t *Ns, const double *const *points,
                                 const double *const *ks, int64_t nks, const double *const *weights_in,
                                 double *const *centers, double *const *ranges, double *const *weights,
                                 int64_t *const *left_child, int64_t *const *right_child,
                                 int64_t *const *lowest_leaf, int64_t *const *highest_leaf,
                                 int64_t *const *permutation, double *const *means, double *const *bandwidth,
                                 double *const *bandwidthMin, double *const *bandwidthMax, int device);

/* ---- (5) direct evaluation and automatic bandwidth (the callers either side of the product) ----
 * kdehip_evaluate: `evaluateDualTree(bd, pos)` / `bd(pos)` with the reference's default
 * FORCE_EVAL_DIRECT = true (src/DualTree01.jl:130-162, 303-346, 370-446): p_out[q] = density of `bd` at
 * column q of pos (D x Nq, column-major).  leave_one_out != 0 is the `bd == locations` case
 * (`makeDualTree(bd, errTol)`, :361-368): pos is ignored, the density is evaluated at its own points
 * without the self term and divided by (1 - w_q) (:335); p_out has npts entries in the ORIGINAL point
 * order.  Host buffers, blocking. */
int kdehip_evaluate(const kdehip_density *bd, const double *pos, int64_t Nq, int leave_one_out,
                    double *p_out, int device);
/* kdehip_auto_bandwidth: the bandwidth `kde!(points)` selects (src/KDE01.jl:3-27): per dimension,
 * `ksize` of the 1-D marginal = golden-section search (tol 1e-2) over the leave-one-out
 * log-likelihood (src/CrossValidation.jl:15-120).  points: D x N column-major; bw_out: D standard
 * deviations; nevals (optional): number of likelihood evaluations -- the reference's count: for the smaller marginals
 * the search evaluates the two possible successors of the point in flight in the same launch (csrc/evaluate.hip
 * loo_round_spec_kernel); the ones golden does not ask for are neither booked nor counted. */
int kdehip_auto_bandwidth(int64_t D, int64_t N, const double *points, double *bw_out, int32_t *nevals,
                          int device);
/* `kde!(points)` in one call (src/KDE01.jl:3-27: LOOCV bandwidth per dimension, then kde!(points, bwds)): the host tree
 * builder runs on the library's worker threads WHILE the GPU searches the bandwidth -- topology, bounding boxes,
 * weights and means do not depend on it -- and the variances are filled in afterwards.  Arrays as
 * kdehip_make_density (unit weights), bw_out and nevals as kdehip_auto_bandwidth; bit-identical to the two calls one
 * after the other.  N >= 2. */
int kdehip_make_density_auto(int64_t D, int64_t N, const double *points, double *bw_out, int32_t *nevals, int device,
                             double *centers, double *ranges, double *weights, int64_t *left_child, int64_t *right_child,
                             int64_t *lowest_leaf, int64_t *highest_leaf, int64_t *permutation, double *means,
                             double *bandwidth, double *bandwidthMin, double *bandwidthMax);

#ifdef __cplusplus
}
#endif
#endif /* KDEHIP_H */
