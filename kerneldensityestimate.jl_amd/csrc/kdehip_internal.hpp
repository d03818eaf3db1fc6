// kdehip_internal.hpp -- shared declarations of libkdehip.so (host side + PODs passed to kernels).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/kdehip.h"

namespace kdehip {

// ---- error reporting (thread-local message behind kdehip_last_error) ---------------------------
int set_error(int code, const std::string &msg);
const char *last_error_cstr();

// ---- packed per-level layout ("pack_levels") ---------------------------------------------------
// The frontier of density j after l levelDown! calls (reference src/MSGibbs01.jl:500-523) is
// data-independent, so it is materialised once on the host.  A frontier of n nodes is stored as a
// tile of B rows x 64 lanes, B = ceil(n/64): frontier entry z (0-based, reference order) sits in
// lane z / B, row z % B, i.e. lane `ln` of a wavefront owns the CONTIGUOUS entries
// ln*B .. ln*B+B-1.  A tile is one contiguous block of T elements:
//   header  kTileHeader (8) elements: the bandwidth vector of entry 0 (= of every entry if uniform_bw)
//   fp64:   row i = F fields of 64 lanes each + 1 pad element (row stride RS = F*64 + 1); the pad makes the
//           column walk of the second selection pass bank-conflict free once the tile sits in LDS;
//           element (row i, field f, lane ln) at hdr_off + kTileHeader + i*RS + f*64 + ln
//   fp32 (round 4): rows in PAIRS -- the values of rows 2p and 2p+1 for one (field, lane) are adjacent, so the packed
//           first pass (two rows per v_pk_* instruction) fetches a register pair with ONE ds_read_b64 (2 LDS cycles
//           for 512 bytes; two ds_read_b32 take 4: the LDS array was 55 % busy at config 5); pair stride
//           RS = 2*F*64 + 2 (the 2 pad elements: the column walk stays conflict free);
//           element (row i, field f, lane ln) at hdr_off + kTileHeader + (i/2)*RS + (f*64 + ln)*2 + i%2
//   (TileAddr<T> below is the one place that knows this)
//   fields: [0, D) mean per dimension; then, unless the level has ONE bandwidth vector shared by
//   all its nodes (uniform_bw: always true for the leaf level of a reference-built density),
//   [D, 2D) bandwidth (variance) per dimension; last field (F-1): weight.
// `pos` = i*64 + ln identifies an entry inside its tile (also the index into the permutation row).
// Padding entries (z >= n) carry weight 0, variance 1, mean 0 and never win a draw.
// The LDS image of a tile is a byte copy of [hdr_off, hdr_off + stage_bytes).
constexpr int kTileHeader = 8;
#if defined(__HIPCC__)
#define KDEHIP_TILE_HD __host__ __device__ inline
#else
#define KDEHIP_TILE_HD inline
#endif
template <int ELEM_BYTES>
struct TileAddrBytes {
  static constexpr bool kPaired = ELEM_BYTES == 4;
  static constexpr int kLane = kPaired ? 2 : 1;   // elements between the same field of adjacent lanes
  static constexpr int kField = 64 * kLane;       // elements between adjacent fields of one entry
  // RS: elements per row (fp64) / per PAIR of rows (fp32)
  KDEHIP_TILE_HD static constexpr int stride(int F) { return kPaired ? 2 * F * 64 + 2 : F * 64 + 1; }
  // element offset of (row r, field 0, lane 0) from the tile's first row
  // (in the integer type of its arguments: 32-bit on the kernels' step paths)
  template <typename I>
  KDEHIP_TILE_HD static constexpr I row(I r, I RS) { return kPaired ? (r >> 1) * RS + (r & 1) : r * RS; }
  // the same for row base + k where `base` is EVEN and k a small constant (relative to the base row's offset)
  KDEHIP_TILE_HD static constexpr int rel(int k, int RS) { return kPaired ? (k >> 1) * RS + (k & 1) : k * RS; }
  // elements of the rows of a tile with B rows per lane
  KDEHIP_TILE_HD static constexpr int64_t body(int64_t B, int F) { return (kPaired ? (B + 1) / 2 : B) * stride(F); }
  // elements covered by rows [r0, r0 + nrows), r0 even
  KDEHIP_TILE_HD static constexpr int64_t span(int64_t nrows, int64_t RS) { return (kPaired ? (nrows + 1) / 2 : nrows) * RS; }
};
template <typename T> using TileAddr = TileAddrBytes<int(sizeof(T))>;
constexpr double kMaxUniformRatio = 1e5;  // see pack_layout_shapes: when a shared-bandwidth frontier gets the compact tile
enum StageMode : int32_t {
  kStageGlobal = 0,    // tile too large for LDS: wavefronts read it from global memory (L1/L2)
  kStageResident = 1,  // all densities' tiles of the level fit the LDS pool at once
  kStageStream = 2,    // one tile per step, double-buffered in the LDS pool
  kStageChunked = 3,   // tile larger than half the pool: rows streamed through the two halves in chunks
  kStageScreen = 4,    // (screen descriptors only) the level is SCREENED in fp32: see "fp32 screening" below
  kStageScreenStream = 5,  // ... with its screen tiles streamed through the pool halves one per step (they do not fit together)
  kStageScreenChunked = 6  // ... in chunks of chunk_rows rows (a screen tile is larger than half the pool); chunk 0 carries the header
};
constexpr int kLdsPoolBytes = 120 * 1024;   // LDS bytes for staged tiles (of 160 KiB per CU; the rest: chain state)

struct LevelDesc {
  int32_t n;            // frontier size n_{j,l}
  int32_t B;            // rows = entries per lane
  int32_t F;            // fields per row: 2D+1, or D+1 when uniform_bw
  int32_t uniform_bw;   // 1: every node of the frontier has the header's bandwidth vector
  int64_t hdr_off;      // element offset (units of T) of the tile header; rows start kTileHeader later
  int64_t perm_off;     // offset of the tile's int32 permutation row (indexed by pos, 0 = internal)
  uint32_t mask_bits;   // bit d: this density informs dimension d (partialDimMask)
  uint32_t others_bits; // bit d: some other density informs dimension d
  int32_t stage_mode;   // StageMode of this LEVEL (same for every density)
  int32_t lds_off;      // byte offset of the tile image in the LDS pool (resident mode)
  int32_t stage_bytes;  // bytes to copy (header + rows, rounded up to 1 KiB)
  int32_t last_lane;    // (n - 1) / B: the lane that owns the last entry (kept here: no integer division per step)
  int32_t chunk_rows;   // rows per LDS chunk in chunked mode (multiple of 4: 16-byte aligned chunk starts, whole row pairs)
  // chunked mode, the one-round second pass (gibbs_device.hpp "chunked tiles"): chunks per segment in bits 0..15 (0: the
  // segment form does not apply to this tile), segments per lane block in bits 16..31 -- worked out by the packer so
  // that no step of the kernel divides integers (four uniform divisions a step were 3 % of config 4's instructions)
  int32_t seg;
};
// the segment geometry of a chunked tile of B rows per lane in chunks of rc rows: a segment = a whole number of chunks, at
// most 64 rows, at most kMaxSegDesc segments (else 0)
constexpr int kMaxSegDesc = 8;
inline int32_t seg_geometry(int B, int rc) {
  if (rc <= 0 || rc > 64) return 0;
  const int cps = 64 / rc, sr = cps * rc, nseg = (B + sr - 1) / sr;
  return nseg <= kMaxSegDesc ? (cps | (nseg << 16)) : 0;
}
static_assert(sizeof(LevelDesc) == 64, "LevelDesc is read with scalar loads; keep it 64 bytes");

// ---- fp32 screening of the deep levels, fp64 certification (round 5; DESIGN.md "fp32 screening") --------------------
// On a level whose fp64 tiles do not fit the LDS pool together (streamed / chunked levels) an fp64 plan ALSO carries
// the level's tiles in fp32 -- means centred at the density's root mean (m' = m - mu0, subtracted in fp64, rounded once),
// variances, weights, in the fp32 row-pair layout (TileAddr<float>) -- when all M of THOSE fit the pool at once.  A draw
// step then evaluates the frontier in packed fp32 from LDS, together with a rigorous bound on the error of every
// cumulative sum (each term carries the relative error A + Bc |x|, x its base-2 exponent; the wave-uniform A, Bc follow
// from the header's bounds below), and accepts the fp32 decision only when the target u * total is farther than that bound
// from every boundary it could cross; otherwise -- about 1 % of the steps at BASELINE config 3 -- the step is repeated in
// fp64 from the plan's fp64 tile in global memory.  The drawn kernel's mean and variance are always adopted from the fp64
// tile: labels and points are bit for bit those of the fp64 path.
// A screen tile: [header kScreenHeaderFloats floats][row pairs as TileAddr<float>].  Header (written by the GPU,
// pack_device.hip screen_build_kernel): mu0[8] (fp64), cmin[8] (fp64: per dimension the smallest variance of the tile =
// THE variance of a shared-bandwidth tile), mmax[8] (fp32: max |m'_d|, rounded up), then flags[8] (fp32): [0] = 1 when
// every value of the tile is inside the ranges the error analysis assumes (kScreen* below), else 0 = never screened.
// Descriptors: a second [M][L+1] table of LevelDesc right behind the plan's level table; entry (j, l) has stage_mode ==
// kStageScreen when LEVEL l is screened, n / B / F / uniform_bw / last_lane of the fp64 tile, hdr_off = offset of the
// screen tile's header in FLOATS from the plan's data, lds_off / stage_bytes of its LDS image.
// ---- step descriptors (round 5) --------------------------------------------------------------------------------------
// What a draw step of the register-resident sampler needs of a tile: 32 bytes, one s_load_dwordx8 per step (half the scalar
// registers of the 16-dword LevelDesc it replaces on the step path, no shuffling of fields: c3 kernel -0.6 %).  A third
// [M][L+1] table behind the level and screen tables (both always present: the screen table is all zero when no level is
// screened).  Tried on top and dropped (profiles/r05_experiments.md): the descriptor requested one step ahead (the compiler
// sinks the load back to its use), and one packed dword per tile kept in scalar registers for the whole level (+1.8 %).
struct StepDesc {
  int32_t n;            // frontier size
  int32_t flags;        // last_lane | uniform_bw << 8
  int32_t lds_off;      // LDS offset of the tile's image: the fp64 tile's on a resident level, the SCREEN tile's on a screened one
  int32_t stage_bytes;  // bytes of the fp64 image (streamed mode)
  int32_t chunk_rows;   // rows per chunk (chunked mode)
  int32_t seg;          // LevelDesc.seg
  int32_t hdr_lo, hdr_hi;  // element offset of the fp64 tile's header in the plan's data
};
static_assert(sizeof(StepDesc) == 32, "StepDesc is read with one 32-byte scalar load");

constexpr int kScreenHeaderFloats = 48;
constexpr int kScreenMaxRows = 64;           // rows per lane up to which a level is screened (one second-pass round)
constexpr int kScreenMaxRowsChunked = 128;   // ... when its screen tiles are chunked (second pass from global memory, two rounds)
constexpr float kScreenMaxAbsMean = 65536.0f;  // |m'_d|, |centre'_d| <= 2^16
constexpr double kScreenMinVar = 1.0 / 128.0, kScreenMaxVar = 256.0;  // variances (tile, leave-one-out) in [2^-7, 2^8]
// The bound's centring term is linear in na (1 + |x|) with a 1 % allowance for the second-order part and the margin's 5 %
// (2.1 E against 2 E): exp(d) - 1 <= 1.06 d needs d = ln2 na (1 + |x|) <= 0.11, and |x| < 127 wherever a value is not
// flushed to zero -- so na <= 2^-11 (d <= 0.044) keeps it rigorous; typical data have na ~ 2^-19.
constexpr float kScreenMaxNa = 0x1p-11f;
// ... enforced per dimension, with the step's other range checks: na = 1.01 u sqrt(c0) sqrt(sum_d a2_d), a2_d = (g_d / sigma_d)^2,
// so a2_d <= (kScreenMaxNa / (1.01 u sqrt(c0)))^2 / 8 = 1.14e7 in each of the (at most 8) dimensions is sufficient
// (g / sigma <= 3376 per dimension; typical data: < 100)
constexpr float kScreenMaxA2 = 1.14e7f;

// Conditional table of density j on level l (see gibbs_kernel.hip "conditional tables"): rows of n+1
// values (inclusive scan over the n frontier nodes, then the total).  Only levels whose frontier sizes
// are powers of two (and <= 64) are tabulated: the labels of all densities are packed into one word,
// density k occupying bits [shift, shift+bits) of it, and the row index of density j is that word with
// j's own digit removed.
struct TabDesc {
  int64_t off;        // element offset (units of T) of row 0 in the plan's table buffer; -1 = not tabulated
  int64_t row_base;   // index of row 0 in the global row enumeration (table build: one wavefront per row)
  int32_t n;          // frontier size of density j on this level (= 1 << bits)
  int32_t ncfg;       // number of rows = product of the other densities' frontier sizes
  int32_t shift;      // position of this density's digit in the packed label word
  int32_t bits;       // width of the digit
};
static_assert(sizeof(TabDesc) == 32, "TabDesc is read with one 32-byte scalar load");
constexpr int64_t kTabMaxEntries = 512 * 1024;  // table budget per plan (4 MiB in fp64)
constexpr int64_t kTabMinChains = 16;           // tables are built at the first run with this many chains (a run is
                                                // latency bound: the ~70 us build pays off within one run)

struct PlanDev {
  const void *data;          // T[...]
  const int32_t *perm;       // int32[...]
  const LevelDesc *levels;   // [M][L+1], level 0 = root; the 8 bytes in front of it: counter of uniform-fallback draws
  const void *tables;        // T[...] conditional tables (may be unbuilt: RunArgs.use_tables)
  const TabDesc *tabdesc;    // [M][L+1]
  int64_t tab_rows_total;
  int32_t M, L, D, Lt;       // Lt: levels 1..Lt are tabulated (0 = none)
  int32_t screened;          // 1: the plan carries screen tiles (descriptors behind the level table) and they are built; 2: some chunked
  // bit d: dimension d is CIRCULAR (2 pi) -- the enumerated on-manifold operators of include/kdehip.h "manifolds", applied at
  // the reference's hook points (src/MSGibbs01.jl:290, 183-184 / 210-213, 456) by the general sampler's generic mode only
  uint32_t circ_bits;
  int32_t reserved_[2];
};

constexpr int kMaxPeers = 7;  // other GPUs of one node
struct BatchEntry;
struct RunArgs {
  int64_t Np;
  int32_t Niter;
  int32_t addEntropy;
  int32_t rng_philox;   // 0: read d_randU/d_randN, 1: on-device Philox
  int32_t variant;
  int32_t table_build;  // 1: this launch fills the conditional tables instead of sampling
  int32_t use_tables;   // 1: the tables are built and may be used
  int32_t use_screen;   // 1: the screen tiles are built and may be used (fp64 plans, gibbs_lean.hip)
  const double *randU;
  const double *randN;
  int64_t K, R;         // per-sample consumption
  int64_t nU, nN;       // stream lengths (streams mode)
  uint64_t seed;
  int64_t sample_offset;
  double *points;
  int64_t *indices;
  int32_t *labels;
  // The all-gather of a multi-GPU product (kdehip_product_multi_*): the kernel stores every final point and label
  // not only into its own device's arrays but straight into the arrays of the `npeers` other devices as well
  // (peer-mapped pointers, stores travel over xGMI) -- no copy engines, no extra launches.
  int32_t npeers;
  double *peer_points[kMaxPeers];
  int64_t *peer_indices[kMaxPeers];
  // A batched launch (kdehip_prod_philox_batch; the BATCH instantiations of gibbs_lean.hip): workgroup b belongs to
  // product batch_map[b], whose plan and run parameters are entry batch_map[b] of `batch` (device memory, read through
  // the scalar cache); everything above except rng_philox / variant is then taken from the entry.
  const BatchEntry *batch;
  const int32_t *batch_map;
};

// One product of a batched launch: what differs from product to product, in three 64-byte pieces (each is read with one
// scalar load and unpacked on its own, like LevelDesc).
struct BatchPlanHead {   // = the first 64 bytes of PlanDev
  const void *data;
  const int32_t *perm;
  const LevelDesc *levels;
  const void *tables;
  const TabDesc *tabdesc;
  int64_t tab_rows_total;
  int32_t M, L, D, Lt;
};
struct BatchRun {
  int32_t reserved_[4];    // (the tail of PlanDev)
  int64_t Np;
  uint64_t seed;
  int64_t sample_offset;
  double *points;
  int64_t *indices;
  int32_t *labels;
};
struct BatchFlags {
  int32_t Niter, addEntropy, use_tables;
  int32_t first_block;   // the product's first workgroup in the launch
  int32_t pad_[12];
};
struct BatchEntry {
  BatchPlanHead head;
  BatchRun run;
  BatchFlags flags;
};
static_assert(sizeof(PlanDev) == 80, "PlanDev layout");
static_assert(sizeof(BatchPlanHead) == 64 && sizeof(BatchRun) == 64 && sizeof(BatchFlags) == 64 && sizeof(BatchEntry) == 192,
              "BatchEntry is read with three 64-byte scalar loads");

// Host result of packing one product (precision-independent description + fp64 payload; the fp32
// payload is a rounding of it).
struct PackedProduct {
  int M = 0, D = 0, L = 0;
  int precision = 64;              // element type of the tiles the geometry was computed for
  std::vector<LevelDesc> levels;   // [M][L+1]
  std::vector<int32_t> front;      // every frontier's node ids (1-based), frontier (j, l) at front_off[j*(L+1)+l]
  std::vector<int64_t> front_off;  // [M*(L+1) + 1]
  int64_t data_elems = 0;          // elements of the tile payload (incl. the screen tiles and the readable tail)
  int64_t tile_elems = 0;          // elements up to the end of the last tile proper (the packers write up to here)
  int64_t perm_elems = 0;          // int32 entries of the permutation rows
  std::vector<double> data;        // fp64 payload (pack_levels only)
  std::vector<int32_t> perm;
  int64_t nodes_per_sweep = 0;     // sum_j sum_{l>=1} n_{j,l}
  bool fast = true;                // product/rsqrt arithmetic + compact uniform tiles in use
  bool all_active = true;          // every dimension of every density is informed by another density
  std::vector<TabDesc> tabdesc;    // [M][L+1]
  std::vector<LevelDesc> screens;  // [M][L+1] screen descriptors ("fp32 screening"), empty = no level is screened
  std::vector<StepDesc> steps;     // [M][L+1] step descriptors
  int nscreened = 0;               // screened levels
  int Lt = 0;                      // tabulated levels 1..Lt
  int64_t tab_entries = 0, tab_rows = 0;
  bool masked = false;
};

// Validates the densities and builds the packed layout.  Returns KDEHIP_OK or an error code.
// `precision` (64/32) decides whether the fast arithmetic path -- and with it the compact
// uniform-bandwidth tiles -- may be used (out.fast).
int pack_levels(int Ndens, const kdehip_density *trees, int ndims, const uint8_t *mask, int precision,
                PackedProduct &out);
// The same in two steps, for callers that own the destination (product.hip packs straight into the pinned upload
// buffer): geometry, descriptors and frontier ids; then the payload in the layout's precision.
// kPackChecked decides fast vs generic arithmetic by looking at every node; kPackOptimistic lays out the fast form
// and leaves the conditions to pack_fill (false = they do not hold: lay out again with kPackGeneric and refill).
enum PackMode : int { kPackChecked = 0, kPackOptimistic = 1, kPackGeneric = 2 };
int pack_layout(int Ndens, const kdehip_density *trees, int ndims, const uint8_t *mask, int precision,
                PackedProduct &out, PackMode pmode = kPackChecked);
bool pack_fill(const PackedProduct &pp, const kdehip_density *trees, void *data, int32_t *perm);

// floor(log(maxNp)/log(2) + 1), reference src/MSGibbs01.jl:568
int nlevels_for(int64_t maxNp);


// ---- kernel launch (gibbs_kernel.hip) ----------------------------------------------------------
// Arithmetic form of the kernel evaluation (see gibbs_kernel.hip).
enum ArithMode : int {
  kModeGeneric = 0,    // the reference's per-dimension divide + log with its NaN rules
  kModeFast = 1,       // product/rsqrt + uniform-bandwidth forms, every dimension active
  kModeFastMasked = 2  // the same with partialDimMask / uninformed dimensions
};
// Device / pinned-host allocations through the library's cache (devmem.cpp).  `bytes` of the free must be the
// `bytes` of the allocation.  Only free a block once the work using it has completed.
hipError_t cached_malloc(void **out, size_t bytes);
void cached_free(void *p, size_t bytes);
hipError_t cached_host_malloc(void **out, size_t bytes);
void drain_pending();  // (product.hip) releases the plans of enqueue-only device products once their work is over
void cached_host_free(void *p, size_t bytes);

// Makes `device` the thread's current HIP device for the lifetime of the guard and restores the caller's
// device on every exit path: no entry point of the library (including kdehip_product_destroy, which a
// garbage collector may call at any time) leaves the thread on another device than it found.
class DeviceGuard {
 public:
  DeviceGuard() = default;
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
  ~DeviceGuard();
  int enter(int device);  // KDEHIP_OK, KDEHIP_ERR_NO_DEVICE or KDEHIP_ERR_ARG (ordinal out of range)
 private:
  int prev_ = -1;
  bool switched_ = false;
};

// Compute units of the current device (cached per device ordinal); 256 on MI355X.
int device_cu_count();

int launch_gibbs(int precision, int mode, const PlanDev &plan, const RunArgs &args, void *stream);
// a group of fp64 products of M (2..4) densities in one launch (gibbs_dispatch.cpp; RunArgs.batch / batch_map)
int launch_gibbs_batch(int D, int M, const PlanDev &plan, const RunArgs &args, void *stream);
int launch_tables_batch(int D, const PlanDev &plan, const RunArgs &args, void *stream);

// kde!(points)'s LOOCV bandwidth search (evaluate.hip) on `stream` of the current device, from the host's copy of the
// D x N matrix and/or a copy that already lives in HBM (`d_points`: nothing is uploaded then).  Blocking.
// `overlap` (optional) is called ONCE, on the calling thread, after the preparation and the first batch of rounds have
// been enqueued and before the host waits for them: work that needs the host but not the bandwidth runs under the search.
// (kLoocvPrepMaxN: marginals up to this size are prepared by the device from the matrix as it is -- no host copy needed.)
constexpr int64_t kLoocvPrepMaxN = 2048;
int auto_bandwidth_run(int D, int64_t N, const double *points, const double *d_points, void *stream, double *bw_out,
                       int32_t *nevals_out, const std::function<void()> *overlap = nullptr);

// Chains per workgroup (= wavefronts per CU, one workgroup per CU at a time) of a sampling launch: 4, 8 or 16,
// the width with the smallest estimated time rounds(width) * cost(width) unless `variant` pins it
// (kdehip_product_set_variant).  Shared by both sampler kernels.
int chains_per_workgroup(int64_t Np, int variant);
// The same for the register-resident sampler (gibbs_lean.hip): diagnostic builds combine a level cut-off (variant % 1000 =
// 100 + k) with a width in the thousands digit (6: sixteen, 8: eight chains per workgroup).
// (Rounds 3-4 also had "wavefront teams" -- one chain on 2 or 4 wavefronts of a 16-wavefront workgroup, plan variants
// 52 / 54: bit-identical, measured slower than one wavefront per chain at every BASELINE shape in both rounds
// (profiles/r03_experiments.md), never selected by a plan; removed in round 5.)
int lean_waves(int64_t Np, int variant);

// Variant codes 30..49 = "gibbs_kernel.hip even where gibbs_lean.hip applies" + (code - 30) as the plain variant.
constexpr int kVariantGenericBase = 30;
constexpr int kLeanNotCovered = 1;  // launch_lean_d*: the run is outside the lean kernel's domain

}  // namespace kdehip
