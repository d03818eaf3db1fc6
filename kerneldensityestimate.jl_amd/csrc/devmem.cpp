// devmem.cpp -- a small cache of device (and pinned host) allocations.
//
// hipMalloc / hipFree / hipHostMalloc cost tens to hundreds of microseconds each and may synchronise the
// device; a one-shot small product (pack, upload, run, copy back: ~0.1 ms of GPU work) would spend most
// of its time in them.  Freed blocks are therefore kept, per device, in power-of-two size classes and
// handed out again; kdehip_clear_cache() returns everything to the driver (SURVEY.md 8b "Ownership").
// Blocks are only recycled after the work that used them has completed: every entry point that frees
// into the cache has synchronised its stream / finished its blocking copies before doing so.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "kdehip_internal.hpp"
#include "phase_timer.hpp"

namespace kdehip {
namespace {

constexpr int kClasses = 40;                              // size class c holds blocks of 2^c bytes
constexpr size_t kMinBlock = size_t(1) << 16;             // 64 KiB: everything smaller shares one class
constexpr size_t kCacheLimit = size_t(1) << 30;           // per device and kind: at most 1 GiB kept
constexpr int kMaxDevices = 64;

struct Cache {
  std::mutex mu;
  std::vector<void *> free_blocks[kClasses];
  size_t cached_bytes = 0;
};
Cache g_dev[kMaxDevices];
Cache g_pinned;  // pinned host memory is not tied to a device

int size_class(size_t bytes) {
  if (bytes < kMinBlock) bytes = kMinBlock;
  int c = 0;
  while ((size_t(1) << c) < bytes) ++c;
  return c;
}

void *take(Cache &c, int cls) {
  std::lock_guard<std::mutex> lock(c.mu);
  auto &v = c.free_blocks[cls];
  if (v.empty()) return nullptr;
  void *p = v.back();
  v.pop_back();
  c.cached_bytes -= size_t(1) << cls;
  return p;
}

bool give(Cache &c, int cls, void *p) {
  std::lock_guard<std::mutex> lock(c.mu);
  if (c.cached_bytes + (size_t(1) << cls) > kCacheLimit) return false;
  c.free_blocks[cls].push_back(p);
  c.cached_bytes += size_t(1) << cls;
  return true;
}

}  // namespace

DeviceGuard::~DeviceGuard() {
  if (switched_) (void)hipSetDevice(prev_);
}

int DeviceGuard::enter(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(KDEHIP_ERR_NO_DEVICE, "no HIP device available (libkdehip has no CPU fallback by design)");
  if (device < 0 || device >= n) return set_error(KDEHIP_ERR_ARG, "device ordinal out of range");
  int cur = -1;
  e = hipGetDevice(&cur);
  if (e != hipSuccess) return set_error(KDEHIP_ERR_NO_DEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
  if (cur == device) return KDEHIP_OK;
  e = hipSetDevice(device);
  if (e != hipSuccess) return set_error(KDEHIP_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  if (!switched_) prev_ = cur;  // (a guard may be entered repeatedly -- multi-GPU loops --: it restores the FIRST device)
  switched_ = true;
  return KDEHIP_OK;
}

// KDEHIP_NO_CACHE=1 (diagnostics): every block straight from / back to the driver
static bool no_cache() {
  static const bool on = [] { const char *e = std::getenv("KDEHIP_NO_CACHE"); return e && e[0] == '1'; }();
  return on;
}

hipError_t cached_malloc(void **out, size_t bytes) {
  *out = nullptr;
  if (no_cache()) return hipMalloc(out, bytes ? bytes : 1);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const int cls = size_class(bytes);
  if (cls >= kClasses || dev < 0 || dev >= kMaxDevices) return hipMalloc(out, bytes ? bytes : 1);
  if (void *p = take(g_dev[dev], cls)) { *out = p; return hipSuccess; }
  return hipMalloc(out, size_t(1) << cls);
}

void cached_free(void *p, size_t bytes) {
  if (!p) return;
  if (no_cache()) { (void)hipFree(p); return; }
  int dev = 0;
  const int cls = size_class(bytes);
  if (hipGetDevice(&dev) == hipSuccess && cls < kClasses && dev >= 0 && dev < kMaxDevices && give(g_dev[dev], cls, p))
    return;
  (void)hipFree(p);
}

hipError_t cached_host_malloc(void **out, size_t bytes) {
  *out = nullptr;
  const int cls = size_class(bytes);
  if (cls >= kClasses) return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (void *p = take(g_pinned, cls)) { *out = p; return hipSuccess; }
  return hipHostMalloc(out, size_t(1) << cls, hipHostMallocDefault);
}

void cached_host_free(void *p, size_t bytes) {
  if (!p) return;
  const int cls = size_class(bytes);
  if (cls < kClasses && give(g_pinned, cls, p)) return;
  (void)hipHostFree(p);
}

}  // namespace kdehip


namespace kdehip {
// kdehip_profile_phase_read (phase_timer.hpp)
namespace {
std::atomic<int> g_phases_on{0};
std::mutex g_phase_mu;
double g_phase_ms[kPhaseCount] = {};
int64_t g_phase_n[kPhaseCount] = {};
}  // namespace
bool profile_phases_on() { return g_phases_on.load(std::memory_order_relaxed) != 0; }
void profile_phases_set(bool on) { g_phases_on.store(on ? 1 : 0, std::memory_order_relaxed); }
void profile_phase_add(int which, double ms) {
  if (which < 0 || which >= kPhaseCount) return;
  std::lock_guard<std::mutex> lock(g_phase_mu);
  g_phase_ms[which] += ms;
  g_phase_n[which] += 1;
}
}  // namespace kdehip

extern "C" int kdehip_profile_phase_read(int which, double *total_ms, int64_t *count) {
  using namespace kdehip;
  if (which < 0 || which >= kPhaseCount) return set_error(KDEHIP_ERR_ARG, "kdehip_profile_phase_read: which in 0..2");
  std::lock_guard<std::mutex> lock(g_phase_mu);
  if (total_ms) *total_ms = g_phase_ms[which];
  if (count) *count = g_phase_n[which];
  g_phase_ms[which] = 0.0;
  g_phase_n[which] = 0;
  return KDEHIP_OK;
}

namespace kdehip { std::atomic<unsigned> g_peer_epoch{0}; }  // product.hip: the multi-GPU plans' peer-store verdicts

extern "C" void kdehip_clear_cache(void) {
  using namespace kdehip;
  g_peer_epoch.fetch_add(1, std::memory_order_relaxed);  // verdicts cached per raw pointer do not survive a cache reset
  kdehip::drain_pending();  // product.hip: plans of enqueue-only device products still waiting for their work
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  for (int d = 0; d < kMaxDevices; ++d) {
    std::vector<void *> blocks;
    {
      std::lock_guard<std::mutex> lock(g_dev[d].mu);
      for (auto &v : g_dev[d].free_blocks) { blocks.insert(blocks.end(), v.begin(), v.end()); v.clear(); }
      g_dev[d].cached_bytes = 0;
    }
    if (blocks.empty() || d >= n) continue;
    if (hipSetDevice(d) != hipSuccess) continue;
    for (void *p : blocks) (void)hipFree(p);
  }
  if (have_cur) (void)hipSetDevice(cur);
  std::vector<void *> blocks;
  {
    std::lock_guard<std::mutex> lock(g_pinned.mu);
    for (auto &v : g_pinned.free_blocks) { blocks.insert(blocks.end(), v.begin(), v.end()); v.clear(); }
    g_pinned.cached_bytes = 0;
  }
  for (void *p : blocks) (void)hipHostFree(p);
}
