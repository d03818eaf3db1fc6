// phase_timer.hpp -- kdehip_profile_phase_read: device time of the blocking entries either side of the product
// (LOOCV search, direct evaluation, GPU tree build), bracketed with HIP events on the stream their launches go to while
// kdehip_profile_sampler is on.  Diagnostics only (bench.py --frow); off: one relaxed load per phase.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace kdehip {

enum ProfilePhase : int { kPhaseLoocv = 0, kPhaseEvaluate = 1, kPhaseTreeBuild = 2, kPhaseCount = 3 };
bool profile_phases_on();                        // devmem.cpp (set by kdehip_profile_sampler)
void profile_phases_set(bool on);
void profile_phase_add(int which, double ms);

class PhaseTimer {
 public:
  PhaseTimer(int which, hipStream_t st) : which_(which), st_(st) {
    if (!profile_phases_on()) return;
    if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) { drop(); return; }
    if (hipEventRecord(a_, st_) != hipSuccess) drop();
  }
  PhaseTimer(const PhaseTimer &) = delete;
  PhaseTimer &operator=(const PhaseTimer &) = delete;
  void stop() { if (a_ && hipEventRecord(b_, st_) == hipSuccess) stopped_ = true; }
  // once the stream has been synchronised by the caller
  void collect() {
    float ms = 0.0f;
    if (a_ && stopped_ && hipEventElapsedTime(&ms, a_, b_) == hipSuccess) profile_phase_add(which_, ms);
    drop();
  }
  ~PhaseTimer() { drop(); }

 private:
  void drop() {
    if (a_) (void)hipEventDestroy(a_);
    if (b_) (void)hipEventDestroy(b_);
    a_ = b_ = nullptr;
  }
  int which_;
  hipStream_t st_;
  hipEvent_t a_ = nullptr, b_ = nullptr;
  bool stopped_ = false;
};

}  // namespace kdehip
