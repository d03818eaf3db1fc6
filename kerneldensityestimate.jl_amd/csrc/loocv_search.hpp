// loocv_search.hpp -- the LOOCV bandwidth search of kde!(points) (src/KDE01.jl:3-27, src/CrossValidation.jl:15-120) as
// non-blocking pieces (evaluate.hip), for callers that keep several searches in flight (pack_device.hip).
#pragma once
#include <cstdint>

namespace kdehip {
// The same search for `nb` matrices of D x N that sit one behind the other in HBM (N <= kLoocvPrepMaxN), all nb * D
// marginals in the same launches; non-blocking pieces: begin enqueues the preparation and the first batch of rounds on
// `stream`, the caller synchronises the stream and polls (done, or the next batch enqueued), finish hands out nb * D
// bandwidths and nb evaluation counts.  kdehip_mul_device_batch keeps one search per distinct N in flight.
// (a launch indexes marginals in blockIdx.z, three probes each in the speculative rounds: at most this many per search)
constexpr int kLoocvMaxMarginals = 21000;
class LoocvSearch;
LoocvSearch *loocv_new();
void loocv_delete(LoocvSearch *s);  // (waits for the stream when the search was abandoned half-way)
int loocv_begin(LoocvSearch *s, int nb, int D, int64_t N, const double *d_points, void *stream);
int loocv_poll(LoocvSearch *s, bool *done);
int loocv_finish(LoocvSearch *s, double *bw_out, int32_t *nevals_out);
}  // namespace kdehip
