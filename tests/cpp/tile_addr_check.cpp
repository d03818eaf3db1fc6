// Host check of the tile addressing (csrc/kdehip_internal.hpp TileAddr): every (row, field, lane) of a tile maps to its
// own element inside the tile's body, the pad elements stay free, fp32 row pairs are 8-byte aligned and adjacent, and a
// chunk of rows [r0, r0 + n) (r0 a multiple of 4) is one contiguous span.  Built and run by tests/test_tile_addr.py.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kdehip_internal.hpp"

using namespace kdehip;

template <int ELEM>
static int check(int F, int B) {
  using TA = TileAddrBytes<ELEM>;
  const long RS = TA::stride(F);
  const long body = TA::body(B, F);
  std::vector<int> used(static_cast<size_t>(body), 0);
  for (long r = 0; r < B; ++r)
    for (int f = 0; f < F; ++f)
      for (int ln = 0; ln < 64; ++ln) {
        const long off = TA::row(r, RS) + static_cast<long>(f) * TA::kField + static_cast<long>(ln) * TA::kLane;
        if (off < 0 || off >= body) { std::printf("out of the body: ELEM %d F %d B %d r %ld f %d ln %d\n", ELEM, F, B, r, f, ln); return 1; }
        if (used[off]++) { std::printf("two entries on one element: ELEM %d F %d B %d r %ld f %d ln %d\n", ELEM, F, B, r, f, ln); return 1; }
      }
  // the pad elements (one per row / two per pair) are nobody's
  for (long r = 0; r < B; ++r)
    if (used[TA::row(r, RS) + static_cast<long>(F) * TA::kField]) { std::printf("pad element in use: ELEM %d F %d r %ld\n", ELEM, F, r); return 1; }
  if (TA::kPaired) {
    if (RS % 2) { std::printf("pair stride not 8-byte aligned\n"); return 1; }
    for (long r = 0; r + 1 < B; r += 2)
      if (TA::row(r + 1, RS) != TA::row(r, RS) + 1) { std::printf("rows of a pair not adjacent\n"); return 1; }
    if (RS % 32 != 2) { std::printf("pair stride: the column walk would hit one bank\n"); return 1; }
  } else if (RS % 2 == 0) { std::printf("row stride even: the column walk would conflict\n"); return 1; }
  // rel() agrees with row() from an even base; a chunk is one span
  for (long r0 = 0; r0 < B; r0 += 4)
    for (int k = 0; k < 4 && r0 + k < B; ++k)
      if (TA::row(r0 + k, RS) != TA::row(r0, RS) + TA::rel(k, static_cast<int>(RS))) { std::printf("rel() disagrees with row()\n"); return 1; }
  for (long r0 = 0; r0 < B; r0 += 4)
    for (long n = 1; r0 + n <= B; ++n) {
      const long lo = TA::row(r0, RS), hi = lo + TA::span(n, RS);
      for (long r = r0; r < r0 + n; ++r) {
        const long last = TA::row(r, RS) + static_cast<long>(F - 1) * TA::kField + 63 * TA::kLane;
        if (TA::row(r, RS) < lo || last >= hi) { std::printf("row outside its chunk's span\n"); return 1; }
      }
      if (hi > body) { std::printf("chunk span beyond the body\n"); return 1; }
    }
  return 0;
}

int main() {
  int bad = 0;
  const int Fs[] = {2, 3, 7, 9, 13, 17};
  for (int F : Fs)
    for (int B = 1; B <= 21; ++B) bad |= check<8>(F, B) | check<4>(F, B);
  std::printf(bad ? "FAILED\n" : "tile addressing ok\n");
  return bad;
}
