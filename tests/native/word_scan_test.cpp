// TEST ONLY: SeqsetView::next_forward_lt / next_backward_lt, eight entries a load (what the device compiles; here through
// BG_WORD_SCAN_HOST), against their definition -- first / last position with shared[pos] < v -- on arrays with long stretches
// above v (whole 64- and 4096-entry blocks the min index skips), at every alignment, for v over 0..255.
#define BG_WORD_SCAN_HOST 1
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../biograph_amd/csrc/bg_device_funcs.h"

using namespace bg;

int main() {
  std::mt19937_64 rng(7);
  long checks = 0;
  for (int round = 0; round < 60; ++round) {
    const uint64_t n = round < 8 ? uint64_t(1 + round * 3) : uint64_t(rng() % 30000 + 1);
    std::vector<uint8_t> sh(((n + 7) / 8) * 8 + 8, 0);
    const int style = round % 5;
    for (uint64_t i = 0; i < n; ++i) {
      uint8_t x;
      switch (style) {
        case 0: x = uint8_t(rng()); break;
        case 1: x = uint8_t(100 + rng() % 100); break;
        case 2: x = (rng() % 9000 == 0) ? uint8_t(rng() % 50) : uint8_t(200 + rng() % 56); break;  // blocks to skip
        case 3: x = uint8_t(rng() % 3 ? 255 : 0); break;
        default: x = uint8_t(128 + (i * 7) % 128); break;
      }
      sh[i] = x;
    }
    for (uint64_t i = n; i < sh.size(); ++i) sh[i] = 0;  // bytes past n must never be taken for a hit
    const uint64_t n1 = (n + 63) / 64, n2 = (n1 + 63) / 64;
    std::vector<uint8_t> m1(n1 + 1, 255), m2(n2 + 1, 255);
    for (uint64_t i = 0; i < n; ++i) m1[i / 64] = std::min(m1[i / 64], sh[i]);
    for (uint64_t i = 0; i < n1; ++i) m2[i / 64] = std::min(m2[i / 64], m1[i]);
    SeqsetView v;
    v.n = n;
    v.shared = sh.data();
    v.min1 = m1.data();
    v.min2 = m2.data();
    for (int q = 0; q < 4000; ++q) {
      const uint64_t pos = q < 64 && uint64_t(q) < n ? uint64_t(q) : rng() % n;
      const unsigned val = q % 7 == 0 ? unsigned(rng() % 256) : unsigned(sh[rng() % n]) + unsigned(rng() % 3);
      const unsigned vv = val > 255 ? 255 : val;
      uint64_t f = pos;
      while (f < n && sh[f] >= vv) ++f;
      uint64_t b = pos;
      while (b > 0 && sh[b] >= vv) --b;
      const uint64_t gf = v.next_forward_lt(pos, vv), gb = v.next_backward_lt(pos, vv);
      if (gf != f || gb != b) {
        printf("MISMATCH n %llu pos %llu v %u: forward %llu want %llu, backward %llu want %llu\n", (unsigned long long)n,
               (unsigned long long)pos, vv, (unsigned long long)gf, (unsigned long long)f, (unsigned long long)gb, (unsigned long long)b);
        return 1;
      }
      ++checks;
    }
  }
  // bytes_below itself, every byte value against every v
  for (unsigned x = 0; x < 256; ++x)
    for (unsigned vv = 0; vv < 256; ++vv) {
      const uint64_t w = 0x0101010101010101ull * x;
      const uint64_t want = x < vv ? 0x0101010101010101ull : 0ull;
      if (SeqsetView::bytes_below(w, vv) != want) { printf("bytes_below(%u, %u)\n", x, vv); return 1; }
    }
  printf("ok %ld checks\n", checks);
  return 0;
}
