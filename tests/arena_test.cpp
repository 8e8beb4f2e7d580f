// CPU unit test of the device-memory sub-allocator (biograph_amd/csrc/arena.h) over a malloc backend that can be
// told to run out of memory.  Built and run by tests/test_arena.py.
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "../biograph_amd/csrc/arena.h"

struct MallocBackend {
  size_t* budget;      // bytes the "driver" still has
  size_t* n_alloc;
  void* seg_alloc(size_t n) {
    if (n > *budget) return nullptr;
    void* p = aligned_alloc(4096, (n + 4095) / 4096 * 4096);
    if (p) { *budget -= n; ++*n_alloc; sizes()[p] = n; }
    return p;
  }
  void seg_free(void* p) { *budget += sizes()[p]; sizes().erase(p); free(p); }
  static std::map<void*, size_t>& sizes() { static std::map<void*, size_t> m; return m; }
};

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "arena_test: %s failed at line %d\n", #c, __LINE__); return 1; } } while (0)

int main() {
  size_t budget = size_t(3) << 30, n_alloc = 0;
  {
    bg::Arena<MallocBackend> a(MallocBackend{&budget, &n_alloc});
    // small requests share one 32 MiB segment; a freed chunk is reused and neighbours merge
    void* p1 = a.alloc(100);
    void* p2 = a.alloc(5000);
    void* p3 = a.alloc(777);
    CHECK(p1 && p2 && p3 && n_alloc == 1 && a.check());
    CHECK(a.live() == 512 + 5120 + 1024);
    a.release(p2);
    CHECK(a.check());
    void* p4 = a.alloc(4096);
    CHECK(p4 == p2 && a.check());  // best fit: the hole
    a.release(p1);
    a.release(p4);
    a.release(p3);
    CHECK(a.live() == 0 && a.check() && a.cached() == bg::Arena<MallocBackend>::kSmallSegment);
    // large requests: a segment of its own size rounded to 256 MiB, split for later requests
    void* b1 = a.alloc(size_t(300) << 20);
    CHECK(b1 && n_alloc == 2 && a.reserved() == (size_t(32) << 20) + (size_t(512) << 20));
    void* b2 = a.alloc(size_t(200) << 20);
    CHECK(b2 && n_alloc == 2);      // out of the rest of the 512 MiB segment
    void* b3 = a.alloc(size_t(100) << 20);
    CHECK(b3 && n_alloc == 3);      // 12 MiB are left there: a new segment
    a.release(b1);
    a.release(b2);                   // merges with b1's chunk and the 12 MiB tail: the whole segment is free again
    CHECK(a.check());
    void* b4 = a.alloc(size_t(500) << 20);
    CHECK(b4 == b1 && n_alloc == 3);
    CHECK(!a.release(reinterpret_cast<void*>(0x1234)));  // not ours
    CHECK(!a.release(b2));                               // already free
    a.release(b3);
    a.release(b4);
    CHECK(a.live() == 0 && a.check());
    // shrink: the chunk keeps its head, the tail is free at once and merges with a free neighbour
    {
      void* s1 = a.alloc(size_t(400) << 20);
      void* s2 = a.alloc(size_t(100) << 20);  // right behind s1 in the 512 MiB segment
      CHECK(s1 && s2 && a.live() == (size_t(500) << 20));
      CHECK(a.shrink(s1, size_t(100) << 20) && a.check() && a.live() == (size_t(200) << 20));
      CHECK(a.size_of(s1) == (size_t(100) << 20));
      void* s3 = a.alloc(size_t(300) << 20);  // exactly the freed tail
      CHECK(s3 == static_cast<char*>(s1) + (size_t(100) << 20) && a.check());
      CHECK(a.shrink(s3, (size_t(300) << 20) - 4096) && a.live() == (size_t(500) << 20));  // too small a rest: as it was
      a.release(s2);
      CHECK(a.shrink(s3, size_t(1) << 20) && a.check());  // the tail joins s2's free chunk and the segment's 12 MiB
      void* s4 = a.alloc(size_t(411) << 20);
      CHECK(s4 == static_cast<char*>(s3) + (size_t(1) << 20) && a.check());
      CHECK(!a.shrink(reinterpret_cast<void*>(0x1234), 1));
      a.release(s1); a.release(s3); a.release(s4);
      CHECK(a.live() == 0 && a.check());
    }
    // the driver runs out: wholly free segments go back, the request is retried (without the rounding if need be)
    const size_t before = a.reserved();
    budget = size_t(750) << 20;  // with the 800 MiB of free segments returned: room for 1500 MiB, not for the rounded 1536
    void* big = a.alloc(size_t(1500) << 20);
    CHECK(big && a.reserved() < before + (size_t(1500) << 20) && a.check());
    CHECK(a.alloc(size_t(4) << 30) == nullptr && a.check());
    a.release(big);
    a.trim();
    CHECK(a.reserved() == 0 && a.check());
    // the cap: a segment that becomes wholly free beyond it is returned at once
    a.set_cap(size_t(64) << 20);
    void* c1 = a.alloc(size_t(300) << 20);
    a.release(c1);
    CHECK(a.reserved() == 0);
    a.set_cap(~size_t(0));
    // random workload: invariants hold, no two live chunks overlap, contents survive
    std::mt19937_64 rng(7);
    budget = size_t(3) << 30;
    struct Live { unsigned char* p; size_t n; unsigned char tag; };
    std::vector<Live> live;
    for (int it = 0; it < 20000; ++it) {
      if (live.empty() || (rng() % 100 < 55 && live.size() < 400)) {
        size_t n = rng() % 3 == 0 ? rng() % (size_t(40) << 20) + 1 : rng() % 70000 + 1;
        unsigned char* p = static_cast<unsigned char*>(a.alloc(n));
        if (!p) continue;
        unsigned char tag = (unsigned char)(rng() & 0xff);
        p[0] = tag;
        p[n - 1] = tag;
        p[n / 2] = tag;
        live.push_back(Live{p, n, tag});
      } else {
        size_t k = rng() % live.size();
        Live l = live[k];
        CHECK(l.p[0] == l.tag && l.p[l.n - 1] == l.tag && l.p[l.n / 2] == l.tag);
        CHECK(a.size_of(l.p) >= l.n);
        CHECK(a.release(l.p));
        live[k] = live.back();
        live.pop_back();
      }
      if (it % 500 == 0) CHECK(a.check());
    }
    for (auto& l : live) CHECK(a.release(l.p));
    CHECK(a.live() == 0 && a.check());
    printf("arena_test ok: %zu driver allocations, peak %zu MiB, reserved %zu MiB\n", a.driver_allocs(), a.peak() >> 20,
           a.reserved() >> 20);
  }
  CHECK(budget == (size_t(3) << 30));  // everything went back
  return 0;
}
