// Stress of biograph_amd/csrc/cli/fastalloc.h (the thread-caching operator new / delete of bgbinary): class arithmetic of the large
// blocks, eight threads allocating 1 byte .. 10 MB in three rounds, every block filled and checked, freed by another thread than
// the one that allocated it (the previous round's neighbour, after that thread has ended).  Built and run by tests/test_fastalloc.py.
#define FASTALLOC_DEFINE
#include "fastalloc.h"
#include <vector>
#include <thread>
#include <string>
#include <random>
#include <cstdio>
#include <cassert>
int main() {
  using namespace fastalloc;
  // class arithmetic
  for (size_t sz = 2049; sz <= kMaxLarge; sz += (sz < 100000 ? 1 : sz / 97)) {
    int lc = large_class(sz);
    assert(lc >= 0 && lc < kLargeClasses);
    assert(large_size(lc) >= sz);
    if (lc > 0) assert(large_size(lc - 1) < sz);
    assert(large_size(lc) * 4 <= sz * 5 + 64 || large_size(lc) - sz < (size_t(1) << 10));
  }
  std::vector<std::thread> th;
  std::vector<std::vector<std::pair<char*, size_t>>> hand(8);
  for (int round = 0; round < 3; ++round) {
    for (int t = 0; t < 8; ++t) th.emplace_back([&, t] {
      std::mt19937_64 rng(t * 77 + round);
      std::vector<std::pair<char*, size_t>> mine;
      // free what the previous round's neighbour left (cross-thread frees, from threads that ended)
      for (auto& x : hand[(t + 1) % 8]) { for (size_t i = 0; i < x.second; i += 511) assert(x.first[i] == char(x.second)); delete[] x.first; }
      hand[(t + 1) % 8].clear();
      for (int i = 0; i < 200000; ++i) {
        size_t sz = (rng() % 100 < 90) ? 1 + rng() % 300 : (rng() % 100 < 90 ? 2000 + rng() % 100000 : 1000000 + rng() % 9000000);
        char* p = new char[sz];
        for (size_t k = 0; k < sz; k += 511) p[k] = char(sz);
        mine.push_back({p, sz});
        if (mine.size() > 300) { size_t j = rng() % mine.size(); auto x = mine[j]; mine[j] = mine.back(); mine.pop_back();
          for (size_t k = 0; k < x.second; k += 511) assert(x.first[k] == char(x.second)); delete[] x.first; }
      }
      hand[t] = mine;
    });
    for (auto& x : th) x.join();
    th.clear();
  }
  for (auto& h : hand) for (auto& x : h) delete[] x.first;
  // many short-lived threads (what `bgbinary discovery` starts per scaffold): threads that are served from the lists
  // others handed over, or that only free, give their lists back when they end, and the rest of an ended thread's region
  // goes to the next thread -- the resident set must not grow with the number of threads
  if (enabled()) {
    auto rss_mib = [] {
      long pages = 0, res = 0;
      FILE* f = fopen("/proc/self/statm", "r");
      if (!f || fscanf(f, "%ld %ld", &pages, &res) != 2) res = 0;
      if (f) fclose(f);
      return double(res) * 4096.0 / (1 << 20);
    };
    std::vector<char*> left;  // blocks a thread allocates and the NEXT thread frees (a thread that only frees)
    auto wave = [&](int n) {
      for (int i = 0; i < n; ++i) {
        std::thread a([&] { for (int k = 0; k < 2000; ++k) { char* p = new char[1 + (k * 37) % 900]; p[0] = 1; left.push_back(p); } });
        a.join();
        std::thread b([&] { for (char* p : left) delete[] p; left.clear(); });
        b.join();
      }
    };
    wave(200);
    const double before = rss_mib();
    wave(3000);
    const double after = rss_mib();
    if (before > 0 && after - before > 64.0) { printf("resident set grew %.0f MiB over 6000 short-lived threads\n", after - before); return 1; }
  }
  printf("ok\n");
}
