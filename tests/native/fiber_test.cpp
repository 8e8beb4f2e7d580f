// biograph_amd/csrc/fiber.h on the CPU: a thread runs many fibers round by round the way csrc/coverage.cpp's workers do -- every
// live fiber is resumed until it parks or ends, then the "round" happens (here: the requests get their answers), and again.
// Built and run by tests/test_fiber.py.
#include "fiber.h"

#include <cassert>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

struct Request { int value; int answer; };

struct Thread {
  ucontext_t sched;
  std::vector<Request*> reqs;
  bool failed = false;
};

static int run_thread(int n_fibers, int seed, bool fail_midway) {
  Thread th;
  struct Island { bg::Fiber fiber; long sum = 0; int parks = 0; bool threw = false; };
  std::vector<std::unique_ptr<Island>> islands;
  for (int k = 0; k < n_fibers; ++k) {
    islands.emplace_back(new Island);
    Island* is = islands.back().get();
    const int rounds = (seed + k * 7) % 23;  // 0 .. 22 parks: some fibers end at once
    auto body = [is, rounds, k, &th] {
      try {
        // some stack use below the park point: the values must survive the switches
        volatile int local[64];
        for (int i = 0; i < 64; ++i) local[i] = k * 100 + i;
        std::string s = "fiber " + std::to_string(k);
        for (int r = 0; r < rounds; ++r) {
          Request q{k * 1000 + r, -1};
          th.reqs.push_back(&q);
          is->fiber.park();
          if (th.failed) throw std::runtime_error("round failed");
          assert(q.answer == q.value * 2 + 1);
          is->sum += q.answer;
          ++is->parks;
        }
        for (int i = 0; i < 64; ++i) assert(local[i] == k * 100 + i);
        assert(s == "fiber " + std::to_string(k));
      } catch (const std::runtime_error&) {
        is->threw = true;  // (caught inside the body: nothing unwinds past the fiber's first frame)
      }
    };
    const bool ok = is->fiber.start(&th.sched, size_t(256) << 10, body);
    assert(ok);
  }
  int n_rounds = 0;
  for (;;) {
    th.reqs.clear();
    bool any = false;
    for (auto& is : islands)
      if (!is->fiber.done) { any = true; is->fiber.resume(); }
    if (!any) break;
    if (th.reqs.empty()) continue;
    ++n_rounds;
    if (fail_midway && n_rounds == 5) th.failed = true;
    for (Request* q : th.reqs) q->answer = q->value * 2 + 1;  // the round
  }
  long total = 0;
  for (int k = 0; k < n_fibers; ++k) {
    const int rounds = (seed + k * 7) % 23;
    Island* is = islands[size_t(k)].get();
    assert(is->fiber.done);
    if (!fail_midway) {
      assert(is->parks == rounds && !is->threw);
      long want = 0;
      for (int r = 0; r < rounds; ++r) want += (k * 1000 + r) * 2 + 1;
      assert(is->sum == want);
    } else {
      assert(is->threw == (rounds >= 5));  // whoever was parked when the round failed threw when it woke
    }
    total += is->sum;
  }
  return n_rounds + int(total % 7);
}

int main() {
  // one thread, then eight side by side (fibers never leave their thread), then a failing round
  run_thread(1, 3, false);
  run_thread(700, 11, false);
  std::vector<std::thread> ts;
  for (int t = 0; t < 8; ++t) ts.emplace_back([t] { run_thread(200 + t * 37, t, false); });
  for (auto& t : ts) t.join();
  run_thread(300, 5, true);
  printf("ok\n");
  return 0;
}
