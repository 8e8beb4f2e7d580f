// biograph_amd/csrc/cli/cpuprof.h on the CPU: two threads burn time in one named function with the profiler armed; the profile
// must be written and tools/cpuprof_report.py must find most samples there.  Built and run by tests/test_cpuprof.py.
#include "cpuprof.h"

#include <cstdio>
#include <thread>

extern "C" __attribute__((noinline)) unsigned long long spin_here_for_the_profiler(unsigned long long n) {
  unsigned long long x = 88172645463325252ull;
  for (unsigned long long i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; }
  return x;
}

int main() {
  cpuprof::start();  // armed by BG_CPU_PROFILE=FILE
  unsigned long long sink = 0;
  std::thread other([&] { sink ^= spin_here_for_the_profiler(600000000ull); });
  sink ^= spin_here_for_the_profiler(600000000ull);
  other.join();
  cpuprof::stop_and_write();
  printf("done %llu\n", sink & 1);
  return 0;
}
