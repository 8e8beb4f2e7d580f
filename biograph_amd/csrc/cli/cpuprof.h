// cpuprof.h — BG_CPU_PROFILE=FILE: where the CPU time of a bgbinary command goes, without any tool on the box.
//
// ITIMER_PROF fires every millisecond of CPU time the process burns and the kernel delivers SIGPROF to a thread that is
// running: the handler stores the interrupted program counter, the thread and the time (and the return addresses of up to
// five frames found through the frame-pointer chain, where there is one) into a preallocated ring.  In practice a few hundred
// samples a second arrive, each on ONE of the running threads: shares of samples are shares of wall time.  At the end the
// samples and /proc/self/maps go to FILE; tools/cpuprof_report.py turns them into time per module, per function (nm) and,
// for a library built with -gline-tables-only, per source line (llvm-symbolizer).  tests/test_cpuprof.py runs both on the
// CPU.  Not part of the product path: nothing is installed unless the variable is set.
#pragma once
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>
#include <ucontext.h>

#include <atomic>

namespace cpuprof {

constexpr size_t kMaxSamples = size_t(1) << 21, kFrames = 6;
struct Sample { uint64_t pc[kFrames]; uint64_t tid, ns; };
inline Sample*& ring() { static Sample* r = nullptr; return r; }
inline std::atomic<uint64_t>& cursor() { static std::atomic<uint64_t> c(0); return c; }
inline const char*& path() { static const char* p = nullptr; return p; }

inline void on_prof(int, siginfo_t*, void* uc_) {
  Sample* r = ring();
  if (!r) return;
  const uint64_t i = cursor().fetch_add(1, std::memory_order_relaxed);
  if (i >= kMaxSamples) return;
  const ucontext_t* uc = static_cast<const ucontext_t*>(uc_);
  Sample& s = r[i];
  s.pc[0] = uint64_t(uc->uc_mcontext.gregs[REG_RIP]);
  s.tid = uint64_t(syscall(SYS_gettid));
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  s.ns = uint64_t(ts.tv_sec) * 1000000000ull + uint64_t(ts.tv_nsec);
  uint64_t fp = uint64_t(uc->uc_mcontext.gregs[REG_RBP]);
  const uint64_t sp = uint64_t(uc->uc_mcontext.gregs[REG_RSP]);
  for (size_t k = 1; k < kFrames; ++k) {
    s.pc[k] = 0;
    // a frame pointer that is a plausible stack address just above the stack pointer; anything else ends the walk
    if (fp < sp || fp - sp > (uint64_t(8) << 20) || (fp & 7)) { fp = 0; continue; }
    // read through the kernel: a thread may be on a 512 KiB fiber stack (csrc/fiber.h) with nothing mapped above it, and a
    // stale rbp must end the walk, not the process
    uint64_t f[2] = {0, 0};
    iovec lv{f, sizeof f}, rv{reinterpret_cast<void*>(fp), sizeof f};
    if (process_vm_readv(getpid(), &lv, 1, &rv, 1, 0) != ssize_t(sizeof f)) { fp = 0; continue; }
    s.pc[k] = f[1];
    const uint64_t next = f[0];
    fp = next > fp ? next : 0;
  }
}

inline void start() {
  const char* p = getenv("BG_CPU_PROFILE");
  if (!p || !*p) return;
  path() = p;
  ring() = static_cast<Sample*>(calloc(kMaxSamples, sizeof(Sample)));
  if (!ring()) return;
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_prof;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, nullptr);
  itimerval tv;
  tv.it_interval.tv_sec = 0; tv.it_interval.tv_usec = 1000;
  tv.it_value = tv.it_interval;
  setitimer(ITIMER_PROF, &tv, nullptr);
}

inline void stop_and_write() {
  if (!ring() || !path()) return;
  itimerval tv;
  memset(&tv, 0, sizeof tv);
  setitimer(ITIMER_PROF, &tv, nullptr);
  FILE* f = fopen(path(), "w");
  if (!f) return;
  FILE* m = fopen("/proc/self/maps", "r");
  if (m) {
    char line[1024];
    while (fgets(line, sizeof line, m))
      if (strstr(line, " r-xp ") || strstr(line, " r-x")) fprintf(f, "M %s", line);
    fclose(m);
  }
  uint64_t n = cursor().load();
  if (n > kMaxSamples) n = kMaxSamples;
  for (uint64_t i = 0; i < n; ++i) {
    fprintf(f, "S %llu %llu", (unsigned long long)ring()[i].tid, (unsigned long long)ring()[i].ns);
    for (size_t k = 0; k < kFrames && ring()[i].pc[k]; ++k) fprintf(f, " %llx", (unsigned long long)ring()[i].pc[k]);
    fprintf(f, "\n");
  }
  fclose(f);
}

}  // namespace cpuprof
