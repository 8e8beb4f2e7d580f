// fiber.h — a function on a stack of its own that can park and be resumed by the thread that started it (ucontext).
//
// csrc/coverage.cpp runs the state machine of one island of variants per fiber: an island that needs bases walked on the device
// parks, the worker thread goes on with its next island, and when all of its islands are parked or finished the thread takes part
// in its group's round as ONE participant (coverage.cpp: IslandFiber).  A fiber never leaves its thread.  Whatever the body throws
// must be caught inside the body: nothing may unwind past the fiber's first frame.
#pragma once
#include <stdint.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <functional>

namespace bg {

struct Fiber {
  ucontext_t uc;             // the fiber while it is parked
  ucontext_t* sched = nullptr;  // the thread's loop: where park() and the end of the body go back to
  void* stack = nullptr;
  size_t stack_bytes = 0;
  bool done = false;
  std::function<void()> body;
  Fiber() {}
  Fiber(const Fiber&) = delete;
  Fiber& operator=(const Fiber&) = delete;
  ~Fiber() { if (stack) munmap(stack, stack_bytes); }

  // Called by the thread: sched is where resume() saves the thread's own context.  false: no memory for the stack.
  bool start(ucontext_t* sched_ctx, size_t stack_size, std::function<void()> f) {
    sched = sched_ctx;
    body = std::move(f);
    const size_t page = size_t(sysconf(_SC_PAGESIZE));
    stack_bytes = stack_size + page;  // a guard page below the stack
    stack = mmap(nullptr, stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (stack == MAP_FAILED) { stack = nullptr; return false; }
    (void)mprotect(stack, page, PROT_NONE);
    if (getcontext(&uc) != 0) return false;
    uc.uc_stack.ss_sp = static_cast<char*>(stack) + page;
    uc.uc_stack.ss_size = stack_size;
    uc.uc_link = sched;  // the body's return resumes the thread's loop
    const uint64_t self = reinterpret_cast<uint64_t>(this);
    makecontext(&uc, reinterpret_cast<void (*)()>(&Fiber::entry), 2, unsigned(self & 0xffffffffu), unsigned(self >> 32));
    return true;
  }
  // Thread side: run the fiber until it parks or ends.
  void resume() { swapcontext(sched, &uc); }
  // Fiber side: back to the thread; returns when the thread resumes this fiber.
  void park() { swapcontext(&uc, sched); }

 private:
  static void entry(unsigned lo, unsigned hi) {
    Fiber* f = reinterpret_cast<Fiber*>((uint64_t(hi) << 32) | uint64_t(lo));
    f->body();
    f->done = true;
  }
};

}  // namespace bg
