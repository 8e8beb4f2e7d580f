// fastalloc.h — a thread-caching replacement of the global operator new / delete for bgbinary.
//
// `bgbinary discovery` runs the reference's cost-ordered search (variants/tracer.cpp) as a host program: 16 tracer sessions,
// each with pool threads, and up to 128 coverage workers allocate and free hundreds of millions of small search states (paths,
// their sequences, pair lists) and, per work item, tables of kilobytes to tens of megabytes.  With glibc's malloc the process
// spent 56-130 s of SYSTEM time per 12-19 s of tracing on the chr20-shaped sample (futex waits on the arenas: 1.9 M voluntary
// context switches with the default arenas, 152 M with one; 7.6 M page faults of heaps that are unmapped when a work item ends
// and mapped again for the next) -- profiles/r04_discovery_sessions_cpu_time.txt; with the small blocks alone served here, a
// sixth of the command's wall time still sat in malloc's internals and munmap (the sampling profile of csrc/cli/cpuprof.h).
//
// Small blocks (<= 2 KiB, 16-byte classes): every thread carves them out of 8 MiB regions of its own and keeps the blocks it
// frees on per-class lists of its own: no lock, no system call.  Blocks migrate (the pool threads allocate a trace's state,
// the session thread frees it): a list that grows past a bound hands part of itself to a per-class global stack as one batch,
// and an empty list takes a batch from there before it carves.  Chunks are never returned (the process is a command).
// Large blocks (2 KiB .. 64 MiB, four classes per octave: at most a quarter of a block is slack): carved from the region (up to
// 1 MiB) or mapped on their own once, then kept --
// a thread keeps up to 32 MiB per class (at least one block), the rest goes to a per-class global stack; their pages are
// touched for the first time once, not once per work item.  Anything larger goes to malloc.  A thread that ends hands every
// list it holds to the global stacks.  A 16-byte header in front of every block names its class.  BG_FAST_ALLOC=0: plain
// malloc / free.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <atomic>
#include <new>

namespace fastalloc {

constexpr size_t kStep = 16, kMaxSmall = 2048;
constexpr int kClasses = int(kMaxSmall / kStep);
constexpr int kLargeLog0 = 11, kLargeLog1 = 26;            // large classes cover (2^11, 2^26] bytes
constexpr int kLargeClasses = (kLargeLog1 - kLargeLog0) * 4;
constexpr size_t kMaxLarge = size_t(1) << kLargeLog1;
constexpr size_t kLargeKeep = size_t(32) << 20;            // bytes per class a thread keeps for itself
constexpr size_t kHuge = size_t(2) << 20;                  // regions are 2 MiB aligned and asked to be huge pages
constexpr size_t kChunk = size_t(8) << 20;                 // a thread's region: small blocks and large ones up to 1 MiB are carved from it
constexpr size_t kCarveMax = size_t(1) << 20;              // larger blocks get a mapping of their own
constexpr uint32_t kBatch = 512;                           // small blocks handed over at a time
constexpr uint32_t kLargeTag = 0xffffffffu, kMagic = 0xb10c5eedu;

// 16 bytes: the user pointer stays 16-byte aligned.  `count` is only meaningful in the first block of a batch on a global stack.
struct Header { uint32_t cls, magic; uint64_t count; };
struct Block { Block* next; };
// a batch is a singly linked list of blocks; the stack links batches through the second word of their first block (class 0
// blocks hold 16 bytes: both words fit)
struct BatchHead { Block* next; BatchHead* next_batch; };

struct Global {
  std::atomic_flag lock = ATOMIC_FLAG_INIT;
  Block* batches = nullptr;
};
extern Global g_global[kClasses + kLargeClasses];
// what ended threads left of their regions: the next thread that needs a region takes one of these first (a command that
// starts fresh threads per scaffold would otherwise abandon most of 8 MiB per thread and never stop growing)
struct Remainder { Remainder* next; size_t bytes; };
struct RemainderPool {
  std::atomic_flag lock = ATOMIC_FLAG_INIT;
  Remainder* head = nullptr;
};
extern RemainderPool g_remainders;
constexpr size_t kMinRemainder = size_t(64) << 10;  // smaller leftovers are not worth a list node's look

struct Local {
  Block* free_list[kClasses + kLargeClasses];
  uint32_t count[kClasses + kLargeClasses];
  char* chunk;
  size_t left;
  bool registered;
};

inline bool enabled() {
  static const bool on = [] { const char* e = getenv("BG_FAST_ALLOC"); return !(e && e[0] == '0'); }();
  return on;
}

inline Local& local() {
  static thread_local Local l = {};
  return l;
}

// Memory from the system: 2 MiB aligned, with MADV_HUGEPAGE (BG_FAST_ALLOC_THP=0: without).  The tracer sessions touch
// thirteen gigabytes once each: as 4 KiB pages that is 3.2 M faults on one address space and, when the command ends, a
// second of the kernel taking them apart again; as huge pages both shrink five-hundredfold.  Never returned.
inline void* os_alloc(size_t bytes) {
  static const bool thp = [] { const char* e = getenv("BG_FAST_ALLOC_THP"); return !(e && e[0] == '0'); }();
  bytes = (bytes + kHuge - 1) & ~(kHuge - 1);
  char* raw = static_cast<char*>(mmap(nullptr, bytes + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  if (raw == MAP_FAILED) throw std::bad_alloc();
  char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(raw) + kHuge - 1) & ~uintptr_t(kHuge - 1));
  if (p > raw) munmap(raw, size_t(p - raw));
  if (p + bytes < raw + bytes + kHuge) munmap(p + bytes, size_t(raw + bytes + kHuge - (p + bytes)));
  if (thp) (void)madvise(p, bytes, MADV_HUGEPAGE);
  return p;
}

inline void lock(Global& g) { while (g.lock.test_and_set(std::memory_order_acquire)) { } }
inline void unlock(Global& g) { g.lock.clear(std::memory_order_release); }

inline void push_batch(int cls, Block* first, uint32_t count) {
  (reinterpret_cast<Header*>(first) - 1)->count = count;
  BatchHead* bh = reinterpret_cast<BatchHead*>(first);
  Global& g = g_global[cls];
  lock(g);
  bh->next_batch = reinterpret_cast<BatchHead*>(g.batches);
  g.batches = first;
  unlock(g);
}

inline Block* pop_batch(int cls, uint32_t* count) {
  Global& g = g_global[cls];
  if (!g.batches) return nullptr;  // (racy peek: a miss only means carving)
  lock(g);
  BatchHead* bh = reinterpret_cast<BatchHead*>(g.batches);
  if (bh) g.batches = reinterpret_cast<Block*>(bh->next_batch);
  unlock(g);
  if (!bh) return nullptr;
  Block* first = reinterpret_cast<Block*>(bh);
  *count = uint32_t((reinterpret_cast<Header*>(first) - 1)->count);
  return first;
}

// a thread that ends gives what it holds to the others
struct Flusher {
  ~Flusher() {
    Local& l = local();
    for (int c = 0; c < kClasses + kLargeClasses; ++c) {
      if (!l.free_list[c]) continue;
      Block* first = l.free_list[c];
      const uint32_t n = l.count[c];
      l.free_list[c] = nullptr;
      l.count[c] = 0;
      if (c < kClasses) { push_batch(c, first, n); continue; }
      while (first) { Block* next = first->next; first->next = nullptr; push_batch(c, first, 1); first = next; }
    }
    // the rest of the thread's region, 16-byte aligned as carve() left it
    if (l.left >= kMinRemainder) {
      Remainder* r = reinterpret_cast<Remainder*>(l.chunk);
      r->bytes = l.left;
      while (g_remainders.lock.test_and_set(std::memory_order_acquire)) { }
      r->next = g_remainders.head;
      g_remainders.head = r;
      g_remainders.lock.clear(std::memory_order_release);
    }
    l.chunk = nullptr;
    l.left = 0;
  }
};
inline void register_thread(Local& l) {
  l.registered = true;  // (first: registering may allocate)
  static thread_local Flusher f;
  (void)&f;
}

inline size_t large_size(int lc) { return (size_t(1) << (kLargeLog0 + lc / 4)) + size_t(lc % 4 + 1) * (size_t(1) << (kLargeLog0 + lc / 4 - 2)); }
inline int large_class(size_t size) {  // 2 KiB < size <= 64 MiB
  const int k = 63 - __builtin_clzll(uint64_t(size - 1));  // 2^k < size <= 2^(k+1)
  const size_t step = size_t(1) << (k - 2);
  const int idx = int((size - (size_t(1) << k) + step - 1) / step);  // 1..4
  return (k - kLargeLog0) * 4 + idx - 1;
}

inline void* carve(Local& l, size_t need) {  // need is a multiple of 16
  if (l.left < need) {
    // a region an ended thread did not use up, if one is large enough (the first that fits; the list is short)
    Remainder* got = nullptr;
    if (g_remainders.head) {  // (racy peek: a miss only means a fresh region)
      while (g_remainders.lock.test_and_set(std::memory_order_acquire)) { }
      for (Remainder** pp = &g_remainders.head; *pp; pp = &(*pp)->next)
        if ((*pp)->bytes >= need) { got = *pp; *pp = got->next; break; }
      g_remainders.lock.clear(std::memory_order_release);
    }
    if (got) {
      l.left = got->bytes;
      l.chunk = reinterpret_cast<char*>(got);
    } else {
      l.chunk = static_cast<char*>(os_alloc(kChunk));
      l.left = kChunk;
    }
  }
  void* p = l.chunk;
  l.chunk += need;
  l.left -= need;
  return p;
}

inline void* allocate(size_t size) {
  if (size == 0) size = 1;
  if (size > kMaxLarge || !enabled()) {
    void* raw = malloc(size + sizeof(Header));
    if (!raw) throw std::bad_alloc();
    Header* h = static_cast<Header*>(raw);
    h->cls = kLargeTag;
    h->magic = kMagic;
    return h + 1;
  }
  Local& l = local();
  if (!l.registered) register_thread(l);  // (on the first call, not only when carving: a thread served from the global
                                          // batches, or one that only frees, has lists to hand back when it ends too)
  if (size > kMaxSmall) {
    const int lc = large_class(size), cls = kClasses + lc;
    Block* b = l.free_list[cls];
    if (b) {
      l.free_list[cls] = b->next;
      --l.count[cls];
      return b;
    }
    uint32_t n = 0;
    if (Block* first = pop_batch(cls, &n)) return first;  // (large batches hold one block)
    const size_t need = large_size(lc) + sizeof(Header);
    Header* h = static_cast<Header*>(need > kCarveMax ? os_alloc(need) : carve(l, need));
    h->cls = uint32_t(cls);
    h->magic = kMagic;
    return h + 1;
  }
  const int cls = int((size + kStep - 1) / kStep) - 1;
  Block* b = l.free_list[cls];
  if (b) {
    l.free_list[cls] = b->next;
    --l.count[cls];
    return b;
  }
  // a batch somebody handed over?
  uint32_t n = 0;
  if (Block* first = pop_batch(cls, &n)) {
    l.free_list[cls] = first->next;
    l.count[cls] = n - 1;
    return first;
  }
  Header* h = static_cast<Header*>(carve(l, sizeof(Header) + size_t(cls + 1) * kStep));
  h->cls = uint32_t(cls);
  h->magic = kMagic;
  return h + 1;
}

inline void deallocate(void* p) {
  if (!p) return;
  Header* h = static_cast<Header*>(p) - 1;
  if (h->magic != kMagic) abort();  // not ours: a mismatched free must not go unnoticed
  if (h->cls == kLargeTag) { free(h); return; }
  const int cls = int(h->cls);
  Local& l = local();
  if (!l.registered && enabled()) register_thread(l);
  Block* b = static_cast<Block*>(p);
  if (cls >= kClasses) {
    if (l.count[cls] && size_t(l.count[cls] + 1) * large_size(cls - kClasses) > kLargeKeep) {
      b->next = nullptr;
      push_batch(cls, b, 1);
      return;
    }
    b->next = l.free_list[cls];
    l.free_list[cls] = b;
    ++l.count[cls];
    return;
  }
  b->next = l.free_list[cls];
  l.free_list[cls] = b;
  if (++l.count[cls] >= 4 * kBatch) {
    // hand the newest kBatch blocks over as one batch
    Block* first = l.free_list[cls];
    Block* last = first;
    for (uint32_t i = 1; i < kBatch; ++i) last = last->next;
    l.free_list[cls] = last->next;
    last->next = nullptr;
    l.count[cls] -= kBatch;
    push_batch(cls, first, kBatch);
  }
}

}  // namespace fastalloc

#ifdef FASTALLOC_DEFINE
namespace fastalloc {
Global g_global[kClasses + kLargeClasses];
RemainderPool g_remainders;
}
void* operator new(size_t n) { return fastalloc::allocate(n); }
void* operator new[](size_t n) { return fastalloc::allocate(n); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { try { return fastalloc::allocate(n); } catch (...) { return nullptr; } }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { try { return fastalloc::allocate(n); } catch (...) { return nullptr; } }
void operator delete(void* p) noexcept { fastalloc::deallocate(p); }
void operator delete[](void* p) noexcept { fastalloc::deallocate(p); }
void operator delete(void* p, size_t) noexcept { fastalloc::deallocate(p); }
void operator delete[](void* p, size_t) noexcept { fastalloc::deallocate(p); }
void operator delete(void* p, const std::nothrow_t&) noexcept { fastalloc::deallocate(p); }
void operator delete[](void* p, const std::nothrow_t&) noexcept { fastalloc::deallocate(p); }
#endif
