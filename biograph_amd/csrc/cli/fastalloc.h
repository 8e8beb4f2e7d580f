// fastalloc.h — a thread-caching replacement of the global operator new / delete for bgbinary.
//
// `bgbinary discovery` runs the reference's cost-ordered search (variants/tracer.cpp) as a host program: 24 tracer sessions,
// each with pool threads, allocate and free hundreds of millions of small search states (paths, their sequences, pair lists).
// With glibc's malloc the process spent 56-130 s of SYSTEM time per 12-19 s of tracing on the chr20-shaped sample (futex waits
// on the arenas: 1.9 M voluntary context switches with the default arenas, 152 M with one; 7.6 M page faults of heaps that are
// unmapped when a work item ends and mapped again for the next) -- profiles/r04_discovery_sessions_cpu_time.txt.
//
// Here every thread carves small blocks (<= 2 KiB, 16-byte classes) out of 1 MiB chunks of its own and keeps the blocks it
// frees on per-class lists of its own: no lock, no system call.  Blocks migrate (the pool threads allocate a trace's state,
// the session thread frees it): a list that grows past a bound hands half of itself to a per-class global stack as one batch,
// and an empty list takes a batch from there before it carves.  Chunks are never returned (the process is a command).  Larger
// requests go to malloc.  A 16-byte header in front of every block names its class.  BG_FAST_ALLOC=0: plain malloc / free.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <new>

namespace fastalloc {

constexpr size_t kStep = 16, kMaxSmall = 2048;
constexpr int kClasses = int(kMaxSmall / kStep);
constexpr size_t kChunk = size_t(1) << 20;
constexpr uint32_t kBatch = 512;            // blocks handed over at a time
constexpr uint32_t kLargeTag = 0xffffffffu, kMagic = 0xb10c5eedu;

struct Header { uint32_t cls, magic; uint64_t pad; };  // 16 bytes: the user pointer stays 16-byte aligned
struct Block { Block* next; };
struct Batch { Batch* next_batch; Block* blocks; };     // a batch lives in its first block's payload (>= 16 bytes)

struct Global {
  std::atomic_flag lock = ATOMIC_FLAG_INIT;
  Block* batches = nullptr;  // stack of batches: each entry is a block whose `next` chains the batch, linked through `link`
};
// per class: a lock-protected stack of batches.  A batch is a singly linked list of kBatch blocks; the stack links batches
// through the second word of their first block.
struct BatchHead { Block* next; BatchHead* next_batch; };
extern Global g_global[kClasses];

struct Local {
  Block* free_list[kClasses];
  uint32_t count[kClasses];
  char* chunk;
  size_t left;
};

inline bool enabled() {
  static const bool on = [] { const char* e = getenv("BG_FAST_ALLOC"); return !(e && e[0] == '0'); }();
  return on;
}

inline Local& local() {
  static thread_local Local l = {};
  return l;
}

inline void lock(Global& g) { while (g.lock.test_and_set(std::memory_order_acquire)) { } }
inline void unlock(Global& g) { g.lock.clear(std::memory_order_release); }

inline void* allocate(size_t size) {
  if (size == 0) size = 1;
  if (size > kMaxSmall || !enabled()) {
    void* raw = malloc(size + sizeof(Header));
    if (!raw) throw std::bad_alloc();
    Header* h = static_cast<Header*>(raw);
    h->cls = kLargeTag;
    h->magic = kMagic;
    return h + 1;
  }
  const int cls = int((size + kStep - 1) / kStep) - 1;
  Local& l = local();
  Block* b = l.free_list[cls];
  if (b) {
    l.free_list[cls] = b->next;
    --l.count[cls];
    return b;
  }
  // a batch somebody handed over?
  Global& g = g_global[cls];
  if (g.batches) {  // (racy peek: a miss only means carving)
    lock(g);
    BatchHead* bh = reinterpret_cast<BatchHead*>(g.batches);
    if (bh) g.batches = reinterpret_cast<Block*>(bh->next_batch);
    unlock(g);
    if (bh) {
      Block* first = reinterpret_cast<Block*>(bh);
      l.free_list[cls] = first->next;
      l.count[cls] = kBatch - 1;
      return first;
    }
  }
  const size_t need = sizeof(Header) + size_t(cls + 1) * kStep;
  if (l.left < need) {
    l.chunk = static_cast<char*>(malloc(kChunk));
    if (!l.chunk) throw std::bad_alloc();
    l.left = kChunk;
  }
  Header* h = reinterpret_cast<Header*>(l.chunk);
  l.chunk += need;
  l.left -= need;
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
  Block* b = static_cast<Block*>(p);
  b->next = l.free_list[cls];
  l.free_list[cls] = b;
  if (++l.count[cls] >= 4 * kBatch) {
    // hand the newest kBatch blocks over as one batch (class 0 blocks hold 16 bytes: next + next_batch fit)
    Block* first = l.free_list[cls];
    Block* last = first;
    for (uint32_t i = 1; i < kBatch; ++i) last = last->next;
    l.free_list[cls] = last->next;
    last->next = nullptr;
    l.count[cls] -= kBatch;
    BatchHead* bh = reinterpret_cast<BatchHead*>(first);
    Global& g = g_global[cls];
    lock(g);
    bh->next_batch = reinterpret_cast<BatchHead*>(g.batches);
    g.batches = first;
    unlock(g);
  }
}

}  // namespace fastalloc

#ifdef FASTALLOC_DEFINE
namespace fastalloc {
Global g_global[kClasses];
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
