// arena.h — the device-memory sub-allocator of a bg_ctx.
//
// hipMalloc maps (and the driver clears) memory at about 30 us per MiB on this GPU: a 17 GiB buffer costs half
// a second.  A build asks for a few hundred buffers of a few dozen different sizes, phase after phase, so the
// allocator keeps what it got from the driver as *segments* and cuts every request out of them: best fit over
// the free chunks, split on allocation, coalesced with free neighbours on release.  (The first version cached
// freed blocks by exact size: the next phase's sizes never matched the last one's, a hundred gigabytes of
// dead blocks sat beside the live ones and every phase still went to the driver.)
//
// Small requests (< 1 MiB: counters, scan partials, per-object arrays of small inputs) come out of 32 MiB
// segments of their own so that they do not cut the big segments into pieces.  A large request that no free
// chunk holds gets a new segment of its own size rounded up to 256 MiB; when the driver refuses, wholly free
// segments are returned and the request is tried again.
//
// The backend (segment allocation) is a template parameter: the library uses hipMalloc / hipFree, the CPU
// unit test (tests/arena_test.cpp) malloc / free.  Not thread-safe: a context belongs to one host thread
// at a time, like its stream.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <map>
#include <set>
#include <utility>
#include <vector>

namespace bg {

// a backend may name the multiple its large segments come in (kLargeRound); 256 MiB otherwise
template <class B, class = void>
struct ArenaLargeRound { static constexpr size_t value = size_t(256) << 20; };
template <class B>
struct ArenaLargeRound<B, decltype(void(B::kLargeRound))> { static constexpr size_t value = B::kLargeRound; };

template <class Backend>
class Arena {
 public:
  static constexpr size_t kAlign = 512;
  static constexpr size_t kSmallLimit = size_t(1) << 20;    // requests below this use the small segments
  static constexpr size_t kSmallSegment = size_t(32) << 20;
  static constexpr size_t kLargeRound = ArenaLargeRound<Backend>::value;
  static constexpr size_t kMinSplit = size_t(1) << 20;      // a large chunk is not split for a smaller remainder

  explicit Arena(Backend b = Backend()) : backend_(b) {}
  ~Arena() { release_all(); }
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;

  // A segment the caller got from the backend itself (outside whatever lock guards this arena: mapping gigabytes takes the
  // driver hundreds of milliseconds) joins the large pool as one free chunk.
  void adopt(void* base, size_t seg_bytes) {
    segments_[uintptr_t(base)] = Segment{seg_bytes, false};
    reserved_ += seg_bytes;
    ++n_driver_allocs_;
    insert_free(large_, uintptr_t(base), seg_bytes, uintptr_t(base));
  }

  // nullptr: the driver is out of memory even after the free segments went back
  void* alloc(size_t bytes) {
    bytes = round_up(bytes ? bytes : 1, kAlign);
    const bool small = bytes < kSmallLimit;
    Pool& pool = small ? small_ : large_;
    auto it = pool.by_size.lower_bound(std::make_pair(bytes, uintptr_t(0)));
    if (it == pool.by_size.end()) {
      size_t seg = small ? kSmallSegment : round_up(bytes, kLargeRound);
      void* base = backend_.seg_alloc(seg);
      if (!base) {
        trim();
        base = backend_.seg_alloc(seg);
        if (!base && seg != bytes) {  // without the rounding
          seg = bytes;
          base = backend_.seg_alloc(seg);
        }
        if (!base) return nullptr;
      }
      segments_[uintptr_t(base)] = Segment{seg, small};
      reserved_ += seg;
      ++n_driver_allocs_;
      insert_free(pool, uintptr_t(base), seg, uintptr_t(base));
      it = pool.by_size.lower_bound(std::make_pair(bytes, uintptr_t(0)));
    }
    const size_t have = it->first;
    const uintptr_t addr = it->second;
    Chunk c = pool.chunks[addr];
    erase_free(pool, addr, have);
    size_t take = have;
    const size_t min_split = small ? kAlign : kMinSplit;
    if (have - bytes >= min_split) {
      take = bytes;
      insert_free(pool, addr + bytes, have - bytes, c.segment);
    }
    pool.chunks[addr] = Chunk{take, c.segment, false};
    live_ += take;
    if (live_ > peak_) peak_ = live_;
    return reinterpret_cast<void*>(addr);
  }

  // false: not a pointer of this arena
  bool release(void* p) {
    const uintptr_t addr = uintptr_t(p);
    Pool* pool = nullptr;
    for (Pool* q : {&large_, &small_}) {
      auto it = q->chunks.find(addr);
      if (it != q->chunks.end() && !it->second.free) { pool = q; break; }
    }
    if (!pool) return false;
    Chunk c = pool->chunks[addr];
    live_ -= c.size;
    uintptr_t lo = addr;
    size_t size = c.size;
    pool->chunks.erase(addr);
    // coalesce with the free neighbours of the same segment
    auto nx = pool->chunks.find(addr + c.size);
    if (nx != pool->chunks.end() && nx->second.free && nx->second.segment == c.segment) {
      const size_t ns = nx->second.size;
      erase_free(*pool, nx->first, ns);
      size += ns;
    }
    auto pv = pool->chunks.lower_bound(addr);
    if (pv != pool->chunks.begin()) {
      --pv;
      if (pv->second.free && pv->second.segment == c.segment && pv->first + pv->second.size == addr) {
        lo = pv->first;
        const size_t ps = pv->second.size;
        erase_free(*pool, lo, ps);
        size += ps;
      }
    }
    insert_free(*pool, lo, size, c.segment);
    // a wholly free segment goes back to the driver once the arena holds more than the cap allows
    if (lo == c.segment && size == segments_[c.segment].size && reserved_ - live_ > cap_) drop_segment(*pool, c.segment);
    return true;
  }

  // Keeps the first `bytes` of a live chunk and frees the rest of it (nothing happens when the rest is smaller than
  // what a chunk is split for).  false: not a live pointer of this arena.
  bool shrink(void* p, size_t bytes) {
    const uintptr_t addr = uintptr_t(p);
    Pool* pool = nullptr;
    for (Pool* q : {&large_, &small_}) {
      auto it = q->chunks.find(addr);
      if (it != q->chunks.end() && !it->second.free) { pool = q; break; }
    }
    if (!pool) return false;
    Chunk& c = pool->chunks[addr];
    bytes = round_up(bytes ? bytes : 1, kAlign);
    const size_t min_split = pool == &small_ ? kAlign : kMinSplit;
    if (bytes >= c.size || c.size - bytes < min_split) return true;
    uintptr_t tail = addr + bytes;
    size_t tail_size = c.size - bytes;
    const uintptr_t segment = c.segment;
    live_ -= tail_size;
    auto nx = pool->chunks.find(addr + c.size);
    c.size = bytes;
    if (nx != pool->chunks.end() && nx->second.free && nx->second.segment == segment) {
      const size_t ns = nx->second.size;
      erase_free(*pool, nx->first, ns);
      tail_size += ns;
    }
    insert_free(*pool, tail, tail_size, segment);
    return true;
  }

  // returns every wholly free segment to the driver
  void trim() {
    for (Pool* pool : {&large_, &small_}) {
      std::vector<uintptr_t> whole;
      for (auto& kv : pool->chunks)
        if (kv.second.free && kv.first == kv.second.segment && kv.second.size == segments_[kv.second.segment].size)
          whole.push_back(kv.first);
      for (uintptr_t s : whole) drop_segment(*pool, s);
    }
  }

  void set_cap(size_t cached_bytes_cap) { cap_ = cached_bytes_cap; }
  size_t live() const { return live_; }
  size_t cached() const { return reserved_ - live_; }  // held from the driver but not in use
  size_t reserved() const { return reserved_; }
  size_t peak() const { return peak_; }
  void reset_peak() { peak_ = live_; }
  size_t driver_allocs() const { return n_driver_allocs_; }
  size_t size_of(void* p) const {
    for (const Pool* q : {&large_, &small_}) {
      auto it = q->chunks.find(uintptr_t(p));
      if (it != q->chunks.end() && !it->second.free) return it->second.size;
    }
    return 0;
  }
  // consistency of the bookkeeping (tests): chunks tile their segments, free neighbours are merged, sizes add up
  bool check() const {
    size_t live = 0, total = 0;
    for (const Pool* pool : {&large_, &small_}) {
      size_t n_free = 0;
      const Chunk* prev = nullptr;
      uintptr_t prev_addr = 0;
      for (auto& kv : pool->chunks) {
        const Chunk& c = kv.second;
        auto sg = segments_.find(c.segment);
        if (sg == segments_.end() || kv.first < c.segment || kv.first + c.size > c.segment + sg->second.size) return false;
        if (prev && prev->segment == c.segment) {
          if (prev_addr + prev->size != kv.first) return false;       // a hole or an overlap
          if (prev->free && c.free) return false;                      // unmerged neighbours
        } else if (kv.first != c.segment) {
          return false;                                                // a segment must start with a chunk
        }
        if (c.free) {
          ++n_free;
          if (!pool->by_size.count(std::make_pair(c.size, kv.first))) return false;
        } else {
          live += c.size;
        }
        total += c.size;
        prev = &c;
        prev_addr = kv.first;
      }
      if (n_free != pool->by_size.size()) return false;
    }
    return live == live_ && total == reserved_;
  }

 private:
  struct Chunk { size_t size; uintptr_t segment; bool free; };
  struct Segment { size_t size; bool small; };
  struct Pool {
    std::map<uintptr_t, Chunk> chunks;                 // every chunk by address
    std::set<std::pair<size_t, uintptr_t>> by_size;    // the free ones by (size, address)
  };

  static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
  void insert_free(Pool& pool, uintptr_t addr, size_t size, uintptr_t segment) {
    pool.chunks[addr] = Chunk{size, segment, true};
    pool.by_size.insert(std::make_pair(size, addr));
  }
  void erase_free(Pool& pool, uintptr_t addr, size_t size) {
    pool.by_size.erase(std::make_pair(size, addr));
    pool.chunks.erase(addr);
  }
  void drop_segment(Pool& pool, uintptr_t seg) {
    auto sg = segments_.find(seg);
    erase_free(pool, seg, sg->second.size);
    reserved_ -= sg->second.size;
    backend_.seg_free(reinterpret_cast<void*>(seg));
    segments_.erase(sg);
  }
  void release_all() {
    for (auto& kv : segments_) backend_.seg_free(reinterpret_cast<void*>(kv.first));
    segments_.clear();
    large_ = Pool();
    small_ = Pool();
    live_ = reserved_ = 0;
  }

  Backend backend_;
  Pool large_, small_;
  std::map<uintptr_t, Segment> segments_;
  size_t live_ = 0, reserved_ = 0, peak_ = 0, cap_ = ~size_t(0), n_driver_allocs_ = 0;
};

}  // namespace bg
