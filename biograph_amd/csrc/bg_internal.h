// bg_internal.h — host-side plumbing shared by the translation units of libbgseq.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/bgseq.h"
#include <atomic>
#include <mutex>

#include "arena.h"
#include "bg_device_funcs.h"

namespace bg {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define BG_HIP(expr)                                                  \
  do {                                                                \
    hipError_t e_ = (expr);                                           \
    if (e_ != hipSuccess) return ::bg::hip_fail(e_, #expr, __FILE__, __LINE__); \
  } while (0)
#define BG_TRY(expr)           \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != BG_OK) return rc_; \
  } while (0)
#define BG_LAUNCH_CHECK() BG_HIP(hipGetLastError())

struct TimingRec {
  std::string name;
  hipEvent_t start, stop;
  uint64_t bytes;
};

}  // namespace bg

// The opaque C types live in the global namespace (they are named in the extern "C" header).
struct bg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;  // bg_ctx_create_own_stream
  bool profiling = false;
  // device memory: segments from the driver (hipMalloc), sub-allocated best-fit with coalescing (arena.h).  Large segments
  // come in multiples of 1 GiB.  Measured (tools/map_rate.hip, tools/e2e_profile.py): hipMalloc of gigabytes costs anything
  // between nothing and 75 us per MiB depending on what the process and the device have been through (a 27 GB reservation:
  // 1.3 - 2.2 s; a cold `create` right after another process gave back 50 GB: 0.7 - 0.95 s inside the build), hipMemCreate +
  // hipMemMap in 1 GiB pieces 1.6 us per MiB (32 GiB: 0.05 s) -- and in 256 MiB pieces 60.  BG_ARENA_VMM=1 takes the segments
  // that way too (see seg_alloc for why it is not the default); bg_ctx_reserve always does.
  struct HipSegments {
    static constexpr size_t kLargeRound = size_t(1) << 30;
    void* seg_alloc(size_t n);
    void seg_free(void* p);
  };
  using DeviceArena = bg::Arena<HipSegments>;
  DeviceArena own_arena;
  std::mutex own_mutex;  // guards own_arena (bg_ctx_reserve may run on another thread); the shared arena has the global one
  std::mutex& arena_mutex();
  DeviceArena* arena = &own_arena;  // the process-wide arena of the device when torch shares it (bg_torch_malloc)
  std::vector<bg::TimingRec> recs;
  std::vector<hipEvent_t> event_pool;
  // two pinned host staging buffers for large host <-> device copies (allocated at first use: pinning 128 MiB costs tens
  // of milliseconds; released with the context)
  void* pinned[2] = {nullptr, nullptr};
  static constexpr size_t kPinnedChunk = size_t(64) << 20;
  int pinned_buffers();

  int alloc(size_t bytes, void** out);
  void release(void* p);
  void shrink(void* p, size_t bytes);  // the chunk keeps its first `bytes`, the rest goes back to the arena
  int trim();
  // profiling bracket around one kernel family
  void tic(const char* name);
  void toc(uint64_t bytes);
  void clear_timings();
  hipEvent_t get_event();
};

namespace bg {

// Large copies through the context's pinned buffers, double-buffered, the host side of every piece spread over threads
// (csrc/fastq.hip).  threads <= 0: hardware_concurrency (at most 32).
int staged_download(bg_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes, int threads);

// Host <-> device copies ordered on the context's stream (a null-stream hipMemcpy does not wait for a non-blocking
// stream such as bg_ctx_create_own_stream's or a torch side stream): the copy is queued behind the kernels launched so
// far and the call returns when it has landed.
inline hipError_t copy_d2h(bg_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (!bytes) return hipSuccess;
  hipError_t e = hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}
inline hipError_t copy_h2d(bg_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (!bytes) return hipSuccess;
  hipError_t e = hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream);
  return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}

// RAII device buffer drawn from the context cache.
template <class T>
struct DevBuf {
  bg_ctx* ctx = nullptr;
  T* p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : ctx(o.ctx), p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { reset(); ctx = o.ctx; p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { reset(); }
  int alloc(bg_ctx* c, size_t count) {
    reset();
    ctx = c;
    n = count;
    void* q = nullptr;
    int rc = c->alloc((count ? count : 1) * sizeof(T), &q);
    p = static_cast<T*>(q);
    return rc;
  }
  void reset() {
    if (p && ctx) ctx->release(p);
    p = nullptr;
    n = 0;
  }
  T* release_ownership() { T* q = p; p = nullptr; n = 0; return q; }
  // keeps the first `count` elements (what follows them goes back to the arena at once: see bg_ctx::release)
  void shrink(size_t count) {
    if (p && ctx && count < n) ctx->shrink(p, (count ? count : 1) * sizeof(T));
    if (count < n) n = count;
  }
  size_t bytes() const { return n * sizeof(T); }
};

inline unsigned cdiv(uint64_t a, uint64_t b) { return unsigned((a + b - 1) / b); }

// ---- primitives (prims.hip)
// Exclusive scan of n uint32 counters into uint64 offsets (out may not alias in).  If d_total is
// non-null the grand total is stored there (device).
int exclusive_scan_u32_to_u64(bg_ctx* ctx, const uint32_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* d_total);
// Same with 32-bit outputs (total must fit).
int exclusive_scan_u32(bg_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint32_t* d_total);
// Stable LSD radix sort of 64-bit keys (optionally carrying 32-bit values) on key bits
// [bit_lo, bit_hi).  keys_a/vals_a hold the input; *_b are scratch of the same size.  On return
// *in_b tells which buffer holds the result.  vals_* may be null (keys only).
// flags: RS_LOW_ENTROPY = keys take few distinct values in the sorted bits (no census, ballot-aggregated
// histogram).
constexpr int RS_LOW_ENTROPY = 1;
int radix_sort_u64(bg_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                   uint64_t n, int bit_lo, int bit_hi, bool* in_b, int flags = 0);

// Stable partition of keys into 65536 buckets by the top 16 bits of key * kPartitionMult.
constexpr uint64_t kPartitionMult = 0x9E3779B97F4A7C15ull;
int radix_partition_hash16(bg_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint64_t n, bool* in_b);
int radix_partition_hash16_from(bg_ctx* ctx, const uint64_t* in, uint64_t* scratch_a, uint64_t* scratch_b, uint64_t n);
int radix_partition_bounds(bg_ctx* ctx, const uint64_t* keys, uint64_t n, int bits, uint64_t* d_bounds);
// One more level: keys grouped by 16 hash bits (kin) -> by 16 + extra_bits (kout), extra_bits in 1..8.
int radix_partition_hash_refine(bg_ctx* ctx, uint64_t* kin, uint64_t* kout, uint64_t n, int extra_bits);
// Same for the k-mer instances of packed reads, generated inside pass 1 (n = their number); result in keys_b.
// First pass alone: keys grouped by the top hash byte, h_starts[257] = bucket starts.
int radix_partition_first_reads(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n, uint64_t* keys_out,
                                uint64_t* h_starts);
int radix_partition_hash16_reads(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n, uint64_t* keys_a,
                                 uint64_t* keys_b);

// ---- device-side object layouts
}  // namespace bg

struct bg_reads {
  bg_ctx* ctx = nullptr;
  uint64_t n = 0;
  int W = 0;                 // words per read (fixed stride)
  uint32_t max_len = 0;
  uint64_t total_bases = 0;
  bg::DevBuf<uint64_t> words;   // n*W
  bg::DevBuf<uint32_t> nmask;   // n*W   (all zero for corrected reads)
  bool any_n = true;            // false: nmask is known to be all zero (nobody needs to read it to find out)
  bg::DevBuf<uint16_t> len;     // n
};

struct bg_kmerset {
  bg_ctx* ctx = nullptr;
  int k = 0;
  uint64_t size = 0, distinct = 0, instances = 0;
  bool sorted = true;            // kmers ascending (the LDS counting path leaves them unordered)
  bg::DevBuf<uint64_t> kmers;    // canonical kept k-mers
  bg::DevBuf<uint32_t> counts;   // total count per kept k-mer
  bg::DevBuf<uint64_t> slots;    // hash table (1 << log2cap)
  int log2cap = 0;
  // Window presence of the reads the set was counted from (4 x 64 bits per read: bit j = window j holds no N
  // and its canonical k-mer is kept): a by-product of the super-k-mer count that spares bg_correct_reads the
  // look-up of every window.  Valid for the reads object identified by present_of / present_n only.
  bg::DevBuf<uint64_t> present;
  const void* present_of = nullptr;
  uint64_t present_n = 0;
  int present_stride = 4;  // words per read: ceil(windows of the longest read / 64) — 150-base reads take 2, and
                           // a 206 MB bitmap stays in the Infinity Cache under the count's atomics where 412 MB did not
  // records (2 words each) + origins of the buckets the LDS count gave up on: their presence bits are set
  // from the finished hash set (kmer_skm_finish_present)
  bg::DevBuf<uint64_t> pend_rec, pend_org;
  uint64_t pend_n = 0;
};

struct bg_seqset {
  bg_ctx* ctx = nullptr;
  uint64_t n = 0;
  uint32_t max_len = 0;
  int W = 0;                        // words per entry key (0 when opened from a file)
  bg::DevBuf<uint64_t> keys;        // n*W entry sequences (only after a build)
  bg::DevBuf<uint64_t> pre2;        // n*2: the first two words of every entry, dense (search accelerator of the build)
  bg::DevBuf<uint8_t> sizes, shared;
  bg::DevBuf<uint64_t> bits[4], subaccum[4], accum[4];
  bg::DevBuf<uint64_t> rank64[4];   // one 64-byte rank block per 384 entries (SeqsetView::count): what the walk kernels read
  bg::DevBuf<uint32_t> rank4;       // one 64-byte rank block per 96 entries, all four bases in it (SeqsetView::count)
  bg::DevBuf<uint64_t> rank4_super; // base counts of every group of 65536 such blocks, four per group
  bg::DevBuf<uint64_t> sel256[4];   // the entry that carries every 256th set prev bit of a base (pop_front's select), + a sentinel n
  // walk blocks (SeqsetView::wblk): ONE 128-byte line per 40 entries holding everything a push_front_drop reads of them -- the
  // four bases' prev bits with their running counts, the entries' sizes, their shared.  Made by the first walk over the seqset
  // (seqset_ensure_walk_blocks: not part of a build), 3.2 bytes an entry.
  bg::DevBuf<uint64_t> wblk, wblk_super;
  std::atomic<int> wblk_state{0};   // 0: not made yet; 1: there; 2: not to be made (switched off, too many entries, no memory)
  bg::DevBuf<uint8_t> min1, min2;
  uint64_t fixed[5] = {0, 0, 0, 0, 0};
  std::string uuid;
  // a shard of a sharded build whose records carried the flags of k_emit_records (bg_records_set_meta): prev bits already set
  // where the records met (emask), the entries whose tail still has to travel (need), and the look-ups that decide a few more
  // (cond: rows of W + 1 words, base + record; n_cond of them, room for 4 more: bg_seqset_fix_boundary)
  bg::DevBuf<uint8_t> emask, need;
  bg::DevBuf<uint64_t> cond;
  uint64_t n_cond = 0;
  // a build from items (items_build.hip): no key table yet -- every entry's first word and record (oriented read << 24 |
  // offset << 16 | ...), two words an entry, and the oriented reads the records point into; seqset_ensure_keys makes keys / pre2 from them the first time somebody needs them
  bg::DevBuf<uint64_t> erec, oreads;
  bg::DevBuf<uint32_t> edir;  // ... and the prefix directory over the entries' first words (items_entries.h: ItEntries), while they are there
  int edir_bases = 0;
  std::mutex keys_mutex;  // seqset_ensure_keys: the first keyed access writes an object the API hands out as const
};


struct bg_records {
  bg_ctx* ctx = nullptr;
  uint64_t n = 0;
  int W = 0;
  int RS = 0;                 // words from one row of rec to the next: 0 = W; W + 1 = the length sits behind the key words
  bool has_meta = false;      // ... and above the length's byte the flags kRecHasPred / kRecLast (bg_records_set_meta)
  bg::DevBuf<uint64_t> rec;   // n rows of zero-padded key words
  bg::DevBuf<uint8_t> rlen;   // n (not used when RS = W + 1)
};

struct bg_readmap {
  bg_ctx* ctx = nullptr;
  uint64_t n_reads = 0, n_entries = 0;
  uint32_t max_read_len = 0;
  std::string seqset_uuid;
  uint32_t min_read_len = 0;    // readmap::min_read_len()
  bool pair_mode = false;       // legacy file without a mate loop table: mate_loop_ptr holds mate_pair_ptr (readmap.cpp:48-51)
  bg::DevBuf<uint64_t> read_len, mate_loop_ptr, is_forward;  // one u64 per read id (packed on demand)
  bg::DevBuf<uint64_t> entry_id;                              // seqset entry of every read id (sparse_multi::reverse_lookup); 64-bit:
                                                              // a whole-human seqset holds > 2^32 entries (make_readmap.h:56-90 packs 37 bits)
  bg::DevBuf<uint64_t> s2m_bits, s2m_sub, s2m_acc, d2m_bits, d2m_sub, d2m_acc;
};

namespace bg {
struct CoverStore {  // device buffers behind a CoverIndex (bg_device_funcs.h)
  DevBuf<uint64_t> skeys, slots;
  DevBuf<uint32_t> soid;
  CoverIndex ci;
  const uint64_t* owords = nullptr;
};
}  // namespace bg

struct bg_cover {
  bg_ctx* ctx = nullptr;
  uint64_t n = 0;
  int W = 0, K = 0;
  bg::DevBuf<uint64_t> owords;
  bg::DevBuf<uint8_t> olen;
  bg::CoverStore cs;
};

struct bg_cover_query {
  bg_ctx* ctx = nullptr;
  uint64_t n = 0, n_pending = 0;
  int W = 0, K = 0;
  int j_lo = 0, span = 0;  // the round in flight
  bg::DevBuf<uint64_t> owords;
  bg::DevBuf<uint8_t> olen;
  bg::DevBuf<uint32_t> pending;
  // filtered rounds (bg_cover_query_scan): per oriented read the next suffix to look at and, once settled, the first
  // covered suffix (= its length when none is); per slot of the round in flight the suffixes that went out
  bg::DevBuf<uint8_t> cursor, hit_j;
  bg::DevBuf<uint16_t> end_j;  // 0xffff = not settled (a read may be 255 bases long)
  uint32_t max_hits = 0;
};

namespace bg {
// prims.hip: one arena per device shared by torch's allocator hook and every context created after the hook was
// installed (bg_torch_allocator_enable); g_arena_mutex guards every arena call in that mode
extern std::mutex g_arena_mutex;
extern std::mutex g_vmm_mutex;                 // segments built through the virtual-memory API (bg_ctx_reserve)
struct VmmSegment { size_t reserved, mapped, piece; };  // address range, bytes of it with memory behind them, bytes per mapping
extern std::map<void*, VmmSegment> g_vmm_segments;
void* vmm_segment(int device, size_t n, size_t piece);  // prims.hip
// merge.hip: rows (W key words, then the length) of the entries [lo, lo + cnt) of a finalized seqset
// the cover stage's products (build_seqset.hip cover_counts_impl): the oriented reads and, for those of [lo, hi), the number of
// suffix records each emits, where they start, and their total
struct LocalPrev;
struct CoverCounts {
  DevBuf<uint64_t> owords;
  DevBuf<uint8_t> olen;
  DevBuf<uint32_t> cnt;
  DevBuf<uint64_t> roff;
  uint64_t n_emit = 0, o_base = 0, n_rec = 0;
};
int cover_counts_impl(bg_ctx* ctx, const bg_reads* reads, uint64_t lo, uint64_t hi, int cover_k, CoverCounts* cc);
// items_build.hip: the entries (keys, pre2, sizes, shared, the local prev bits and the list of tails) from 16-byte items;
// *done = false: the input has buckets or runs the items path does not take yet -- nothing was changed, use the rows
int items_entries_impl(bg_ctx* ctx, const bg_reads* reads, CoverCounts* cc, bg_seqset* ss, LocalPrev* lp, bool* done);
// build_seqset.hip: the order of records by full key (rows of RS words), sidx[i] = the row at sorted place i
int sort_records_impl(bg_ctx* ctx, const uint64_t* rec_p, const uint8_t* rlen_p, uint64_t n_rec, int W, DevBuf<uint64_t>* k0,
                      DevBuf<uint32_t>* sidx_out, uint64_t* n_tied_out, int RS = 0, bool defer_small = false);
// build_seqset.hip: the suffix records as rows from the cover stage's counts (the second half of suffix_records_impl)
int suffix_rows_from_counts(bg_ctx* ctx, const bg_reads* reads, CoverCounts* cc, DevBuf<uint64_t>* rec, DevBuf<uint8_t>* rlen,
                            DevBuf<uint64_t>* k0, int RS);
// The readmap's columns as the discovery engine consults them (tracer sessions, calc_coverage): a bit per entry that carries
// reads, the number of such entries before every 64-entry word, the first read id of the k-th such entry (readmap::entry_to_index
// without a binary search), a 32-bit mate loop and two bytes a read -- made on the device from the 64-bit columns, downloaded
// once through the pinned double buffer and shared by everybody who asks for the same readmap (readmap_tables keeps a weak
// reference per readmap; bg_readmap_free drops it).  The d_* copies (plain hipMalloc: they outlive any context) serve
// k_trace_simple.
// (vectors whose resize leaves the new elements as they are: a download fills them; zeroing 270 MB first was a single thread's
// page faults and stores before the copy threads wrote the same pages)
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
    else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
  }
};
struct ReadmapTables {
  std::vector<uint8_t, NoInitAlloc<uint8_t>> is_forward, read_len;
  std::vector<uint32_t, NoInitAlloc<uint32_t>> mate_loop, word_rank, group_start;
  std::vector<uint64_t, NoInitAlloc<uint64_t>> has_reads;
  uint32_t min_read_len = 0;
  uint32_t *d_word_rank = nullptr, *d_group_start = nullptr, *d_mate_loop = nullptr;
  uint8_t* d_is_forward = nullptr;
  ReadmapTables() {}
  ReadmapTables(const ReadmapTables&) = delete;
  ReadmapTables& operator=(const ReadmapTables&) = delete;
  ~ReadmapTables();
};
int readmap_tables(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, std::shared_ptr<const ReadmapTables>* out);
void readmap_tables_forget(const bg_readmap* rm);
// keys / pre2 of a seqset built from items, on demand (no-op when they are there or cannot be made).  The first keyed access of
// a fresh seqset is NOT const: it allocates from the seqset's own context (ctx only when the seqset has none) under keys_mutex.
int seqset_ensure_keys(bg_ctx* ctx, bg_seqset* ss);
// ... for an entry point that cannot work without them: BG_EINVAL (a seqset opened from a file holds no key table)
int seqset_require_keys(bg_ctx* ctx, const bg_seqset* ss, const char* who);
// items_build.hip: the front half of the items path (BG_ITEMS_PROBE=1), beside the build
int items_probe(bg_ctx* ctx, const uint64_t* owords, const uint16_t* len, const uint32_t* cnt, const uint64_t* roff, uint64_t o_base,
                uint64_t n_emit, int W, uint64_t n_rec);
int seqset_stripe_rows(bg_ctx* ctx, const bg_seqset* s, uint64_t lo, uint64_t cnt, int W, uint64_t* d_rows);
bg_ctx::DeviceArena* shared_arena(int device);  // null unless the hook is installed
// walk.hip: the read-ahead tables of one walked stretch of a tracer work item (offset after base i = first_off + i),
// appended to the three lists; the two sorted ones are sorted within this stretch
// Several pop_tracer passes ahead for n poppers (walk.hip k_pop_passes): record [i * passes + j] is popper i after its (j + 1)-th
// pop_front -- the range, and ids[ids_at, ids_at + n_ids) the reverse complements of its first `cap` prefix reads; a popper stops
// at the first pass that leaves it less than min_pop_overlap bases (that record is the last one written).
constexpr uint32_t kPopPassMaxIds = 32;
struct PopPassRec { uint64_t b, e; uint32_t s, n_ids, ids_at, pad; };
int tracer_extend_launch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n, const bg_trace_query* d_queries,
                         uint32_t step_cap, bg_trace_reply* d_replies, uint32_t* d_steps);
// h_*: host arrays; pin_in / pin_out: page-locked staging of at least pop_passes_in_bytes / _out_bytes.
size_t pop_passes_in_bytes(uint64_t n);
size_t pop_passes_out_bytes(uint64_t n, uint32_t passes, uint32_t cap);
int pop_passes(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n, const uint64_t* h_begin, const uint64_t* h_end,
               const uint32_t* h_size, uint32_t passes, uint32_t min_pop_overlap, int read_len_limit, uint32_t cap, void* pin_in,
               void* pin_out, std::vector<PopPassRec>* recs, std::vector<uint32_t>* ids);
// The prefix reads of n ranges (device arrays) with the pop tracer's per-read attributes, packed (walk.hip): range q's first
// min(count[q], cap) reads are [first[q], first[q + 1]) of id / rc / mate_rc (0xffffffff: no mate) / flags (1 has mate, 2
// is forward) / len; count[q] is the full number of prefix reads.
struct PackedPrefixReads {
  std::vector<uint32_t> count, first, id, rc, mate_rc;
  std::vector<uint8_t> flags, len;
};
int prefix_reads_packed(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n, const uint64_t* d_begin,
                        const uint64_t* d_end, const uint32_t* d_size, int read_len_limit, uint32_t cap, PackedPrefixReads* out);
int tracer_read_tables(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const uint8_t* d_ref_map, bool skip_ambiguous,
                       const uint64_t* d_ent, uint64_t n, int64_t first_off, std::vector<std::pair<int32_t, uint64_t>>* pos_entries,
                       std::vector<std::pair<uint64_t, int32_t>>* entry_pos, std::vector<std::pair<uint32_t, int32_t>>* mate_pos,
                       DevBuf<uint64_t>* keep_entry_pos = nullptr, DevBuf<uint64_t>* keep_mate_pos = nullptr,
                       std::vector<uint16_t>* pos_info = nullptr);  // per pos_entries row: the entry's size | 256 when the reference holds it more than once  // (the device's copies of the two sorted tables, for the caller to go on using)
// walk.hip: the push tracer's search for the traces that never branch, whole on the device (tracer.cpp:477-519, 528-673,
// 675-724, 737-845, 1023-1250 for ONE path): start, the walk to the next read, add_pairs there, the rejoin, the assembly's
// fields.  A trace that does anything else (a branch, a read that does not rejoin, a step cap) is marked for the host's
// state machine, which runs it from its start.
struct TraceSeed {  // one trace = one offset of the sweep (tracer::trace)
  uint64_t entry;
  int32_t cur_offset;
  uint32_t size;            // the left anchor's length = the entry's size
  uint32_t ref_ambiguous;   // m_ref_ambiguous_bases at this offset
  int32_t win_lo, win_hi;   // window of the sliding tables
  uint32_t pad;
};
constexpr int kSimpleSteps = 64, kSimpleIds = 24;
enum { kSimpleHost = 0, kSimpleAssembly = 1, kSimpleNothing = 2 };
enum { kSimRefReads = 1, kSimAmbiguousRefReads = 2, kSimStepCount = 4, kSimOutput = 8, kSimMatchedPairs = 16, kSimMaxBranchCost = 32,
       kSimTooManyAmbiguousBases = 64, kSimTooManySteps = 128, kSimTooFarWithoutPair = 256, kSimPruneAmbiguousRef = 512 };
struct TraceSimpleOut {
  uint32_t status, n_steps;
  int32_t right_offset;
  uint32_t right_anchor_len, min_overlap, n_ids, counters, found_pairs;
  uint32_t ids[kSimpleIds];
  uint8_t bases[kSimpleSteps];
};
struct TraceSimpleTables {
  const uint8_t* ref_map;         // per entry: match count in the low 6 bits
  const uint32_t* word_rank;      // entries with reads before each 64-entry word
  const uint32_t* group_start;    // first read id of the k-th entry that carries reads (+ one past the end)
  const uint32_t* mate_loop;      // per read id
  const uint8_t* is_forward;      // per read id
  const void* entry_positions;    // std::pair<uint64_t, int32_t>[n_entry_positions], sorted
  uint64_t n_entry_positions;
  const void* mate_positions;     // std::pair<uint32_t, int32_t>[n_mate_positions], sorted
  uint64_t n_mate_positions;
};
int tracer_simple_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const TraceSimpleTables& tb, const bg_trace_options& o,
                        const TraceSeed* d_seeds, uint64_t n, TraceSimpleOut* d_out);
// build_seqset.hip
int cover_index_build(bg_ctx* ctx, const uint64_t* owords, const uint8_t* olen, uint64_t n_o, int W, int cover_k,
                      DevBuf<uint64_t>* k0o, CoverStore* cs);
int seqset_finalize_device(bg_ctx* ctx, bg_seqset* ss, bool slice = false);  // bitcount finalize + min index + fixed
// the walk blocks of a finalized seqset, made by the first walk that asks (under keys_mutex, from the seqset's own context, on
// its stream, waited for); a seqset they cannot be made for keeps walking through rank4 / sizes / shared
void seqset_ensure_walk_blocks(const bg_seqset* ss);
int suffix_records_impl(bg_ctx* ctx, const bg_reads* reads, uint64_t lo, uint64_t hi, int cover_k,
                        DevBuf<uint64_t>* rec, DevBuf<uint8_t>* rlen, uint64_t* n_rec_out,
                        DevBuf<uint64_t>* k0 = nullptr, int RS = 0);  // k0: also the sort keys of the records (optional)
                        // RS = words from one record row to the next (0: W; W + 1: a row ends with the record's length)
// word W of a record row (RS = W + 1): the record's length in the low byte, and above it what k_gather_dedup_build needs to set
// prev bits where the records meet (written by k_emit_records and by the sharded build's RecordSource)
constexpr uint64_t kRecHasPred = 1ull << 8;   // the read has a base in front of this suffix
constexpr int kRecPredShift = 9;              // ... that base (2 bits)
constexpr uint64_t kRecLast = 1ull << 11;     // the last suffix its read emitted (the next one is covered by another read)

// prev bits set where the sorted records meet (k_gather_dedup_build) + the tails that still need a search (k_prev_list)
struct LocalPrev {
  bool shard = false;        // a shard of a sharded build: survivors whose tail must travel are flagged in `need`, and the
                             // list holds, for every record dropped as a proper prefix, base + record (looked up where it sorts)
  DevBuf<uint8_t> need;
  DevBuf<uint8_t> emask;     // one byte per entry: bit b = prev_b
  DevBuf<uint64_t> list;     // rows of W + 1 words: tail, length | base << 8 | checked << 16
  DevBuf<uint64_t> count;    // rows asked for (device)
  uint64_t cap = 0, n_items = 0;
  bool valid = false;        // the kernel ran in this mode and the list held every row
  bool searched = false;     // the list's tails are in emask already (items_build.hip): only the packing is left
};
int entries_from_records_impl(bg_ctx* ctx, DevBuf<uint64_t>& rec, DevBuf<uint8_t>& rlen, uint64_t n_rec, int W,
                              bg_seqset* ss, uint64_t* n_tied_out, DevBuf<uint64_t>* k0 = nullptr, int RS = 0,
                              LocalPrev* lp = nullptr);
// leading-bases directory over sorted W-word keys (*bases = 8..13 chosen from n): dir[p] = first key with
// prefix >= p, dir[4^bases] = n
int build_prefix_dir_keys(bg_ctx* ctx, const uint64_t* keys, uint64_t n, int W, DevBuf<uint32_t>* dir, int* bases);
// items_build.hip: the oriented reads of >= cover_k bases in (string, length) order, as first K-mers and read numbers
int items_cover_order(bg_ctx* ctx, const uint64_t* owords, const uint8_t* olen, uint64_t n_o, int W, int cover_k, DevBuf<uint64_t>* premade,
                      DevBuf<uint64_t>* skeys, DevBuf<uint32_t>* soid, uint64_t* m_out, bool* done);
int prefix_dir_prepare(bg_ctx* ctx, uint64_t n, DevBuf<uint32_t>* dir, int* bases);  // (the caller writes the heads between the two)
int prefix_dir_finish(bg_ctx* ctx, uint64_t n, int bases, DevBuf<uint32_t>* dir);
int alloc_prev_bits(bg_ctx* ctx, bg_seqset* ss);
int prev_bits_local_impl(bg_ctx* ctx, bg_seqset* ss, LocalPrev* lp = nullptr);
int apply_tails_impl(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows, uint32_t* d_err);
int apply_tails2_impl(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows, uint8_t* d_need_full, uint32_t* d_err);
int kmerset_sort(bg_ctx* ctx, bg_kmerset* ks);
int kmer_count_impl(bg_ctx* ctx, const bg_reads* reads, int k, uint32_t min_count, bool with_flags, bg_kmerset** out);
int read_locus_order(bg_ctx* ctx, const bg_reads* reads, uint64_t first, uint64_t count, DevBuf<uint32_t>* perm);
// spiral_zip.cpp
struct ZipPart { std::string name; std::string data; bool raw_crc0; uint64_t hole_size = 0; };
int spiral_zip_write(const char* path, const std::vector<ZipPart>& parts);
int spiral_zip_write_layout(const char* path, const std::vector<ZipPart>& parts, std::vector<uint64_t>* data_offsets);
int spiral_zip_read(const char* path, std::map<std::string, std::string>* parts);
int spiral_zip_read_views(const char* path, std::map<std::string, std::string>* parts,
                          std::map<std::string, std::pair<const char*, uint64_t>>* big, std::shared_ptr<void>* keep);
std::string make_uuid();
int sha1_file_hex(const char* path, char* hex41);
}  // namespace bg
