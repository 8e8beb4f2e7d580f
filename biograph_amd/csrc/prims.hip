// prims.hip — context, device-wide scan and LSD radix sort for gfx950 (wave64).
//
// The radix sort is the bandwidth-dominant primitive of the build: it orders the canonical k-mer
// instances (row B of SURVEY §8a; reference = hash tables kmer_count_table.h:54-103) and the suffix
// records (row D3; reference = std::sort of 14-byte records, repo_seq.cpp:660-684).
//
// One pass over an 8-bit digit = 3 kernels:
//   rs_hist     per-tile digit histogram                      (reads keys once)
//   scan        exclusive scan of the [bin][tile] table
//   rs_scatter  stable rank inside the tile via wave64 match-any ballots, LDS-staged reorder, then
//               coalesced runs to HBM                          (reads keys once, writes once)
// A one-off rs_digit_census pass finds digits on which all keys agree so those passes are skipped.
#include "bg_device_funcs.h"
#include <chrono>

#include "bg_internal.h"

namespace bg {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
const char* last_error_cstr() { return g_err.c_str(); }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d: %s", int(e), hipGetErrorString(e), file, line, what);
  return e == hipErrorOutOfMemory ? BG_ENOMEM : BG_ENODEV;
}

}  // namespace bg

// ------------------------------------------------------------------------------------------------
// context

namespace bg {
std::mutex g_vmm_mutex;
std::map<void*, VmmSegment> g_vmm_segments;

// n bytes (a multiple of `piece`) as one address range with `piece` bytes of physical memory mapped at a time; nullptr when
// the driver refuses (nothing stays behind).  Registered in g_vmm_segments: seg_free takes it apart the same way.
void* vmm_segment(int device, size_t n, size_t piece) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  hipMemAccessDesc access = {};
  access.location = prop.location;
  access.flags = hipMemAccessFlagsProtReadWrite;
  void* base = nullptr;
  // (the range aligned like its pieces: a gigabyte of contiguous physical memory behind a gigabyte-aligned address is what the
  // page tables can describe with their largest fragments)
  if (hipMemAddressReserve(&base, n, piece, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  size_t done = 0;
  for (; done < n; done += piece) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) break;
    const bool mapped = hipMemMap(static_cast<char*>(base) + done, piece, 0, h, 0) == hipSuccess;
    (void)hipMemRelease(h);  // the mapping keeps the memory
    if (!mapped) break;
  }
  if (done == n && hipMemSetAccess(base, n, &access, 1) == hipSuccess) {
    std::lock_guard<std::mutex> l(g_vmm_mutex);
    g_vmm_segments[base] = VmmSegment{n, n, piece};
    return base;
  }
  (void)hipGetLastError();
  for (size_t at = 0; at < done; at += piece) (void)hipMemUnmap(static_cast<char*>(base) + at, piece);
  (void)hipMemAddressFree(base, n);
  (void)hipGetLastError();
  return nullptr;
}
}  // namespace bg

void* bg_ctx::HipSegments::seg_alloc(size_t n) {
  static const bool trace = getenv("BG_ALLOC_TRACE") != nullptr;
  // Off unless asked for (BG_ARENA_VMM=1).  Measured on the bench step, memory mapped this way serves the kernels 2 % slower
  // than hipMalloc's (59.6 against 58.4 ms: k_correct_dfs 3.55 against 3.23), and a process that shares the arena with torch's
  // allocator aborted inside the runtime.  What it is for is the one-shot command: bg_ctx_reserve (bgbinary create) maps the
  // build's memory this way beside the import.
  static const bool use_vmm = [] { const char* e = getenv("BG_ARENA_VMM"); return e && e[0] == '1'; }();
  const auto t0 = std::chrono::steady_clock::now();
  void* p = nullptr;
  if (use_vmm && n >= kLargeRound && n % kLargeRound == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    p = bg::vmm_segment(dev, n, kLargeRound);
    if (p) {
      if (trace)
        fprintf(stderr, "[bg alloc] %zu MiB mapped in 1 GiB pieces: %.2f ms\n", n >> 20,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      return p;
    }
  }
  hipError_t e = hipMalloc(&p, n);
  if (trace)
    fprintf(stderr, "[bg alloc] hipMalloc %zu MiB: %.2f ms%s\n", n >> 20,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
            e == hipSuccess ? "" : " FAILED");
  if (e != hipSuccess) {
    (void)hipGetLastError();  // the failed attempt must not surface at the next launch check
    return nullptr;
  }
  return p;
}

// Segments built piece by piece through the virtual-memory API (vmm_segment above, bg_ctx_reserve): they go back the same way.

void bg_ctx::HipSegments::seg_free(void* p) {  // waits for the device
  bg::VmmSegment vmm{0, 0, 0};
  {
    std::lock_guard<std::mutex> l(bg::g_vmm_mutex);
    auto it = bg::g_vmm_segments.find(p);
    if (it != bg::g_vmm_segments.end()) { vmm = it->second; bg::g_vmm_segments.erase(it); }
  }
  if (!vmm.reserved) { (void)hipFree(p); return; }
  (void)hipDeviceSynchronize();
  // every piece was mapped with a handle of its own: it is unmapped on its own (an unmap spanning several mappings may be
  // refused), and the address range goes back whole -- also the tail a partial failure left unmapped
  for (size_t at = 0; at < vmm.mapped; at += vmm.piece)
    if (hipMemUnmap(static_cast<char*>(p) + at, vmm.piece) != hipSuccess) {
      (void)hipGetLastError();
      fprintf(stderr, "libbgseq: hipMemUnmap of a reserved piece failed (%zu MiB stay mapped)\n", vmm.piece >> 20);
    }
  if (hipMemAddressFree(p, vmm.reserved) != hipSuccess) (void)hipGetLastError();
}

// A context with an arena of its own locks only itself: the tracer sessions of `bgbinary discovery` (16 contexts on 16
// threads, hundreds of small buffers per work item each) used to meet on one process-wide mutex.
std::mutex& bg_ctx::arena_mutex() { return arena == &own_arena ? own_mutex : bg::g_arena_mutex; }

int bg_ctx::alloc(size_t bytes, void** out) {
  void* p;
  {
    std::lock_guard<std::mutex> lock(arena_mutex());
    p = arena->alloc(bytes);
  }
  if (!p) {
    (void)hipStreamSynchronize(stream);
    bg::set_error("device allocation of %zu bytes failed (live %zu MiB, held %zu MiB)", bytes, arena->live() >> 20,
                  arena->reserved() >> 20);
    return BG_ENOMEM;
  }
  *out = p;
  return BG_OK;
}

// The chunk goes back to the arena at once: everything a context does is ordered on its one stream, so whoever gets
// the chunk next is ordered after the work that used it; a wholly free segment may go back to the driver
// (hipFree waits for the device).  An arena is never shared across streams: the per-device arena of
// bg_torch_allocator_enable serves torch's current stream and the contexts created ON that stream (bg_ctx_create's
// `stream` argument must be torch's current stream in that mode); bg_ctx_create_own_stream keeps a private arena.
void bg_ctx::release(void* p) {
  std::lock_guard<std::mutex> lock(arena_mutex());
  (void)arena->release(p);
}

void bg_ctx::shrink(void* p, size_t bytes) {
  std::lock_guard<std::mutex> lock(arena_mutex());
  (void)arena->shrink(p, bytes);
}

int bg_ctx::trim() {
  (void)hipStreamSynchronize(stream);
  std::lock_guard<std::mutex> lock(arena_mutex());
  arena->trim();
  return BG_OK;
}

// ---- torch's allocator hook (torch.cuda.memory.CUDAPluggableAllocator): the exchange buffers of the multi-GPU
// driver and the library's own buffers come out of ONE arena per device, so that neither allocator sits on memory the
// other needs (with two, each keeps its cache until one of its own requests fails, and the driver maps memory at
// ~30 us per MiB: a build of 11.6 Gbases per rank spent seconds re-acquiring what the other side had just let go).
// Everything is ordered on one stream per process here (torch's current stream = the context's stream; torch waits
// for a collective before the call returns), so a freed chunk may be handed out again at once.
namespace bg {
std::mutex g_arena_mutex;
static bg_ctx::DeviceArena* g_shared[16] = {};
static bool g_share = false;
bg_ctx::DeviceArena* shared_arena(int device) {
  if (!g_share || device < 0 || device >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  if (!g_shared[device]) g_shared[device] = new bg_ctx::DeviceArena();  // lives as long as the process
  return g_shared[device];
}
}  // namespace bg

extern "C" void bg_torch_allocator_enable(void) { bg::g_share = true; }
extern "C" int bg_torch_allocator_enabled(void) { return bg::g_share ? 1 : 0; }

extern "C" void* bg_torch_malloc(ssize_t size, int device, void*) {
  bg::g_share = true;
  bg_ctx::DeviceArena* a = bg::shared_arena(device);
  if (!a) return nullptr;
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != device) (void)hipSetDevice(device);
  void* p;
  {
    std::lock_guard<std::mutex> lock(bg::g_arena_mutex);
    p = a->alloc(size > 0 ? size_t(size) : 1);
  }
  if (cur != device) (void)hipSetDevice(cur);
  return p;  // null: torch raises its out-of-memory error
}

extern "C" void bg_torch_free(void* ptr, ssize_t, int device, void*) {
  bg_ctx::DeviceArena* a = bg::shared_arena(device);
  if (!a || !ptr) return;
  std::lock_guard<std::mutex> lock(bg::g_arena_mutex);
  (void)a->release(ptr);
}

hipEvent_t bg_ctx::get_event() {
  if (!event_pool.empty()) {
    hipEvent_t e = event_pool.back();
    event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void bg_ctx::tic(const char* name) {
  if (!profiling) return;
  bg::TimingRec r;
  r.name = name;
  r.start = get_event();
  r.stop = get_event();
  r.bytes = 0;
  (void)hipEventRecord(r.start, stream);
  recs.push_back(r);
}

void bg_ctx::toc(uint64_t bytes) {
  if (!profiling || recs.empty()) return;
  bg::TimingRec& r = recs.back();
  r.bytes = bytes;
  (void)hipEventRecord(r.stop, stream);
}

void bg_ctx::clear_timings() {
  for (auto& r : recs) {
    event_pool.push_back(r.start);
    event_pool.push_back(r.stop);
  }
  recs.clear();
}

namespace bg {

// ------------------------------------------------------------------------------------------------
// wave / block helpers (wave = 64 lanes)

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// Exclusive scan of one value per thread over a 256-thread block; returns the exclusive prefix and
// the block total through *total.  sm must hold 4 elements.
template <class T>
__device__ __forceinline__ T block_excl_scan_256(T v, T* sm, T* total) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  T inc;
  if constexpr (sizeof(T) == 8) inc = wave_incl_scan_u64(v, lane);
  else inc = wave_incl_scan_u32(v, lane);
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    T s = sm[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ------------------------------------------------------------------------------------------------
// device-wide exclusive scan: reduce -> spine -> apply

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;  // 2048

template <class OutT>
__global__ __launch_bounds__(SC_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in, uint64_t n,
                                                                 OutT* __restrict__ partial) {
  __shared__ OutT sm[4];
  uint64_t base = uint64_t(blockIdx.x) * SC_TILE + uint64_t(threadIdx.x) * SC_ITEMS;
  OutT s = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i)
    if (base + i < n) s += in[base + i];
  OutT tot;
  (void)block_excl_scan_256<OutT>(s, sm, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// Single block: exclusive scan of up to SC_TILE partials in place; writes total.
template <class OutT>
__global__ __launch_bounds__(SC_THREADS) void scan_spine_kernel(OutT* __restrict__ partial, uint32_t np,
                                                                OutT* __restrict__ total) {
  __shared__ OutT sm[4];
  OutT v[SC_ITEMS];
  uint32_t base = threadIdx.x * SC_ITEMS;
  OutT s = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    v[i] = (base + i < np) ? partial[base + i] : 0;
    s += v[i];
  }
  OutT tot;
  OutT ex = block_excl_scan_256<OutT>(s, sm, &tot);
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    if (base + i < np) partial[base + i] = ex;
    ex += v[i];
  }
  if (threadIdx.x == 0 && total) *total = tot;
}

template <class OutT>
__global__ __launch_bounds__(SC_THREADS) void scan_apply_kernel(const uint32_t* __restrict__ in, uint64_t n,
                                                                const OutT* __restrict__ partial,
                                                                OutT* __restrict__ out) {
  __shared__ OutT sm[4];
  uint64_t base = uint64_t(blockIdx.x) * SC_TILE + uint64_t(threadIdx.x) * SC_ITEMS;
  uint32_t v[SC_ITEMS];
  OutT s = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  OutT tot;
  OutT ex = block_excl_scan_256<OutT>(s, sm, &tot) + partial[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
}

// Scan an array of OutT partials in place (recursive when it does not fit one spine block).
template <class OutT>
__global__ __launch_bounds__(SC_THREADS) void scan_reduce_wide_kernel(const OutT* __restrict__ in, uint64_t n,
                                                                      OutT* __restrict__ partial) {
  __shared__ OutT sm[4];
  uint64_t base = uint64_t(blockIdx.x) * SC_TILE + uint64_t(threadIdx.x) * SC_ITEMS;
  OutT s = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i)
    if (base + i < n) s += in[base + i];
  OutT tot;
  (void)block_excl_scan_256<OutT>(s, sm, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
template <class OutT>
__global__ __launch_bounds__(SC_THREADS) void scan_apply_wide_kernel(OutT* __restrict__ data, uint64_t n,
                                                                     const OutT* __restrict__ partial) {
  __shared__ OutT sm[4];
  uint64_t base = uint64_t(blockIdx.x) * SC_TILE + uint64_t(threadIdx.x) * SC_ITEMS;
  OutT v[SC_ITEMS];
  OutT s = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    v[i] = (base + i < n) ? data[base + i] : 0;
    s += v[i];
  }
  OutT tot;
  OutT ex = block_excl_scan_256<OutT>(s, sm, &tot) + partial[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    if (base + i < n) data[base + i] = ex;
    ex += v[i];
  }
}

template <class OutT>
static int scan_partials_inplace(bg_ctx* ctx, OutT* d_part, uint64_t np, OutT* d_total) {
  if (np <= SC_TILE) {
    scan_spine_kernel<OutT><<<1, SC_THREADS, 0, ctx->stream>>>(d_part, uint32_t(np), d_total);
    BG_LAUNCH_CHECK();
    return BG_OK;
  }
  uint64_t np2 = (np + SC_TILE - 1) / SC_TILE;
  DevBuf<OutT> p2;
  BG_TRY(p2.alloc(ctx, np2));
  scan_reduce_wide_kernel<OutT><<<unsigned(np2), SC_THREADS, 0, ctx->stream>>>(d_part, np, p2.p);
  BG_LAUNCH_CHECK();
  BG_TRY(scan_partials_inplace<OutT>(ctx, p2.p, np2, d_total));
  scan_apply_wide_kernel<OutT><<<unsigned(np2), SC_THREADS, 0, ctx->stream>>>(d_part, np, p2.p);
  BG_LAUNCH_CHECK();
  return BG_OK;
}

template <class OutT>
static int exclusive_scan_impl(bg_ctx* ctx, const uint32_t* d_in, OutT* d_out, uint64_t n, OutT* d_total) {
  if (n == 0) {
    if (d_total) BG_HIP(hipMemsetAsync(d_total, 0, sizeof(OutT), ctx->stream));
    return BG_OK;
  }
  uint64_t nb = (n + SC_TILE - 1) / SC_TILE;
  DevBuf<OutT> part;
  BG_TRY(part.alloc(ctx, nb));
  scan_reduce_kernel<OutT><<<unsigned(nb), SC_THREADS, 0, ctx->stream>>>(d_in, n, part.p);
  BG_LAUNCH_CHECK();
  BG_TRY(scan_partials_inplace<OutT>(ctx, part.p, nb, d_total));
  scan_apply_kernel<OutT><<<unsigned(nb), SC_THREADS, 0, ctx->stream>>>(d_in, n, part.p, d_out);
  BG_LAUNCH_CHECK();
  return BG_OK;
}

int exclusive_scan_u32_to_u64(bg_ctx* ctx, const uint32_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* d_total) {
  return exclusive_scan_impl<uint64_t>(ctx, d_in, d_out, n, d_total);
}
int exclusive_scan_u32(bg_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint32_t* d_total) {
  return exclusive_scan_impl<uint32_t>(ctx, d_in, d_out, n, d_total);
}

// BG_FORCE_OFF64=1 runs the 64-bit-offset instantiations at any size (they are otherwise reached only
// beyond 2^32 keys, which no test can afford)
static bool use_off32(uint64_t n) {
  static const bool force64 = [] { const char* e = getenv("BG_FORCE_OFF64"); return e && e[0] == '1'; }();
  return n < (1ull << 32) && !force64;
}

// ------------------------------------------------------------------------------------------------
// radix sort

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 keys per workgroup
constexpr int RS_BINS = 256;

// Census: the AND and the OR of all keys.  A bit position on which they agree is constant over the input, so
// the host skips the passes whose digit is constant (tie rounds sort keys that agree almost everywhere; a
// per-byte histogram of such keys serialises on one LDS address).
__global__ __launch_bounds__(RS_THREADS) void rs_and_or_kernel(const uint64_t* __restrict__ keys, uint64_t n,
                                                               unsigned long long* __restrict__ and_or) {
  unsigned long long a = ~0ull, o = 0ull;
  const uint64_t stride = uint64_t(gridDim.x) * RS_THREADS;
  for (uint64_t i = uint64_t(blockIdx.x) * RS_THREADS + threadIdx.x; i < n; i += stride) {
    const unsigned long long k = keys[i];
    a &= k;
    o |= k;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a &= __shfl_xor(a, d, 64);
    o |= __shfl_xor(o, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAnd(&and_or[0], a);
    atomicOr(&and_or[1], o);
  }
}

// Lanes of the wave holding the same 8-bit digit (and the same validity) as this lane.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
  uint64_t m = ~0ull;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    bool bit = (d >> b) & 1;
    uint64_t bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  uint64_t bv = __ballot(valid);
  m &= valid ? bv : ~bv;
  return m;
}

// digit of a key; HASHED sorts on the bits of key * kPartitionMult instead (hash partitioning)
template <bool HASHED>
__device__ __forceinline__ uint32_t rs_digit(uint64_t key, int shift, uint32_t dmask) {
  if (HASHED) key *= kPartitionMult;
  return uint32_t(key >> shift) & dmask;
}

template <bool HASHED>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, int shift,
                                                             uint32_t dmask, uint32_t* __restrict__ hist,
                                                             uint32_t ntiles) {
  const uint64_t tile_x = xcd_chunk(blockIdx.x, gridDim.x);  // XCD-contiguous tiles: adjacent runs of a bin meet in one L2
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  uint64_t base = tile_x * RS_TILE;
#pragma unroll 4
  for (int it = 0; it < RS_ITEMS; ++it) {
    uint64_t idx = base + uint64_t(it) * RS_THREADS + threadIdx.x;
    if (idx < n) atomicAdd(&h[rs_digit<HASHED>(keys[idx], shift, dmask)], 1u);
  }
  __syncthreads();
  hist[uint64_t(threadIdx.x) * ntiles + tile_x] = h[threadIdx.x];
}

// Same histogram for keys with few distinct digit values (partition destinations, group ids): lanes
// holding the same digit are found with ballots and one lane adds the whole group, so a wave costs
// at most (#distinct digits) LDS atomics instead of 64 serialised ones on one address.
__global__ __launch_bounds__(RS_THREADS) void rs_hist_agg_kernel(const uint64_t* __restrict__ keys, uint64_t n,
                                                                 int shift, uint32_t dmask,
                                                                 uint32_t* __restrict__ hist, uint32_t ntiles) {
  const uint64_t tile_x = xcd_chunk(blockIdx.x, gridDim.x);  // XCD-contiguous tiles: adjacent runs of a bin meet in one L2
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint64_t lane_lt = (1ull << lane) - 1;
  uint64_t base = tile_x * RS_TILE;
  for (int it = 0; it < RS_ITEMS; ++it) {
    uint64_t idx = base + uint64_t(it) * RS_THREADS + threadIdx.x;
    bool valid = idx < n;
    uint32_t d = valid ? uint32_t(keys[idx] >> shift) & dmask : 0u;
    uint64_t m = match_digit(d, valid);
    if (valid && (m & lane_lt) == 0) atomicAdd(&h[d], uint32_t(__popcll(m)));
  }
  __syncthreads();
  hist[uint64_t(threadIdx.x) * ntiles + tile_x] = h[threadIdx.x];
}


template <class OffT, bool HAS_VAL, bool HASHED>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint64_t* __restrict__ kin,
                                                                const uint32_t* __restrict__ vin,
                                                                uint64_t* __restrict__ kout,
                                                                uint32_t* __restrict__ vout, uint64_t n, int shift,
                                                                uint32_t dmask, const OffT* __restrict__ offs,
                                                                uint32_t ntiles) {
  const uint64_t tile_x = xcd_chunk(blockIdx.x, gridDim.x);  // XCD-contiguous tiles: adjacent runs of a bin meet in one L2
  __shared__ uint64_t skeys[RS_TILE];
  __shared__ uint32_t svals[HAS_VAL ? RS_TILE : 1];
  __shared__ uint32_t whist[4][RS_BINS];
  __shared__ uint32_t binstart[RS_BINS];
  __shared__ OffT goff[RS_BINS];
  __shared__ uint32_t scan_sm[4];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint64_t base = tile_x * RS_TILE;
  const uint64_t lane_lt = (1ull << lane) - 1;

  for (int i = tid; i < 4 * RS_BINS; i += RS_THREADS) (&whist[0][0])[i] = 0;
  __syncthreads();

  uint64_t key[RS_ITEMS];
  uint32_t val[RS_ITEMS];
  uint16_t rnk[RS_ITEMS];
  volatile uint32_t* wh = &whist[w][0];

  // phase 1: load (each wave owns a contiguous 1024-key chunk, so tile order == global order) and
  // rank every key among equal digits seen so far by this wave.
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    uint64_t idx = base + uint64_t(w) * (RS_ITEMS * 64) + uint64_t(it) * 64 + lane;
    bool valid = idx < n;
    key[it] = valid ? kin[idx] : ~0ull;
    if (HAS_VAL) val[it] = valid ? vin[idx] : 0u;
    uint32_t d = rs_digit<HASHED>(key[it], shift, dmask);
    uint64_t m = match_digit(d, valid);
    uint32_t before = wh[d];
    uint32_t lt = __popcll(m & lane_lt);
    rnk[it] = uint16_t(before + lt);
    if (valid && lt == 0) wh[d] = before + uint32_t(__popcll(m));
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // phase 2: thread = bin.  Wave bases, bin starts inside the tile, global run offsets.
  {
    uint32_t c0 = whist[0][tid], c1 = whist[1][tid], c2 = whist[2][tid], c3 = whist[3][tid];
    uint32_t tot = c0 + c1 + c2 + c3;
    whist[0][tid] = 0;
    whist[1][tid] = c0;
    whist[2][tid] = c0 + c1;
    whist[3][tid] = c0 + c1 + c2;
    uint32_t dummy;
    uint32_t ex = block_excl_scan_256<uint32_t>(tot, scan_sm, &dummy);
    binstart[tid] = ex;
    goff[tid] = offs[uint64_t(tid) * ntiles + tile_x] - OffT(ex);
  }
  __syncthreads();

  // phase 3: reorder inside LDS.
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    uint64_t idx = base + uint64_t(w) * (RS_ITEMS * 64) + uint64_t(it) * 64 + lane;
    if (idx < n) {
      uint32_t d = rs_digit<HASHED>(key[it], shift, dmask);
      uint32_t pos = binstart[d] + whist[w][d] + rnk[it];
      skeys[pos] = key[it];
      if (HAS_VAL) svals[pos] = val[it];
    }
  }
  __syncthreads();

  // phase 4: coalesced runs out.
  uint32_t cnt = (n - base) < uint64_t(RS_TILE) ? uint32_t(n - base) : uint32_t(RS_TILE);
  for (uint32_t i = tid; i < cnt; i += RS_THREADS) {
    uint64_t k = skeys[i];
    uint32_t d = rs_digit<HASHED>(k, shift, dmask);
    OffT o = goff[d] + OffT(i);
    kout[o] = k;
    if (HAS_VAL) vout[o] = svals[i];
  }
}


// ---- one sweep per digit (decoupled look-back instead of a histogram pass + scan per digit).
// The digit histograms of an LSD sort do not depend on the order of the keys: ONE pass over the input counts all of them
// (rs_hist_all_kernel) and a 256-wide scan per digit gives every bin its base.  A scatter pass then ranks its tile as
// before, publishes the tile's bin counts (state = AGG | count), adds up its predecessors' states until one carries its
// inclusive prefix (INCL | prefix) and publishes its own.  Tiles take their number from a ticket counter: every predecessor
// a tile waits for is running.  (Per-XCD chains with bases of their own would keep neighbouring tiles in one L2, but which
// keys a stretch of the array holds in pass d depends on the passes before: only the whole array's histogram is known up
// front.)
template <class StateT> struct rs_state_bits;
template <> struct rs_state_bits<uint32_t> { static constexpr int shift = 30; };
template <> struct rs_state_bits<unsigned long long> { static constexpr int shift = 62; };

// hist[d * 256 + bin] += keys whose digit d is bin (digit d = bits shift[d].. of the key)
struct RsDigits { int n; int shift[8]; uint32_t mask[8]; };
__global__ __launch_bounds__(RS_THREADS) void rs_hist_all_kernel(const uint64_t* __restrict__ keys, uint64_t n, RsDigits dg,
                                                                 unsigned long long* __restrict__ hist, uint64_t ntiles) {
  __shared__ uint32_t h[8][RS_BINS];
  for (int d = 0; d < dg.n; ++d) h[d][threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t base = t * RS_TILE;
#pragma unroll 4
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint64_t idx = base + uint64_t(it) * RS_THREADS + threadIdx.x;
      if (idx < n) {
        const uint64_t k = keys[idx];
        for (int d = 0; d < dg.n; ++d) atomicAdd(&h[d][uint32_t(k >> dg.shift[d]) & dg.mask[d]], 1u);
      }
    }
  }
  __syncthreads();
  for (int d = 0; d < dg.n; ++d) {
    const uint32_t c = h[d][threadIdx.x];
    if (c) atomicAdd(&hist[uint64_t(d) * RS_BINS + threadIdx.x], (unsigned long long)c);
  }
}

// one workgroup per digit, thread = bin: counts -> exclusive bases
__global__ __launch_bounds__(RS_BINS) void rs_bin_base_kernel(unsigned long long* __restrict__ hist) {
  __shared__ unsigned long long sm[4];
  const int d = blockIdx.x, bin = threadIdx.x;
  const unsigned long long c = hist[uint64_t(d) * RS_BINS + bin];
  unsigned long long dummy;
  hist[uint64_t(d) * RS_BINS + bin] = block_excl_scan_256<unsigned long long>(c, sm, &dummy);
}

template <class OffT, class StateT, bool HAS_VAL>
__global__ __launch_bounds__(RS_THREADS) void rs_onesweep_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                                 uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, uint64_t n,
                                                                 int shift, uint32_t dmask, const unsigned long long* __restrict__ bin_base,
                                                                 StateT* __restrict__ state, uint32_t* __restrict__ ticket) {
  constexpr int SB = rs_state_bits<StateT>::shift;
  constexpr StateT kVal = (StateT(1) << SB) - 1;
  __shared__ uint64_t skeys[RS_TILE];
  __shared__ uint32_t svals[HAS_VAL ? RS_TILE : 1];
  __shared__ uint32_t whist[4][RS_BINS];
  __shared__ uint32_t binstart[RS_BINS];
  __shared__ OffT goff[RS_BINS];
  __shared__ uint32_t scan_sm[4];
  __shared__ uint32_t s_tile;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(ticket, 1u);
  for (int i = tid; i < 4 * RS_BINS; i += RS_THREADS) (&whist[0][0])[i] = 0;
  __syncthreads();
  const uint64_t tile_x = s_tile;
  const uint64_t base = tile_x * RS_TILE;
  const uint64_t lane_lt = (1ull << lane) - 1;

  uint64_t key[RS_ITEMS];
  uint32_t val[RS_ITEMS];
  uint16_t rnk[RS_ITEMS];
  volatile uint32_t* wh = &whist[w][0];
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    uint64_t idx = base + uint64_t(w) * (RS_ITEMS * 64) + uint64_t(it) * 64 + lane;
    bool valid = idx < n;
    key[it] = valid ? kin[idx] : ~0ull;
    if (HAS_VAL) val[it] = valid ? vin[idx] : 0u;
    uint32_t d = rs_digit<false>(key[it], shift, dmask);
    uint64_t m = match_digit(d, valid);
    uint32_t before = wh[d];
    uint32_t lt = __popcll(m & lane_lt);
    rnk[it] = uint16_t(before + lt);
    if (valid && lt == 0) wh[d] = before + uint32_t(__popcll(m));
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // thread = bin: the tile's count goes out at once, then the look-back
  {
    uint32_t c0 = whist[0][tid], c1 = whist[1][tid], c2 = whist[2][tid], c3 = whist[3][tid];
    uint32_t tot = c0 + c1 + c2 + c3;
    StateT* mine = state + tile_x * RS_BINS + tid;
    unsigned long long excl = 0;
    if (tile_x == 0) {
      excl = bin_base[tid];
      __hip_atomic_store(mine, (StateT(2) << SB) | StateT(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_store(mine, (StateT(1) << SB) | StateT(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    whist[0][tid] = 0;
    whist[1][tid] = c0;
    whist[2][tid] = c0 + c1;
    whist[3][tid] = c0 + c1 + c2;
    uint32_t dummy;
    uint32_t ex = block_excl_scan_256<uint32_t>(tot, scan_sm, &dummy);
    binstart[tid] = ex;
    if (tile_x != 0) {
      // eight predecessors per round trip (the loads are independent; a thread walking back one state at a time pays a
      // memory round trip per tile, and hundreds of tiles are in flight)
      int64_t j = int64_t(tile_x) - 1;
      bool done = false;
      while (!done) {
        StateT v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = j - u >= 0 ? __hip_atomic_load(state + uint64_t(j - u) * RS_BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : (StateT(2) << SB);
        int used = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (done || used != u) continue;
          const int flag = int(v[u] >> SB);
          if (flag == 0) continue;  // not published yet: this round stops here
          excl += (unsigned long long)(v[u] & kVal);
          ++used;
          if (flag == 2) done = true;
        }
        j -= used;
        if (!done && used < 8) __builtin_amdgcn_s_sleep(4);
      }
      __hip_atomic_store(mine, (StateT(2) << SB) | StateT(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    goff[tid] = OffT(excl) - OffT(ex);
  }
  __syncthreads();

#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    uint64_t idx = base + uint64_t(w) * (RS_ITEMS * 64) + uint64_t(it) * 64 + lane;
    if (idx < n) {
      uint32_t d = rs_digit<false>(key[it], shift, dmask);
      uint32_t pos = binstart[d] + whist[w][d] + rnk[it];
      skeys[pos] = key[it];
      if (HAS_VAL) svals[pos] = val[it];
    }
  }
  __syncthreads();
  uint32_t cnt = (n - base) < uint64_t(RS_TILE) ? uint32_t(n - base) : uint32_t(RS_TILE);
  for (uint32_t i = tid; i < cnt; i += RS_THREADS) {
    uint64_t k = skeys[i];
    uint32_t d = rs_digit<false>(k, shift, dmask);
    OffT o = goff[d] + OffT(i);
    kout[o] = k;
    if (HAS_VAL) vout[o] = svals[i];
  }
}

template <class OffT, class StateT>
static int radix_onesweep(bg_ctx* ctx, uint64_t*& kin, uint64_t*& kout, uint32_t*& vin, uint32_t*& vout, uint64_t n, const RsDigits& dg,
                          bool* flipped) {
  const bool has_val = vin != nullptr;
  const uint64_t ntiles = (n + RS_TILE - 1) / RS_TILE;
  DevBuf<unsigned long long> hist;
  DevBuf<StateT> state;
  DevBuf<uint32_t> ticket;
  BG_TRY(hist.alloc(ctx, size_t(dg.n) * RS_BINS));
  BG_TRY(state.alloc(ctx, size_t(ntiles) * RS_BINS));
  BG_TRY(ticket.alloc(ctx, size_t(dg.n) * 16));
  BG_HIP(hipMemsetAsync(hist.p, 0, hist.bytes(), ctx->stream));
  BG_HIP(hipMemsetAsync(ticket.p, 0, ticket.bytes(), ctx->stream));
  ctx->tic("radix_hist");
  rs_hist_all_kernel<<<unsigned(std::min<uint64_t>(ntiles, 2048)), RS_THREADS, 0, ctx->stream>>>(kin, n, dg, hist.p, ntiles);
  BG_LAUNCH_CHECK();
  rs_bin_base_kernel<<<dg.n, RS_BINS, 0, ctx->stream>>>(hist.p);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 8);
  for (int d = 0; d < dg.n; ++d) {
    BG_HIP(hipMemsetAsync(state.p, 0, state.bytes(), ctx->stream));
    ctx->tic(has_val ? "radix_scatter_kv" : "radix_scatter_k");
    if (has_val)
      rs_onesweep_kernel<OffT, StateT, true><<<unsigned(ntiles), RS_THREADS, 0, ctx->stream>>>(
          kin, vin, kout, vout, n, dg.shift[d], dg.mask[d], hist.p + size_t(d) * RS_BINS, state.p, ticket.p + d * 16);
    else
      rs_onesweep_kernel<OffT, StateT, false><<<unsigned(ntiles), RS_THREADS, 0, ctx->stream>>>(
          kin, nullptr, kout, nullptr, n, dg.shift[d], dg.mask[d], hist.p + size_t(d) * RS_BINS, state.p, ticket.p + d * 16);
    BG_LAUNCH_CHECK();
    ctx->toc(n * (has_val ? 24 : 16));
    std::swap(kin, kout);
    std::swap(vin, vout);
    *flipped = !*flipped;
  }
  return BG_OK;
}

template <class OffT>
static int radix_sort_impl(bg_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                           uint64_t n, int bit_lo, int bit_hi, bool* in_b, int flags) {
  *in_b = false;
  if (n <= 1 || bit_hi <= bit_lo) return BG_OK;
  const bool has_val = vals_a != nullptr;
  const uint64_t ntiles = (n + RS_TILE - 1) / RS_TILE;

  // census -> which byte-aligned digits actually vary (skipped for low-entropy keys, whose single
  // pass is always needed and whose census would serialise on one LDS address)
  const bool low_entropy = (flags & RS_LOW_ENTROPY) != 0;
  unsigned long long h_ao[2] = {~0ull, 0ull};  // AND, OR of the keys
  if (!low_entropy) {
    DevBuf<unsigned long long> ao;
    BG_TRY(ao.alloc(ctx, 2));
    BG_HIP(hipMemcpyAsync(ao.p, h_ao, 16, hipMemcpyHostToDevice, ctx->stream));
    // A digit that varies among the first million keys varies: a big input is only read in full (0.4 ms per
    // 150 M keys) when that sample leaves some digit of the sorted bits looking constant.
    const uint64_t kSample = 1ull << 20;
    for (uint64_t m : {n > 4 * kSample ? kSample : n, n}) {
      const uint64_t mt = (m + RS_TILE - 1) / RS_TILE;
      ctx->tic("radix_census");
      rs_and_or_kernel<<<unsigned(mt < 2048 ? mt : 2048), RS_THREADS, 0, ctx->stream>>>(keys_a, m, ao.p);
      BG_LAUNCH_CHECK();
      ctx->toc(m * 8);
      BG_HIP(hipMemcpyAsync(h_ao, ao.p, 16, hipMemcpyDeviceToHost, ctx->stream));
      BG_HIP(hipStreamSynchronize(ctx->stream));
      if (m == n) break;
      bool all_vary = true;
      for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        const int bits = bit_hi - shift < 8 ? bit_hi - shift : 8;
        if ((((h_ao[0] ^ h_ao[1]) >> shift) & ((1u << bits) - 1)) == 0) all_vary = false;
      }
      if (all_vary) break;  // (AND / OR of the sample: every pass is needed, which is all the census is for)
      // ao keeps the sample's AND / OR: the full pass only refines them
    }
  }

  uint64_t* kin = keys_a;
  uint64_t* kout = keys_b;
  uint32_t* vin = vals_a;
  uint32_t* vout = vals_b;
  bool flipped = false;
  // Measured on chr20's sorts (4 passes over 158 M and over 51 M keys): the histogram passes and scans go (2.4 -> 0.7 ms) but
  // the scatter passes take 6.6 ms instead of 4.6 -- tiles in ticket order land on the XCDs round-robin, so the two tiles that
  // share a bin's line write it from different L2s -- and the step is no faster (62.0 against 61.8 ms): off unless asked for.
  static const bool no_onesweep = [] { const char* e = getenv("BG_RADIX_ONESWEEP"); return !(e && e[0] == '1'); }();
  if (!low_entropy && !no_onesweep && ntiles >= 64) {
    // every digit that varies, least significant first; eight digits per histogram pass
    RsDigits dg;
    dg.n = 0;
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
      const int bits = bit_hi - shift < 8 ? bit_hi - shift : 8;
      const uint32_t dmask = (1u << bits) - 1;
      if ((((h_ao[0] ^ h_ao[1]) >> shift) & dmask) == 0) continue;
      dg.shift[dg.n] = shift;
      dg.mask[dg.n] = dmask;
      ++dg.n;
    }
    if (dg.n) {
      if (n < (1ull << 30)) BG_TRY((radix_onesweep<OffT, uint32_t>(ctx, kin, kout, vin, vout, n, dg, &flipped)));
      else BG_TRY((radix_onesweep<OffT, unsigned long long>(ctx, kin, kout, vin, vout, n, dg, &flipped)));
    }
    *in_b = flipped;
    return BG_OK;
  }

  DevBuf<uint32_t> hist;
  DevBuf<OffT> offs;
  BG_TRY(hist.alloc(ctx, size_t(RS_BINS) * ntiles));
  BG_TRY(offs.alloc(ctx, size_t(RS_BINS) * ntiles));

  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    int bits = bit_hi - shift < 8 ? bit_hi - shift : 8;
    uint32_t dmask = (1u << bits) - 1;
    // skip when every key has the same value in this digit
    if (!low_entropy && (((h_ao[0] ^ h_ao[1]) >> shift) & dmask) == 0) continue;
    ctx->tic("radix_hist");
    if (low_entropy)
      rs_hist_agg_kernel<<<unsigned(ntiles), RS_THREADS, 0, ctx->stream>>>(kin, n, shift, dmask, hist.p,
                                                                           uint32_t(ntiles));
    else
      rs_hist_kernel<false><<<unsigned(ntiles), RS_THREADS, 0, ctx->stream>>>(kin, n, shift, dmask, hist.p,
                                                                              uint32_t(ntiles));
    BG_LAUNCH_CHECK();
    ctx->toc(n * 8);
    ctx->tic("radix_scan");
    BG_TRY(exclusive_scan_impl<OffT>(ctx, hist.p, offs.p, uint64_t(RS_BINS) * ntiles, nullptr));
    ctx->toc(uint64_t(RS_BINS) * ntiles * (4 + sizeof(OffT)));
    ctx->tic(has_val ? "radix_scatter_kv" : "radix_scatter_k");
    if (has_val)
      rs_scatter_kernel<OffT, true, false><<<unsigned(ntiles), RS_THREADS, 0, ctx->stream>>>(kin, vin, kout, vout, n, shift,
                                                                                      dmask, offs.p, uint32_t(ntiles));
    else
      rs_scatter_kernel<OffT, false, false><<<unsigned(ntiles), RS_THREADS, 0, ctx->stream>>>(
          kin, nullptr, kout, nullptr, n, shift, dmask, offs.p, uint32_t(ntiles));
    BG_LAUNCH_CHECK();
    ctx->toc(n * (has_val ? 24 : 16));
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
    flipped = !flipped;
  }
  *in_b = flipped;
  return BG_OK;
}

int radix_sort_u64(bg_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint64_t n,
                   int bit_lo, int bit_hi, bool* in_b, int flags) {
  if (use_off32(n))
    return radix_sort_impl<uint32_t>(ctx, keys_a, keys_b, vals_a, vals_b, n, bit_lo, bit_hi, in_b, flags);
  return radix_sort_impl<uint64_t>(ctx, keys_a, keys_b, vals_a, vals_b, n, bit_lo, bit_hi, in_b, flags);
}

// Groups equal keys' neighbourhoods: stable partition of the keys into 65536 buckets by the top 16 bits
// of key * kPartitionMult (two 8-bit passes, least significant byte first).  Equal keys land in the
// same bucket; bucket b = keys whose hash prefix is b, buckets ascending.
// The partition only has to GROUP keys, so it runs most-significant byte first and unstable: a key's
// rank inside its (tile, bin) run is the return value of one LDS atomic instead of the ballot match
// a stable sort needs.  Pass 1 splits the whole array on hash bits 56..63; pass 2 splits each of the
// 256 buckets on bits 48..55 with tiles that do not straddle buckets (segments), so one flat scan of
// the [segment][bin][tile] counts yields the final offsets.
constexpr int PT_ITEMS = 16;
constexpr int PT_TILE = RS_THREADS * PT_ITEMS;  // keys per partition tile (runs of PT_TILE/256 keys per bin)
struct __attribute__((aligned(8))) Key2 { uint64_t a, b; };  // 8-byte aligned 16-byte access
struct PartSegs {
  const uint64_t* seg_key0;   // S+1 key offsets
  const uint32_t* seg_tile0;  // S+1 tile offsets
  int nseg;
};

__device__ __forceinline__ void part_locate(const PartSegs& sg, uint32_t tile, int* seg, uint64_t* lo, uint64_t* hi,
                                            uint64_t* hist_base, uint32_t* tiles_in_seg, uint32_t* tile_in_seg) {
  int a = 0, b = sg.nseg;  // last segment with seg_tile0 <= tile
  while (b - a > 1) {
    int m = (a + b) >> 1;
    if (sg.seg_tile0[m] <= tile) a = m; else b = m;
  }
  uint32_t t0 = sg.seg_tile0[a];
  *seg = a;
  *tiles_in_seg = sg.seg_tile0[a + 1] - t0;
  *tile_in_seg = tile - t0;
  *lo = sg.seg_key0[a] + uint64_t(tile - t0) * PT_TILE;
  uint64_t end = sg.seg_key0[a + 1];
  *hi = *lo + PT_TILE < end ? *lo + PT_TILE : end;
  *hist_base = uint64_t(t0) * RS_BINS;
}

__global__ __launch_bounds__(RS_THREADS) void rs_part_hist_kernel(const uint64_t* __restrict__ keys, PartSegs sg,
                                                                  int shift, uint32_t dmask,
                                                                  uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  int seg;
  uint64_t lo, hi, hbase;
  uint32_t nt, ti;
  part_locate(sg, blockIdx.x, &seg, &lo, &hi, &hbase, &nt, &ti);
#pragma unroll 4
  for (int it = 0; it < PT_ITEMS / 2; ++it) {  // two keys (16 bytes) per lane and load
    uint64_t idx = lo + (uint64_t(it) * RS_THREADS + threadIdx.x) * 2;
    if (idx + 1 < hi) {
      Key2 v = *reinterpret_cast<const Key2*>(keys + idx);
      atomicAdd(&h[rs_digit<true>(v.a, shift, dmask)], 1u);
      atomicAdd(&h[rs_digit<true>(v.b, shift, dmask)], 1u);
    } else if (idx < hi) {
      atomicAdd(&h[rs_digit<true>(keys[idx], shift, dmask)], 1u);
    }
  }
  __syncthreads();
  hist[hbase + uint64_t(threadIdx.x) * nt + ti] = h[threadIdx.x];
}

template <class OffT>
__global__ __launch_bounds__(RS_THREADS) void rs_part_scatter_kernel(const uint64_t* __restrict__ kin,
                                                                     uint64_t* __restrict__ kout, PartSegs sg,
                                                                     int shift, uint32_t dmask,
                                                                     const OffT* __restrict__ offs) {
  __shared__ uint64_t skeys[PT_TILE];
  __shared__ uint32_t h[RS_BINS];
  __shared__ uint32_t binstart[RS_BINS];
  __shared__ OffT goff[RS_BINS];
  __shared__ uint32_t scan_sm[4];
  const int tid = threadIdx.x;
  h[tid] = 0;
  int seg;
  uint64_t lo, hi, hbase;
  uint32_t nt, ti;
  part_locate(sg, blockIdx.x, &seg, &lo, &hi, &hbase, &nt, &ti);
  __syncthreads();
  uint64_t key[PT_ITEMS];
  uint16_t rnk[PT_ITEMS];
#pragma unroll
  for (int it = 0; it < PT_ITEMS; it += 2) {  // two keys (16 bytes) per lane and load
    uint64_t idx = lo + (uint64_t(it / 2) * RS_THREADS + tid) * 2;
    if (idx + 1 < hi) {
      Key2 v = *reinterpret_cast<const Key2*>(kin + idx);
      key[it] = v.a;
      key[it + 1] = v.b;
    } else {
      key[it] = idx < hi ? kin[idx] : 0;
      key[it + 1] = 0;
    }
    rnk[it] = idx < hi ? uint16_t(atomicAdd(&h[rs_digit<true>(key[it], shift, dmask)], 1u)) : uint16_t(0);
    rnk[it + 1] = idx + 1 < hi ? uint16_t(atomicAdd(&h[rs_digit<true>(key[it + 1], shift, dmask)], 1u)) : uint16_t(0);
  }
  __syncthreads();
  {
    uint32_t tot = h[tid], dummy;
    uint32_t ex = block_excl_scan_256<uint32_t>(tot, scan_sm, &dummy);
    binstart[tid] = ex;
    goff[tid] = offs[hbase + uint64_t(tid) * nt + ti] - OffT(ex);
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < PT_ITEMS; ++it) {
    uint64_t idx = lo + (uint64_t(it / 2) * RS_THREADS + tid) * 2 + (it & 1);
    if (idx < hi) skeys[binstart[rs_digit<true>(key[it], shift, dmask)] + rnk[it]] = key[it];
  }
  __syncthreads();
  uint32_t cnt = uint32_t(hi - lo);
  for (uint32_t i = tid; i < cnt; i += RS_THREADS) {
    uint64_t k = skeys[i];
    kout[goff[rs_digit<true>(k, shift, dmask)] + OffT(i)] = k;
  }
}

template <class OffT>
__global__ void rs_part_bucket_starts_kernel(const OffT* __restrict__ offs, uint32_t ntiles, uint64_t n,
                                             uint64_t* __restrict__ starts) {
  int b = threadIdx.x;
  starts[b] = uint64_t(offs[uint64_t(b) * ntiles]);
  if (b == 0) starts[RS_BINS] = n;
}

template <class OffT>
static int radix_partition_hash16_impl(bg_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint64_t n, bool* in_b,
                                       const uint64_t* first_in = nullptr) {
  *in_b = false;
  if (n <= 1) return BG_OK;
  const uint64_t ntiles1 = (n + PT_TILE - 1) / PT_TILE;
  const uint64_t max_tiles = ntiles1 + RS_BINS;
  DevBuf<uint32_t> hist;
  DevBuf<OffT> offs;
  DevBuf<uint64_t> seg_key;
  DevBuf<uint32_t> seg_tile;
  BG_TRY(hist.alloc(ctx, size_t(RS_BINS) * max_tiles));
  BG_TRY(offs.alloc(ctx, size_t(RS_BINS) * max_tiles));
  BG_TRY(seg_key.alloc(ctx, RS_BINS + 1));
  BG_TRY(seg_tile.alloc(ctx, RS_BINS + 1));
  std::vector<uint64_t> h_key(RS_BINS + 1, 0);
  std::vector<uint32_t> h_tile(RS_BINS + 1, 0);
  const uint64_t* kin = first_in ? first_in : keys_a;  // (first_in: the keys are only read; keys_a is scratch like keys_b)
  uint64_t* kout = keys_b;
  for (int pass = 0; pass < 2; ++pass) {
    const int shift = pass == 0 ? 56 : 48;
    int nseg;
    if (pass == 0) {
      nseg = 1;
      h_key[0] = 0; h_key[1] = n;
      h_tile[0] = 0; h_tile[1] = uint32_t(ntiles1);
    } else {
      nseg = RS_BINS;
      uint32_t t = 0;
      for (int b = 0; b < RS_BINS; ++b) {
        h_tile[b] = t;
        t += uint32_t((h_key[b + 1] - h_key[b] + PT_TILE - 1) / PT_TILE);
      }
      h_tile[RS_BINS] = t;
    }
    const uint32_t ntiles = h_tile[nseg];
    BG_HIP(hipMemcpyAsync(seg_key.p, h_key.data(), (nseg + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(seg_tile.p, h_tile.data(), (nseg + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors are rewritten below
    PartSegs sg{seg_key.p, seg_tile.p, nseg};
    ctx->tic("radix_hist");
    rs_part_hist_kernel<<<ntiles, RS_THREADS, 0, ctx->stream>>>(kin, sg, shift, 0xffu, hist.p);
    BG_LAUNCH_CHECK();
    ctx->toc(n * 8);
    ctx->tic("radix_scan");
    BG_TRY(exclusive_scan_impl<OffT>(ctx, hist.p, offs.p, uint64_t(RS_BINS) * ntiles, nullptr));
    ctx->toc(uint64_t(RS_BINS) * ntiles * (4 + sizeof(OffT)));
    ctx->tic("radix_scatter_k");
    rs_part_scatter_kernel<OffT><<<ntiles, RS_THREADS, 0, ctx->stream>>>(kin, kout, sg, shift, 0xffu, offs.p);
    BG_LAUNCH_CHECK();
    ctx->toc(n * 16);
    if (pass == 0) {
      rs_part_bucket_starts_kernel<OffT><<<1, RS_BINS, 0, ctx->stream>>>(offs.p, uint32_t(ntiles1), n, seg_key.p);
      BG_LAUNCH_CHECK();
      BG_HIP(hipMemcpyAsync(h_key.data(), seg_key.p, (RS_BINS + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
      BG_HIP(hipStreamSynchronize(ctx->stream));
    }
    kin = kout;
    kout = kout == keys_b ? keys_a : keys_b;
    *in_b = !*in_b;
  }
  return BG_OK;
}

// ---- the same partition with pass 1 generating its keys straight from the packed reads: the canonical
// k-mers of a tile of reads are produced in registers (wave per read, lane per window, as k_extract_kmers
// does), so the 8-byte instances are written once (by pass 1's scatter) instead of written by an
// extraction kernel and read back twice.
struct ReadKeySource {
  const uint64_t* words;
  const uint32_t* nmask;
  const uint16_t* len;
  uint64_t n_reads;
  int W, k, reads_per_tile;
};

// calls f(key) for every window of reads [r_lo, r_hi): canonical k-mer, or all-ones when the window holds
// an N (kmer_counter.h:306-311).  Wave w takes reads r_lo + w, r_lo + w + 4, ...
template <class F>
__device__ __forceinline__ void tile_kmers(const ReadKeySource& src, uint64_t r_lo, uint64_t r_hi, F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = src.W, k = src.k;
  for (uint64_t i = r_lo + wave; i < r_hi; i += RS_THREADS / 64) {
    const int L = src.len[i];
    if (L < k) continue;
    const int nk = L - k + 1;
    uint64_t myw = lane < W ? src.words[i * W + lane] : 0;
    uint32_t myn = lane < W ? src.nmask[i * W + lane] : 0;
    for (int j0 = 0; j0 < nk; j0 += 64) {
      int j = j0 + lane;
      int q = j >> 5, r = j & 31;
      int q0 = q < 63 ? q : 63, q1 = q + 1 < 63 ? q + 1 : 63;
      uint64_t w0 = __shfl(myw, q0, 64), w1 = __shfl(myw, q1, 64);
      uint32_t n0 = __shfl(myn, q0, 64), n1 = __shfl(myn, q1, 64);
      if (j < nk) {
        uint64_t x = r ? ((w0 << (2 * r)) | (w1 >> (64 - 2 * r))) : w0;
        uint32_t nm = r ? ((n0 << r) | (n1 >> (32 - r))) : n0;
        bool fl;
        uint64_t canon = kmer_canon(x >> (64 - 2 * k), k, &fl);
        f((nm >> (32 - k)) != 0 ? ~0ull : canon);
      }
    }
  }
}

__global__ __launch_bounds__(RS_THREADS) void rs_part_hist_reads_kernel(ReadKeySource src, int shift,
                                                                        uint32_t* __restrict__ hist, uint32_t ntiles) {
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  uint64_t r_lo = uint64_t(blockIdx.x) * src.reads_per_tile;
  uint64_t r_hi = r_lo + src.reads_per_tile < src.n_reads ? r_lo + src.reads_per_tile : src.n_reads;
  tile_kmers(src, r_lo, r_hi, [&](uint64_t key) { atomicAdd(&h[rs_digit<true>(key, shift, 0xffu)], 1u); });
  __syncthreads();
  hist[uint64_t(threadIdx.x) * ntiles + blockIdx.x] = h[threadIdx.x];
}

template <class OffT>
__global__ __launch_bounds__(RS_THREADS) void rs_part_scatter_reads_kernel(ReadKeySource src,
                                                                           uint64_t* __restrict__ kout, int shift,
                                                                           const OffT* __restrict__ offs,
                                                                           const uint32_t* __restrict__ hist,
                                                                           uint32_t ntiles) {
  __shared__ uint64_t skeys[PT_TILE];
  __shared__ uint32_t cursor[RS_BINS];
  __shared__ OffT goff[RS_BINS];
  __shared__ uint32_t scan_sm[4];
  const int tid = threadIdx.x;
  uint64_t r_lo = uint64_t(blockIdx.x) * src.reads_per_tile;
  uint64_t r_hi = r_lo + src.reads_per_tile < src.n_reads ? r_lo + src.reads_per_tile : src.n_reads;
  uint32_t total;
  {
    uint32_t tot = hist[uint64_t(tid) * ntiles + blockIdx.x];  // this tile's column of the histogram pass
    uint32_t ex = block_excl_scan_256<uint32_t>(tot, scan_sm, &total);
    cursor[tid] = ex;
    goff[tid] = offs[uint64_t(tid) * ntiles + blockIdx.x] - OffT(ex);
  }
  __syncthreads();
  // arrival order inside a bin is free: one LDS atomic gives a key its slot
  tile_kmers(src, r_lo, r_hi, [&](uint64_t key) {
    skeys[atomicAdd(&cursor[rs_digit<true>(key, shift, 0xffu)], 1u)] = key;
  });
  __syncthreads();
  for (uint32_t i = tid; i < total; i += RS_THREADS) {
    uint64_t k = skeys[i];
    kout[goff[rs_digit<true>(k, shift, 0xffu)] + OffT(i)] = k;
  }
}

// One unstable partition pass over `nseg` segments (key offsets h_key[0..nseg]) on hash bits
// shift..shift+7 under dmask: tiles never straddle segments, so the flat scan of [segment][bin][tile]
// counts yields final offsets.
template <class OffT>
static int radix_partition_segmented(bg_ctx* ctx, uint64_t* kin, uint64_t* kout, uint64_t n, uint32_t nseg,
                                     const uint64_t* h_key, int shift, uint32_t dmask) {
  std::vector<uint32_t> h_tile(size_t(nseg) + 1, 0);
  uint64_t t = 0;
  for (uint32_t b = 0; b < nseg; ++b) {
    h_tile[b] = uint32_t(t);
    t += (h_key[b + 1] - h_key[b] + PT_TILE - 1) / PT_TILE;
  }
  if (t >= (1ull << 31)) { set_error("partition of %llu keys needs too many tiles", (unsigned long long)n); return BG_EINVAL; }
  h_tile[nseg] = uint32_t(t);
  const uint32_t ntiles = uint32_t(t);
  if (!ntiles) return BG_OK;
  DevBuf<uint32_t> hist, seg_tile;
  DevBuf<OffT> offs;
  DevBuf<uint64_t> seg_key;
  BG_TRY(hist.alloc(ctx, size_t(RS_BINS) * ntiles));
  BG_TRY(offs.alloc(ctx, size_t(RS_BINS) * ntiles));
  BG_TRY(seg_key.alloc(ctx, size_t(nseg) + 1));
  BG_TRY(seg_tile.alloc(ctx, size_t(nseg) + 1));
  BG_HIP(hipMemcpyAsync(seg_key.p, h_key, (size_t(nseg) + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemcpyAsync(seg_tile.p, h_tile.data(), (size_t(nseg) + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  PartSegs sg{seg_key.p, seg_tile.p, int(nseg)};
  ctx->tic("radix_hist");
  rs_part_hist_kernel<<<ntiles, RS_THREADS, 0, ctx->stream>>>(kin, sg, shift, dmask, hist.p);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 8);
  ctx->tic("radix_scan");
  BG_TRY(exclusive_scan_impl<OffT>(ctx, hist.p, offs.p, uint64_t(RS_BINS) * ntiles, nullptr));
  ctx->toc(uint64_t(RS_BINS) * ntiles * (4 + sizeof(OffT)));
  ctx->tic("radix_scatter_k");
  rs_part_scatter_kernel<OffT><<<ntiles, RS_THREADS, 0, ctx->stream>>>(kin, kout, sg, shift, dmask, offs.p);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 16);
  return BG_OK;
}

template <class OffT>
static int radix_partition_hash16_second(bg_ctx* ctx, uint64_t* kin, uint64_t* kout, uint64_t n,
                                         const uint64_t* h_key /* 257 bucket starts */) {
  return radix_partition_segmented<OffT>(ctx, kin, kout, n, RS_BINS, h_key, 48, 0xffu);
}

// first position whose top `bits` hash bits are >= b, for b = 0 .. 2^bits (keys grouped by those bits)
__global__ void rs_part_bounds_kernel(const uint64_t* __restrict__ keys, uint64_t n, int bits, uint64_t* __restrict__ bounds) {
  uint64_t b = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b > (1ull << bits)) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (((keys[mid] * kPartitionMult) >> (64 - bits)) < b) lo = mid + 1; else hi = mid;
  }
  bounds[b] = lo;
}

// pass 1 alone: instances grouped by hash bits 56..63 in keys_out, bucket starts (257) on the host
template <class OffT>
static int radix_partition_first_reads_impl(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n, uint64_t* keys_out,
                                            uint64_t* h_starts) {
  const int maxw = int(reads->max_len) - k + 1;
  ReadKeySource src{reads->words.p, reads->nmask.p, reads->len.p, reads->n, reads->W, k,
                    maxw > 0 ? std::max(1, PT_TILE / maxw) : 1};
  const uint32_t ntiles = uint32_t((reads->n + src.reads_per_tile - 1) / src.reads_per_tile);
  DevBuf<uint32_t> hist;
  DevBuf<OffT> offs;
  DevBuf<uint64_t> starts;
  BG_TRY(hist.alloc(ctx, size_t(RS_BINS) * ntiles));
  BG_TRY(offs.alloc(ctx, size_t(RS_BINS) * ntiles));
  BG_TRY(starts.alloc(ctx, RS_BINS + 1));
  ctx->tic("radix_hist");
  rs_part_hist_reads_kernel<<<ntiles, RS_THREADS, 0, ctx->stream>>>(src, 56, hist.p, ntiles);
  BG_LAUNCH_CHECK();
  ctx->toc(reads->n * uint64_t(reads->W) * 12);
  ctx->tic("radix_scan");
  BG_TRY(exclusive_scan_impl<OffT>(ctx, hist.p, offs.p, uint64_t(RS_BINS) * ntiles, nullptr));
  ctx->toc(uint64_t(RS_BINS) * ntiles * (4 + sizeof(OffT)));
  ctx->tic("kmer_extract_partition");
  rs_part_scatter_reads_kernel<OffT><<<ntiles, RS_THREADS, 0, ctx->stream>>>(src, keys_out, 56, offs.p, hist.p, ntiles);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 8 + reads->n * uint64_t(reads->W) * 12);
  rs_part_bucket_starts_kernel<OffT><<<1, RS_BINS, 0, ctx->stream>>>(offs.p, ntiles, n, starts.p);
  BG_LAUNCH_CHECK();
  BG_HIP(hipMemcpyAsync(h_starts, starts.p, (RS_BINS + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

template <class OffT>
static int radix_partition_hash16_reads_impl(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n,
                                             uint64_t* keys_a, uint64_t* keys_b) {
  std::vector<uint64_t> h_key(RS_BINS + 1, 0);
  BG_TRY(radix_partition_first_reads_impl<OffT>(ctx, reads, k, n, keys_a, h_key.data()));
  return radix_partition_hash16_second<OffT>(ctx, keys_a, keys_b, n, h_key.data());
}

// First pass only (the sharded build sends whole hash-byte buckets to their owner ranks).
int radix_partition_first_reads(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n, uint64_t* keys_out,
                                uint64_t* h_starts) {
  for (int b = 0; b <= RS_BINS; ++b) h_starts[b] = 0;
  if (!n) return BG_OK;
  if (use_off32(n)) return radix_partition_first_reads_impl<uint32_t>(ctx, reads, k, n, keys_out, h_starts);
  return radix_partition_first_reads_impl<uint64_t>(ctx, reads, k, n, keys_out, h_starts);
}

// Partition of the k-mer instances of `reads` (n of them in all, sentinels included) into keys_b.
int radix_partition_hash16_reads(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n, uint64_t* keys_a,
                                 uint64_t* keys_b) {
  if (!n) return BG_OK;
  if (use_off32(n)) return radix_partition_hash16_reads_impl<uint32_t>(ctx, reads, k, n, keys_a, keys_b);
  return radix_partition_hash16_reads_impl<uint64_t>(ctx, reads, k, n, keys_a, keys_b);
}

// Keys grouped by 16 hash bits (kin) -> grouped by 16 + extra_bits (kout), extra_bits in 1..8.
int radix_partition_hash_refine(bg_ctx* ctx, uint64_t* kin, uint64_t* kout, uint64_t n, int extra_bits) {
  if (!n) return BG_OK;
  DevBuf<uint64_t> bounds;
  BG_TRY(bounds.alloc(ctx, 65537));
  rs_part_bounds_kernel<<<cdiv(65537, 256), 256, 0, ctx->stream>>>(kin, n, 16, bounds.p);
  BG_LAUNCH_CHECK();
  std::vector<uint64_t> h(65537);
  BG_HIP(hipMemcpyAsync(h.data(), bounds.p, 65537 * 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  const uint32_t dmask = (1u << extra_bits) - 1;
  if (use_off32(n)) return radix_partition_segmented<uint32_t>(ctx, kin, kout, n, 65536, h.data(), 48 - extra_bits, dmask);
  return radix_partition_segmented<uint64_t>(ctx, kin, kout, n, 65536, h.data(), 48 - extra_bits, dmask);
}

int radix_partition_hash16(bg_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint64_t n, bool* in_b) {
  if (use_off32(n)) return radix_partition_hash16_impl<uint32_t>(ctx, keys_a, keys_b, n, in_b);
  return radix_partition_hash16_impl<uint64_t>(ctx, keys_a, keys_b, n, in_b);
}
// the same with the keys read from `in` (left as they are); the result is in scratch_a (two passes: in -> b -> a)
int radix_partition_hash16_from(bg_ctx* ctx, const uint64_t* in, uint64_t* scratch_a, uint64_t* scratch_b, uint64_t n) {
  bool in_b = false;
  if (n <= 1) {
    if (n) BG_HIP(hipMemcpyAsync(scratch_a, in, 8, hipMemcpyDeviceToDevice, ctx->stream));
    return BG_OK;
  }
  if (use_off32(n)) return radix_partition_hash16_impl<uint32_t>(ctx, scratch_a, scratch_b, n, &in_b, in);
  return radix_partition_hash16_impl<uint64_t>(ctx, scratch_a, scratch_b, n, &in_b, in);
}
// first position whose top `bits` bits of key * kPartitionMult are >= b, for b = 0 .. 2^bits (device array of 2^bits + 1)
int radix_partition_bounds(bg_ctx* ctx, const uint64_t* keys, uint64_t n, int bits, uint64_t* d_bounds) {
  rs_part_bounds_kernel<<<cdiv((1ull << bits) + 1, 256), 256, 0, ctx->stream>>>(keys, n, bits, d_bounds);
  BG_LAUNCH_CHECK();
  return BG_OK;
}

}  // namespace bg
