// items_build.hip — the entry set from 16-byte ITEMS instead of materialised rows (the default build).
//
// A suffix record (oriented read r, offset j) of the exact cover travels as
//     fw  = its first 32 bases (one word, zero padded), and
//     id  = r << 24 | j << 16 | length << 8 | flags        (flags: bit 0 the suffix one base earlier is a record too,
//                                                            bits 1-2 the base in front of it, bit 3 the read's last record;
//                                                            bits 55..62: its LCP with its successor, bit 63: that note is set)
// and is put in order by two 256-way partition passes on the leading bits of fw (the first makes the items from the reads as
// it scatters them; a third pass beyond ~170 M records) and one LDS sort per bucket; only items that tie on all 32 bases go back
// to the read for the rest of their string (k_items_tie_small / _wave / _block; what those cannot take goes through the rows
// machinery as a subset).  Stream passes then drop the prefixes (k_items_keep), number and write the entries with their sizes,
// shared lengths, local prev bits and the prefix directory's heads (k_items_entries), and search the tails the local rule
// leaves through the entries' first words and the reads (k_prev_list_items).  Nothing of 48 bytes per record is written,
// sorted by index and gathered back (k_emit_records / radix passes / k_gather_dedup_build: BG_ITEMS_BUILD=0), and no key table
// is made in the step (seqset_ensure_keys makes it for the consumers that ask).  items_cover_order sorts the oriented reads
// for the cover index with the same kernels (a read is the item of its suffix 0).
// The same strings as SURVEY rows D1-D5 / E (expand.cpp:285-380, builder.cpp:85-108) in another order of work; no reference
// code has this shape.  BG_ITEMS_PROBE=1 runs the front half beside the build and reports times and bucket sizes.
#include "bg_device_funcs.h"
#include "bg_internal.h"
#include "items_entries.h"
#include "compact.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace bg {


constexpr int IT_THREADS = 512;  // (256 / 512 / 1 024 threads over the same 2 048 items: the four partition kernels 3.49 / 3.32 / 3.46 ms)
constexpr int IT_TILE = 2048;  // items staged in LDS: 32 KB
constexpr int IT_PER = IT_TILE / IT_THREADS;
constexpr int IT_BINS = 256;
static_assert(IT_THREADS >= IT_BINS && IT_THREADS % 64 == 0, "a thread a bin in the partition kernels");

// One wave expands 64 oriented reads (k_emit_records' way of walking the records: a lane per RECORD, its read found by a
// search over the 64 record offsets held one per lane): 16 bytes a record, written in runs.
__global__ __launch_bounds__(256) void k_emit_items(const uint64_t* __restrict__ owords, const uint16_t* __restrict__ len,
                                                    const uint32_t* __restrict__ cnt, const uint64_t* __restrict__ roff,
                                                    uint64_t o_base, uint64_t n_emit, int W, Item* __restrict__ items) {
  const int lane = threadIdx.x & 63;
  const uint64_t t0 = (uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 64;
  if (t0 >= n_emit) return;
  const uint64_t t = t0 + lane;
  const bool live = t < n_emit;
  const uint64_t my_off = live ? roff[t] : ~0ull;
  const uint32_t my_cnt = live ? cnt[t] : 0u;
  const int my_len = live ? int(len[(o_base + t) >> 1]) : 0;
  const uint64_t r0 = __shfl(my_off, 0, 64);
  uint64_t r1 = live ? my_off + my_cnt : 0;
#pragma unroll
  for (int d = 32; d; d >>= 1) {
    const uint64_t o = __shfl_xor(r1, d, 64);
    r1 = o > r1 ? o : r1;
  }
  const uint32_t nrec = uint32_t(r1 - r0);
  const uint32_t my_rel = live ? uint32_t(my_off - r0) : ~0u;
  for (uint32_t x0 = 0; x0 < nrec; x0 += 64) {
    const uint32_t xq = x0 + uint32_t(lane);
    int lo = 0;
#pragma unroll
    for (int step = 32; step; step >>= 1) {
      const uint32_t v = uint32_t(__shfl(int(my_rel), lo + step, 64));
      if (v <= xq) lo += step;
    }
    const uint32_t off = uint32_t(__shfl(int(my_rel), lo, 64));
    const int L = __shfl(my_len, lo, 64);
    const uint32_t rcnt = uint32_t(__shfl(int(my_cnt), lo, 64));
    if (xq < nrec) {
      const int j = int(xq - off);
      const uint64_t ro = o_base + t0 + uint64_t(lo);
      const uint64_t* s = owords + ro * uint64_t(W);
      const int sq = j >> 5, sr = j & 31;
      const uint64_t cur = s[sq], next = sq + 1 < W ? s[sq + 1] : 0;
      Item it;
      it.fw = sr ? (cur << (2 * sr)) | (next >> (64 - 2 * sr)) : cur;
      uint64_t flags = 0;
      if (j > 0) flags |= 1u | (((s[(j - 1) >> 5] >> (62 - 2 * ((j - 1) & 31))) & 3) << 1);
      if (uint32_t(j) + 1 == rcnt) flags |= 8u;
      it.id = (ro << 24) | (uint64_t(j) << 16) | (uint64_t(L - j) << 8) | flags;
      items[r0 + xq] = it;
    }
  }
}

// ---- one 256-way partition pass over items grouped into segments (a segment = the items of one bin of the pass before)
struct ItSegs {
  const uint64_t* seg0;   // nseg + 1 item offsets
  const uint32_t* tile0;  // nseg + 1 tile offsets
  int nseg;
};
// (the same tables inside the kernel's arguments when there are at most 256 segments: no upload, no wait for it)
struct ItSegsArg {
  uint64_t seg0[257];
  uint32_t tile0[257];
  int nseg;
};
template <class Segs>
__device__ __forceinline__ void it_locate(const Segs& sg, uint32_t tile, uint64_t* lo, uint64_t* hi, uint64_t* hist_base, uint32_t* nt,
                                          uint32_t* ti) {
  int a = 0, b = sg.nseg;
  while (b - a > 1) {
    const int m = (a + b) >> 1;
    if (sg.tile0[m] <= tile) a = m; else b = m;
  }
  const uint32_t t0 = sg.tile0[a];
  *nt = sg.tile0[a + 1] - t0;
  *ti = tile - t0;
  *lo = sg.seg0[a] + uint64_t(tile - t0) * IT_TILE;
  const uint64_t end = sg.seg0[a + 1];
  *hi = *lo + IT_TILE < end ? *lo + IT_TILE : end;
  *hist_base = uint64_t(t0) * IT_BINS;
}

template <class Segs>
__global__ __launch_bounds__(IT_THREADS) void it_hist_kernel(const Item* __restrict__ items, const Segs sg, int shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[IT_BINS];
  if (threadIdx.x < IT_BINS) h[threadIdx.x] = 0;
  __syncthreads();
  uint64_t lo, hi, hb;
  uint32_t nt, ti;
  it_locate(sg, uint32_t(xcd_chunk(blockIdx.x, gridDim.x)), &lo, &hi, &hb, &nt, &ti);
#pragma unroll
  for (int k = 0; k < IT_PER; ++k) {
    const uint64_t i = lo + uint64_t(k) * IT_THREADS + threadIdx.x;
    if (i < hi) atomicAdd(&h[uint32_t(items[i].fw >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < IT_BINS) hist[hb + uint64_t(threadIdx.x) * nt + ti] = h[threadIdx.x];
}

__device__ __forceinline__ uint32_t it_block_excl_scan(uint32_t v, uint32_t* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) sm[wave] = x;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += sm[w];
  return base + x - v;
}

template <class Segs>
__global__ __launch_bounds__(IT_THREADS) void it_scatter_kernel(const Item* __restrict__ in, Item* __restrict__ out, const Segs sg, int shift,
                                                                const uint64_t* __restrict__ offs) {
  __shared__ Item stage[IT_TILE];
  __shared__ uint32_t h[IT_BINS];
  __shared__ uint32_t binstart[IT_BINS];
  __shared__ uint64_t goff[IT_BINS];
  __shared__ uint32_t scan_sm[IT_THREADS / 64];
  const int tid = threadIdx.x;
  if (tid < IT_BINS) h[tid] = 0;
  uint64_t lo, hi, hb;
  uint32_t nt, ti;
  it_locate(sg, uint32_t(xcd_chunk(blockIdx.x, gridDim.x)), &lo, &hi, &hb, &nt, &ti);
  __syncthreads();
  Item it[IT_PER];
  uint16_t rnk[IT_PER];
#pragma unroll
  for (int k = 0; k < IT_PER; ++k) {
    const uint64_t i = lo + uint64_t(k) * IT_THREADS + tid;
    rnk[k] = 0;
    it[k].fw = 0;
    it[k].id = 0;
    if (i < hi) {
      it[k] = in[i];
      rnk[k] = uint16_t(atomicAdd(&h[uint32_t(it[k].fw >> shift) & 255u], 1u));
    }
  }
  __syncthreads();
  {
    const uint32_t ex = it_block_excl_scan(tid < IT_BINS ? h[tid] : 0u, scan_sm);
    if (tid < IT_BINS) {
      binstart[tid] = ex;
      goff[tid] = offs[hb + uint64_t(tid) * nt + ti] - ex;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IT_PER; ++k) {
    const uint64_t i = lo + uint64_t(k) * IT_THREADS + tid;
    if (i < hi) stage[binstart[uint32_t(it[k].fw >> shift) & 255u] + rnk[k]] = it[k];
  }
  __syncthreads();
  const uint32_t cnt = uint32_t(hi - lo);
  for (uint32_t i = tid; i < cnt; i += IT_THREADS) {
    const Item x = stage[i];
    out[goff[uint32_t(x.fw >> shift) & 255u] + i] = x;
  }
}

// ---- the first level straight from the reads: a workgroup takes IR_READS oriented reads, makes their records' items in
// stretches of IT_TILE and scatters them by the top byte -- the items are never written in read order (k_emit_items: 2.5 GB
// out, read again by the histogram and again by the scatter; here the reads' words are read twice instead, 1 GB each).
#ifndef BG_IR_READS
#define BG_IR_READS 1024
#endif
constexpr int IR_READS = BG_IR_READS;
__device__ __forceinline__ uint32_t ir_offsets(const uint64_t* __restrict__ roff, uint64_t n_emit, uint64_t n_rec, uint64_t t0, uint32_t* rel,
                                               uint64_t* r0) {
  const uint64_t base = roff[t0];
  for (int i = threadIdx.x; i <= IR_READS; i += IT_THREADS) {
    const uint64_t t = t0 + uint64_t(i);
    rel[i] = uint32_t((t < n_emit ? roff[t] : n_rec) - base);
  }
  __syncthreads();
  *r0 = base;
  return rel[IR_READS];
}
__device__ __forceinline__ int ir_read_of(const uint32_t* rel, uint32_t x) {  // last read whose first record is at or before x
  int lo = 0, hi = IR_READS;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rel[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(IT_THREADS) void it_hist_reads_kernel(const uint64_t* __restrict__ owords, const uint64_t* __restrict__ roff,
                                                                   uint64_t o_base, uint64_t n_emit, uint64_t n_rec, int W, uint32_t nt,
                                                                   uint32_t* __restrict__ hist) {
  __shared__ uint32_t rel[IR_READS + 1];
  __shared__ uint32_t h[IT_BINS];
  if (threadIdx.x < IT_BINS) h[threadIdx.x] = 0;
  const uint64_t t0 = uint64_t(blockIdx.x) * IR_READS;
  uint64_t r0;
  (void)ir_offsets(roff, n_emit, n_rec, t0, rel, &r0);
  // a thread a read: its records' top bytes slide along the first words (no search; the wave waits for its longest read)
  for (int q = threadIdx.x; q < IR_READS; q += IT_THREADS) {
    const uint32_t c = rel[q + 1] - rel[q];
    if (!c) continue;
    const uint64_t* s = owords + (o_base + t0 + uint64_t(q)) * uint64_t(W);
    int wi = 0;
    uint64_t cur = s[0], nxt = W > 1 ? s[1] : 0;
    for (uint32_t j = 0; j < c; ++j) {
      const int sr = int(j & 31u);
      if (j && !sr) {
        cur = nxt;
        ++wi;
        nxt = wi + 1 < W ? s[wi + 1] : 0;
      }
      uint32_t top = uint32_t((cur << (2 * sr)) >> 56);
      if (sr > 28) top |= uint32_t(nxt >> (64 - 2 * (sr - 28)));
      atomicAdd(&h[top & 255u], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < IT_BINS) hist[uint64_t(threadIdx.x) * nt + blockIdx.x] = h[threadIdx.x];
}

template <int MODE>  // 0; 1: everything but the global writes (timing experiments, BG_ITEMS_PROBE)
__global__ __launch_bounds__(IT_THREADS) void it_scatter_reads_kernel(const uint64_t* __restrict__ owords, const uint16_t* __restrict__ len,
                                                                      const uint64_t* __restrict__ roff, uint64_t o_base, uint64_t n_emit,
                                                                      uint64_t n_rec, int W, uint32_t nt, const uint64_t* __restrict__ offs,
                                                                      Item* __restrict__ out) {
  __shared__ Item stage[IT_TILE];
  __shared__ uint32_t rel[IR_READS + 1];
  __shared__ uint32_t h[IT_BINS];
  __shared__ uint32_t binstart[IT_BINS];
  __shared__ uint64_t goff[IT_BINS];
  __shared__ uint32_t scan_sm[IT_THREADS / 64];
  const int tid = threadIdx.x;
  const uint64_t t0 = uint64_t(blockIdx.x) * IR_READS;
  uint64_t r0;
  const uint32_t total = ir_offsets(roff, n_emit, n_rec, t0, rel, &r0);
  uint64_t run = tid < IT_BINS ? offs[uint64_t(tid) * nt + blockIdx.x] : 0;  // where this workgroup's next item of bin `tid` goes
  for (uint32_t c0 = 0; c0 < total; c0 += IT_TILE) {
    if (tid < IT_BINS) h[tid] = 0;
    __syncthreads();
    Item it[IT_PER];
    uint16_t rnk[IT_PER];
    // (a thread's IT_PER records are consecutive: one search, then along the reads)
    const uint32_t x0 = c0 + uint32_t(tid) * IT_PER;
    int q = 0, j = 0;
    int cq = -1, csq = -1, cL = 0;
    uint32_t crc = 0;
    uint64_t w0 = 0, w1 = 0, wp = 0, cro = 0;
    if (x0 < total) {
      q = ir_read_of(rel, x0);
      j = int(x0 - rel[q]);
    }
#pragma unroll
    for (int k = 0; k < IT_PER; ++k) {
      const uint32_t x = x0 + uint32_t(k);
      rnk[k] = 0;
      it[k].fw = 0;
      it[k].id = 0;
      if (x < total) {
        while (rel[q + 1] <= x) { ++q; j = 0; }
        const int sq = j >> 5, sr = j & 31;
        if (q != cq || sq != csq) {  // (the words under this record: kept while the thread's records stay on them)
          cro = o_base + t0 + uint64_t(q);
          const uint64_t* s = owords + cro * uint64_t(W);
          w0 = s[sq];
          w1 = sq + 1 < W ? s[sq + 1] : 0;
          wp = sq ? s[sq - 1] : 0;
          cL = int(len[cro >> 1]);
          crc = rel[q + 1] - rel[q];
          cq = q;
          csq = sq;
        }
        it[k].fw = sr ? (w0 << (2 * sr)) | (w1 >> (64 - 2 * sr)) : w0;
        uint64_t flags = 0;
        if (j > 0) flags |= 1u | ((sr ? (w0 >> (64 - 2 * sr)) & 3 : wp & 3) << 1);
        if (uint32_t(j) + 1 == crc) flags |= 8u;
        it[k].id = (cro << 24) | (uint64_t(j) << 16) | (uint64_t(cL - j) << 8) | flags;
        rnk[k] = uint16_t(atomicAdd(&h[uint32_t(it[k].fw >> 56)], 1u));
        ++j;
      }
    }
    __syncthreads();
    {
      const uint32_t hv = tid < IT_BINS ? h[tid] : 0u;
      const uint32_t ex = it_block_excl_scan(hv, scan_sm);
      if (tid < IT_BINS) {
        binstart[tid] = ex;
        goff[tid] = run - ex;
        run += hv;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IT_PER; ++k) {
      if (x0 + uint32_t(k) < total) stage[binstart[uint32_t(it[k].fw >> 56)] + rnk[k]] = it[k];
    }
    __syncthreads();
    const uint32_t cnt = total - c0 < uint32_t(IT_TILE) ? total - c0 : uint32_t(IT_TILE);
    for (uint32_t i = tid; i < cnt; i += IT_THREADS) {
      const Item x = stage[i];
      if (MODE == 0 || (x.fw == 0x123456789abcdefull && x.id == 77))
      out[goff[uint32_t(x.fw >> 56)] + i] = x;
    }
    __syncthreads();
  }
}

// first item whose leading `bits` bits are >= v, for v = 0 .. 2^bits (items grouped by those bits)
__global__ void it_bounds_kernel(const Item* __restrict__ items, uint64_t n, int bits, uint64_t* __restrict__ bounds) {
  const uint64_t v = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v > (1ull << bits)) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if ((items[mid].fw >> (64 - bits)) < v) lo = mid + 1; else hi = mid;
  }
  bounds[v] = lo;
}

// ---- one bucket (equal leading 16 bits) in order of the first word; then the runs of equal first words by the rest
// id bits 55..63 are free: bit 63 = "the 8 bits below are this item's LCP with the item after it" (set for an item whose
// successor has the same first word: their order and LCP took the rest of their strings); the stream pass that follows needs
// no reads at all -- every other pair of neighbours differs inside the first word.
constexpr int IB_THREADS = 512;
constexpr int IB_CAP = 4096;       // items a workgroup sorts in LDS
constexpr int IB_LOG2BINS = 11;
constexpr int IB_BINS = 1 << IB_LOG2BINS;  // counting sort on the next bits, then ranks inside a bin by comparison
constexpr int IB_BIN_MAX = 96;     // members of a bin beyond which the bucket is sorted by a bitonic network instead
constexpr int IB_BIN_ROWS = 2048;  // ... and beyond which it is left to the rows machinery

// status[bucket]: 0 done, 1 too many items (copied as they were: the caller's fall-back)
// runs[]: every run of two or more equal first words, as (place of its first item) | (members, at most 65535) << 48;
// *n_runs counts them (runs past run_cap are not listed: the caller sees the count).
__global__ __launch_bounds__(IB_THREADS) void k_items_bucket_sort(const Item* __restrict__ in, Item* __restrict__ out,
                                                                  const uint64_t* __restrict__ bounds, uint8_t* __restrict__ status,
                                                                  uint64_t* __restrict__ runs, unsigned long long* __restrict__ n_runs,
                                                                  uint64_t run_cap, int bin_shift) {
  __shared__ uint64_t sfw[IB_CAP];
  // 53 KB of LDS: three workgroups a CU (66 KB with the bins' counts and starts as words of their own: two, 2.24 ms against 1.91;
  // 1 024 bins in 51 KB: 1.96)
  __shared__ uint16_t start[IB_BINS + 1];
  __shared__ uint16_t ord[IB_CAP];   // sorted place -> item
  uint32_t* const hist = reinterpret_cast<uint32_t*>(ord);  // the bins' counts, then cursors: done with before `ord` is written
  static_assert(sizeof(uint32_t) * IB_BINS <= sizeof(uint16_t) * IB_CAP, "the bins' counts live in ord");
  __shared__ uint16_t grp[IB_CAP];   // items grouped by bin (unordered inside)
  __shared__ uint32_t s_scan[IB_THREADS / 64];
  const uint32_t b = blockIdx.x;
  const uint64_t lo = bounds[b], hi = bounds[b + 1];
  const uint32_t n = uint32_t(hi - lo);
  const int tid = threadIdx.x;
  if (hi == lo) return;
  if (hi - lo > uint64_t(IB_CAP)) {
    if (tid == 0) { status[b] = 1; atomicAdd(n_runs + 3, 1ull); }
    for (uint64_t i = lo + tid; i < hi; i += IB_THREADS) out[i] = in[i];
    return;
  }
  for (int i = tid; i < IB_BINS; i += IB_THREADS) hist[i] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += IB_THREADS) {
    const uint64_t f = in[lo + i].fw;
    sfw[i] = f;
    atomicAdd(&hist[uint32_t(f >> bin_shift) & (IB_BINS - 1)], 1u);
  }
  __syncthreads();
  bool crowded = false;
  {  // exclusive scan of the bins: 4 per thread
    uint32_t v[IB_BINS / IB_THREADS], sum = 0;
#pragma unroll
    for (int k = 0; k < IB_BINS / IB_THREADS; ++k) { v[k] = hist[tid * (IB_BINS / IB_THREADS) + k]; sum += v[k]; }
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) s_scan[wave] = x;
    uint32_t fullest = 0;
#pragma unroll
    for (int k = 0; k < IB_BINS / IB_THREADS; ++k) fullest = v[k] > fullest ? v[k] : fullest;
    // a bin of a hundred and more: ranking its members against each other is quadratic -- the bucket is sorted by a bitonic
    // network over the item numbers instead (twenty buckets of the bench input; the rows machinery took them before, a
    // millisecond of launches and read-backs).  A bin of thousands is a repeat family's first word: its runs of equal
    // first words are beyond k_items_tie_block anyway -- the bucket goes the way of the oversized ones (a genome with 7.5 % of it
    // in 5000 copies of one element: 40.3 ms so, 41.6 with every crowded bucket kept here; 40.9 / 40.5 with 1 024 / 512 as the
    // bound, which leave four buckets of the bench input to the rows)
    crowded = __syncthreads_or(fullest > uint32_t(IB_BIN_MAX));
    if (__syncthreads_or(fullest > uint32_t(IB_BIN_ROWS))) {
      if (tid == 0) { status[b] = 1; atomicAdd(n_runs + 3, 1ull); }
      for (uint64_t i = lo + tid; i < hi; i += IB_THREADS) out[i] = in[i];
      return;
    }
    uint32_t base = 0;
    for (int w2 = 0; w2 < wave; ++w2) base += s_scan[w2];
    uint32_t run = base + x - sum;
#pragma unroll
    for (int k = 0; k < IB_BINS / IB_THREADS; ++k) {
      start[tid * (IB_BINS / IB_THREADS) + k] = uint16_t(run);
      hist[tid * (IB_BINS / IB_THREADS) + k] = 0;  // (now the bin's cursor)
      run += v[k];
    }
    if (tid == IB_THREADS - 1) start[IB_BINS] = uint16_t(run);
  }
  __syncthreads();
  if (crowded) {
    uint32_t np = 1;
    while (np < n) np <<= 1;  // (at most IB_CAP; the places past n hold 0xffff, which sorts last)
    for (uint32_t i = tid; i < np; i += IB_THREADS) ord[i] = i < n ? uint16_t(i) : uint16_t(0xffff);
    __syncthreads();
    auto before = [&](uint32_t a, uint32_t c) {
      if (a == 0xffffu) return false;
      if (c == 0xffffu) return true;
      const uint64_t fa = sfw[a], fc = sfw[c];
      return fa < fc || (fa == fc && a < c);
    };
    for (uint32_t k = 2; k <= np; k <<= 1)
      for (uint32_t j = k >> 1; j; j >>= 1) {
        for (uint32_t i = tid; i < np; i += IB_THREADS) {
          const uint32_t x = i ^ j;
          if (x > i) {
            const uint32_t a = ord[i], c = ord[x];
            if ((i & k) == 0 ? before(c, a) : before(a, c)) { ord[i] = uint16_t(c); ord[x] = uint16_t(a); }
          }
        }
        __syncthreads();
      }
  } else {
  for (uint32_t i = tid; i < n; i += IB_THREADS) {
    const uint32_t bin = uint32_t(sfw[i] >> bin_shift) & (IB_BINS - 1);
    grp[start[bin] + atomicAdd(&hist[bin], 1u)] = uint16_t(i);
  }
  __syncthreads();
  // place inside the bin: the number of bin mates that sort before this item (first word, then the item's index)
  for (uint32_t p = tid; p < n; p += IB_THREADS) {
    const uint32_t i = grp[p];
    const uint64_t f = sfw[i];
    const uint32_t bin = uint32_t(f >> bin_shift) & (IB_BINS - 1);
    const uint32_t bs = start[bin], be = start[bin + 1];
    uint32_t rank = 0;
    for (uint32_t q = bs; q < be; ++q) {
      const uint32_t k = grp[q];
      const uint64_t g = sfw[k];
      rank += (g < f || (g == f && k < i)) ? 1u : 0u;
    }
    ord[bs + rank] = uint16_t(i);
  }
  }
  __syncthreads();
  __shared__ uint32_t s_nruns;
  __shared__ unsigned long long s_rbase;
  if (tid == 0) s_nruns = 0;
  __syncthreads();
  uint32_t my_run[IB_CAP / IB_THREADS], my_at[IB_CAP / IB_THREADS];
  int nmine = 0;
  for (uint32_t p = tid; p < n; p += IB_THREADS) {
    const uint32_t src = ord[p];
    const uint64_t f = sfw[src];
    out[lo + p] = Item{f, in[lo + src].id};  // (the ids once more from the L2: 32 KB of LDS less a workgroup)
    if ((p == 0 || sfw[ord[p - 1]] != f) && p + 1 < n && sfw[ord[p + 1]] == f) {
      uint32_t g = 2;
      while (p + g < n && g < 65535u && sfw[ord[p + g]] == f) ++g;
      my_run[nmine] = (p << 16) | g;
      my_at[nmine] = atomicAdd(&s_nruns, 1u);
      ++nmine;
    }
  }
  __syncthreads();
  if (tid == 0 && s_nruns) s_rbase = atomicAdd(n_runs, (unsigned long long)s_nruns);
  __syncthreads();
  for (int k = 0; k < nmine; ++k) {
    const unsigned long long at = s_rbase + my_at[k];
    if (at < run_cap) runs[at] = (lo + (my_run[k] >> 16)) | (uint64_t(my_run[k] & 0xffffu) << 48);
  }
}

// The runs of equal first words (a suffix and the longer suffixes of other reads over the same place: two or three members;
// repeats make longer ones).  A lane takes a run of at most four: the rest of every member's string from its read, a sorting
// network over (rest, length), every member's LCP with its successor noted in its id.  Longer runs go to `more` (a wave a run,
// k_items_tie_wave).
struct ItRow {
  uint64_t id;
  uint64_t r[kMaxWords - 1];
};
__device__ __forceinline__ void it_load_row(ItRow* x, uint64_t id, const uint64_t* __restrict__ owords, int W) {
  x->id = id;
  const uint64_t* s = owords + (id >> 24) * uint64_t(W);
  const int j = int(id >> 16) & 255;
#pragma unroll
  for (int w = 0; w < kMaxWords - 1; ++w) x->r[w] = w < W - 1 ? it_suffix_word(s, W, j, w + 1) : 0;
}
__device__ __forceinline__ bool it_row_less(const ItRow& a, const ItRow& b) {  // (rest, length), then the id: a total order
  int c = 0;
#pragma unroll
  for (int w = 0; w < kMaxWords - 1; ++w)
    if (!c) c = a.r[w] < b.r[w] ? -1 : (a.r[w] > b.r[w] ? 1 : 0);
  if (!c) {
    const int la = int(a.id >> 8) & 255, lb = int(b.id >> 8) & 255;
    c = la < lb ? -1 : (la > lb ? 1 : 0);
  }
  return c < 0 || (c == 0 && a.id < b.id);
}
__device__ __forceinline__ void it_cswap(ItRow& a, ItRow& b) {
  if (it_row_less(b, a)) { const ItRow t = a; a = b; b = t; }
}
__device__ __forceinline__ uint64_t it_note_lcp(const ItRow& a, const ItRow& b) {  // a's id with its LCP with b, its successor
  int l = 32;
#pragma unroll
  for (int w = 0; w < kMaxWords - 1; ++w) {
    if (l == 32 * (w + 1)) l += lcp_word(a.r[w], b.r[w]);
  }
  const int la = int(a.id >> 8) & 255, lb = int(b.id >> 8) & 255;
  const int mn = la < lb ? la : lb;
  l = l < mn ? l : mn;
  return a.id | kItLcpValid | (uint64_t(l) << kItLcpShift);
}

// (the three run kernels take their list's length from the device and stride over it: no read-back between them)
__global__ __launch_bounds__(256) void k_items_tie_small(Item* __restrict__ items, const uint64_t* __restrict__ runs,
                                                         const unsigned long long* __restrict__ n_runs_dev, uint64_t run_cap,
                                                         const uint64_t* __restrict__ owords, int W, uint64_t* __restrict__ more,
                                                         unsigned long long* __restrict__ n_more, uint64_t more_cap) {
  const uint64_t n_runs = *n_runs_dev < run_cap ? *n_runs_dev : run_cap;
  for (uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n_runs; t += uint64_t(gridDim.x) * blockDim.x) {
  const uint64_t rr = runs[t];
  const uint64_t p = rr & ((1ull << 48) - 1);
  const uint32_t g = uint32_t(rr >> 48);
  if (g > 4) {
    const unsigned long long at = atomicAdd(n_more, 1ull);
    if (at < more_cap) more[at] = rr;
    continue;
  }
  const uint64_t f = items[p].fw;
  ItRow x[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (uint32_t(m) < g) {
      it_load_row(&x[m], items[p + m].id & kItIdMask, owords, W);
    } else {  // past the run: sorts behind every member
      x[m].id = ~0ull;
#pragma unroll
      for (int w = 0; w < kMaxWords - 1; ++w) x[m].r[w] = ~0ull;
    }
  }
  it_cswap(x[0], x[1]); it_cswap(x[2], x[3]); it_cswap(x[0], x[2]); it_cswap(x[1], x[3]); it_cswap(x[1], x[2]);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (uint32_t(m) >= g) break;
    uint64_t id = x[m].id;
    if (m + 1 < 4 && uint32_t(m + 1) < g) id = it_note_lcp(x[m], x[m + 1]);
    items[p + m] = Item{f, id};
  }
  }
}

// a run of 5 .. 64 members: a lane a member; every lane counts the members that sort before its own through shuffles
__global__ __launch_bounds__(256) void k_items_tie_wave(Item* __restrict__ items, const uint64_t* __restrict__ more,
                                                        const unsigned long long* __restrict__ n_more_dev, uint64_t more_cap,
                                                        const uint64_t* __restrict__ owords, int W, uint64_t* __restrict__ longer,
                                                        unsigned long long* __restrict__ n_long, uint64_t long_cap) {
  const int lane = threadIdx.x & 63;
  const uint64_t n_more = *n_more_dev < more_cap ? *n_more_dev : more_cap;
  for (uint64_t t = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; t < n_more; t += (uint64_t(gridDim.x) * blockDim.x) >> 6) {
  const uint64_t rr = more[t];
  const uint64_t p = rr & ((1ull << 48) - 1);
  const uint32_t g = uint32_t(rr >> 48);
  if (g > 64) {
    if (lane == 0) {
      const unsigned long long at = atomicAdd(n_long, 1ull);
      if (at < long_cap) longer[at] = rr;
    }
    continue;
  }
  const bool live = uint32_t(lane) < g;
  ItRow me;
  uint64_t f = 0;
  if (live) {
    const Item it = items[p + lane];
    f = it.fw;
    it_load_row(&me, it.id & kItIdMask, owords, W);
  } else {
    me.id = ~0ull;
#pragma unroll
    for (int w = 0; w < kMaxWords - 1; ++w) me.r[w] = ~0ull;
  }
  uint32_t rank = 0;
  for (uint32_t q = 0; q < g; ++q) {
    ItRow o;
    o.id = __shfl(me.id, int(q), 64);
#pragma unroll
    for (int w = 0; w < kMaxWords - 1; ++w) o.r[w] = __shfl(me.r[w], int(q), 64);
    rank += it_row_less(o, me) ? 1u : 0u;
  }
  // the member at sorted place m = the lane whose rank is m; its successor = the lane whose rank is m + 1
  ItRow nx;  // my successor's row
  {
    // lane L publishes at index rank: find, for my rank + 1, the lane that holds it
    int src = -1;
    for (uint32_t q = 0; q < g; ++q) {
      const uint32_t rq = uint32_t(__shfl(int(rank), int(q), 64));
      if (rq == rank + 1) src = int(q);
    }
    const int s2 = src < 0 ? lane : src;
    nx.id = __shfl(me.id, s2, 64);
#pragma unroll
    for (int w = 0; w < kMaxWords - 1; ++w) nx.r[w] = __shfl(me.r[w], s2, 64);
    if (live) {
      uint64_t id = me.id;
      if (src >= 0) id = it_note_lcp(me, nx);
      items[p + rank] = Item{f, id};
    }
  }
  }
}

// a run of more than 64 members (repeats; a bucket holds at most IB_CAP items): a workgroup a run; the members' rest rows and
// ids in LDS (lds_members of them fit: longer runs fetch them from the reads at every comparison), every member counts the
// members that sort before it (quadratic: such runs are few).
__global__ __launch_bounds__(1024) void k_items_tie_block(Item* __restrict__ items, const uint64_t* __restrict__ longer,
                                                         const unsigned long long* __restrict__ n_long_dev, uint64_t long_cap,
                                                         const uint64_t* __restrict__ owords, int W, uint32_t lds_members,
                                                         uint32_t max_members, unsigned long long* __restrict__ n_left) {
  extern __shared__ uint64_t tb_smem[];  // lds_members x (1 + kMaxWords - 1) words: id, rest
  __shared__ uint16_t place[IB_CAP];     // sorted place -> member
  const uint64_t n_long = *n_long_dev < long_cap ? *n_long_dev : long_cap;
  for (uint64_t li = blockIdx.x; li < n_long; li += gridDim.x) {
  const uint64_t rr = longer[li];
  const uint64_t p = rr & ((1ull << 48) - 1);
  const uint32_t g = uint32_t(rr >> 48);
  if (g > max_members) {  // (left to the caller: the rows machinery)
    if (threadIdx.x == 0) atomicAdd(n_left, 1ull);
    continue;
  }
  const bool in_lds = g <= lds_members;
  constexpr int RS = kMaxWords;  // words a member takes in LDS
  auto load = [&](uint32_t m, ItRow* x) {
    if (in_lds) {
      x->id = tb_smem[m * RS];
#pragma unroll
      for (int w = 0; w < kMaxWords - 1; ++w) x->r[w] = tb_smem[m * RS + 1 + w];
    } else {
      it_load_row(x, items[p + m].id & kItIdMask, owords, W);
    }
  };
  if (in_lds) {
    for (uint32_t m = threadIdx.x; m < g; m += blockDim.x) {
      ItRow x;
      it_load_row(&x, items[p + m].id & kItIdMask, owords, W);
      tb_smem[m * RS] = x.id;
#pragma unroll
      for (int w = 0; w < kMaxWords - 1; ++w) tb_smem[m * RS + 1 + w] = x.r[w];
    }
  }
  __syncthreads();
  {  // the members in order: a bitonic network over their numbers (every member against every other took 2 048 row
     // comparisons a thread for a run of 2 048; this takes 66)
    uint32_t np = 1;
    while (np < g) np <<= 1;  // (at most IB_CAP: max_members is)
    for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) place[i] = i < g ? uint16_t(i) : uint16_t(0xffff);
    __syncthreads();
    auto before = [&](uint32_t a, uint32_t c) {
      if (a == 0xffffu) return false;
      if (c == 0xffffu) return true;
      ItRow x, y;
      load(a, &x);
      load(c, &y);
      return it_row_less(x, y);
    };
    for (uint32_t k = 2; k <= np; k <<= 1)
      for (uint32_t j = k >> 1; j; j >>= 1) {
        for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
          const uint32_t x = i ^ j;
          if (x > i) {
            const uint32_t a = place[i], c = place[x];
            if ((i & k) == 0 ? before(c, a) : before(a, c)) { place[i] = uint16_t(c); place[x] = uint16_t(a); }
          }
        }
        __syncthreads();
      }
  }
  const uint64_t f = items[p].fw;
  // (every member's row is read before any item of the run is overwritten: the rows come from LDS, or -- a run too long for
  // it -- the ids are staged in `place`'s order first)
  uint64_t out_id[IB_CAP / 1024];
  int k = 0;
  for (uint32_t r = threadIdx.x; r < g; r += blockDim.x, ++k) {
    ItRow me;
    load(place[r], &me);
    uint64_t id = me.id;
    if (r + 1 < g) {
      ItRow nx;
      load(place[r + 1], &nx);
      id = it_note_lcp(me, nx);
    }
    out_id[k] = id;
  }
  __syncthreads();
  k = 0;
  for (uint32_t r = threadIdx.x; r < g; r += blockDim.x, ++k) items[p + r] = Item{f, out_id[k]};
  __syncthreads();  // (the rows and places are the next run's)
  }
}

// checker (the probe only): neighbours in the whole array are in (string, length) order -- strings through the reads
__global__ void k_items_check_order(const Item* __restrict__ items, uint64_t n, const uint64_t* __restrict__ owords, int W,
                                    unsigned long long* __restrict__ bad, unsigned long long* __restrict__ idsum) {
  const uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const Item a = items[p];
  atomicAdd(idsum, (unsigned long long)(a.id & ((1ull << kItLcpShift) - 1)) * 0x9E3779B97F4A7C15ull);
  if (p + 1 >= n) return;
  const Item c = items[p + 1];
  int cmp = a.fw < c.fw ? -1 : (a.fw > c.fw ? 1 : 0);
  const int la = int(a.id >> 8) & 255, lb = int(c.id >> 8) & 255;
  int l = lcp_word(a.fw, c.fw);
  if (!cmp) {
    const uint64_t* sa = owords + ((a.id & ((1ull << kItLcpShift) - 1)) >> 24) * uint64_t(W);
    const uint64_t* sc = owords + ((c.id & ((1ull << kItLcpShift) - 1)) >> 24) * uint64_t(W);
    for (int w = 1; w < W && !cmp; ++w) {
      const uint64_t x = it_suffix_word(sa, W, int(a.id >> 16) & 255, w), y = it_suffix_word(sc, W, int(c.id >> 16) & 255, w);
      cmp = x < y ? -1 : (x > y ? 1 : 0);
      const int cw = lcp_word(x, y);
      l += cw;
      if (cw < 32) break;
    }
    if (!cmp) cmp = la < lb ? -1 : (la > lb ? 1 : 0);
    const int m = la < lb ? la : lb;
    l = l < m ? l : m;
    const bool has = (a.id & kItLcpValid) != 0;
    if (!has || int((a.id >> kItLcpShift) & 255) != l) atomicAdd(bad + 1, 1ull);  // the LCP note is missing or wrong
  } else if (a.id & kItLcpValid) {
    atomicAdd(bad + 1, 1ull);
  }
  if (cmp > 0) atomicAdd(bad, 1ull);
}

static int it_partition(bg_ctx* ctx, const Item* in, Item* out, uint64_t n, uint32_t nseg, const uint64_t* h_seg0, int shift) {
  std::vector<uint32_t> h_tile(size_t(nseg) + 1, 0);
  uint64_t t = 0;
  for (uint32_t b = 0; b < nseg; ++b) {
    h_tile[b] = uint32_t(t);
    t += (h_seg0[b + 1] - h_seg0[b] + IT_TILE - 1) / IT_TILE;
  }
  if (t >= (1ull << 31)) { set_error("items: %llu records need too many partition tiles", (unsigned long long)n); return BG_EINVAL; }
  h_tile[nseg] = uint32_t(t);
  const uint32_t ntiles = uint32_t(t);
  if (!ntiles) return BG_OK;
  DevBuf<uint32_t> hist;
  DevBuf<uint64_t> offs;
  BG_TRY(hist.alloc(ctx, size_t(IT_BINS) * ntiles));
  BG_TRY(offs.alloc(ctx, size_t(IT_BINS) * ntiles));
  if (nseg <= 256) {  // the tables travel in the kernels' arguments (two waits for the stream less a pass)
    ItSegsArg sg;
    for (uint32_t b = 0; b <= 256; ++b) {
      sg.seg0[b] = h_seg0[b < nseg ? b : nseg];
      sg.tile0[b] = h_tile[b < nseg ? b : nseg];
    }
    sg.nseg = int(nseg);
    it_hist_kernel<ItSegsArg><<<ntiles, IT_THREADS, 0, ctx->stream>>>(in, sg, shift, hist.p);
    BG_LAUNCH_CHECK();
    BG_TRY(exclusive_scan_u32_to_u64(ctx, hist.p, offs.p, uint64_t(IT_BINS) * ntiles, nullptr));
    it_scatter_kernel<ItSegsArg><<<ntiles, IT_THREADS, 0, ctx->stream>>>(in, out, sg, shift, offs.p);
    BG_LAUNCH_CHECK();
    return BG_OK;  // (what goes out of scope here goes back to an arena that is ordered on the stream)
  }
  DevBuf<uint32_t> seg_tile;
  DevBuf<uint64_t> seg0;
  BG_TRY(seg0.alloc(ctx, size_t(nseg) + 1));
  BG_TRY(seg_tile.alloc(ctx, size_t(nseg) + 1));
  BG_HIP(hipMemcpyAsync(seg0.p, h_seg0, (size_t(nseg) + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemcpyAsync(seg_tile.p, h_tile.data(), (size_t(nseg) + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));  // (the host tables go out of scope)
  const ItSegs sg{seg0.p, seg_tile.p, int(nseg)};
  it_hist_kernel<ItSegs><<<ntiles, IT_THREADS, 0, ctx->stream>>>(in, sg, shift, hist.p);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32_to_u64(ctx, hist.p, offs.p, uint64_t(IT_BINS) * ntiles, nullptr));
  it_scatter_kernel<ItSegs><<<ntiles, IT_THREADS, 0, ctx->stream>>>(in, out, sg, shift, offs.p);
  BG_LAUNCH_CHECK();
  return BG_OK;
}

// ---- what the bucket sort and the run kernels leave (a bucket of more than IB_CAP items, a run of more than kItBlockRun equal
// first words: repeats) goes through the rows machinery, which takes any input: the items of those stretches (`segs`: start,
// count; ascending, disjoint) become rows, sort_records_impl orders them by (string, length), and they come back in that order
// with the LCP notes of neighbours that share their first word.  A stretch is a whole range of the final order (a bucket; all
// the items of one first word), so the sorted subset maps onto the stretches in turn.
constexpr uint32_t kItBlockRun = 2048;  // runs up to here: k_items_tie_block (quadratic, the members' rows in 128 KB of LDS)
__device__ __forceinline__ uint64_t it_seg_place(const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_off, uint32_t nseg, uint64_t t) {
  uint32_t a = 0, z = nseg;  // last stretch with seg_off <= t
  while (z - a > 1) {
    const uint32_t mid = (a + z) >> 1;
    if (seg_off[mid] <= t) a = mid; else z = mid;
  }
  return seg_start[a] + (t - seg_off[a]);
}
__global__ __launch_bounds__(256) void k_items_subset_rows(const Item* __restrict__ items, const uint64_t* __restrict__ seg_start,
                                                           const uint64_t* __restrict__ seg_off, uint32_t nseg, uint64_t m,
                                                           const uint64_t* __restrict__ owords, int W, Item* __restrict__ sub,
                                                           uint64_t* __restrict__ rec) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const Item it = items[it_seg_place(seg_start, seg_off, nseg, t)];
  const uint64_t id = it.id & kItIdMask;
  sub[t] = Item{it.fw, id};
  const uint64_t* s = owords + (id >> 24) * uint64_t(W);
  const int j = int(id >> 16) & 255;
  uint64_t* row = rec + t * uint64_t(W + 1);
  for (int w = 0; w < W; ++w) row[w] = it_suffix_word(s, W, j, w);
  row[W] = (id >> 8) & 255;
}
__global__ __launch_bounds__(256) void k_items_subset_back(Item* __restrict__ items, const uint64_t* __restrict__ seg_start,
                                                           const uint64_t* __restrict__ seg_off, uint32_t nseg, uint64_t m,
                                                           const Item* __restrict__ sub, const uint64_t* __restrict__ rec,
                                                           const uint32_t* __restrict__ sidx, int W) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const uint32_t src = sidx[t];
  Item it = sub[src];
  if (t + 1 < m) {
    const uint32_t nx = sidx[t + 1];
    if (sub[nx].fw == it.fw) {
      const uint64_t* a = rec + uint64_t(src) * (W + 1);
      const uint64_t* c = rec + uint64_t(nx) * (W + 1);
      const int l = key_lcp(a, int(a[W] & 255), c, int(c[W] & 255), W);
      it.id |= kItLcpValid | (uint64_t(l) << kItLcpShift);
    }
  }
  items[it_seg_place(seg_start, seg_off, nseg, t)] = it;
}

static int items_subset_through_rows(bg_ctx* ctx, Item* items, std::vector<std::pair<uint64_t, uint64_t>>& segs, const uint64_t* owords, int W) {
  if (segs.empty()) return BG_OK;
  std::sort(segs.begin(), segs.end());
  std::vector<uint64_t> h_start(segs.size()), h_off(segs.size() + 1, 0);
  for (size_t k = 0; k < segs.size(); ++k) {
    h_start[k] = segs[k].first;
    h_off[k + 1] = h_off[k] + segs[k].second;
  }
  const uint64_t m = h_off.back();
  const uint32_t nseg = uint32_t(segs.size());
  DevBuf<uint64_t> d_start, d_off, rec;
  DevBuf<Item> sub;
  DevBuf<uint32_t> sidx;
  DevBuf<uint8_t> no_len;
  BG_TRY(d_start.alloc(ctx, nseg));
  BG_TRY(d_off.alloc(ctx, nseg + 1));
  BG_TRY(sub.alloc(ctx, m));
  BG_TRY(rec.alloc(ctx, m * uint64_t(W + 1)));
  BG_HIP(hipMemcpyAsync(d_start.p, h_start.data(), nseg * 8, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemcpyAsync(d_off.p, h_off.data(), (size_t(nseg) + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  ctx->tic("items_through_rows");
  k_items_subset_rows<<<cdiv(m, 256), 256, 0, ctx->stream>>>(items, d_start.p, d_off.p, nseg, m, owords, W, sub.p, rec.p);
  BG_LAUNCH_CHECK();
  BG_HIP(hipStreamSynchronize(ctx->stream));
  uint64_t n_tied = 0;
  BG_TRY(sort_records_impl(ctx, rec.p, no_len.p, m, W, nullptr, &sidx, &n_tied, W + 1, false));
  k_items_subset_back<<<cdiv(m, 256), 256, 0, ctx->stream>>>(items, d_start.p, d_off.p, nseg, m, sub.p, rec.p, sidx.p, W);
  BG_LAUNCH_CHECK();
  ctx->toc(m * (uint64_t(W + 1) * 8 * 3 + 64));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// ---- the stream passes over the items in order: no read is touched except for the list rows and the keys
// keep[p] = the item survives (it is not a prefix of, or equal to, its successor: expand.cpp:19-48 skip_dups); lcpn[p] = its LCP
// with its successor (the note of k_items_tie_*, or what the two first words show)
__global__ __launch_bounds__(256) void k_items_keep(const Item* __restrict__ items, uint64_t n, uint8_t* __restrict__ keep,
                                                    uint8_t* __restrict__ lcpn, uint32_t* __restrict__ err,
                                                    uint32_t* __restrict__ cnt256, uint32_t* __restrict__ firstk,
                                                    uint32_t* __restrict__ lastk) {
  __shared__ uint32_t s_c[4], s_f[4], s_l[4];
  const uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  bool k = false;
  if (p < n) {
    const Item a = items[p];
    const int la = int(a.id >> 8) & 255;
    int l = 0;
    k = true;
    if (p + 1 < n) {
      const Item b = items[p + 1];
      const int lb = int(b.id >> 8) & 255;
      if (a.id & kItLcpValid) {
        l = int(a.id >> kItLcpShift) & 255;
      } else {
        if (a.fw == b.fw) atomicOr(err, 8u);  // a run the tie kernels did not see
        if (a.fw > b.fw) atomicOr(err, 1u);   // not sorted
        l = lcp_word(a.fw, b.fw);
        const int m = la < lb ? la : lb;
        l = l < m ? l : m;
      }
      k = l != la;
    }
    keep[p] = k ? 1 : 0;
    lcpn[p] = uint8_t(l);
  }
  // the workgroup's survivors: how many, the first and the last place (k_items_entries numbers the entries from these)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t m = __ballot(k);
  if (lane == 0) {
    const uint32_t at = blockIdx.x * 256u + uint32_t(w) * 64u;
    s_c[w] = uint32_t(__popcll(m));
    s_f[w] = m ? at + uint32_t(__ffsll((unsigned long long)m) - 1) : ~0u;
    s_l[w] = m ? at + uint32_t(63 - __clzll((long long)m)) : ~0u;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = 0, f = ~0u, l = ~0u;
    for (int u = 0; u < 4; ++u) {
      c += s_c[u];
      if (f == ~0u) f = s_f[u];
      if (s_l[u] != ~0u) l = s_l[u];
    }
    cnt256[blockIdx.x] = c;
    firstk[blockIdx.x] = f;
    lastk[blockIdx.x] = l;
  }
}

// One lane an ITEM.  A survivor writes its entry: size, its record, shared = its LCP with the entry before = the smallest LCP on
// the way there = min(the LCP of the entry before with its successor, the length of the first record this entry absorbed) -- the
// absorbed records are prefixes of each other, shortest first.  Every record brings the prev bit it knows (k_gather_dedup_build's
// rule: a record with a predecessor (r, j - 1) whose string is an entry sets prev_b of that entry): its own entry's if it
// survives or equals the survivor; what the rule cannot decide goes to the list as (place of the item) << 2 | kind -- 1 the
// tail of a survivor that is its read's last record, 2 a record dropped as a proper prefix, checked (k_items_list_rows makes the
// rows).  emask: bytes, four to a word (atomicOr on the word).
// The entry an item belongs to = the survivors before it: the workgroup's base (offs256: a scan of k_items_keep's counts) and a
// count over its lanes; the survivor before and the one at or after an item: over the lanes, the waves, and -- past the
// workgroup's ends -- the neighbours' first / last survivors (k_items_keep) -- no rank and no place written per item (0.7 ms
// of their own and 1.2 GB).
__global__ __launch_bounds__(256) void k_items_entries(const Item* __restrict__ items, uint64_t n, const uint8_t* __restrict__ keep,
                                                       const uint64_t* __restrict__ offs256, const uint32_t* __restrict__ firstk,
                                                       const uint32_t* __restrict__ lastk,
                                                       const uint8_t* __restrict__ lcpn, uint8_t* __restrict__ sizes, uint8_t* __restrict__ shared,
                                                       Item* __restrict__ ent, uint32_t* __restrict__ emask32, uint8_t* __restrict__ kindf,
                                                       uint32_t* __restrict__ dir, int dir_bases) {
  __shared__ uint32_t s_cnt[4], s_first[4], s_last[4], s_prev, s_next;
  const uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const bool k = p < n && keep[p] != 0;
  const uint64_t km = __ballot(k);
  {
    const uint32_t at = blockIdx.x * 256u + uint32_t(w) * 64u;
    if (lane == 0) {
      s_cnt[w] = uint32_t(__popcll(km));
      s_first[w] = km ? at + uint32_t(__ffsll((unsigned long long)km) - 1) : ~0u;
      s_last[w] = km ? at + uint32_t(63 - __clzll((long long)km)) : ~0u;
    }
    if (threadIdx.x == 1) {  // the last survivor before this workgroup's items
      uint32_t v = ~0u;
      for (uint32_t bb = blockIdx.x; bb > 0 && v == ~0u;) v = lastk[--bb];
      s_prev = v;
    }
    if (threadIdx.x == 65) {  // the first survivor behind them
      uint32_t v = ~0u;
      for (uint32_t bb = blockIdx.x + 1; bb < gridDim.x && v == ~0u; ++bb) v = firstk[bb];
      s_next = v;
    }
  }
  __syncthreads();
  uint32_t kind = 0, contrib = 0, my_e = 0xffffffffu;
  bool mine_kept = false, spans = false;
  if (p < n) {
    const Item me = items[p];
    const uint64_t id = me.id & kItIdMask;
    const uint32_t fl = uint32_t(id) & 255u;
    const int len = int(id >> 8) & 255;
    uint32_t e = uint32_t(offs256[blockIdx.x]) + uint32_t(__popcll(km & lane_lt));
    for (int u = 0; u < w; ++u) e += s_cnt[u];
    const uint32_t at = blockIdx.x * 256u + uint32_t(w) * 64u;
    bool sets = false;
    if (k) {
      // the entry before: the last survivor before this item
      uint32_t pp = ~0u;
      const uint64_t mb = km & lane_lt;
      if (mb) pp = at + uint32_t(63 - __clzll((long long)mb));
      else {
        for (int u = w - 1; u >= 0 && pp == ~0u; --u) pp = s_last[u];
        if (pp == ~0u) pp = s_prev;
      }
      int sh = 0;
      if (pp != ~0u) {
        sh = lcpn[pp];
        if (pp + 1 < p) { const int v = lcpn[pp + 1]; sh = v < sh ? v : sh; }
      }
      sizes[e] = uint8_t(len);
      shared[e] = uint8_t(sh);
      ent[e] = Item{me.fw, id};  // the entry's first word and its record, side by side: what the searches probe
      if (dir) {  // the prefix directory's heads (k_dir_heads): the first entry of every prefix that occurs
        const uint32_t cur = uint32_t(me.fw >> (64 - 2 * dir_bases));
        if (pp == ~0u || uint32_t(items[pp].fw >> (64 - 2 * dir_bases)) != cur) dir[cur] = e;
      }
      sets = (fl & 1u) != 0;
      if (fl & 8u) kind = 1;
      spans = (pp != ~0u ? pp + 1 : 0u) + uint32_t(lane) < uint32_t(p);  // its first record lies in an earlier wave
    } else if (fl & 1u) {
      // the survivor: the first one behind this item (a few places on)
      uint32_t sp = ~0u;
      const uint64_t ma = km & ~lane_lt;
      if (ma) sp = at + uint32_t(__ffsll((unsigned long long)ma) - 1);
      else {
        for (int u = w + 1; u < 4 && sp == ~0u; ++u) sp = s_first[u];
        if (sp == ~0u) sp = s_next;
      }
      const int slen = int(items[sp].id >> 8) & 255;
      if (slen == len) sets = true;
      else kind = 2;
    }
    contrib = sets ? 1u << ((fl >> 1) & 3u) : 0u;
    my_e = e;
    mine_kept = k;
  }
  // the records of an entry sit side by side: their bits are put together over the lanes (an atomic a record was 30 ms: four
  // billion a second is all the L2 does), and only an entry whose records straddle two waves takes atomics
  {
    uint32_t acc = contrib;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = uint32_t(__shfl_up(int(acc), d, 64));
      const uint32_t oe = uint32_t(__shfl_up(int(my_e), d, 64));
      if (lane >= d && oe == my_e) acc |= o;
    }
    const uint32_t ne = uint32_t(__shfl_down(int(my_e), 1, 64));
    const bool seg_last = lane == 63 || ne != my_e;
    if (p < n && seg_last) {
      if (mine_kept && !spans) reinterpret_cast<uint8_t*>(emask32)[my_e] = uint8_t(acc);
      else if (acc) atomicOr(&emask32[my_e >> 2], acc << (8 * (my_e & 3u)));
    }
  }
  if (p < n) kindf[p] = uint8_t(kind);  // (the list's rows are counted and placed by a compaction: one counter for all waves was 24 ms)
}

// rows of W + 1 words for k_prev_list (the formats of k_gather_dedup_build): a lane a word
__global__ __launch_bounds__(256) void k_items_list_rows(const Item* __restrict__ items, const uint32_t* __restrict__ lpos,
                                                         const uint8_t* __restrict__ kindf, uint64_t n_rows,
                                                         const uint64_t* __restrict__ owords, int W, uint64_t* __restrict__ list) {
  const uint64_t x = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t RW = uint64_t(W + 1);
  if (x >= n_rows * RW) return;
  const uint64_t r = x / RW;
  const int w = int(x - r * RW);
  const uint32_t at = lpos[r];
  const uint32_t d = kindf[at];
  const uint64_t id = items[at].id & kItIdMask;
  const uint64_t* s = owords + (id >> 24) * uint64_t(W);
  const int j = int(id >> 16) & 255, len = int(id >> 8) & 255;
  uint64_t v;
  if ((d & 3) == 1) {  // s[1:], its length, s[0]
    v = w < W ? it_suffix_word(s, W, j + 1, w) : (uint64_t(len - 1) | (((s[j >> 5] >> (62 - 2 * (j & 31))) & 3) << 8));
  } else {             // s, its length, the base in front of it; bit 16: only if that base + s is an entry
    v = w < W ? it_suffix_word(s, W, j, w) : (uint64_t(len) | (uint64_t((uint32_t(id) >> 1) & 3u) << 8) | (1ull << 16));
  }
  list[x] = v;
}

// ---- the list's tails searched through the entries' first words and, where those tie, through the reads (no key table)
// k_prev_list's work (builder.cpp:85-108) for the items' list: item `at` of kind 1 = the tail of a survivor that is its read's
// last record (suffix j + 1 of the read; the bit is the base at j), kind 2 = a record dropped as a proper prefix (its own
// string s; the bit is the base in front, and only if that base + s -- suffix j - 1 of the read -- is an entry)
__global__ __launch_bounds__(256) void k_prev_list_items(const Item* __restrict__ items, const uint32_t* __restrict__ lpos,
                                                         const uint8_t* __restrict__ kindf, uint64_t n_rows, ItEntries t,
                                                         uint8_t* __restrict__ emask, uint32_t* __restrict__ err, int verify) {
  const uint64_t r = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const uint32_t at = lpos[r];
  const uint32_t kind = kindf[at];
  const uint64_t id = items[at].id & kItIdMask;
  const uint64_t* s = t.owords + (id >> 24) * uint64_t(t.W);
  const int j = int(id >> 16) & 255, len = int(id >> 8) & 255;
  const int tj = kind == 1 ? j + 1 : j, tl = kind == 1 ? len - 1 : len;
  const int b = kind == 1 ? int((s[j >> 5] >> (62 - 2 * (j & 31))) & 3) : int((uint32_t(id) >> 1) & 3u);
  uint64_t key[kMaxWords];
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w) key[w] = w < t.W ? it_suffix_word(s, t.W, tj, w) : 0;
  bool alone = false;
  const uint64_t lo = it_lower_bound(t, key, tl, &alone);
  if (lo >= t.n || ((!alone || verify) && it_entry_lcp(t, lo, key, tl) < tl)) { atomicOr(err, 2u); return; }  // "Missing expansion?"
  if (kind == 2) {
    uint64_t k2[kMaxWords];
#pragma unroll
    for (int w = 0; w < kMaxWords; ++w) k2[w] = w < t.W ? it_suffix_word(s, t.W, j - 1, w) : 0;
    const int kl = len + 1;
    const uint64_t e2 = it_lower_bound(t, k2, kl, &alone);
    if (e2 >= t.n || int(t.sizes[e2]) != kl || it_entry_lcp(t, e2, k2, kl) < kl) return;  // (is it an entry: always checked)
  }
  atomicOr(reinterpret_cast<unsigned int*>(emask + (lo & ~3ull)), (1u << b) << (8 * unsigned(lo & 3)));
}

// the key table from the entries' records (what k_gather_dedup_build writes as it goes): W words an entry, and the first two
__global__ __launch_bounds__(256) void k_items_keys(const Item* __restrict__ ent, uint64_t n_ent, const uint64_t* __restrict__ owords,
                                                    int W, uint64_t* __restrict__ keys, uint64_t* __restrict__ pre2) {
  const uint64_t x = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (x >= n_ent * uint64_t(W)) return;
  const uint64_t e = x / uint64_t(W);
  const int w = int(x - e * uint64_t(W));
  const uint64_t id = ent[e].id;
  const uint64_t word = it_suffix_word(owords + (id >> 24) * uint64_t(W), W, int(id >> 16) & 255, w);
  keys[x] = word;
  if (w < 2) pre2[e * 2 + w] = word;
  if (W == 1 && w == 0) pre2[e * 2 + 1] = 0;
}

// From items partitioned by their top byte (in b; a is scratch of the same size and is released) to the items in (string, length)
// order in b, every member of a run of equal first words carrying its LCP note: second (and third) partition level, bucket sort,
// the run kernels, the rows machinery for what they leave (*segs_out).  `owords`: the rows the ids point into (W words each).
// tics: the caller's "items_partition" bracket is open and is closed here; the bucket sort and the runs get their own.
static int items_sort_from_level1(bg_ctx* ctx, DevBuf<Item>& a, DevBuf<Item>& b, uint64_t n_rec, const uint64_t* owords, int W, bool tics,
                                  std::vector<std::pair<uint64_t, uint64_t>>* segs_out, bool* ok) {
  *ok = true;
  // Buckets of a few thousand records at most are what the LDS sort takes: two levels (65 536 buckets) up to ~170 M records, a
  // third on as many more bits as bring the average bucket back under 2 600 (a workgroup a bucket: many small buckets cost more
  // than they sort) beyond (BG_ITEMS_LEVELS=3: a third level of four bits at any size, for the tests)
  const char* lv = getenv("BG_ITEMS_LEVELS");
  int extra = 0;
  while (extra < 8 && (n_rec >> (16 + extra)) > 2600) ++extra;
  if (lv && lv[0] == '3' && extra < 4) extra = 4;
  const int levels = extra ? 3 : 2;
  // (and fewer, fuller buckets for fewer records: the second level on as few bits as keep the average under 2 600 -- 25.7 M
  // oriented reads in 16 384 buckets, not 65 536 of 390: a workgroup a bucket costs what it costs)
  int bucket_bits = 16 + extra;
  if (!extra)
    while (bucket_bits > 8 && (n_rec >> (bucket_bits - 1)) <= 2600) --bucket_bits;
  const uint64_t n_buckets = 1ull << bucket_bits;
  DevBuf<uint64_t> b8, b16, bnd;
  BG_TRY(b8.alloc(ctx, 257));
  BG_TRY(b16.alloc(ctx, 65537));
  if (bucket_bits > 8) {
    it_bounds_kernel<<<2, 256, 0, ctx->stream>>>(b.p, n_rec, 8, b8.p);
    BG_LAUNCH_CHECK();
    std::vector<uint64_t> h8(257);
    BG_HIP(copy_d2h(ctx, h8.data(), b8.p, 257 * 8));
    // (the byte the pass sorts on: the second level's bits under as many of the first byte's, which a segment shares)
    BG_TRY(it_partition(ctx, b.p, a.p, n_rec, 256, h8.data(), 64 - std::min(bucket_bits, 16)));
  } else {
    std::swap(a, b);  // (the first level is all: the items are where the bucket sort reads them)
  }
  it_bounds_kernel<<<cdiv((1ull << std::min(bucket_bits, 16)) + 1, 256), 256, 0, ctx->stream>>>(a.p, n_rec, std::min(bucket_bits, 16), b16.p);
  BG_LAUNCH_CHECK();
  std::vector<uint64_t> h_bounds;  // (on the host only when a stretch has to be named)
  if (levels == 3) {
    std::vector<uint64_t> h16(65537);
    BG_HIP(copy_d2h(ctx, h16.data(), b16.p, 65537 * 8));
    BG_TRY(it_partition(ctx, a.p, b.p, n_rec, 65536, h16.data(), 48 - extra));  // (eight bits from there: the upper ones are the segment's own)
    BG_TRY(bnd.alloc(ctx, n_buckets + 1));
    it_bounds_kernel<<<cdiv(n_buckets + 1, 256), 256, 0, ctx->stream>>>(b.p, n_rec, bucket_bits, bnd.p);
    BG_LAUNCH_CHECK();
    std::swap(a, b);  // (the partitioned items are in a again)
  }
  const uint64_t* d_bounds = levels == 3 ? bnd.p : b16.p;
  if (tics) ctx->toc(n_rec * 16 * 3 * uint64_t(levels));
  // ---- every bucket in order of the first word, the runs of equal first words by the rest of their strings
  DevBuf<uint8_t> status;
  DevBuf<unsigned long long> ctr;  // [0] runs, [1] runs of more than four, [2] of more than 64, [3] buckets left as they were, [4] runs left
  DevBuf<uint64_t> runs, more, longer;
  const uint64_t run_cap = n_rec / 2 + 1, more_cap = n_rec / 5 + 1, long_cap = n_rec / 65 + 1;
  BG_TRY(status.alloc(ctx, n_buckets));
  BG_TRY(ctr.alloc(ctx, 8));
  BG_TRY(runs.alloc(ctx, run_cap));
  BG_TRY(more.alloc(ctx, more_cap));
  BG_TRY(longer.alloc(ctx, long_cap));
  BG_HIP(hipMemsetAsync(status.p, 0, n_buckets, ctx->stream));
  BG_HIP(hipMemsetAsync(ctr.p, 0, 64, ctx->stream));
  if (tics) ctx->tic("items_bucket_sort");
  k_items_bucket_sort<<<unsigned(n_buckets), IB_THREADS, 0, ctx->stream>>>(a.p, b.p, d_bounds, status.p, runs.p, ctr.p, run_cap, 64 - bucket_bits - IB_LOG2BINS);
  BG_LAUNCH_CHECK();
  if (tics) ctx->toc(n_rec * 32);
  // the runs: lanes, waves, workgroups -- each kernel strides over a list whose length the one before left on the device, and
  // the host reads the counters once, behind all three (five read-backs and their idle gaps before)
  if (tics) ctx->tic("items_tie_runs");
  {
    const unsigned small_grid = unsigned(std::min<uint64_t>(cdiv(run_cap, 256), 1u << 16));
    k_items_tie_small<<<small_grid, 256, 0, ctx->stream>>>(b.p, runs.p, ctr.p, run_cap, owords, W, more.p, ctr.p + 1, more_cap);
    BG_LAUNCH_CHECK();
    k_items_tie_wave<<<2048, 256, 0, ctx->stream>>>(b.p, more.p, ctr.p + 1, more_cap, owords, W, longer.p, ctr.p + 2, long_cap);
    BG_LAUNCH_CHECK();
    // runs of 65 .. kItBlockRun members: a workgroup each; longer ones: the rows machinery
    const uint32_t lds_members = kItBlockRun;
    BG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_items_tie_block), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(lds_members * kMaxWords * 8)));
    k_items_tie_block<<<256, 1024, size_t(lds_members) * kMaxWords * 8, ctx->stream>>>(b.p, longer.p, ctr.p + 2, long_cap, owords, W, lds_members,
                                                                                         kItBlockRun, ctr.p + 4);
    BG_LAUNCH_CHECK();
  }
  unsigned long long h_ctr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  BG_HIP(copy_d2h(ctx, h_ctr, ctr.p, 64));
  if (tics) ctx->toc(h_ctr[0] * 2 * (16 + uint64_t(W) * 8));
  a.reset();
  if (h_ctr[0] > run_cap || h_ctr[1] > more_cap || h_ctr[2] > long_cap) { *ok = false; return BG_OK; }  // (cannot happen: a run has 2 / 5 / 65 members at least)
  std::vector<std::pair<uint64_t, uint64_t>>& segs = *segs_out;  // stretches left to the rows machinery: (start, count)
  segs.clear();
  if (h_ctr[3]) {
    std::vector<uint8_t> h_status(n_buckets);
    BG_HIP(copy_d2h(ctx, h_status.data(), status.p, n_buckets));
    h_bounds.resize(n_buckets + 1);
    BG_HIP(copy_d2h(ctx, h_bounds.data(), d_bounds, (n_buckets + 1) * 8));
    for (uint64_t v = 0; v < n_buckets; ++v)
      if (h_status[size_t(v)]) segs.emplace_back(h_bounds[size_t(v)], h_bounds[size_t(v) + 1] - h_bounds[size_t(v)]);
  }
  if (h_ctr[4]) {
    std::vector<uint64_t> h_long(h_ctr[2]);
    BG_HIP(copy_d2h(ctx, h_long.data(), longer.p, h_ctr[2] * 8));
    for (uint64_t rr : h_long)
      if (uint32_t(rr >> 48) > kItBlockRun) segs.emplace_back(rr & ((1ull << 48) - 1), rr >> 48);
  }
  longer.reset();
  runs.reset();
  more.reset();
  BG_TRY(items_subset_through_rows(ctx, b.p, segs, owords, W));
  return BG_OK;
}

// ---- the cover index's order of the oriented reads by the same machinery: a read is the item of its suffix 0
// (a read shorter than K is no cover and is dropped from the order afterwards: its first word is replaced by a scattered value,
// or the reads the correction emptied -- thousands of equal strings -- would be one run for the rows machinery)
__global__ void k_read_items(const uint64_t* __restrict__ owords, const uint8_t* __restrict__ olen, uint64_t n_o, int W, int K,
                             Item* __restrict__ out) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= n_o) return;
  const int L = olen[o];
  out[o] = Item{L >= K ? owords[o * uint64_t(W)] : cover_item_scatter(o), (o << 24) | (uint64_t(L) << 8)};
}
__global__ void k_cover_flag_items(const Item* __restrict__ items, uint64_t n, int K, uint8_t* __restrict__ flags) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (int(items[i].id >> 8) & 255) >= K ? 1 : 0;
}
__global__ void k_cover_keys_items(const Item* __restrict__ items, const uint32_t* __restrict__ keep, uint64_t m, int K,
                                   uint64_t* __restrict__ skeys, uint32_t* __restrict__ soid) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Item it = items[keep[i]];
  soid[i] = uint32_t((it.id & kItIdMask) >> 24);
  skeys[i] = it.fw >> (64 - 2 * K);
}
// The oriented reads of at least cover_k bases in (string, length) order: their first K-mers (skeys) and numbers (soid), what
// cover_index_build made from sort_records_impl's order with two gathers (k_flag_long_enough, k_cover_keys: 1 ms of random
// bytes and rows) -- an item carries its length and its first 32 bases.  *done = false: the caller goes the old way.
int items_cover_order(bg_ctx* ctx, const uint64_t* owords, const uint8_t* olen, uint64_t n_o, int W, int cover_k, DevBuf<uint64_t>* premade,
                      DevBuf<uint64_t>* skeys, DevBuf<uint32_t>* soid, uint64_t* m_out, bool* done) {
  *done = false;
  if (!n_o || n_o >= (1ull << 31) || cover_k > 32) return BG_OK;
  DevBuf<Item> a, b;
  if (premade && premade->p && premade->n == 2 * n_o) {  // (k_oriented_words wrote the items beside the reads: taken over)
    a.ctx = ctx;
    a.n = n_o;
    a.p = reinterpret_cast<Item*>(premade->release_ownership());
  } else {
    BG_TRY(a.alloc(ctx, n_o));
    k_read_items<<<cdiv(n_o, 256), 256, 0, ctx->stream>>>(owords, olen, n_o, W, cover_k, a.p);
    BG_LAUNCH_CHECK();
  }
  BG_TRY(b.alloc(ctx, n_o));
  const uint64_t whole[2] = {0, n_o};
  BG_TRY(it_partition(ctx, a.p, b.p, n_o, 1, whole, 56));
  std::vector<std::pair<uint64_t, uint64_t>> segs;
  bool ok = true;
  BG_TRY(items_sort_from_level1(ctx, a, b, n_o, owords, W, false, &segs, &ok));
  if (!ok) return BG_OK;
  DevBuf<uint8_t> flags;
  DevBuf<uint32_t> keep;
  uint64_t m = 0;
  BG_TRY(flags.alloc(ctx, n_o));
  k_cover_flag_items<<<cdiv(n_o, 256), 256, 0, ctx->stream>>>(b.p, n_o, cover_k, flags.p);
  BG_LAUNCH_CHECK();
  BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, n_o, &keep, &m));
  BG_TRY(skeys->alloc(ctx, m));
  BG_TRY(soid->alloc(ctx, m));
  if (m) {
    k_cover_keys_items<<<cdiv(m, 256), 256, 0, ctx->stream>>>(b.p, keep.p, m, cover_k, skeys->p, soid->p);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));  // (the items go out of scope)
  *m_out = m;
  *done = true;
  return BG_OK;
}

int items_entries_impl(bg_ctx* ctx, const bg_reads* reads, CoverCounts* cc, bg_seqset* ss, LocalPrev* lp, bool* done) {
  *done = false;
  const uint64_t n_rec = cc->n_rec;
  const int W = reads->W;
  // (an item's id holds the oriented read in 31 bits above offset, length and flags, under the LCP note: more reads than that --
  // or records than 32-bit places -- go through rows)
  if (!n_rec || n_rec >= (1ull << 32) || cc->o_base + cc->n_emit >= (1ull << 31) || getenv("BG_PREV_SEARCH_ALL")) return BG_OK;
  DevBuf<Item> a, b;
  BG_TRY(a.alloc(ctx, n_rec));
  BG_TRY(b.alloc(ctx, n_rec));
  const char* em = getenv("BG_ITEMS_EMIT");
  if (em && !strcmp(em, "plain")) {  // (the items written in read order first, then partitioned like any other level: for comparison)
    ctx->tic("items_emit");
    k_emit_items<<<cdiv(cdiv(cc->n_emit, 64), 4), 256, 0, ctx->stream>>>(cc->owords.p, reads->len.p, cc->cnt.p, cc->roff.p, cc->o_base, cc->n_emit, W, a.p);
    BG_LAUNCH_CHECK();
    ctx->toc(n_rec * 16 + cc->n_emit * uint64_t(W) * 8);
    ctx->tic("items_partition");
    const uint64_t whole[2] = {0, n_rec};
    BG_TRY(it_partition(ctx, a.p, b.p, n_rec, 1, whole, 56));
  } else {
    const uint64_t nt = cdiv(cc->n_emit, IR_READS);
    if (nt >= (1ull << 31)) { set_error("items: %llu oriented reads need too many partition tiles", (unsigned long long)cc->n_emit); return BG_EINVAL; }
    DevBuf<uint32_t> hist;
    DevBuf<uint64_t> offs;
    BG_TRY(hist.alloc(ctx, size_t(IT_BINS) * nt));
    BG_TRY(offs.alloc(ctx, size_t(IT_BINS) * nt));
    ctx->tic("items_partition");
    it_hist_reads_kernel<<<unsigned(nt), IT_THREADS, 0, ctx->stream>>>(cc->owords.p, cc->roff.p, cc->o_base, cc->n_emit, n_rec, W, uint32_t(nt), hist.p);
    BG_LAUNCH_CHECK();
    BG_TRY(exclusive_scan_u32_to_u64(ctx, hist.p, offs.p, uint64_t(IT_BINS) * nt, nullptr));
    it_scatter_reads_kernel<0><<<unsigned(nt), IT_THREADS, 0, ctx->stream>>>(cc->owords.p, reads->len.p, cc->roff.p, cc->o_base, cc->n_emit, n_rec, W,
                                                                            uint32_t(nt), offs.p, b.p);
    BG_LAUNCH_CHECK();
    if (getenv("BG_ITEMS_PROBE")) {
      it_scatter_reads_kernel<1><<<unsigned(nt), IT_THREADS, 0, ctx->stream>>>(cc->owords.p, reads->len.p, cc->roff.p, cc->o_base, cc->n_emit, n_rec, W,
                                                                              uint32_t(nt), offs.p, a.p);
      BG_LAUNCH_CHECK();
    }
  }
  std::vector<std::pair<uint64_t, uint64_t>> segs;  // stretches that went through the rows machinery (BG_ITEMS_CHECK's report)
  {
    bool ok = true;
    BG_TRY(items_sort_from_level1(ctx, a, b, n_rec, cc->owords.p, W, true, &segs, &ok));
    if (!ok) return BG_OK;
  }
  if (getenv("BG_ITEMS_CHECK")) {  // every neighbour pair compared through the reads (slow; stderr)
    DevBuf<unsigned long long> chk;
    BG_TRY(chk.alloc(ctx, 4));
    BG_HIP(hipMemsetAsync(chk.p, 0, 32, ctx->stream));
    k_items_check_order<<<cdiv(n_rec, 256), 256, 0, ctx->stream>>>(b.p, n_rec, cc->owords.p, W, chk.p, chk.p + 2);
    BG_LAUNCH_CHECK();
    unsigned long long h[4];
    BG_HIP(copy_d2h(ctx, h, chk.p, 32));
    uint64_t in_segs = 0;
    for (auto& sg : segs) in_segs += sg.second;
    fprintf(stderr, "items check: %llu records, %zu stretches (%llu records) through the rows; out of order %llu, LCP notes missing or wrong %llu\n",
            (unsigned long long)n_rec, segs.size(), (unsigned long long)in_segs, h[0], h[1]);
  }
  // ---- entries
  DevBuf<uint8_t> keep, lcpn;
  DevBuf<uint32_t> err;
  BG_TRY(keep.alloc(ctx, n_rec));
  BG_TRY(lcpn.alloc(ctx, n_rec));
  BG_TRY(err.alloc(ctx, 2));
  BG_HIP(hipMemsetAsync(err.p, 0, 8, ctx->stream));
  ctx->tic("items_entries");
  const uint64_t nb256 = cdiv(n_rec, 256);
  DevBuf<uint32_t> cnt256, firstk, lastk;
  DevBuf<uint64_t> offs256, total_ent;
  BG_TRY(cnt256.alloc(ctx, nb256));
  BG_TRY(firstk.alloc(ctx, nb256));
  BG_TRY(lastk.alloc(ctx, nb256));
  BG_TRY(offs256.alloc(ctx, nb256));
  BG_TRY(total_ent.alloc(ctx, 1));
  k_items_keep<<<unsigned(nb256), 256, 0, ctx->stream>>>(b.p, n_rec, keep.p, lcpn.p, err.p, cnt256.p, firstk.p, lastk.p);
  BG_LAUNCH_CHECK();
  uint64_t n_ent = 0;
  BG_TRY(exclusive_scan_u32_to_u64(ctx, cnt256.p, offs256.p, nb256, total_ent.p));
  BG_HIP(copy_d2h(ctx, &n_ent, total_ent.p, 8));
  DevBuf<uint64_t> erec;  // n_ent x {first word, record}
  ss->W = W;
  BG_TRY(ss->sizes.alloc(ctx, n_ent));
  BG_TRY(ss->shared.alloc(ctx, n_ent));
  BG_TRY(erec.alloc(ctx, n_ent * 2));
  BG_TRY(lp->emask.alloc(ctx, (n_ent + 3) & ~3ull));
  BG_HIP(hipMemsetAsync(lp->emask.p, 0, (n_ent + 3) & ~3ull, ctx->stream));
  BG_TRY(lp->count.alloc(ctx, 1));
  DevBuf<uint8_t> kindf;
  DevBuf<uint32_t> lpos;
  BG_TRY(kindf.alloc(ctx, n_rec));
  const char* keys_env = getenv("BG_ITEMS_KEYS");
  const bool keys_now = keys_env && keys_env[0] == '1';
  DevBuf<uint32_t> dir;  // the entries' prefix directory for the tails' searches: its heads come out of k_items_entries
  int bases = 0;
  if (!keys_now) BG_TRY(prefix_dir_prepare(ctx, n_ent, &dir, &bases));
  k_items_entries<<<unsigned(nb256), 256, 0, ctx->stream>>>(b.p, n_rec, keep.p, offs256.p, firstk.p, lastk.p, lcpn.p, ss->sizes.p, ss->shared.p,
                                                             reinterpret_cast<Item*>(erec.p), reinterpret_cast<uint32_t*>(lp->emask.p), kindf.p,
                                                             dir.p, bases);
  BG_LAUNCH_CHECK();
  BG_TRY(compact_flagged<uint32_t>(ctx, kindf.p, n_rec, &lpos, &lp->n_items));
  ctx->toc(n_rec * 32 + n_ent * 12);
  uint32_t h_err[2] = {0, 0};
  BG_HIP(copy_d2h(ctx, h_err, err.p, 8));
  if (h_err[0]) { set_error("items: the sorted items are not in order (internal error %u)", h_err[0]); return BG_EINTERNAL; }
  lp->cap = lp->n_items;
  lp->valid = true;
  if (keys_now) {
    // (BG_ITEMS_KEYS=1: the rows of the list and the key table at once, searched by k_prev_list as the rows path does)
    if (lp->n_items) {
      BG_TRY(lp->list.alloc(ctx, lp->n_items * uint64_t(W + 1)));
      ctx->tic("items_list_rows");
      k_items_list_rows<<<cdiv(lp->n_items * uint64_t(W + 1), 256), 256, 0, ctx->stream>>>(b.p, lpos.p, kindf.p, lp->n_items, cc->owords.p, W, lp->list.p);
      BG_LAUNCH_CHECK();
      ctx->toc(lp->n_items * (uint64_t(W + 1) * 8 * 2 + 24));
    }
  } else {
    // the tails searched through the entries' first words and the reads
    ctx->tic("prev_bits");
    BG_TRY(prefix_dir_finish(ctx, n_ent, bases, &dir));
    if (lp->n_items) {
      const ItEntries t{reinterpret_cast<const Item*>(erec.p), ss->sizes.p, cc->owords.p, dir.p, n_ent, W, bases};
      const char* v = getenv("BG_VERIFY_CLOSURE");  // (the lone-first-word shortcut checked against the reads as well)
      k_prev_list_items<<<cdiv(lp->n_items, 256), 256, 0, ctx->stream>>>(b.p, lpos.p, kindf.p, lp->n_items, t, lp->emask.p, err.p, v && v[0] == '1');
      BG_LAUNCH_CHECK();
    }
    ctx->toc(lp->n_items * (uint64_t(W) * 8 * 3 + 64));
    BG_HIP(copy_d2h(ctx, h_err, err.p, 8));
    if (h_err[0] & 2) { set_error("missing expansion (entry set not closed under pop_front)"); return BG_EINTERNAL; }
    lp->searched = true;
  }
  lpos.reset();
  kindf.reset();
  b.reset();
  keep.reset();
  lcpn.reset();
  ss->n = n_ent;
  ss->erec = std::move(erec);
  ss->oreads = std::move(cc->owords);
  if (!keys_now) {  // (the directory the tails were searched through serves the readmap's look-ups the same way: readmap.hip)
    ss->edir = std::move(dir);
    ss->edir_bases = bases;
  }
  if (keys_now) BG_TRY(seqset_ensure_keys(ctx, ss));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  ss->n = n_ent;
  *done = true;
  return BG_OK;
}

int seqset_ensure_keys(bg_ctx* caller, bg_seqset* ss) {
  if (!ss) return BG_OK;
  std::lock_guard<std::mutex> lk(ss->keys_mutex);
  if (ss->keys.p || !ss->erec.p || !ss->oreads.p || !ss->W) return BG_OK;
  // the buffers outlive the call: they come from the context the seqset lives in, not from a caller's temporary one
  bg_ctx* ctx = ss->ctx ? ss->ctx : caller;
  const uint64_t n_ent = ss->n;
  const int W = ss->W;
  BG_HIP(hipSetDevice(ctx->device));
  DevBuf<uint64_t> keys, pre2;
  BG_TRY(keys.alloc(ctx, n_ent * W));
  BG_TRY(pre2.alloc(ctx, n_ent * 2));
  if (n_ent) {
    ctx->tic("items_keys");
    k_items_keys<<<cdiv(n_ent * uint64_t(W), 256), 256, 0, ctx->stream>>>(reinterpret_cast<const Item*>(ss->erec.p), n_ent, ss->oreads.p, W, keys.p, pre2.p);
    BG_LAUNCH_CHECK();
    ctx->toc(n_ent * (uint64_t(W) * 8 * 2 + 16));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  ss->pre2 = std::move(pre2);
  ss->keys = std::move(keys);
  ss->erec.reset();
  ss->oreads.reset();
  ss->edir.reset();
  ss->edir_bases = 0;
  return BG_OK;
}

int seqset_require_keys(bg_ctx* ctx, const bg_seqset* ss, const char* who) {
  BG_TRY(seqset_ensure_keys(ctx, const_cast<bg_seqset*>(ss)));
  if (ss && ss->n && !(ss->keys.p && ss->W)) { set_error("%s: the seqset holds no key table (opened from a file?)", who); return BG_EINVAL; }
  return BG_OK;
}

// BG_ITEMS_PROBE=1 (suffix_records_impl): the front half of the items path beside the build -- times and bucket sizes to stderr
int items_probe(bg_ctx* ctx, const uint64_t* owords, const uint16_t* len, const uint32_t* cnt, const uint64_t* roff, uint64_t o_base,
                uint64_t n_emit, int W, uint64_t n_rec) {
  if (!n_rec) return BG_OK;
  DevBuf<Item> a, b;
  BG_TRY(a.alloc(ctx, n_rec));
  BG_TRY(b.alloc(ctx, n_rec));
  hipEvent_t ev[6];
  for (auto& e : ev) BG_HIP(hipEventCreate(&e));
  BG_HIP(hipEventRecord(ev[0], ctx->stream));
  k_emit_items<<<cdiv(cdiv(n_emit, 64), 4), 256, 0, ctx->stream>>>(owords, len, cnt, roff, o_base, n_emit, W, a.p);
  BG_LAUNCH_CHECK();
  BG_HIP(hipEventRecord(ev[1], ctx->stream));
  const uint64_t whole[2] = {0, n_rec};
  BG_TRY(it_partition(ctx, a.p, b.p, n_rec, 1, whole, 56));
  BG_HIP(hipEventRecord(ev[2], ctx->stream));
  DevBuf<uint64_t> b8, b16;
  BG_TRY(b8.alloc(ctx, 257));
  BG_TRY(b16.alloc(ctx, 65537));
  it_bounds_kernel<<<2, 256, 0, ctx->stream>>>(b.p, n_rec, 8, b8.p);
  BG_LAUNCH_CHECK();
  std::vector<uint64_t> h8(257);
  BG_HIP(copy_d2h(ctx, h8.data(), b8.p, 257 * 8));
  BG_HIP(hipEventRecord(ev[3], ctx->stream));
  BG_TRY(it_partition(ctx, b.p, a.p, n_rec, 256, h8.data(), 48));
  BG_HIP(hipEventRecord(ev[4], ctx->stream));
  it_bounds_kernel<<<cdiv(65537, 256), 256, 0, ctx->stream>>>(a.p, n_rec, 16, b16.p);
  BG_LAUNCH_CHECK();
  std::vector<uint64_t> h16(65537);
  BG_HIP(copy_d2h(ctx, h16.data(), b16.p, 65537 * 8));
  BG_HIP(hipEventRecord(ev[5], ctx->stream));
  // the buckets in order (a -> b), then the checker
  DevBuf<uint8_t> status;
  DevBuf<unsigned long long> chk;
  BG_TRY(status.alloc(ctx, 65536));
  BG_TRY(chk.alloc(ctx, 8));
  BG_HIP(hipMemsetAsync(status.p, 0, 65536, ctx->stream));
  BG_HIP(hipMemsetAsync(chk.p, 0, 64, ctx->stream));
  hipEvent_t es[3];
  for (auto& e : es) BG_HIP(hipEventCreate(&e));
  BG_HIP(hipEventRecord(es[0], ctx->stream));
  DevBuf<uint64_t> runs, more, longer;
  DevBuf<unsigned long long> pc;  // the run kernels' counters (items_entries_impl's layout)
  const uint64_t run_cap = n_rec / 2 + 1, more_cap = n_rec / 5 + 1, long_cap = n_rec / 65 + 1;
  BG_TRY(runs.alloc(ctx, run_cap));
  BG_TRY(more.alloc(ctx, more_cap));
  BG_TRY(longer.alloc(ctx, long_cap));
  BG_TRY(pc.alloc(ctx, 8));
  BG_HIP(hipMemsetAsync(pc.p, 0, 64, ctx->stream));
  k_items_bucket_sort<<<65536, IB_THREADS, 0, ctx->stream>>>(a.p, b.p, b16.p, status.p, runs.p, pc.p, run_cap, 64 - 16 - IB_LOG2BINS);
  BG_LAUNCH_CHECK();
  BG_HIP(hipEventRecord(es[1], ctx->stream));
  {
    k_items_tie_small<<<unsigned(std::min<uint64_t>(cdiv(run_cap, 256), 1u << 16)), 256, 0, ctx->stream>>>(b.p, runs.p, pc.p, run_cap, owords, W, more.p, pc.p + 1,
                                                                                                      more_cap);
    BG_LAUNCH_CHECK();
    k_items_tie_wave<<<2048, 256, 0, ctx->stream>>>(b.p, more.p, pc.p + 1, more_cap, owords, W, longer.p, pc.p + 2, long_cap);
    BG_LAUNCH_CHECK();
    const uint32_t lds_members = kItBlockRun;
    BG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_items_tie_block), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(lds_members * kMaxWords * 8)));
    k_items_tie_block<<<256, 1024, size_t(lds_members) * kMaxWords * 8, ctx->stream>>>(b.p, longer.p, pc.p + 2, long_cap, owords, W, lds_members, kItBlockRun,
                                                                                         pc.p + 4);
    BG_LAUNCH_CHECK();
  }
  unsigned long long h_pc[8];
  BG_HIP(copy_d2h(ctx, h_pc, pc.p, 64));
  const unsigned long long n_runs = h_pc[0], n_more = h_pc[1];
  BG_HIP(hipMemsetAsync(chk.p + 3, 0, 8, ctx->stream));
  BG_HIP(hipEventRecord(es[2], ctx->stream));
  k_items_check_order<<<cdiv(n_rec, 256), 256, 0, ctx->stream>>>(a.p, n_rec, owords, W, chk.p + 6, chk.p + 2);  // (ids before; its order counts are unused)
  BG_LAUNCH_CHECK();
  k_items_check_order<<<cdiv(n_rec, 256), 256, 0, ctx->stream>>>(b.p, n_rec, owords, W, chk.p, chk.p + 5);
  BG_LAUNCH_CHECK();
  unsigned long long h_chk[6];
  std::vector<uint8_t> h_status(65536);
  BG_HIP(copy_d2h(ctx, h_chk, chk.p, 48));
  BG_HIP(copy_d2h(ctx, h_status.data(), status.p, 65536));
  float ms_sort = 0, ms_ties = 0;
  BG_HIP(hipEventElapsedTime(&ms_sort, es[0], es[1]));
  BG_HIP(hipEventElapsedTime(&ms_ties, es[1], es[2]));
  for (auto& e : es) (void)hipEventDestroy(e);
  uint64_t st1 = 0;
  for (uint8_t v : h_status) st1 += v == 1;
  fprintf(stderr, "items probe: bucket sort %.2f ms, %llu runs of equal first words (%llu of more than four) %.2f ms; out of order %llu, LCP notes missing or "
                  "wrong %llu, same items %s; left for the fall-back: %llu buckets too big, %llu runs longer than 64\n",
          ms_sort, n_runs, n_more, ms_ties, h_chk[0], h_chk[1], h_chk[2] == h_chk[5] ? "yes" : "NO", (unsigned long long)st1, h_chk[4]);
  BG_HIP(hipEventSynchronize(ev[5]));
  float ms[5];
  for (int i = 0; i < 5; ++i) BG_HIP(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
  for (auto& e : ev) (void)hipEventDestroy(e);
  uint64_t mx = 0, over4k = 0, over8k = 0, over16k = 0, in_over8k = 0, empty = 0;
  for (int v = 0; v < 65536; ++v) {
    const uint64_t c = h16[v + 1] - h16[v];
    mx = std::max(mx, c);
    over4k += c > 4096;
    over8k += c > 8192;
    over16k += c > 16384;
    if (c > 8192) in_over8k += c;
    empty += c == 0;
  }
  fprintf(stderr, "items probe: %llu records; emit %.2f ms, level 1 %.2f ms, bounds %.2f ms, level 2 %.2f ms, bounds %.2f ms; buckets: mean %.0f, largest %llu, "
                  "%llu over 4096, %llu over 8192 (holding %llu records), %llu over 16384, %llu empty\n",
          (unsigned long long)n_rec, ms[0], ms[1], ms[2], ms[3], ms[4], double(n_rec) / 65536.0, (unsigned long long)mx, (unsigned long long)over4k,
          (unsigned long long)over8k, (unsigned long long)in_over8k, (unsigned long long)over16k, (unsigned long long)empty);
  return BG_OK;
}

}  // namespace bg
