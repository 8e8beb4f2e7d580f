// items_build.hip — the entry set from 16-byte ITEMS instead of materialised rows (first lead).
//
// A suffix record (oriented read r, offset j) of the exact cover travels as
//     fw  = its first 32 bases (one word, zero padded), and
//     id  = r << 24 | j << 16 | length << 8 | flags        (flags: bit 0 the suffix one base earlier is a record too,
//                                                            bits 1-2 the base in front of it, bit 3 the read's last record)
// and is put in order by two 256-way partition passes on the leading 8 + 8 bits of fw and one LDS sort per bucket of the
// 65 536 that result; only items that tie on all 32 bases go back to the read for the rest of their string.  Nothing of
// 48 bytes per record is written, sorted by index and gathered back (k_emit_records / radix passes / k_gather_dedup_build).
//
// What is here so far is the front half -- items, the two partition levels, the bucket bounds -- behind BG_ITEMS_PROBE=1,
// which runs it beside the build (the build's result does not change) and reports the stages' times and the bucket sizes
// the LDS sort will have to take.  The same strings as SURVEY rows D1-D5 / E (expand.cpp:285-380, builder.cpp:85-108) in
// another order of work; no reference code has this shape.
#include "bg_device_funcs.h"
#include "bg_internal.h"

#include <stdlib.h>

#include <algorithm>
#include <vector>

namespace bg {

struct __attribute__((aligned(16))) Item { uint64_t fw, id; };

constexpr int IT_THREADS = 256;
constexpr int IT_PER = 8;
constexpr int IT_TILE = IT_THREADS * IT_PER;  // 2048 items = 32 KB staged in LDS
constexpr int IT_BINS = 256;

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
__device__ __forceinline__ void it_locate(const ItSegs& sg, uint32_t tile, uint64_t* lo, uint64_t* hi, uint64_t* hist_base, uint32_t* nt,
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

__global__ __launch_bounds__(IT_THREADS) void it_hist_kernel(const Item* __restrict__ items, ItSegs sg, int shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[IT_BINS];
  h[threadIdx.x] = 0;
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
  hist[hb + uint64_t(threadIdx.x) * nt + ti] = h[threadIdx.x];
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

__global__ __launch_bounds__(IT_THREADS) void it_scatter_kernel(const Item* __restrict__ in, Item* __restrict__ out, ItSegs sg, int shift,
                                                                const uint64_t* __restrict__ offs) {
  __shared__ Item stage[IT_TILE];
  __shared__ uint32_t h[IT_BINS];
  __shared__ uint32_t binstart[IT_BINS];
  __shared__ uint64_t goff[IT_BINS];
  __shared__ uint32_t scan_sm[4];
  const int tid = threadIdx.x;
  h[tid] = 0;
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
    const uint32_t ex = it_block_excl_scan(h[tid], scan_sm);
    binstart[tid] = ex;
    goff[tid] = offs[hb + uint64_t(tid) * nt + ti] - ex;
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
  DevBuf<uint32_t> hist, seg_tile;
  DevBuf<uint64_t> offs, seg0;
  BG_TRY(hist.alloc(ctx, size_t(IT_BINS) * ntiles));
  BG_TRY(offs.alloc(ctx, size_t(IT_BINS) * ntiles));
  BG_TRY(seg0.alloc(ctx, size_t(nseg) + 1));
  BG_TRY(seg_tile.alloc(ctx, size_t(nseg) + 1));
  BG_HIP(hipMemcpyAsync(seg0.p, h_seg0, (size_t(nseg) + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemcpyAsync(seg_tile.p, h_tile.data(), (size_t(nseg) + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  const ItSegs sg{seg0.p, seg_tile.p, int(nseg)};
  it_hist_kernel<<<ntiles, IT_THREADS, 0, ctx->stream>>>(in, sg, shift, hist.p);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32_to_u64(ctx, hist.p, offs.p, uint64_t(IT_BINS) * ntiles, nullptr));
  it_scatter_kernel<<<ntiles, IT_THREADS, 0, ctx->stream>>>(in, out, sg, shift, offs.p);
  BG_LAUNCH_CHECK();
  BG_HIP(hipStreamSynchronize(ctx->stream));  // (the segment tables go out of scope)
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
