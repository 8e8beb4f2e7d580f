// kmer_skm.hip — canonical k-mer counting (S2) through super-k-mers, for gfx950.
//
// Same result as the instance path in build_kmer.hip (rows B1-B3 of SURVEY.md §8a: kmer_counter.h:275-326,
// kmer_count_table.h:54-103, kmerize_bf.cpp:290-294) with a fifth of its memory traffic.  The instance path
// moves every k-mer occurrence (8 bytes) through two partition passes and the counting pass; here the unit
// that moves is a *super-k-mer*: a run of consecutive windows of one read that share their minimiser (the
// smallest order value over the canonical 12-mers of the window), stored as its bases (2 bits each) in one
// 16-byte record.  A window and its reverse complement have the same minimiser, so all occurrences of a
// canonical k-mer meet in the bucket its minimiser hashes to; a bucket is counted in an LDS table exactly as
// before, the windows being unrolled from the records on the fly.  At k = 30 a record carries ~10 windows:
// 16 bytes instead of 80.
//
// Record (two 64-bit words): a = bases 0..31; b = bases 32..50 (38 bits) | bucket (21 bits) | windows-1 (5 bits).
// A bucket whose distinct k-mers overflow the LDS table is unrolled into plain instances and counted by the
// sort-based path (kmer_count_keys_sort_impl), so the result never depends on how minimisers spread.
#include "bg_device_funcs.h"
#include "bg_internal.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace bg {

int kmer_count_keys_sort_impl(bg_ctx* ctx, uint64_t* ka, uint64_t* kb, uint64_t N, uint32_t min_count, bg_kmerset* ks);

#ifndef SK_GEN_SGPR
#define SK_GEN_SGPR 80
#endif
constexpr int SK_M = 12;          // minimiser length
constexpr int SK_THREADS = 256;
constexpr int SK_BINS = 256;
constexpr int SK_CAP = 1536;      // records one tile can stage in LDS
constexpr int SK_TARGET = 1024;   // records a tile is sized for (expected)
constexpr uint32_t SK_INF = 0xffffffffu;
constexpr int SK_BUCKET_FIELD = 21;
constexpr int SK_WORDS = 12;       // 8 words of a 255-base read + zero padding for the look-ahead fetches

struct __attribute__((aligned(16))) SkRec { uint64_t a, b; };

BG_HD int sk_windows(uint64_t b) { return int(b & 31) + 1; }
BG_HD uint32_t sk_bucket(uint64_t b) { return uint32_t(b >> 5) & ((1u << SK_BUCKET_FIELD) - 1); }
// order of a canonical m-mer: a fixed pseudo-random permutation, so that low-complexity m-mers are not
// everybody's minimiser; 31 bits, SK_INF marks "holds an N"
BG_HD uint32_t sk_order(uint32_t canon) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t x = __umul24(canon, 0x9E3779u);  // 24-bit multiplies run at full rate, 32-bit ones at a quarter
#else
  uint32_t x = (canon & 0xffffffu) * 0x9E3779u;
#endif
  x ^= x >> 15;
  return x >> 1;
}
// canonical form of a 12-mer in 32-bit arithmetic (the 64-bit kmer_canon costs three times as much)
__device__ __forceinline__ uint32_t sk_canon_mmer(uint32_t mm) {
  uint32_t rc = __brev(~mm);  // the 24 payload bits, complemented and reversed, now sit in bits 31..8
  rc = ((rc & 0xAAAAAAAAu) >> 1) | ((rc & 0x55555555u) << 1);
  rc >>= 32 - 2 * SK_M;
  return rc < mm ? rc : mm;
}
BG_HD uint32_t sk_bucket_of(uint32_t order) { return (order * 0x85EBCA6Bu) >> (32 - SK_BUCKET_FIELD); }

struct SkSource {
  const uint64_t* words;
  const uint32_t* nmask;
  const uint16_t* len;
  uint64_t n_reads;
  int W, k, reads_per_tile, nmax;
};

__device__ __forceinline__ uint32_t sk_block_excl_scan(uint32_t v, uint32_t* sm, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) sm[wave] = x;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += sm[w];
  *total = sm[0] + sm[1] + sm[2] + sm[3];
  return base + x - v;
}

__device__ __forceinline__ uint32_t sk2_block_excl_scan(uint32_t v, uint32_t* sm) {  // (any number of waves)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) sm[wave] = x;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += sm[w];
  return base + x - v;
}

__device__ __forceinline__ uint64_t sk_readlane64(uint64_t v, int l) {
  const uint32_t lo = __builtin_amdgcn_readlane(int(uint32_t(v)), l), hi = __builtin_amdgcn_readlane(int(uint32_t(v >> 32)), l);
  return (uint64_t(hi) << 32) | lo;
}
// LDS written by some lanes of a wave, read by others: keep the compiler from moving accesses across
__device__ __forceinline__ void sk_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Calls f(emit, read, first window, windows, minimiser order) for every super-k-mer of reads [r_lo, r_hi), from
// the lane of its last window.  Wave w takes reads r_lo + w, + 4, ...
// Windows holding an N (kmer_counter.h:306-311) belong to no record.  A chunk of 64 lanes computes the order
// values of 64 consecutive 12-mers and, by log-step shuffles, the minimisers of 65 - (k - 11) windows; the
// last of them only serves as look-ahead, so chunks advance by 64 - (k - 11).  sw / sn = this wave's LDS copy
// of the read's words and N masks.  The lane of a
// record's LAST window emits it: it needs the start of its run (a bit scan, or a carry from earlier chunks)
// and whether the next window starts a new one.
template <class F>
__device__ __forceinline__ void tile_skms(const SkSource& src, uint64_t r_lo, uint64_t r_hi, uint64_t* sw, uint32_t* sn, F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = src.W, k = src.k, w = src.k - SK_M + 1, nmax = src.nmax;
  const int S = 65 - w, stride = S - 1;
  int pw = 1;
  while (pw * 2 <= w) pw *= 2;
  const uint32_t rcp = (65536u + uint32_t(nmax) - 1) / uint32_t(nmax);  // x / nmax = (x * rcp) >> 16 for x < 400
  // byte addresses of the lanes the log-step minimum pulls from (loop invariant; lanes past the end pull
  // from lane 63 and end up with partial windows, which are not used)
  auto lane_addr = [&](int l) { return (l < 0 ? 0 : (l > 63 ? 63 : l)) << 2; };
  const int a1 = lane_addr(lane + 1), a2 = lane_addr(lane + 2), a4 = lane_addr(lane + 4), a8 = lane_addr(lane + 8),
            ax = lane_addr(lane + w - pw), aup = lane_addr(lane - 1);
  auto pull = [](int addr, uint32_t v) { return uint32_t(__builtin_amdgcn_ds_bpermute(addr, int(v))); };
  for (uint64_t i = r_lo + wave; i < r_hi; i += SK_THREADS / 64) {
    const int L = src.len[i];
    if (L < k) continue;  // wave-uniform
    const int nk = L - k + 1, nm = L - SK_M + 1;
    sk_wave_sync();  // the previous read's words are no longer needed
    uint32_t my_n = 0;
    if (lane < SK_WORDS) {  // the read's words (zero padded) where every lane can fetch the ones it needs, halves swapped:
      // as 32-bit words the copy then reads 16 bases after 16 bases, first base in the top bits
      const uint64_t wd = lane < W ? src.words[i * W + lane] : 0;
      sw[lane] = (wd << 32) | (wd >> 32);
      my_n = lane < W ? src.nmask[i * W + lane] : 0;
      sn[lane] = my_n;
    }
    const bool has_n = __ballot(my_n != 0) != 0;  // wave-uniform: nearly every read has none and skips the masks
    sk_wave_sync();
    uint32_t carry_mz = SK_INF;
    int carry_rs = -1;
    for (int base = 0; base < nk; base += stride) {
      const int p = base + lane;
      const int q = p >> 5, r = p & 31;
      // the 12-mer at base p: 24 bits out of two neighbouring 32-bit words (64-bit shifts of two 64-bit words cost three times
      // the instructions)
      const uint32_t* sd = reinterpret_cast<const uint32_t*>(sw);
      const uint32_t mm = __funnelshift_l(sd[(p >> 4) + 1], sd[p >> 4], (p & 15) << 1) >> (32 - 2 * SK_M);
      uint32_t nw = 0;
      if (has_n) {
        const uint32_t n0 = sn[q], n1 = sn[q + 1];
        nw = (n0 << r) | ((n1 >> 1) >> (31 - r));
      }
      uint32_t m = SK_INF;
      if (p < nm && (nw >> (32 - SK_M)) == 0) m = sk_order(sk_canon_mmer(mm));
      // minimum over the w order values from this lane on, in log steps
      uint32_t t;
      if (pw > 1) { t = pull(a1, m); m = t < m ? t : m; }
      if (pw > 2) { t = pull(a2, m); m = t < m ? t : m; }
      if (pw > 4) { t = pull(a4, m); m = t < m ? t : m; }
      if (pw > 8) { t = pull(a8, m); m = t < m ? t : m; }
      if (w > pw) { t = pull(ax, m); m = t < m ? t : m; }
      const bool inwin = lane < S && p < nk;
      const bool valid = inwin && (nw >> (32 - k)) == 0;
      const uint32_t mz = valid ? m : SK_INF;  // a valid window's minimiser is never SK_INF
      uint32_t prev = pull(aup, mz);
      if (lane == 0) prev = carry_mz;
      // boundary = window that starts a run of equal minimisers, or holds an N
      const uint64_t bm = __ballot(inwin && (!valid || prev != mz));
      const uint64_t le = bm & (~0ull >> (63 - lane));
      const int rs = le ? base + 63 - __clzll((long long)le) : carry_rs;  // start of this window's run, if valid
      const uint32_t dd = uint32_t(p - rs);
      // records of nmax windows ending at or before this one, times nmax (24-bit multiplies run at full rate, 32-bit
      // ones at a quarter: dd < 400, rcp < 3200)
      const uint32_t full_n = uint32_t(__umul24(uint32_t(__umul24(dd + 1, rcp)) >> 16, uint32_t(nmax)));
      const bool at_full = full_n == dd + 1;
      const bool next_starts = p + 1 >= nk || ((bm >> ((lane + 1) & 63)) & 1);
      const bool emit = valid && lane < stride && (next_starts || at_full);
      // the record's first window = rs + (dd / nmax) * nmax; the whole wave calls f (so that it can aggregate over
      // the emitting lanes)
      const int s0 = rs + int(at_full ? full_n - uint32_t(nmax) : full_n);
      f(emit, i, s0, p - s0 + 1, mz);
      carry_mz = uint32_t(__builtin_amdgcn_readlane(int(mz), stride - 1));
      const uint64_t lo = bm & ((1ull << stride) - 1);
      if (lo) carry_rs = base + 63 - __clzll((long long)lo);
    }
  }
}

// record of windows [s0, s0 + n) of a read (words = its W packed words)
__device__ __forceinline__ SkRec sk_build_record(const uint64_t* __restrict__ words, int W, int s0, int n, int k,
                                                 uint32_t order) {
  const int qs = s0 >> 5, rr = s0 & 31;
  const uint64_t r0 = words[qs], r1 = qs + 1 < W ? words[qs + 1] : 0, r2 = qs + 2 < W ? words[qs + 2] : 0;
  const int nbases = n + k - 1;  // <= 51
  SkRec r;
  r.a = (r0 << (2 * rr)) | ((r1 >> 1) >> (63 - 2 * rr));
  uint64_t bw = (r1 << (2 * rr)) | ((r2 >> 1) >> (63 - 2 * rr));
  if (nbases < 32) r.a &= ~0ull << (64 - 2 * nbases);
  bw = nbases > 32 ? (bw & (~0ull << (64 - 2 * (nbases - 32)))) : 0ull;
  r.b = bw | (uint64_t(sk_bucket_of(order)) << 5) | uint64_t(n - 1);
  return r;
}

// Level 0.  Phase 1 finds the super-k-mers of a tile of reads (one lane per window, most lanes idle when a
// record is cut) and only notes (read, first window, windows, minimiser) in LDS; phase 2 builds the records
// with every lane busy and appends them to `out` as one block (one global atomic per tile; the order of tiles
// is whatever the scheduler makes it — only grouping matters downstream).
__global__ __launch_bounds__(SK_THREADS) __attribute__((amdgpu_num_sgpr(SK_GEN_SGPR))) void sk_generate_kernel(SkSource src, SkRec* __restrict__ out,
                                                                 uint64_t* __restrict__ org, uint64_t cap,
                                                                 unsigned long long* __restrict__ cursor) {
  __shared__ uint64_t sdesc[SK_CAP];  // read in tile : 16 | first window : 8 | windows - 1 : 8 | order : 32
  __shared__ uint64_t sw[SK_THREADS / 64][SK_WORDS];
  __shared__ uint32_t sn[SK_THREADS / 64][SK_WORDS];
  __shared__ uint32_t s_cnt;
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  const uint64_t r_lo = uint64_t(blockIdx.x) * src.reads_per_tile;
  const uint64_t r_hi = r_lo + src.reads_per_tile < src.n_reads ? r_lo + src.reads_per_tile : src.n_reads;
  tile_skms(src, r_lo, r_hi, sw[tid >> 6], sn[tid >> 6], [&](bool emit, uint64_t read, int s0, int n, uint32_t order) {
    const uint64_t em = __ballot(emit);
    if (!em) return;  // wave-uniform
    // one LDS atomic per wave and chunk (the counter is a single address: per-lane atomics on it serialise)
    const int lane = tid & 63, leader = __ffsll((unsigned long long)em) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&s_cnt, uint32_t(__popcll(em)));
    base = uint32_t(__builtin_amdgcn_readlane(int(base), leader));
    if (!emit) return;
    const uint32_t s = base + uint32_t(__popcll(em & ((1ull << lane) - 1)));
    if (s < uint32_t(SK_CAP)) {
      sdesc[s] = (uint64_t(read - r_lo) << 48) | (uint64_t(s0) << 40) | (uint64_t(n - 1) << 32) | order;
    } else {  // the staging area is full (minimisers changing at nearly every window): one record at a time
      const unsigned long long g = atomicAdd(cursor, 1ull);
      if (g < cap) {
        out[g] = sk_build_record(src.words + read * src.W, src.W, s0, n, src.k, order);
        org[g] = (read << 8) | uint64_t(s0);
      }
    }
  });
  __syncthreads();
  const uint32_t total = s_cnt < uint32_t(SK_CAP) ? s_cnt : uint32_t(SK_CAP);
  if (tid == 0) s_base = atomicAdd(cursor, (unsigned long long)total);
  __syncthreads();
  const uint64_t base = s_base;
  for (uint32_t i = tid; i < total; i += SK_THREADS) {
    const uint64_t d = sdesc[i];
    const uint64_t read = r_lo + (d >> 48);
    const SkRec r = sk_build_record(src.words + read * src.W, src.W, int(d >> 40) & 0xff, (int(d >> 32) & 0xff) + 1, src.k,
                                    uint32_t(d));
    if (base + i < cap) {
      out[base + i] = r;
      org[base + i] = (read << 8) | ((d >> 40) & 0xff);  // origin: read : first window
    }
  }
}

// Level 0 and level 1 in one: the tile's records leave for the REGION of their bucket's top 8 bits (region b = places
// [b * bin_cap, (b + 1) * bin_cap) of out / org; bin_cursor[b] counts what it holds) instead of one block in arrival order that
// a histogram pass and a scatter pass then regroup -- a read of all records, a scan and a read + write of all of them less
// (0.52 + 0.13 + 1.26 ms on the bench's chr20; the generator pays 1.17 of it back: its writes are 16-byte pieces now).  A bin's records of one tile are reserved with one global atomic (256 a tile,
// on 256 addresses) and written side by side.  A region that would overflow sets *flag and keeps nothing past its end: the
// caller takes the two-pass way.
__global__ __launch_bounds__(SK_THREADS) __attribute__((amdgpu_num_sgpr(SK_GEN_SGPR))) void sk_generate_binned_kernel(
    SkSource src, SkRec* __restrict__ out, uint64_t* __restrict__ org, uint64_t bin_cap, unsigned long long* __restrict__ bin_cursor,
    uint32_t* __restrict__ flag) {
  __shared__ uint64_t sdesc[SK_CAP];  // read in tile : 16 | first window : 8 | windows - 1 : 8 | order : 32
  __shared__ uint64_t sw[SK_THREADS / 64][SK_WORDS];
  __shared__ uint32_t sn[SK_THREADS / 64][SK_WORDS];
  __shared__ uint32_t s_cnt;
  __shared__ uint32_t bh[SK_BINS], bfill[SK_BINS];
  __shared__ unsigned long long bbase[SK_BINS];
  static_assert(SK_THREADS == SK_BINS, "a thread a bin");
  const int tid = threadIdx.x;
  if (tid == 0) s_cnt = 0;
  bh[tid] = 0;
  bfill[tid] = 0;
  __syncthreads();
  const uint64_t r_lo = uint64_t(blockIdx.x) * src.reads_per_tile;
  const uint64_t r_hi = r_lo + src.reads_per_tile < src.n_reads ? r_lo + src.reads_per_tile : src.n_reads;
  constexpr int kBinShift = SK_BUCKET_FIELD - 8;
  tile_skms(src, r_lo, r_hi, sw[tid >> 6], sn[tid >> 6], [&](bool emit, uint64_t read, int s0, int n, uint32_t order) {
    const uint64_t em = __ballot(emit);
    if (!em) return;  // wave-uniform
    const int lane = tid & 63, leader = __ffsll((unsigned long long)em) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&s_cnt, uint32_t(__popcll(em)));
    base = uint32_t(__builtin_amdgcn_readlane(int(base), leader));
    if (!emit) return;
    const uint32_t s = base + uint32_t(__popcll(em & ((1ull << lane) - 1)));
    if (s < uint32_t(SK_CAP)) {
      sdesc[s] = (uint64_t(read - r_lo) << 48) | (uint64_t(s0) << 40) | (uint64_t(n - 1) << 32) | order;
    } else {  // the staging area is full (minimisers changing at nearly every window): one record at a time
      const uint32_t bin = sk_bucket_of(order) >> kBinShift;
      const unsigned long long g = atomicAdd(&bin_cursor[bin], 1ull);
      if (g < bin_cap) {
        out[uint64_t(bin) * bin_cap + g] = sk_build_record(src.words + read * src.W, src.W, s0, n, src.k, order);
        org[uint64_t(bin) * bin_cap + g] = (read << 8) | uint64_t(s0);
      } else {
        atomicOr(flag, 1u);
      }
    }
  });
  __syncthreads();
  // (Measured and put away, 7.50 ms as it stands against 6.34 for the plain generator: the tile's descriptions put in bin order
  // in LDS first, so that consecutive lanes write consecutive places -- 8.96 ms, the registers and the second staging array
  // cost more waves than the writes gain; a record's place in its bin counted where the record comes about -- 7.72 ms.)
  const uint32_t total = s_cnt < uint32_t(SK_CAP) ? s_cnt : uint32_t(SK_CAP);
  for (uint32_t i = tid; i < total; i += SK_THREADS) atomicAdd(&bh[sk_bucket_of(uint32_t(sdesc[i])) >> kBinShift], 1u);
  __syncthreads();
  {  // every bin's share of its region, reserved with one global atomic
    const uint32_t c = bh[tid];
    unsigned long long g = 0;
    if (c) {
      g = atomicAdd(&bin_cursor[tid], (unsigned long long)c);
      if (g + c > bin_cap) { atomicOr(flag, 1u); g = ~0ull; }
    }
    bbase[tid] = g;
  }
  __syncthreads();
  for (uint32_t i = tid; i < total; i += SK_THREADS) {
    const uint64_t d = sdesc[i];
    const uint32_t bin = sk_bucket_of(uint32_t(d)) >> kBinShift;
    const uint32_t slot = atomicAdd(&bfill[bin], 1u);
    const unsigned long long g = bbase[bin];
    if (g == ~0ull) continue;
    const uint64_t read = r_lo + (d >> 48);
    const uint64_t o = uint64_t(bin) * bin_cap + g + slot;
    out[o] = sk_build_record(src.words + read * src.W, src.W, int(d >> 40) & 0xff, (int(d >> 32) & 0xff) + 1, src.k, uint32_t(d));
    org[o] = (read << 8) | ((d >> 40) & 0xff);  // origin: read : first window
  }
}

// ---- further levels: one unstable pass over records already grouped into segments ------------------
// A tile of 3 072 records is 64 KB of LDS: two workgroups a CU.  With 256 threads of 12 records each that was 8 waves a CU and
// the two scatter passes took 1.84 + 1.82 ms; 256 x 8 (three workgroups): 1.61 + 1.62; 512 x 6: 1.40 + 1.45; 1 024 x 3
// (32 waves a CU): 1.26 + 1.35 -- 7.5 GB moved at 5.8 TB/s; 1 024 x 2: 1.37 + 1.39.
constexpr int SK2_THREADS = 1024;
constexpr int SK2_ITEMS = 3;
constexpr int SK2_TILE = SK2_THREADS * SK2_ITEMS;
static_assert(SK2_THREADS >= SK_BINS && SK2_THREADS % 64 == 0, "a thread a bin in the partition kernels");
struct SkSegs {
  const uint64_t* seg_rec0;   // S+1 record offsets
  const uint32_t* seg_tile0;  // S+1 tile offsets
  int nseg;
  const uint64_t* seg_end;    // null: segment s ends where s + 1 starts; else its end (segments with room behind them: the
                              // regions sk_generate_binned_kernel fills)
};

__device__ __forceinline__ void sk_locate(const SkSegs& sg, uint32_t tile, uint64_t* lo, uint64_t* hi, uint64_t* hist_base,
                                          uint32_t* tiles_in_seg, uint32_t* tile_in_seg) {
  int a = 0, b = sg.nseg;  // last segment with seg_tile0 <= tile
  while (b - a > 1) {
    int m = (a + b) >> 1;
    if (sg.seg_tile0[m] <= tile) a = m; else b = m;
  }
  const uint32_t t0 = sg.seg_tile0[a];
  *tiles_in_seg = sg.seg_tile0[a + 1] - t0;
  *tile_in_seg = tile - t0;
  *lo = sg.seg_rec0[a] + uint64_t(tile - t0) * SK2_TILE;
  const uint64_t end = sg.seg_end ? sg.seg_end[a] : sg.seg_rec0[a + 1];
  *hi = *lo + SK2_TILE < end ? *lo + SK2_TILE : end;
  *hist_base = uint64_t(t0) * SK_BINS;
}

__global__ __launch_bounds__(SK2_THREADS) void sk_part_hist_kernel(const SkRec* __restrict__ recs, SkSegs sg, int shift,
                                                                  uint32_t dmask, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[SK_BINS];
  if (threadIdx.x < SK_BINS) h[threadIdx.x] = 0;
  __syncthreads();
  uint64_t lo, hi, hbase;
  uint32_t nt, ti;
  sk_locate(sg, uint32_t(xcd_chunk(blockIdx.x, gridDim.x)), &lo, &hi, &hbase, &nt, &ti);  // XCD-contiguous tiles
#pragma unroll
  for (int it = 0; it < SK2_ITEMS; ++it) {
    const uint64_t idx = lo + uint64_t(it) * SK2_THREADS + threadIdx.x;
    if (idx < hi) atomicAdd(&h[(sk_bucket(recs[idx].b) >> shift) & dmask], 1u);
  }
  __syncthreads();
  if (threadIdx.x < SK_BINS) hist[hbase + uint64_t(threadIdx.x) * nt + ti] = h[threadIdx.x];
}

// Org = what travels with a record: its 64-bit origin (read : first window), or — the count's own levels — its
// 32-bit index in the order the records came in (kIndex: made here, oin is not read), which is all the count
// needs to find the origin of the one record in ten it has something to tell.  full_bits (kIndex only, may be null):
// receives "all windows kept" for every record, in input order.
template <typename Org, bool kIndex>
__global__ __launch_bounds__(SK2_THREADS) void sk_part_scatter_kernel(const SkRec* __restrict__ in, SkRec* __restrict__ out,
                                                                     const Org* __restrict__ oin,
                                                                     Org* __restrict__ oout, SkSegs sg, int shift,
                                                                     uint32_t dmask, const uint64_t* __restrict__ offs,
                                                                     uint32_t* __restrict__ full_bits) {
  __shared__ SkRec srec[SK2_TILE];
  __shared__ Org sorg[SK2_TILE];
  __shared__ uint32_t h[SK_BINS];
  __shared__ uint32_t binstart[SK_BINS];
  __shared__ uint64_t goff[SK_BINS];
  __shared__ uint32_t scan_sm[SK2_THREADS / 64];
  const int tid = threadIdx.x;
  if (tid < SK_BINS) h[tid] = 0;
  uint64_t lo, hi, hbase;
  uint32_t nt, ti;
  sk_locate(sg, uint32_t(xcd_chunk(blockIdx.x, gridDim.x)), &lo, &hi, &hbase, &nt, &ti);  // XCD-contiguous tiles
  __syncthreads();
  SkRec rec[SK2_ITEMS];
  Org og[SK2_ITEMS];
  uint16_t rnk[SK2_ITEMS];
#pragma unroll
  for (int it = 0; it < SK2_ITEMS; ++it) {
    const uint64_t idx = lo + uint64_t(it) * SK2_THREADS + tid;
    og[it] = 0;
    if (idx < hi) {
      rec[it] = in[idx];
      og[it] = kIndex ? Org(idx) : oin[idx];
      if (kIndex && full_bits) {  // "every window kept" for the record at this place of the input (sk_spread_kernel)
        const int n = sk_windows(rec[it].b);
        full_bits[idx] = n == 32 ? ~0u : (1u << n) - 1;
      }
      rnk[it] = uint16_t(atomicAdd(&h[(sk_bucket(rec[it].b) >> shift) & dmask], 1u));
    } else {
      rec[it].a = 0;
      rec[it].b = 0;
      rnk[it] = 0;
    }
  }
  __syncthreads();
  {
    const uint32_t ex = sk2_block_excl_scan(tid < SK_BINS ? h[tid] : 0u, scan_sm);
    if (tid < SK_BINS) {
      binstart[tid] = ex;
      goff[tid] = offs[hbase + uint64_t(tid) * nt + ti] - ex;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < SK2_ITEMS; ++it) {
    const uint64_t idx = lo + uint64_t(it) * SK2_THREADS + tid;
    if (idx < hi) {
      const uint32_t at = binstart[(sk_bucket(rec[it].b) >> shift) & dmask] + rnk[it];
      srec[at] = rec[it];
      sorg[at] = og[it];
    }
  }
  __syncthreads();
  const uint32_t cnt = uint32_t(hi - lo);
  for (uint32_t i = tid; i < cnt; i += SK2_THREADS) {
    const SkRec r = srec[i];
    const uint64_t o = goff[(sk_bucket(r.b) >> shift) & dmask] + i;
    out[o] = r;
    oout[o] = sorg[i];
  }
}

// first record whose leading `bits` bucket bits are >= v, for v = 0 .. 2^bits (records grouped by those bits)
__global__ void sk_bounds_kernel(const SkRec* __restrict__ recs, uint64_t n, int bits, uint64_t* __restrict__ bounds) {
  const uint64_t v = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (v > (1ull << bits)) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if ((sk_bucket(recs[mid].b) >> (SK_BUCKET_FIELD - bits)) < v) lo = mid + 1; else hi = mid;
  }
  bounds[v] = lo;
}


// ---- equal records are counted once --------------------------------------------------------------------
// At 30x coverage a super-k-mer that lies inside its reads comes ~20 times as the very same record (same bases,
// same window count); only records cut by a read end or holding a sequencing error are alone.  Before the count
// the records of a bucket therefore meet, a chunk at a time, in a small LDS hash table (slot = multiplicity :
// index of the first record of that content, the representative): the representatives go to the count with
// their multiplicity (a quarter to a third of the records), every record keeps the bucket-local index of its
// representative, and the presence bits found for a representative are handed to its copies afterwards
// (sk_spread_kernel).  Equal records of different chunks are simply not merged.
constexpr int SKD_THREADS = 1024;
constexpr int SKD_PER = 3;
constexpr int SKD_CHUNK = SKD_PER * SKD_THREADS;
constexpr int SKD_BITS = 12;
constexpr int SKD_SLOTS = 1 << SKD_BITS;  // >= 4/3 of the chunk: the table never fills

// Representatives that stand for at least min_count records ("full" ones: every window of theirs is a kept
// k-mer whatever else happens) fill the bucket's slice of `reps` from its start, the others from its end
// backwards; nrep[2 * bucket], nrep[2 * bucket + 1] = how many of each.  Within a chunk both kinds are in
// order of decreasing window count, so that the 64 records of a wave of the count need the same number of
// steps, or nearly.  rep_of = a record's representative (its place in `reps`).
__global__ __launch_bounds__(SKD_THREADS) void sk_dedup_kernel(const SkRec* __restrict__ recs,
                                                               const uint64_t* __restrict__ bounds, uint32_t min_count,
                                                               SkRec* __restrict__ reps, uint16_t* __restrict__ mult,
                                                               uint32_t* __restrict__ rep_of, uint32_t* __restrict__ nrep) {
  __shared__ uint32_t s_dh[SKD_SLOTS];
  __shared__ SkRec s_rec[SKD_CHUNK];  // the chunk's records: a copy is compared with its representative here, not in HBM
  __shared__ uint32_t s_hist[64];     // full representatives by window count (32 bins), then the others
  __shared__ uint32_t s_full, s_other;
  const uint32_t bucket = blockIdx.x;
  const uint64_t lo = bounds[bucket], hi = bounds[bucket + 1];
  const uint32_t last = uint32_t(hi - lo) - 1;
  uint32_t base_full = 0, base_other = 0;  // representatives of the earlier chunks
  for (uint64_t c0 = lo; c0 < hi; c0 += SKD_CHUNK) {  // block-uniform trip count
    const uint32_t cn = hi - c0 < uint64_t(SKD_CHUNK) ? uint32_t(hi - c0) : uint32_t(SKD_CHUNK);
    for (uint32_t i = threadIdx.x; i < uint32_t(SKD_SLOTS); i += SKD_THREADS) s_dh[i] = ~0u;
    if (threadIdx.x < 64) s_hist[threadIdx.x] = 0;
    __syncthreads();
    SkRec r[SKD_PER];
    uint32_t myslot[SKD_PER];
    bool first[SKD_PER];
#pragma unroll
    for (int j = 0; j < SKD_PER; ++j) {
      const uint32_t q = uint32_t(j) * SKD_THREADS + threadIdx.x;
      r[j].a = 0;
      r[j].b = 0;
      if (q < cn) {
        r[j] = recs[c0 + q];
        s_rec[q] = r[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SKD_PER; ++j) {
      const uint32_t q = uint32_t(j) * SKD_THREADS + threadIdx.x;
      myslot[j] = 0;
      first[j] = false;
      if (q < cn) {
        uint32_t slot = ((uint32_t(r[j].a) ^ uint32_t(r[j].a >> 32)) * 0x9E3779B1u ^
                         (uint32_t(r[j].b) ^ uint32_t(r[j].b >> 32)) * 0x85EBCA6Bu) >> (32 - SKD_BITS);
        for (;;) {
          uint32_t e = s_dh[slot];
          if (e == ~0u) {
            e = atomicCAS(&s_dh[slot], ~0u, (1u << 16) | q);
            if (e == ~0u) {  // first of its content
              first[j] = true;
              break;
            }
          }
          const SkRec other = s_rec[e & 0xffffu];
          if (other.a == r[j].a && other.b == r[j].b) {
            atomicAdd(&s_dh[slot], 1u << 16);
            break;
          }
          slot = (slot + 1) & uint32_t(SKD_SLOTS - 1);
        }
        myslot[j] = slot;
      }
    }
    __syncthreads();
    uint32_t m[SKD_PER], bin[SKD_PER], rank[SKD_PER];
#pragma unroll
    for (int j = 0; j < SKD_PER; ++j) {
      m[j] = 0;
      bin[j] = 0;
      rank[j] = 0;
      if (first[j]) {
        m[j] = s_dh[myslot[j]] >> 16;  // <= SKD_CHUNK
        bin[j] = (m[j] >= min_count ? 0u : 32u) + 31u - uint32_t(r[j].b & 31);
        rank[j] = atomicAdd(&s_hist[bin[j]], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the bins by the first wave, each kind from zero
      const uint32_t v = s_hist[threadIdx.x];
      uint32_t x = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (int(threadIdx.x) >= d) x += y;
      }
      const uint32_t n_full = __shfl(x, 31, 64), n_all = __shfl(x, 63, 64);
      s_hist[threadIdx.x] = x - v - (threadIdx.x >= 32 ? n_full : 0u);
      if (threadIdx.x == 0) { s_full = n_full; s_other = n_all - n_full; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SKD_PER; ++j) {
      if (first[j]) {
        const uint32_t pos = s_hist[bin[j]] + rank[j];
        const uint32_t idx = bin[j] < 32 ? base_full + pos : last - (base_other + pos);
        reps[lo + idx] = r[j];
        mult[lo + idx] = uint16_t(m[j]);
        s_dh[myslot[j]] = idx;  // nobody else reads this slot's multiplicity
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SKD_PER; ++j) {
      const uint32_t q = uint32_t(j) * SKD_THREADS + threadIdx.x;
      if (q < cn) rep_of[c0 + q] = uint32_t(lo) + s_dh[myslot[j]];  // (absolute: a count takes < 2^32 records)
    }
    base_full += s_full;
    base_other += s_other;
    __syncthreads();  // the tables are reset by the next chunk
  }
  if (threadIdx.x == 0) {
    nrep[2 * bucket] = base_full;
    nrep[2 * bucket + 1] = base_other;
  }
}

// What the count found goes back to the records.  Local count (bits_out == nullptr): the presence words start as
// "every window without an N is kept" (sk_present_valid_kernel) and rep_bits holds the windows NOT known to be
// kept — nine records in ten have none, so one record in ten costs an atomic instead of every one (the 142 M
// atomicOrs of the earlier version wrote 8 GB for a 0.4 GB bitmap).  The records of a bucket that was not
// counted in LDS lose all their bits here and get them back from the finished hash set (kmer_skm_finish_present).
// Remote records (bits_out): the array starts as "every window kept" too (the first partition level writes that as
// the records pass), and only the records with a window that is not kept, or of an uncounted bucket (kSkUnknown),
// are written here.  idx = a record's place in the order the count received the records in (what the
// partition moves along with it), org = the origins (read : first window) in that order.
constexpr uint32_t kSkUnknown = 0x80000000u;
constexpr int SKS_THREADS = 256;
__global__ __launch_bounds__(SKS_THREADS) void sk_spread_kernel(const SkRec* __restrict__ recs, uint64_t n_rec,
                                                                int bucket_bits, const uint8_t* __restrict__ overflow,
                                                                const uint32_t* __restrict__ rep_of,
                                                                const uint32_t* __restrict__ rep_bits,
                                                                const uint32_t* __restrict__ idx,
                                                                const uint64_t* __restrict__ org,
                                                                unsigned long long* __restrict__ present, int pstride,
                                                                uint32_t* __restrict__ bits_out) {
  // four records per thread: their look-ups (representative -> its bits; for the record in ten that has something
  // to clear, index -> origin -> presence word) are four dependent chains in flight instead of one
  const uint64_t i0 = uint64_t(blockIdx.x) * (4 * SKS_THREADS) + threadIdx.x;
  uint32_t v[4], full[4];  // windows not known to be kept, all windows
  bool uncounted[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint64_t i = i0 + uint64_t(u) * SKS_THREADS;
    v[u] = i < n_rec ? rep_of[i] : 0u;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint64_t i = i0 + uint64_t(u) * SKS_THREADS;
    v[u] = i < n_rec ? rep_bits[v[u]] : 0u;
  }
  uint32_t at[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint64_t i = i0 + uint64_t(u) * SKS_THREADS;
    at[u] = 0;
    full[u] = 0;
    uncounted[u] = false;
    if (v[u]) {  // (i < n_rec)
      at[u] = idx[i];
      if (bits_out || v[u] == kSkUnknown) {  // (a local record with windows to clear needs nothing of the record itself)
        const uint64_t rb = recs[i].b;
        const int n = sk_windows(rb);
        full[u] = n == 32 ? ~0u : (1u << n) - 1;
        // kSkUnknown marks the representatives of a bucket that was not counted — or is window 31 alone: ask the bucket
        if (v[u] == kSkUnknown && overflow[sk_bucket(rb) >> (SK_BUCKET_FIELD - bucket_bits)]) {
          uncounted[u] = true;
          v[u] = full[u];  // every window is to be looked up later
        }
      }
    }
  }
  if (bits_out) {  // (they start as "every window kept": sk_part_scatter_kernel)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (v[u]) bits_out[at[u]] = uncounted[u] ? kSkUnknown : (full[u] & ~v[u]);
    return;
  }
  uint64_t o[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) o[u] = v[u] ? org[at[u]] : 0;  // the record's origin, read : first window
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (!v[u]) continue;
    const uint64_t read = o[u] >> 8;
    const int s0 = int(o[u] & 0xff), wi = s0 >> 6, sh = s0 & 63;
    atomicAnd(&present[read * pstride + wi], ~((unsigned long long)v[u] << sh));
    const unsigned long long over = sh ? (unsigned long long)v[u] >> (64 - sh) : 0ull;  // windows past the word
    if (over) atomicAnd(&present[read * pstride + wi + 1], ~over);
  }
}

// presence words of a local count before it: one bit per window that holds no N (the windows records are made of)
__global__ __launch_bounds__(256) void sk_present_valid_kernel(const uint32_t* __restrict__ nmask,
                                                               const uint16_t* __restrict__ len, uint64_t n_reads,
                                                               int W, int k, unsigned long long* __restrict__ present,
                                                               int pstride) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_reads) return;
  const int nk = int(len[i]) - k + 1;
  uint32_t any_n = 0;  // nearly every read has no N at all and needs nothing but its length
  if (nmask) {         // (nullptr: no read of the whole input holds one)
#pragma unroll
    for (int q = 0; q < 8; ++q)  // W <= 8; all the loads in flight at once
      if (q < W) any_n |= nmask[i * W + q];
  }
  unsigned long long m[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t0 = 64 * j;
    m[j] = 0;
    if (nk > t0) m[j] = nk - t0 >= 64 ? ~0ull : (1ull << (nk - t0)) - 1;
  }
  if (any_n) {
    // windows that hold an N = the N bits of the bases (bit b of word w = base 64 w + b) smeared over the k - 1
    // positions before each, in log steps.  (One read in 70 holds an N at a rate of 1e-4: two waves in three have such
    // a lane, and a loop over its windows — 121 rounds of dependent loads — was nine tenths of the kernel.)
    unsigned long long nb[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < W) {
        const unsigned long long rev = __brev(nmask[i * W + q]);  // bit r = base 32 q + r
        nb[q >> 1] |= rev << (32 * (q & 1));
      }
    }
    for (int width = 1; width < k;) {
      const int sft = width < k - width ? width : k - width;
#pragma unroll
      for (int w = 0; w < 4; ++w) nb[w] |= (nb[w] >> sft) | (nb[w + 1] << (64 - sft));
      width += sft;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] &= ~nb[j];
  }
  if (pstride == 2) {
    *reinterpret_cast<ulonglong2*>(present + i * 2) = make_ulonglong2(m[0], m[1]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < pstride) present[i * pstride + j] = m[j];
  }
}

// ---- counting: one workgroup per bucket, windows unrolled from the records into an LDS table ---------
constexpr int SKC_THREADS = 1024;
#ifdef SKC_TIMING  // tools/build_variant.sh <name> kmer_skm.hip -DSKC_TIMING: where a workgroup's time goes (stderr)
__device__ unsigned long long g_skc_times[8];
#define SKC_T(n) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_skc_times[n], now_ - t_last_); t_last_ = now_; } } while (0)
#else
#define SKC_T(n) do {} while (0)
#endif

// recs / mult: the bucket's representatives as sk_dedup_kernel leaves them (nrep[2 * bucket] full ones from the
// start of the slice, nrep[2 * bucket + 1] others from its end backwards) and how many records each stands
// for; bits_out[i] = the windows of representative i that are NOT known to be kept;
// kSkUnknown when the bucket is not counted here (overflow[bucket] is set as well).  bucket_list: the buckets to
// count (one workgroup each) when not all of them.
// home slot of a k-mer in a table of cap slots: the key folded to 32 bits, one multiply, one multiply-high.  (Two
// 24-bit multiplies instead — full rate, not a quarter — change nothing: the probe loop waits on LDS, not on the VALU.)
__device__ __forceinline__ uint32_t skc_slot(unsigned long long key, uint32_t cap) {
  return __umulhi((uint32_t(key) ^ uint32_t(key >> 31)) * 0x9E3779B1u, cap);
}
#ifndef SKC_SPLIT
#define SKC_SPLIT 4  // lanes that share the windows of one representative (a power of two, neighbours in a wave)
#endif
__global__ __launch_bounds__(SKC_THREADS) __attribute__((amdgpu_num_sgpr(80))) void sk_count_lds_kernel(const SkRec* __restrict__ recs,
                                                                   const uint16_t* __restrict__ mult,
                                                                   const uint64_t* __restrict__ bounds,
                                                                   const uint32_t* __restrict__ nrep, uint32_t cap,
                                                                   int k, uint32_t min_count, uint64_t out_cap,
                                                                   uint64_t* __restrict__ out_kmers,
                                                                   uint32_t* __restrict__ out_counts,
                                                                   unsigned long long* __restrict__ out_n,
                                                                   unsigned long long* __restrict__ distinct_total,
                                                                   uint8_t* __restrict__ overflow,
                                                                   uint64_t* __restrict__ ulist,
                                                                   uint32_t* __restrict__ bits_out,
                                                                   uint64_t giant, const uint32_t* __restrict__ bucket_list) {
  extern __shared__ __align__(16) unsigned char skc_smem[];
  unsigned long long* tkeys = reinterpret_cast<unsigned long long*>(skc_smem);
  uint32_t* tcnt = reinterpret_cast<uint32_t*>(skc_smem + size_t(cap) * 8);
  __shared__ uint32_t s_over, s_kept, s_distinct, s_cursor, s_unsettled, s_crossed;
  __shared__ unsigned long long s_base;
  const uint32_t bucket = bucket_list ? bucket_list[blockIdx.x] : blockIdx.x;  // (a list: the second try, below)
  const uint64_t lo = bounds[bucket], end = bounds[bucket + 1];
  if (lo == end) return;  // block-uniform
  const uint32_t n_full = nrep[2 * bucket], n_other = nrep[2 * bucket + 1];
  if (end - lo > giant) {
    // a repeat family: one workgroup would spend tens of milliseconds serialising on a few LDS counters while
    // the rest of the GPU idles — such a bucket goes to the sort-based count with the overflowed ones
    if (threadIdx.x == 0) overflow[bucket] = 1;
    for (uint64_t i = lo + threadIdx.x; i < end; i += SKC_THREADS) bits_out[i] = kSkUnknown;
    return;
  }
#ifdef SKC_TIMING
  unsigned long long t_last_ = wall_clock64();
  const unsigned long long t_first_ = t_last_;
  if (threadIdx.x == 0) atomicMax(&g_skc_times[7], atomicAdd(&g_skc_times[6], 1ull) + 1);
#endif
  for (uint32_t i = threadIdx.x; i < cap; i += SKC_THREADS) { tkeys[i] = ~0ull; tcnt[i] = 0; }
  if (threadIdx.x == 0) { s_over = 0; s_kept = 0; s_distinct = 0; s_cursor = 0; s_unsettled = 0; s_crossed = 0; }
  __syncthreads();
  SKC_T(0);
  const uint32_t max_probe = cap < 256u ? cap : 256u;
  const uint64_t kmask = ~(~0ull << (2 * k));
  // One flat loop per batch of work: every iteration is one probe of every live lane, and a lane whose probe
  // hits moves on to its next window by itself.  (Nesting "for each window { probe until placed }" makes the
  // whole wave wait for its slowest lane at every window, and its exec-mask bookkeeping saturated the scalar
  // unit.)
  // SKC_SPLIT neighbouring lanes share the windows of a record: few records are left, and the time of a pass is
  // the number of steps of its longest lane.
  // Pass 0, the full representatives: their windows are kept k-mers (the count only grows and starts at
  // min_count or more).
  // Pass 1, the others: a window whose own increment reaches min_count is known to be kept —
  // after pass 0 that is every window of a k-mer that a full representative holds too, which leaves the k-mers
  // of sequencing errors — and the rest are listed for the second sweep.  That sweep only has something to find
  // when a pass-1 increment took a counter across min_count (the windows that counted before it are the ones
  // it would correct): most buckets skip it.
  for (int pass = 0; pass < 2; ++pass) {
    const uint32_t items = (pass == 0 ? n_full : n_other) * SKC_SPLIT;
    for (uint32_t it0 = 0; it0 < items; it0 += SKC_THREADS) {  // block-uniform trip count
      const uint32_t it = it0 + threadIdx.x;
      bool live = it < items;
      const uint64_t i = pass == 0 ? lo + it / SKC_SPLIT : end - 1 - it / SKC_SPLIT;
      SkRec r;
      r.a = 0;
      r.b = 0;
      uint32_t m = 0;
      if (live) {
        r = recs[i];
        m = mult[i];
      }
      const int n = sk_windows(r.b);
      const int per = (n + SKC_SPLIT - 1) / SKC_SPLIT;
      int t = int(it % SKC_SPLIT) * per;
      const int t_end = t + per < n ? t + per : n;
      if (t >= t_end) live = false;
      uint32_t bits = 0, und = 0, new_kept = 0, new_distinct = 0;
      bool crossed = false;
      // window t, and the bases that enter the following windows, next one on top (k <= 31, t <= 31: only base
      // bits of b come into it; its field bits only come up after the last window)
      const uint64_t x = t ? ((r.a << (2 * t)) | (r.b >> (64 - 2 * t))) : r.a;
      uint64_t nb = ((r.a << (2 * k)) | (r.b >> (64 - 2 * k))) << (2 * t);
      uint64_t fwd = x >> (64 - 2 * k);
      uint64_t rc = kmer_revcomp(fwd, k);
      unsigned long long key = rc < fwd ? rc : fwd;
      uint32_t slot = skc_slot(key, cap);
      uint32_t probes = 0;
      while (__any(live)) {
        if (live) {
          // LDS atomics run at about a lane per clock: most occurrences find their key already in place, and a
          // plain read (full rate) spares them the compare-and-swap
          unsigned long long cur = __atomic_load_n(&tkeys[slot], __ATOMIC_RELAXED);
          if (cur == ~0ull) {
            cur = atomicCAS(&tkeys[slot], ~0ull, key);
            if (cur == ~0ull) {
              cur = key;
              ++new_distinct;
            }
          }
          if (cur == key) {
            const uint32_t seen = atomicAdd(&tcnt[slot], m) + m;
            if (seen >= min_count) {
              bits |= 1u << t;
              if (seen - m < min_count) {  // this increment made the k-mer a kept one
                crossed = true;
                ++new_kept;
              }
            } else {
              und |= 1u << t;
            }
            if (++t >= t_end) {
              live = false;
            } else {
              const uint64_t base = nb >> 62;  // the base that enters the window
              nb <<= 2;
              fwd = ((fwd << 2) | base) & kmask;
              rc = (rc >> 2) | ((base ^ 3) << (2 * (k - 1)));
              key = rc < fwd ? rc : fwd;
              slot = skc_slot(key, cap);
              probes = 0;
            }
          } else {
            slot = slot + 1 == cap ? 0 : slot + 1;
            if (++probes >= max_probe) { s_over = 1; live = false; }
          }
        }
      }
      // the lanes of a record put their windows together
#pragma unroll
      for (int d = 1; d < SKC_SPLIT; d <<= 1) {
        bits |= uint32_t(__shfl_xor(int(bits), d, 64));
        und |= uint32_t(__shfl_xor(int(und), d, 64));
      }
      if (crossed && pass == 1) s_crossed = 1;
      {  // one LDS atomic per wave (the counters are single addresses); a lane has at most 32 windows
        uint32_t wk = 0, wd = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          wk += uint32_t(__popcll(__ballot((new_kept >> b) & 1))) << b;
          wd += uint32_t(__popcll(__ballot((new_distinct >> b) & 1))) << b;
        }
        if ((threadIdx.x & 63) == 0) {
          if (wk) atomicAdd(&s_kept, wk);
          if (wd) atomicAdd(&s_distinct, wd);
        }
      }
      if (it < items && it % SKC_SPLIT == 0) {
        if (pass == 0) {
          bits_out[i] = 0u;
        } else {
          // the unsettled records are listed densely (in the bucket's own slice of ulist) so that the second
          // sweep keeps every lane busy
          if (und) ulist[lo + atomicAdd(&s_unsettled, 1u)] = (uint64_t(i - lo) << 32) | und;
          bits_out[i] = und;
        }
      }
    }
    if (pass == 0) SKC_T(4);
    __syncthreads();  // pass 1 relies on every increment of pass 0
  }
  SKC_T(1);
  if (s_over) {
    if (threadIdx.x == 0) overflow[bucket] = 1;
    for (uint64_t i = lo + threadIdx.x; i < end; i += SKC_THREADS) bits_out[i] = kSkUnknown;
    return;
  }
  // The counters are final: room for the bucket's kept k-mers is asked for now, and the answer comes while the
  // second sweep runs.  (Kept and distinct k-mers were counted as they came about.)
  if (threadIdx.x == 0) {
    s_base = s_kept ? atomicAdd(out_n, (unsigned long long)s_kept) : 0ull;
    if (s_distinct) atomicAdd(distinct_total, (unsigned long long)s_distinct);
  }
  // Second sweep, the table being final: the windows that were not settled above, SKC_SPLIT lanes to a record.
  const uint32_t n_unsettled = (s_crossed ? s_unsettled : 0u) * SKC_SPLIT;  // nothing crossed: none of them is kept
  for (uint32_t q0 = 0; q0 < n_unsettled; q0 += SKC_THREADS) {
    const uint32_t q = q0 + threadIdx.x;
    const uint64_t ent = q < n_unsettled ? ulist[lo + q / SKC_SPLIT] : 0ull;
    uint32_t und = uint32_t(ent);
    const uint64_t i = lo + (ent >> 32);
    SkRec r;
    r.a = 0;
    r.b = 0;
    if (und) r = recs[i];
    {
      const int per = (sk_windows(r.b) + SKC_SPLIT - 1) / SKC_SPLIT;  // <= 8
      und &= (per >= 32 ? ~0u : (1u << per) - 1) << (int(q % SKC_SPLIT) * per);
    }
    uint32_t bits = 0, slot = 0;
    int t = -1;  // window being looked up, -1: take the next one
    unsigned long long key = 0;
    while (__any(und != 0)) {
      if (und) {
        if (t < 0) {
          t = __ffs(und) - 1;
          const uint64_t x = t ? ((r.a << (2 * t)) | (r.b >> (64 - 2 * t))) : r.a;  // t <= 31: only base bits of b
          const uint64_t fwd = x >> (64 - 2 * k);
          const uint64_t rc = kmer_revcomp(fwd, k);
          key = rc < fwd ? rc : fwd;
          slot = skc_slot(key, cap);
        }
        if (tkeys[slot] == key) {  // every window's key is in the table (no overflow)
          if (tcnt[slot] >= min_count) bits |= 1u << t;
          und &= und - 1;
          t = -1;
        } else {
          slot = slot + 1 == cap ? 0 : slot + 1;
        }
      }
    }
#pragma unroll
    for (int d = 1; d < SKC_SPLIT; d <<= 1) bits |= uint32_t(__shfl_xor(int(bits), d, 64));
    if (bits && q % SKC_SPLIT == 0) {  // this thread is the only one touching the record in this sweep
      bits_out[i] &= ~bits;
    }
  }
  __syncthreads();
  SKC_T(2);
#ifdef SKC_TIMING
  if (threadIdx.x == 0) {
    atomicMax(&g_skc_times[5], ((t_last_ - t_first_) << 24) | bucket);
    atomicAdd(&g_skc_times[6], ~0ull);
  }
#endif
  if (!s_kept) return;
  const uint64_t out_base = s_base;
  for (uint32_t i0 = 0; i0 < cap; i0 += SKC_THREADS) {  // block-uniform trip count
    const uint32_t i = i0 + threadIdx.x;
    const bool keep = i < cap && tkeys[i] != ~0ull && tcnt[i] >= min_count;
    const uint64_t km = __ballot(keep);
    if (!km) continue;  // wave-uniform
    const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long)km) - 1;
    uint32_t at = 0;
    if (lane == leader) at = atomicAdd(&s_cursor, uint32_t(__popcll(km)));
    at = uint32_t(__builtin_amdgcn_readlane(int(at), leader)) + uint32_t(__popcll(km & ((1ull << lane) - 1)));
    if (keep && out_base + at < out_cap) { out_kmers[out_base + at] = tkeys[i]; out_counts[out_base + at] = tcnt[i]; }
  }
  SKC_T(3);
}

// windows of the records of the overflowed buckets as plain canonical instances, nmax slots per record (unused
// slots keep the all-ones sentinel the sort-based count skips).  Segment s = records [seg_start[s], +len) of the
// partitioned array, placed at seg_prefix[s] of the output; also copies records and origins for the presence
// look-ups that follow the hash build (pend_* may be null).
__global__ void sk_expand_kernel(const SkRec* __restrict__ recs, const uint32_t* __restrict__ idx,
                                 const uint64_t* __restrict__ org, const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_prefix,
                                 uint32_t nseg, uint64_t total, int k, int nmax, uint64_t* __restrict__ keys,
                                 SkRec* __restrict__ pend_rec, uint64_t* __restrict__ pend_org) {
  const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= total) return;
  uint32_t a = 0, b = nseg;  // last segment whose prefix is <= g
  while (b - a > 1) {
    const uint32_t m = (a + b) >> 1;
    if (seg_prefix[m] <= g) a = m; else b = m;
  }
  const uint64_t i = seg_start[a] + (g - seg_prefix[a]);
  const SkRec r = recs[i];
  if (pend_rec) {
    pend_rec[g] = r;
    pend_org[g] = org[idx[i]];
  }
  const int n = sk_windows(r.b);
  const uint64_t kmask = ~(~0ull << (2 * k));
  uint64_t fwd = r.a >> (64 - 2 * k);
  uint64_t rc = kmer_revcomp(fwd, k);
  for (int t = 0; t < n; ++t) {
    keys[g * nmax + t] = rc < fwd ? rc : fwd;
    const int idx = k + t;
    const uint64_t base = idx < 32 ? (r.a >> (62 - 2 * idx)) & 3 : (r.b >> (62 - 2 * (idx - 32))) & 3;
    fwd = ((fwd << 2) | base) & kmask;
    rc = (rc >> 2) | ((3 - base) << (2 * (k - 1)));
  }
}

// presence bits of records the LDS count gave up on, from the finished hash set
__global__ void sk_present_lookup_kernel(const SkRec* __restrict__ recs, const uint64_t* __restrict__ org, uint64_t n_rec,
                                         int k, const uint64_t* __restrict__ slots, int log2cap,
                                         unsigned long long* __restrict__ present, int pstride) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  const SkRec r = recs[i];
  const int n = sk_windows(r.b);
  KmerTable tab{slots, log2cap, k};
  const uint64_t kmask = ~(~0ull << (2 * k));
  uint64_t fwd = r.a >> (64 - 2 * k);
  uint64_t rc = kmer_revcomp(fwd, k);
  uint32_t bits = 0;
  for (int t = 0; t < n; ++t) {
    if (tab.find(rc < fwd ? rc : fwd) >= 0) bits |= 1u << t;
    const int idx = k + t;
    const uint64_t base = idx < 32 ? (r.a >> (62 - 2 * idx)) & 3 : (r.b >> (62 - 2 * (idx - 32))) & 3;
    fwd = ((fwd << 2) | base) & kmask;
    rc = (rc >> 2) | ((3 - base) << (2 * (k - 1)));
  }
  if (bits) {
    const uint64_t o = org[i];
    const uint64_t read = o >> 8;
    const int s0 = int(o & 0xff), wi = s0 >> 6, sh = s0 & 63;
    atomicOr(&present[read * pstride + wi], (unsigned long long)bits << sh);
    if (sh + n > 64) atomicOr(&present[read * pstride + wi + 1], (unsigned long long)bits >> (64 - sh));
  }
}

// after kmer_hash_build_impl: presence bits of the pending records (overflowed buckets)
int kmer_skm_finish_present(bg_ctx* ctx, bg_kmerset* ks) {
  if (ks->pend_n && ks->present.p) {
    sk_present_lookup_kernel<<<cdiv(ks->pend_n, 256), 256, 0, ctx->stream>>>(
        reinterpret_cast<const SkRec*>(ks->pend_rec.p), ks->pend_org.p, ks->pend_n, ks->k, ks->slots.p, ks->log2cap,
        reinterpret_cast<unsigned long long*>(ks->present.p), ks->present_stride);
    BG_LAUNCH_CHECK();
  }
  ks->pend_rec.reset();
  ks->pend_org.reset();
  ks->pend_n = 0;
  return BG_OK;
}

template <typename Org, bool kIndex>
static int sk_partition_segmented(bg_ctx* ctx, const SkRec* in, SkRec* out, const Org* oin, Org* oout, uint64_t n,
                                  uint32_t nseg, const uint64_t* h_rec0, int shift, uint32_t dmask,
                                  uint32_t* full_bits = nullptr, const uint64_t* h_end = nullptr) {
  std::vector<uint32_t> h_tile(size_t(nseg) + 1, 0);
  uint64_t t = 0;
  for (uint32_t b = 0; b < nseg; ++b) {
    h_tile[b] = uint32_t(t);
    t += ((h_end ? h_end[b] : h_rec0[b + 1]) - h_rec0[b] + SK2_TILE - 1) / SK2_TILE;
  }
  if (t >= (1ull << 31)) { set_error("partition of %llu super-k-mers needs too many tiles", (unsigned long long)n); return BG_EINVAL; }
  h_tile[nseg] = uint32_t(t);
  const uint32_t ntiles = uint32_t(t);
  if (!ntiles) return BG_OK;
  DevBuf<uint32_t> hist, seg_tile;
  DevBuf<uint64_t> offs, seg_rec, seg_end;
  BG_TRY(hist.alloc(ctx, size_t(SK_BINS) * ntiles));
  BG_TRY(offs.alloc(ctx, size_t(SK_BINS) * ntiles));
  BG_TRY(seg_rec.alloc(ctx, size_t(nseg) + 1));
  BG_TRY(seg_tile.alloc(ctx, size_t(nseg) + 1));
  BG_HIP(hipMemcpyAsync(seg_rec.p, h_rec0, (size_t(nseg) + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemcpyAsync(seg_tile.p, h_tile.data(), (size_t(nseg) + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (h_end) {
    BG_TRY(seg_end.alloc(ctx, nseg));
    BG_HIP(hipMemcpyAsync(seg_end.p, h_end, size_t(nseg) * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  SkSegs sg{seg_rec.p, seg_tile.p, int(nseg), h_end ? seg_end.p : nullptr};
  ctx->tic("skm_part_hist");
  sk_part_hist_kernel<<<ntiles, SK2_THREADS, 0, ctx->stream>>>(in, sg, shift, dmask, hist.p);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 16);
  ctx->tic("radix_scan");
  BG_TRY(exclusive_scan_u32_to_u64(ctx, hist.p, offs.p, uint64_t(SK_BINS) * ntiles, nullptr));
  ctx->toc(uint64_t(SK_BINS) * ntiles * 12);
  ctx->tic("skm_part_scatter");
  sk_part_scatter_kernel<Org, kIndex><<<ntiles, SK2_THREADS, 0, ctx->stream>>>(in, out, oin, oout, sg, shift, dmask, offs.p,
                                                                               full_bits);
  BG_LAUNCH_CHECK();
  ctx->toc(n * (32 + (kIndex ? 1 : 2) * sizeof(Org)));
  return BG_OK;
}

// slots of a bucket's LDS table: 6400 x 12 bytes lets two 1024-thread workgroups share a CU's 160 KB
// (BG_KMER_LDS_LOG2CAP shrinks the table: tests force the overflow path with it; BG_KMER_LDS_SLOTS / _SLOTS2 set the sizes
// of the first and the second try)
static uint32_t sk_lds_slots() {
  const char* e = getenv("BG_KMER_LDS_LOG2CAP");
  if (const char* s = getenv("BG_KMER_LDS_SLOTS")) return uint32_t(std::max(2, std::min(13000, atoi(s))));  // (keeps the second try)
  if (!e) return 6400;
  const int v = atoi(e);
  return 1u << (v < 1 ? 1 : (v > 13 ? 13 : v));
}

// ---- host side ----------------------------------------------------------------------------------------

struct SkGeometry {
  int nmax, w, maxw;
  SkSource src;
  uint32_t ntiles;
  uint64_t cap;
};

// false: the super-k-mer path does not apply (k too small for a 12-base minimiser, no windows)
static bool sk_geometry(const bg_reads* reads, int k, uint64_t N, SkGeometry* g) {
  g->maxw = int(reads->max_len) - k + 1;
  if (k < SK_M + 2 || k > 31 || g->maxw <= 0 || !N) return false;
  g->nmax = std::min(32, 52 - k);
  g->w = k - SK_M + 1;
  const double per_read = double(g->maxw) * 2.0 / double(g->w + 1) + 1.5;
  g->src = SkSource{reads->words.p, reads->nmask.p, reads->len.p, reads->n, reads->W, k,
                    std::max(1, int(double(SK_TARGET) / per_read)), g->nmax};
  // test hooks: BG_SKM_READS_PER_TILE overfills the LDS staging area (one-record-at-a-time path),
  // BG_SKM_CAP shrinks the record buffer (fallback to the instance path)
  if (const char* e = getenv("BG_SKM_READS_PER_TILE")) g->src.reads_per_tile = std::max(1, std::min(60000, atoi(e)));
  const uint64_t ntiles64 = (reads->n + g->src.reads_per_tile - 1) / g->src.reads_per_tile;
  if (ntiles64 >= (1ull << 31)) return false;
  g->ntiles = uint32_t(ntiles64);
  // The number of records is not known beforehand: room for 3x the expected density (2 / (w + 1) records per
  // window); an input that beats it takes the instance path.
  g->cap = std::min<uint64_t>(N, uint64_t(double(N) * 6.0 / double(g->w + 1)) + reads->n + 1024);
  if (const char* e = getenv("BG_SKM_CAP")) g->cap = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  return true;
}

// level 0: records (+ origins) straight from the packed reads into ra / oa (g.cap slots); *R = their number
// (> g.cap: the buffer was too small and its content is incomplete)
static int sk_generate(bg_ctx* ctx, const bg_reads* reads, const SkGeometry& g, SkRec* ra, uint64_t* oa, uint64_t* R_out) {
  DevBuf<unsigned long long> cursor;
  BG_TRY(cursor.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(cursor.p, 0, 8, ctx->stream));
  ctx->tic("skm_generate");
  sk_generate_kernel<<<g.ntiles, SK_THREADS, 0, ctx->stream>>>(g.src, ra, oa, g.cap, cursor.p);
  BG_LAUNCH_CHECK();
  unsigned long long R = 0;
  BG_HIP(hipMemcpyAsync(&R, cursor.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  ctx->toc(R * 24 + reads->n * uint64_t(reads->W) * 12);
  *R_out = R;
  return BG_OK;
}

// level 1: the top 8 bucket bits (in -> out); h_starts[257] = group starts
template <typename Org, bool kIndex>
static int sk_level1(bg_ctx* ctx, const SkRec* in, SkRec* out, const Org* oin, Org* oout, uint64_t R,
                     uint64_t* h_starts, uint32_t* full_bits = nullptr) {
  const uint64_t whole[2] = {0, R};
  BG_TRY((sk_partition_segmented<Org, kIndex>(ctx, in, out, oin, oout, R, 1, whole, SK_BUCKET_FIELD - 8, 0xffu, full_bits)));
  DevBuf<uint64_t> b8;
  BG_TRY(b8.alloc(ctx, SK_BINS + 1));
  sk_bounds_kernel<<<cdiv(SK_BINS + 1, 256), 256, 0, ctx->stream>>>(out, R, 8, b8.p);
  BG_LAUNCH_CHECK();
  BG_HIP(hipMemcpyAsync(h_starts, b8.p, (SK_BINS + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

static int sk_count_records(bg_ctx* ctx, const SkRec* in, const uint64_t* oin, uint64_t R, uint64_t N, int k,
                            uint32_t min_count, bg_kmerset* ks, uint32_t* bits_out, const uint64_t* binned_start = nullptr,
                            const uint64_t* binned_end = nullptr);

// level 0 + 1 in one kernel (sk_generate_binned_kernel): ra / oa hold 256 regions of bin_cap places; h_count[b] = records in
// region b.  *fits = false: a region overflowed (the content is incomplete: take sk_generate and the two-pass level 1).
static int sk_generate_binned(bg_ctx* ctx, const bg_reads* reads, const SkGeometry& g, SkRec* ra, uint64_t* oa, uint64_t bin_cap,
                              uint64_t* h_count, uint64_t* R_out, bool* fits) {
  DevBuf<unsigned long long> cursor;
  DevBuf<uint32_t> flag;
  BG_TRY(cursor.alloc(ctx, SK_BINS));
  BG_TRY(flag.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(cursor.p, 0, SK_BINS * 8, ctx->stream));
  BG_HIP(hipMemsetAsync(flag.p, 0, 4, ctx->stream));
  ctx->tic("skm_generate");
  sk_generate_binned_kernel<<<g.ntiles, SK_THREADS, 0, ctx->stream>>>(g.src, ra, oa, bin_cap, cursor.p, flag.p);
  BG_LAUNCH_CHECK();
  uint32_t h_flag = 0;
  BG_HIP(hipMemcpyAsync(h_count, cursor.p, SK_BINS * 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(&h_flag, flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  uint64_t R = 0;
  bool ok = h_flag == 0;
  for (int b = 0; b < SK_BINS; ++b) {
    if (h_count[b] > bin_cap) ok = false;
    R += h_count[b];
  }
  ctx->toc(R * 24 + reads->n * uint64_t(reads->W) * 12);
  *R_out = R;
  *fits = ok;
  return BG_OK;
}

// Counts the canonical k-mers of `reads` (N windows in all, for sizing) into ks->kmers / counts / distinct and
// fills ks->present.  *done = false when this path does not apply and nothing was done.
int kmer_count_skm(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t N, uint32_t min_count, bg_kmerset* ks, bool* done) {
  *done = false;
  SkGeometry g;
  if (!sk_geometry(reads, k, N, &g)) return BG_OK;
  DevBuf<SkRec> ra;
  DevBuf<uint64_t> oa;  // origin (read : first window) of every record, moved along with it
  BG_TRY(ra.alloc(ctx, g.cap));
  BG_TRY(oa.alloc(ctx, g.cap));
  uint64_t R = 0;
  ks->present_stride = std::max(1, (g.maxw + 63) / 64);
  BG_TRY(ks->present.alloc(ctx, reads->n * ks->present_stride));
  // BG_SKM_BINNED=1: the records leave the generator grouped by the top byte of their bucket (sk_generate_binned_kernel: 256
  // regions of cap / 256 places, three times a bin's expected share; a region that overflows sends the input through the plain
  // generator and the two-pass first level).  Measured on the bench's chr20 and left off: the first level's histogram, scan
  // and scatter go (1.9 ms) but the generator's writes are 16-byte pieces instead of one block (6.34 -> 7.50 ms): 0.6 ms of
  // 48.9 for the step, and the dominant kernel a fifth slower.
  std::vector<uint64_t> bin_start, bin_end;
  {
    // (tests: =big gives every region the whole buffer's length -- small inputs, whose bins are far from even, stay on this
    // path; =tiny gives them 8 places -- the overflow and the way back)
    const char* e = getenv("BG_SKM_BINNED");
    uint64_t bin_cap = g.cap / SK_BINS;
    if (e && !strcmp(e, "big")) bin_cap = g.cap;
    if (e && !strcmp(e, "tiny")) bin_cap = 8;
    if (bin_cap > g.cap / SK_BINS) {  // (room for the regions)
      ra.reset();
      oa.reset();
      BG_TRY(ra.alloc(ctx, bin_cap * SK_BINS));
      BG_TRY(oa.alloc(ctx, bin_cap * SK_BINS));
    }
    if (e && e[0] != '0' && bin_cap >= 8 && bin_cap * SK_BINS < (1ull << 32)) {
      std::vector<uint64_t> cnt(SK_BINS);
      bool fits = false;
      BG_TRY(sk_generate_binned(ctx, reads, g, ra.p, oa.p, bin_cap, cnt.data(), &R, &fits));
      if (fits) {
        bin_start.resize(SK_BINS + 1);
        bin_end.resize(SK_BINS);
        for (int b = 0; b <= SK_BINS; ++b) bin_start[b] = uint64_t(b) * bin_cap;
        for (int b = 0; b < SK_BINS; ++b) bin_end[b] = bin_start[b] + cnt[b];
      }
    }
  }
  if (bin_start.empty()) {
    BG_TRY(sk_generate(ctx, reads, g, ra.p, oa.p, &R));
    if (R > g.cap) {  // *done stays false
      ks->present.reset();
      return BG_OK;
    }
  }
  *done = true;
  if (reads->W > 8) { set_error("k-mer counting: reads of more than 256 bases"); return BG_EINVAL; }
  sk_present_valid_kernel<<<cdiv(reads->n, 256), 256, 0, ctx->stream>>>(
      reads->any_n ? reads->nmask.p : nullptr, reads->len.p, reads->n, reads->W, k,
      reinterpret_cast<unsigned long long*>(ks->present.p), ks->present_stride);
  BG_LAUNCH_CHECK();
  ks->present_of = reads;
  ks->present_n = reads->n;
  return sk_count_records(ctx, ra.p, oa.p, R, N, k, min_count, ks, nullptr, bin_start.empty() ? nullptr : bin_start.data(),
                          bin_start.empty() ? nullptr : bin_end.data());
}

// Partition + count of R records (in / oin are only read).  bits_out == nullptr: oin holds their origins and the
// presence bits go to ks->present (local count); else oin is not used and bits_out[i] receives the presence bits
// of the windows of record i (kSkUnknown for the records of overflowed buckets).
// binned_start / binned_end (257 / 256 values): `in` / `oin` are sk_generate_binned's regions -- the first level is done, the
// records' indices are their places in the regions.
static int sk_count_records(bg_ctx* ctx, const SkRec* in, const uint64_t* oin, uint64_t R, uint64_t N, int k,
                            uint32_t min_count, bg_kmerset* ks, uint32_t* bits_out, const uint64_t* binned_start,
                            const uint64_t* binned_end) {
  const int nmax = std::min(32, 52 - k);
  ks->distinct = 0;
  ks->size = 0;
  ks->sorted = false;
  if (!R) {  // every window holds an N
    BG_TRY(ks->kmers.alloc(ctx, 0));
    BG_TRY(ks->counts.alloc(ctx, 0));
    return BG_OK;
  }
  DevBuf<SkRec> rb, rc;
  if (R >= (1ull << 32)) { set_error("k-mer counting: %llu super-k-mers on one GPU (2^32 at most)", (unsigned long long)R); return BG_EINVAL; }
  DevBuf<uint32_t> ob, oc;  // every record's index in `in`, moved along with it
  BG_TRY(rb.alloc(ctx, R));
  BG_TRY(ob.alloc(ctx, R));
  BG_TRY(rc.alloc(ctx, R));
  BG_TRY(oc.alloc(ctx, R));
  std::vector<uint64_t> h_starts(SK_BINS + 1);
  // level 2: the next 8 bits inside each of the 256 groups; bigger inputs take up to 5 more bits so that a
  // bucket stays around 24 k windows (BG_KMER_EXTRA_BITS overrides, for tests).  A rank of a sharded build
  // owns a slice of the bucket space: the level is chosen from the groups that are in use.
  if (binned_start) {
    // (the generator grouped the records by the top byte already: this level reads its regions and makes the indices)
    for (int b8 = 0; b8 < SK_BINS; ++b8) { h_starts[b8] = binned_start[b8]; }
    h_starts[SK_BINS] = binned_start[SK_BINS];
    BG_TRY((sk_partition_segmented<uint32_t, true>(ctx, in, rc.p, nullptr, oc.p, R, SK_BINS, binned_start, SK_BUCKET_FIELD - 16, 0xffu, nullptr,
                                                   binned_end)));
  } else {
    // (remote records: bits_out starts as "every window kept" and the spread only writes what differs)
    BG_TRY((sk_level1<uint32_t, true>(ctx, in, rb.p, nullptr, ob.p, R, h_starts.data(), bits_out)));
    BG_TRY((sk_partition_segmented<uint32_t, false>(ctx, rb.p, rc.p, ob.p, oc.p, R, SK_BINS, h_starts.data(), SK_BUCKET_FIELD - 16, 0xffu)));
  }
  int extra = 0;
  {
    uint64_t used = 0;
    for (int b8 = 0; b8 < SK_BINS; ++b8) used += binned_start ? binned_end[b8] > binned_start[b8] : h_starts[b8 + 1] > h_starts[b8];
    const uint64_t per_bucket = N / (256 * (used ? used : 1));
    while (extra < SK_BUCKET_FIELD - 16 && (per_bucket >> extra) > 24000) ++extra;
  }
  if (const char* e = getenv("BG_KMER_EXTRA_BITS")) extra = std::max(0, std::min(SK_BUCKET_FIELD - 16, atoi(e)));
  SkRec* part = rc.p;
  uint32_t* part_idx = oc.p;
  if (extra) {
    DevBuf<uint64_t> b16;
    BG_TRY(b16.alloc(ctx, 65537));
    sk_bounds_kernel<<<cdiv(65537, 256), 256, 0, ctx->stream>>>(rc.p, R, 16, b16.p);
    BG_LAUNCH_CHECK();
    std::vector<uint64_t> h(65537);
    BG_HIP(hipMemcpyAsync(h.data(), b16.p, 65537 * 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    BG_TRY((sk_partition_segmented<uint32_t, false>(ctx, rc.p, rb.p, oc.p, ob.p, R, 65536, h.data(),
                                                    SK_BUCKET_FIELD - 16 - extra, (1u << extra) - 1)));
    part = rb.p;
    part_idx = ob.p;
  }
  const int bucket_bits = 16 + extra;
  const uint32_t NB = 1u << bucket_bits;

  DevBuf<uint64_t> bounds;
  DevBuf<uint8_t> overflow;
  DevBuf<unsigned long long> ctr;  // [0] kept cursor, [1] distinct
  BG_TRY(bounds.alloc(ctx, uint64_t(NB) + 1));
  BG_TRY(overflow.alloc(ctx, NB));
  BG_TRY(ctr.alloc(ctx, 2));
  BG_HIP(hipMemsetAsync(overflow.p, 0, NB, ctx->stream));
  BG_HIP(hipMemsetAsync(ctr.p, 0, 16, ctx->stream));
  const uint64_t out_cap = N / min_count + 1;
  DevBuf<uint64_t> ok_a;
  DevBuf<uint32_t> oc_a;
  BG_TRY(ok_a.alloc(ctx, out_cap));
  BG_TRY(oc_a.alloc(ctx, out_cap));
  DevBuf<uint64_t> ulist;  // per bucket: (representative : windows) whose presence the first sweep could not settle
  BG_TRY(ulist.alloc(ctx, R));
  ctx->tic("skm_dedup");
  sk_bounds_kernel<<<cdiv(uint64_t(NB) + 1, 256), 256, 0, ctx->stream>>>(part, R, bucket_bits, bounds.p);
  BG_LAUNCH_CHECK();
  // every bucket's distinct records, in the first places of its slice of `reps`
  SkRec* reps = part == rb.p ? rc.p : rb.p;  // the other partition buffer is free
  DevBuf<uint16_t> rep_mult;
  DevBuf<uint32_t> rep_of, rep_bits, nrep;
  BG_TRY(rep_mult.alloc(ctx, R));
  BG_TRY(rep_of.alloc(ctx, R));
  BG_TRY(rep_bits.alloc(ctx, R));
  BG_TRY(nrep.alloc(ctx, 2 * uint64_t(NB)));
  sk_dedup_kernel<<<NB, SKD_THREADS, 0, ctx->stream>>>(part, bounds.p, min_count, reps, rep_mult.p, rep_of.p, nrep.p);
  BG_LAUNCH_CHECK();
  ctx->toc(R * 38);
  ctx->tic("skm_count_lds");
  const uint32_t slots = sk_lds_slots();
  const size_t smem = size_t(slots) * 12;
  // buckets with more than 64 records per thread (30x the average at chr20) are not counted in LDS
  uint64_t giant = 64ull * SKC_THREADS;
  if (const char* e = getenv("BG_SKM_GIANT")) giant = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  BG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sk_count_lds_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
  sk_count_lds_kernel<<<NB, SKC_THREADS, smem, ctx->stream>>>(reps, rep_mult.p, bounds.p, nrep.p, slots, k, min_count, out_cap,
                                                             ok_a.p, oc_a.p, ctr.p, ctr.p + 1, overflow.p, ulist.p, rep_bits.p,
                                                             giant, nullptr);
  BG_LAUNCH_CHECK();
#ifdef SKC_TIMING
  {
    unsigned long long h_t[8], z[8] = {0};
    BG_HIP(hipStreamSynchronize(ctx->stream));
    BG_HIP(hipMemcpyFromSymbol(h_t, HIP_SYMBOL(g_skc_times), sizeof(h_t)));
    fprintf(stderr, "sk_count_lds_kernel, 100 MHz ticks summed over workgroups: init %llu pass 0 %llu pass 1 %llu second sweep %llu output %llu; slowest workgroup before its output: %llu ticks (bucket %llu); most workgroups at once %llu\n",
            h_t[0], h_t[4], h_t[1], h_t[2], h_t[3], h_t[5] >> 24, h_t[5] & 0xffffff, h_t[7]);
    BG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_skc_times), z, sizeof(z)));
  }
#endif
  ctx->toc(R * 40);
  unsigned long long h_ctr[2] = {0, 0};
  std::vector<uint8_t> h_over(NB);
  uint64_t n_over = 0;
  auto read_back = [&]() -> int {
    BG_HIP(hipMemcpyAsync(h_ctr, ctr.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipMemcpyAsync(h_over.data(), overflow.p, NB, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    n_over = 0;
    for (uint32_t b = 0; b < NB; ++b) n_over += h_over[b];
    return BG_OK;
  };
  BG_TRY(read_back());
  if (n_over && !getenv("BG_KMER_LDS_LOG2CAP") && !getenv("BG_SKM_NO_SECOND_TRY")) {
    // A bucket whose k-mers did not fit the table (a few times the average bucket: a handful at chr20) gets a
    // second try with a CU's whole LDS to itself before its windows go through the sort-based count.
    std::vector<uint32_t> list;
    for (uint32_t b = 0; b < NB; ++b)
      if (h_over[b]) list.push_back(b);
    DevBuf<uint32_t> d_list;
    BG_TRY(d_list.alloc(ctx, list.size()));
    BG_HIP(hipMemcpyAsync(d_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipMemsetAsync(overflow.p, 0, NB, ctx->stream));
    uint32_t big_slots = 13000;  // 156 KB
    if (const char* s = getenv("BG_KMER_LDS_SLOTS2")) big_slots = uint32_t(std::max(2, std::min(13000, atoi(s))));  // tests
    BG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sk_count_lds_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(big_slots * 12)));
    sk_count_lds_kernel<<<unsigned(list.size()), SKC_THREADS, size_t(big_slots) * 12, ctx->stream>>>(
        reps, rep_mult.p, bounds.p, nrep.p, big_slots, k, min_count, out_cap, ok_a.p, oc_a.p, ctr.p, ctr.p + 1, overflow.p,
        ulist.p, rep_bits.p, giant, d_list.p);
    BG_LAUNCH_CHECK();
    BG_TRY(read_back());  // (also keeps `list` alive until the copy has been made)
  }
  ctx->tic("skm_spread");
  sk_spread_kernel<<<cdiv(R, 4 * SKS_THREADS), SKS_THREADS, 0, ctx->stream>>>(part, R, bucket_bits, overflow.p, rep_of.p, rep_bits.p, part_idx, oin,
                                                        reinterpret_cast<unsigned long long*>(ks->present.p),
                                                        ks->present_stride, bits_out);
  BG_LAUNCH_CHECK();
  ctx->toc(R * 24);
  uint64_t nkept = h_ctr[0], distinct = h_ctr[1];
  if (n_over) {  // unroll the overflowed buckets into plain instances and count those by sorting
    std::vector<uint64_t> h_bounds(size_t(NB) + 1);
    BG_HIP(hipMemcpyAsync(h_bounds.data(), bounds.p, (size_t(NB) + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t tot_rec = 0;
    for (uint32_t b = 0; b < NB; ++b)
      if (h_over[b]) tot_rec += h_bounds[b + 1] - h_bounds[b];
    if (getenv("BG_DEBUG_SKM")) {
      uint64_t biggest = 0;
      for (uint32_t b = 0; b < NB; ++b)
        if (h_over[b]) biggest = std::max<uint64_t>(biggest, h_bounds[b + 1] - h_bounds[b]);
      fprintf(stderr, "skm count: %llu of %u buckets not counted in LDS, %llu records in them (biggest %llu)\n",
              (unsigned long long)n_over, NB, (unsigned long long)tot_rec, (unsigned long long)biggest);
    }
    const uint64_t tot = tot_rec * uint64_t(nmax);
    if (tot >= (1ull << 32)) {
      set_error("k-mer counting: %llu buckets of %u overflowed their tables (%llu super-k-mers); raise BG_KMER_EXTRA_BITS",
                (unsigned long long)n_over, NB, (unsigned long long)tot_rec);
      return BG_EINTERNAL;
    }
    DevBuf<uint64_t> ta, tb;
    BG_TRY(ta.alloc(ctx, tot));
    BG_TRY(tb.alloc(ctx, tot));
    BG_HIP(hipMemsetAsync(ta.p, 0xff, tot * 8, ctx->stream));
    if (!bits_out) {  // their presence bits wait for the finished hash set
      BG_TRY(ks->pend_rec.alloc(ctx, tot_rec * 2));
      BG_TRY(ks->pend_org.alloc(ctx, tot_rec));
      ks->pend_n = tot_rec;
    }
    std::vector<uint64_t> seg_start, seg_prefix;
    uint64_t pos = 0;
    for (uint32_t b = 0; b < NB; ++b) {
      if (!h_over[b]) continue;
      seg_start.push_back(h_bounds[b]);
      seg_prefix.push_back(pos);
      pos += h_bounds[b + 1] - h_bounds[b];
    }
    DevBuf<uint64_t> d_seg;
    BG_TRY(d_seg.alloc(ctx, seg_start.size() * 2));
    BG_HIP(hipMemcpyAsync(d_seg.p, seg_start.data(), seg_start.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(d_seg.p + seg_start.size(), seg_prefix.data(), seg_start.size() * 8, hipMemcpyHostToDevice,
                          ctx->stream));
    sk_expand_kernel<<<cdiv(tot_rec, 256), 256, 0, ctx->stream>>>(
        part, part_idx, oin, d_seg.p, d_seg.p + seg_start.size(), uint32_t(seg_start.size()), tot_rec, k, nmax, ta.p,
        bits_out ? nullptr : reinterpret_cast<SkRec*>(ks->pend_rec.p), bits_out ? nullptr : ks->pend_org.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope
    bg_kmerset tmp;
    tmp.ctx = ctx;
    tmp.k = k;
    BG_TRY(kmer_count_keys_sort_impl(ctx, ta.p, tb.p, tot, min_count, &tmp));
    if (nkept + tmp.size > out_cap) { set_error("k-mer output overflow (internal error)"); return BG_EINTERNAL; }
    if (tmp.size) {
      BG_HIP(hipMemcpyAsync(ok_a.p + nkept, tmp.kmers.p, tmp.size * 8, hipMemcpyDeviceToDevice, ctx->stream));
      BG_HIP(hipMemcpyAsync(oc_a.p + nkept, tmp.counts.p, tmp.size * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    BG_HIP(hipStreamSynchronize(ctx->stream));
    nkept += tmp.size;
    distinct += tmp.distinct;
  }
  if (nkept > out_cap) { set_error("k-mer output overflow (internal error)"); return BG_EINTERNAL; }
  // the result arrays are the heads of the buffers the count wrote to (room for every window / min_count): no copy
  ok_a.shrink(nkept);
  oc_a.shrink(nkept);
  ks->kmers = std::move(ok_a);
  ks->counts = std::move(oc_a);
  ks->size = nkept;
  ks->distinct = distinct;
  ks->sorted = false;
  return BG_OK;
}

// ---- sharded build: super-k-mers cross the GPUs instead of k-mer instances -------------------------------


__global__ __launch_bounds__(256) void sk_sum_windows_kernel(const SkRec* __restrict__ recs, uint64_t n,
                                                             unsigned long long* __restrict__ total) {
  unsigned long long v = 0;
  for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x)
    v += uint64_t(sk_windows(recs[i].b));
#pragma unroll
  for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, v);
}

// presence bits of one rank's own records, as returned by their owners (kSkUnknown: look the windows up)
__global__ void sk_present_apply_kernel(const SkRec* __restrict__ recs, const uint64_t* __restrict__ org,
                                        const uint32_t* __restrict__ bits_in, uint64_t n_rec, int k,
                                        const uint64_t* __restrict__ slots, int log2cap,
                                        unsigned long long* __restrict__ present, int pstride) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  uint32_t bits = bits_in[i];
  const SkRec r = recs[i];
  const int n = sk_windows(r.b);
  if (bits & kSkUnknown) {
    KmerTable tab{slots, log2cap, k};
    const uint64_t kmask = ~(~0ull << (2 * k));
    uint64_t fwd = r.a >> (64 - 2 * k);
    uint64_t rc = kmer_revcomp(fwd, k);
    bits = 0;
    for (int t = 0; t < n; ++t) {
      if (tab.find(rc < fwd ? rc : fwd) >= 0) bits |= 1u << t;
      const int idx = k + t;
      const uint64_t base = idx < 32 ? (r.a >> (62 - 2 * idx)) & 3 : (r.b >> (62 - 2 * (idx - 32))) & 3;
      fwd = ((fwd << 2) | base) & kmask;
      rc = (rc >> 2) | ((3 - base) << (2 * (k - 1)));
    }
  }
  // the words start as "a bit per window without an N" (sk_present_valid_kernel: the windows records are made of): the one
  // record in ten with a window that is not kept clears bits, the others touch nothing (see sk_spread_kernel; setting the
  // kept bits of every record instead was 6.1 ms of atomics for a chr20-sized rank)
  const uint32_t miss = ~bits & (n == 32 ? ~0u : (1u << n) - 1);
  if (!miss) return;
  const uint64_t o = org[i];
  const uint64_t read = o >> 8;
  const int s0 = int(o & 0xff), wi = s0 >> 6, sh = s0 & 63;
  atomicAnd(&present[read * pstride + wi], ~((unsigned long long)miss << sh));
  const unsigned long long over = sh ? (unsigned long long)miss >> (64 - sh) : 0ull;  // windows past the word
  if (over) atomicAnd(&present[read * pstride + wi + 1], ~over);
}

}  // namespace bg

using namespace bg;

extern "C" int bg_skm_supported(int k) { return k >= SK_M + 2 && k <= 31; }

extern "C" int bg_skm_capacity(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n_windows, uint64_t* cap_out) {
  if (!ctx || !reads || !cap_out) { set_error("bg_skm_capacity: null argument"); return BG_EINVAL; }
  SkGeometry g;
  *cap_out = sk_geometry(reads, k, n_windows, &g) ? g.cap : 0;
  return BG_OK;
}

extern "C" int bg_skm_extract_by_owner(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t n_windows, uint32_t nparts,
                                       uint64_t* d_recs, uint64_t* d_org, uint64_t cap, uint64_t* counts,
                                       uint64_t* n_out) {
  if (!ctx || !reads || !counts || !n_out || (cap && (!d_recs || !d_org))) { set_error("bg_skm_extract_by_owner: null argument"); return BG_EINVAL; }
  if (nparts < 1 || nparts > 256) { set_error("bg_skm_extract_by_owner: %u parts outside 1..256", nparts); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  for (uint32_t p = 0; p < nparts; ++p) counts[p] = 0;
  *n_out = 0;
  SkGeometry g;
  if (!sk_geometry(reads, k, n_windows, &g)) {
    if (!n_windows || int(reads->max_len) - k + 1 <= 0) return BG_OK;  // no windows at all
    set_error("bg_skm_extract_by_owner: k = %d is outside the super-k-mer path (see bg_skm_supported)", k);
    return BG_EINVAL;
  }
  // cap may be smaller than bg_skm_capacity()'s worst case (3x the expected density): the caller sizes its buffers for
  // what an ordinary input produces and comes back with the full capacity if this one produces more (BG_ENOMEM)
  if (!cap) { set_error("bg_skm_extract_by_owner: no room for records"); return BG_EINVAL; }
  g.cap = std::min<uint64_t>(cap, g.cap);
  DevBuf<SkRec> ra;
  DevBuf<uint64_t> oa;
  BG_TRY(ra.alloc(ctx, g.cap));
  BG_TRY(oa.alloc(ctx, g.cap));
  uint64_t R = 0;
  BG_TRY(sk_generate(ctx, reads, g, ra.p, oa.p, &R));
  if (R > g.cap) { set_error("bg_skm_extract_by_owner: %llu super-k-mers do not fit the %llu the caller made room for",
                             (unsigned long long)R, (unsigned long long)g.cap); return BG_ENOMEM; }
  *n_out = R;
  if (!R) return BG_OK;
  // whole groups of the top bucket byte go to their owner: owner = (byte * nparts) >> 8
  std::vector<uint64_t> h_starts(SK_BINS + 1);
  BG_TRY((sk_level1<uint64_t, false>(ctx, ra.p, reinterpret_cast<SkRec*>(d_recs), oa.p, d_org, R, h_starts.data())));
  for (int b = 0; b < SK_BINS; ++b) counts[(uint32_t(b) * nparts) >> 8] += h_starts[b + 1] - h_starts[b];
  return BG_OK;
}

extern "C" int bg_skm_count(bg_ctx* ctx, const uint64_t* d_recs, uint64_t n, int k, uint32_t min_count,
                            bg_kmerset** out, uint32_t* d_bits) {
  if (!ctx || !out || (n && !d_recs)) { set_error("bg_skm_count: null argument"); return BG_EINVAL; }
  if (!bg_skm_supported(k)) { set_error("bg_skm_count: k = %d is outside the super-k-mer path", k); return BG_EINVAL; }
  if (min_count < 1) min_count = 1;
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_kmerset> ks(new bg_kmerset);
  ks->ctx = ctx;
  ks->k = k;
  const SkRec* recs = reinterpret_cast<const SkRec*>(d_recs);
  DevBuf<unsigned long long> tot;
  DevBuf<uint32_t> own_bits;
  BG_TRY(tot.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(tot.p, 0, 8, ctx->stream));
  if (!d_bits) {  // the caller does not want them
    BG_TRY(own_bits.alloc(ctx, n));
    d_bits = own_bits.p;
  }
  unsigned long long N = 0;
  if (n) {
    sk_sum_windows_kernel<<<(cdiv(n, 256) < 2048u ? cdiv(n, 256) : 2048u), 256, 0, ctx->stream>>>(recs, n, tot.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipMemcpyAsync(&N, tot.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  ks->instances = N;
  BG_TRY(sk_count_records(ctx, recs, nullptr, n, N, k, min_count, ks.get(), d_bits));
  *out = ks.release();
  return BG_OK;
}

extern "C" int bg_kmerset_attach_presence(bg_ctx* ctx, bg_kmerset* ks, const bg_reads* reads, const uint64_t* d_recs,
                                          const uint64_t* d_org, const uint32_t* d_bits, uint64_t n) {
  if (!ctx || !ks || !reads || (n && (!d_recs || !d_org || !d_bits))) { set_error("bg_kmerset_attach_presence: null argument"); return BG_EINVAL; }
  if (!ks->slots.p) { set_error("bg_kmerset_attach_presence: the k-mer set has no hash table"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const int maxw = int(reads->max_len) - ks->k + 1;
  ks->present_stride = std::max(1, (maxw + 63) / 64);
  BG_TRY(ks->present.alloc(ctx, reads->n * ks->present_stride));
  if (reads->W > 8) { set_error("bg_kmerset_attach_presence: reads of more than 256 bases"); return BG_EINVAL; }
  if (reads->n) {
    sk_present_valid_kernel<<<cdiv(reads->n, 256), 256, 0, ctx->stream>>>(
        reads->any_n ? reads->nmask.p : nullptr, reads->len.p, reads->n, reads->W, ks->k,
        reinterpret_cast<unsigned long long*>(ks->present.p), ks->present_stride);
    BG_LAUNCH_CHECK();
  }
  if (n) {
    sk_present_apply_kernel<<<cdiv(n, 256), 256, 0, ctx->stream>>>(reinterpret_cast<const SkRec*>(d_recs), d_org, d_bits, n,
                                                                 ks->k, ks->slots.p, ks->log2cap,
                                                                 reinterpret_cast<unsigned long long*>(ks->present.p),
                                                                 ks->present_stride);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  ks->present_of = reads;
  ks->present_n = reads->n;
  return BG_OK;
}

// ---- sharded build, routed look-ups: the kept k-mers stay with the rank that counted them ---------------

namespace bg {

// bits_out[i] bit t = window t of record i is in the set (the owner's answer to a probe record; with
// only_unknown, just the records its counting table gave up on — bits_out[i] & kSkUnknown — are looked up).
// stop_at_absent: windows past the first absent one are left 0.  Only for callers that consume a record's windows
// strictly in order and never ask past an absent one — NOT the routed correction: its probes answer by k-mer, and a
// search reaches a probe's later windows by other routes too.
// Four neighbouring lanes share a record's windows (four look-ups of a record in flight instead of one); with
// stop_at_absent a record keeps one lane, whose windows depend on each other.
constexpr int SKL_SPLIT = 4;
__global__ void sk_lookup_bits_kernel(const SkRec* __restrict__ recs, uint64_t n_rec, int k, const uint64_t* __restrict__ slots,
                                      int log2cap, uint32_t* __restrict__ bits_out, int only_unknown, int stop_at_absent) {
  const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t i = g / SKL_SPLIT;
  const int part = int(g % SKL_SPLIT);
  bool live = i < n_rec && !(only_unknown && !(bits_out[i] & kSkUnknown));
  SkRec r;
  r.a = 0;
  r.b = 0;
  if (live) r = recs[i];
  const int n = sk_windows(r.b);
  const int per = stop_at_absent ? n : (n + SKL_SPLIT - 1) / SKL_SPLIT;
  const int t0 = part * per, t1 = t0 + per < n ? t0 + per : n;
  uint32_t bits = 0;
  if (live && t0 < t1) {
    KmerTable tab{slots, log2cap, k};
    const uint64_t kmask = ~(~0ull << (2 * k));
    const uint64_t x = t0 ? ((r.a << (2 * t0)) | (r.b >> (64 - 2 * t0))) : r.a;  // t0 <= 31: only base bits of b
    uint64_t fwd = x >> (64 - 2 * k);
    uint64_t rc = kmer_revcomp(fwd, k);
    for (int t = t0; t < t1; ++t) {
      if (tab.find(rc < fwd ? rc : fwd) >= 0) bits |= 1u << t;
      else if (stop_at_absent) break;
      const int idx = k + t;
      const uint64_t base = idx < 32 ? (r.a >> (62 - 2 * idx)) & 3 : (r.b >> (62 - 2 * (idx - 32))) & 3;
      fwd = ((fwd << 2) | base) & kmask;
      rc = (rc >> 2) | ((3 - base) << (2 * (k - 1)));
    }
  }
#pragma unroll
  for (int d = 1; d < SKL_SPLIT; d <<= 1) bits |= uint32_t(__shfl_xor(int(bits), d, 64));
  if (live && part == 0) bits_out[i] = bits;
}

// presence bits of a rank's own records as their owners returned them (every record known): the words start as "a
// bit per window without an N" (sk_present_valid_kernel) and the one record in ten with a window that is not kept
// clears bits — see sk_spread_kernel
__global__ void sk_present_from_bits_kernel(const SkRec* __restrict__ recs, const uint64_t* __restrict__ org,
                                            const uint32_t* __restrict__ bits_in, uint64_t n_rec,
                                            unsigned long long* __restrict__ present, int pstride, uint32_t* __restrict__ err) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  // every record is resolved by its owner (bg_skm_lookup); bit 31 is window 31 of a 32-window record here
  const uint32_t bits = bits_in[i];
  const int n = sk_windows(recs[i].b);
  if (n < 32 && (bits >> n)) { atomicOr(err, 1u); return; }
  const uint32_t miss = ~bits & (n == 32 ? ~0u : (1u << n) - 1);
  if (!miss) return;
  const uint64_t o = org[i];
  const uint64_t read = o >> 8;
  const int s0 = int(o & 0xff), wi = s0 >> 6, sh = s0 & 63;
  atomicAnd(&present[read * pstride + wi], ~((unsigned long long)miss << sh));
  const unsigned long long over = sh ? (unsigned long long)miss >> (64 - sh) : 0ull;  // windows past the word
  if (over) atomicAnd(&present[read * pstride + wi + 1], ~over);
}

// answers of a probe exchange: record i covers windows s0.. of probe string p (origin = p << 8 | s0); string
// 4 d + b belongs to alternative b of probe base + d
__global__ void sk_probe_answers_kernel(const uint64_t* __restrict__ org, const uint32_t* __restrict__ bits_in, uint64_t n_rec,
                                        ProbeDesc* __restrict__ pool, uint64_t base, uint64_t n_probes, uint32_t* __restrict__ err) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  const uint32_t bits = bits_in[i];
  const uint64_t o = org[i];
  const uint64_t p = o >> 8;
  const int s0 = int(o & 0xff);
  if ((p >> 2) >= n_probes || s0 > 31 || (s0 && (bits >> (32 - s0)))) { atomicOr(err, 1u); return; }
  if (bits) atomicOr(&pool[base + (p >> 2)].mask[p & 3], bits << s0);
}

int skm_presence_from_bits(bg_ctx* ctx, const bg_reads* reads, int k, const uint64_t* d_recs, const uint64_t* d_org,
                           const uint32_t* d_bits, uint64_t n, DevBuf<uint64_t>* present, int* stride) {
  const int maxw = int(reads->max_len) - k + 1;
  *stride = std::max(1, (maxw + 63) / 64);
  BG_TRY(present->alloc(ctx, reads->n * uint64_t(*stride)));
  if (reads->W > 8) { set_error("presence bits: reads of more than 256 bases"); return BG_EINVAL; }
  if (reads->n) {
    sk_present_valid_kernel<<<cdiv(reads->n, 256), 256, 0, ctx->stream>>>(
        reads->any_n ? reads->nmask.p : nullptr, reads->len.p, reads->n, reads->W, k,
        reinterpret_cast<unsigned long long*>(present->p), *stride);
    BG_LAUNCH_CHECK();
  }
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  if (n) {
    sk_present_from_bits_kernel<<<cdiv(n, 256), 256, 0, ctx->stream>>>(reinterpret_cast<const SkRec*>(d_recs), d_org, d_bits, n,
                                                                     reinterpret_cast<unsigned long long*>(present->p), *stride, err.p);
    BG_LAUNCH_CHECK();
  }
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("presence bits: bits beyond a record's windows (unresolved by the owner? see bg_skm_lookup)"); return BG_EINVAL; }
  return BG_OK;
}

int skm_probe_answers(bg_ctx* ctx, const uint64_t* d_org, const uint32_t* d_bits, uint64_t n, ProbeDesc* pool, uint64_t base,
                      uint64_t n_probes) {
  if (!n) return BG_OK;
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  sk_probe_answers_kernel<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_org, d_bits, n, pool, base, n_probes, err.p);
  BG_LAUNCH_CHECK();
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("probe answers do not match the probes in flight"); return BG_EINVAL; }
  return BG_OK;
}

int kmer_hash_build_impl(bg_ctx* ctx, bg_kmerset* ks, const bg_reads* reads);

}  // namespace bg

extern "C" int bg_kmerset_build_hash(bg_ctx* ctx, bg_kmerset* ks) {
  if (!ctx || !ks) { set_error("bg_kmerset_build_hash: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (ks->slots.p) return BG_OK;
  return kmer_hash_build_impl(ctx, ks, nullptr);
}

extern "C" int bg_skm_lookup(bg_ctx* ctx, const bg_kmerset* ks, const uint64_t* d_recs, uint64_t n, uint32_t* d_bits,
                             int only_unknown) {
  if (!ctx || !ks || (n && (!d_recs || !d_bits))) { set_error("bg_skm_lookup: null argument"); return BG_EINVAL; }
  if (!ks->slots.p) { set_error("bg_skm_lookup: the k-mer set has no hash table (bg_kmerset_build_hash)"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (n) {
    ctx->tic("kmer_probe_lookup");
    sk_lookup_bits_kernel<<<cdiv(n * SKL_SPLIT, 256), 256, 0, ctx->stream>>>(reinterpret_cast<const SkRec*>(d_recs), n, ks->k, ks->slots.p,
                                                               ks->log2cap, d_bits, only_unknown & 1, (only_unknown >> 1) & 1);
    BG_LAUNCH_CHECK();
    ctx->toc(n * 20);
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}
