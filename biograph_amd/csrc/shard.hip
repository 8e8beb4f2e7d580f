// shard.hip — exchange points of the sharded (multi-GPU) seqset build.
//
// One process per GPU; the host side (biograph_amd/sharded.py) moves the buffers these entry points
// fill with torch.distributed (RCCL all_to_all / all_gather over xGMI).  The library itself never
// communicates: every function here works on one rank's device memory.
//
//   k-mer instances --bg_partition_rows(KMER_HASH)--> all_to_all --> bg_kmer_count_keys --> all_gather kept
//   corrected reads --bg_reads_export--> all_gather --> bg_reads_import (cover index needs all reads)
//   suffix records  --bg_partition_rows(PREFIX)--> all_to_all --> bg_entries_from_records
//   boundary rows   --all_gather--> bg_seqset_fix_boundary      (part_repo.cpp:294-303, expand.cpp:318-327)
//   tails s[1:]     --bg_partition_rows(KEY_BOUNDS)--> all_to_all --> bg_seqset_apply_tails (builder.cpp:85-108)
//   slices          --gather--> bg_seqset_assemble --> seqset::finalize (seqset.cpp:113-129)
#include "bg_device_funcs.h"
#include "bg_internal.h"
#include "compact.h"

namespace bg {

constexpr uint64_t kDropDest = 255;

// ---- destination kernels: dest[i] (as a u64 radix key) and idx[i] = i

__global__ void k_dest_kmer_hash(const uint64_t* __restrict__ rows, uint64_t n, uint32_t nparts,
                                 uint64_t* __restrict__ dest, uint32_t* __restrict__ idx) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t k = rows[i];
  // owner = floor(hash * nparts / 2^32) of the top 32 hash bits: uniform, independent of the table hash
  uint64_t h = (k * 0xD6E8FEB86659FD93ull) >> 32;
  dest[i] = (k == ~0ull) ? kDropDest : ((h * nparts) >> 32);
  idx[i] = uint32_t(i);
}

// rows = W+1 words; bin = leading `hb` bases of word 0; dest = number of splitters <= bin
__global__ void k_dest_prefix(const uint64_t* __restrict__ rows, uint64_t n, int row_words, int hb,
                              const uint32_t* __restrict__ splitters, uint32_t nparts, uint64_t* __restrict__ dest,
                              uint32_t* __restrict__ idx) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t bin = uint32_t(rows[i * row_words] >> (64 - 2 * hb));
  uint32_t d = 0;
  for (uint32_t s = 0; s + 1 < nparts; ++s) d += (splitters[s] <= bin) ? 1u : 0u;
  // a row of length 0 is nothing (a dropped read, an unused candidate slot): it goes nowhere
  dest[i] = (rows[i * row_words + row_words - 1] & 0xff) ? uint64_t(d) : kDropDest;
  idx[i] = uint32_t(i);
}

// rows = W+1 words (last word: low 8 bits = length); bounds = nparts rows (last word ~0 = rank holds no
// entries).  dest = first rank whose last entry is >= the row (global lower_bound lands there).
__global__ void k_dest_key_bounds(const uint64_t* __restrict__ rows, uint64_t n, int W,
                                  const uint64_t* __restrict__ bounds, uint32_t nparts, uint64_t* __restrict__ dest,
                                  uint32_t* __restrict__ idx, uint32_t* __restrict__ err) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* r = rows + i * (W + 1);
  int rl = int(r[W] & 0xff);
  uint64_t d = kDropDest;
  for (uint32_t p = 0; p < nparts; ++p) {
    const uint64_t* b = bounds + uint64_t(p) * (W + 1);
    if (b[W] == ~0ull) continue;
    if (key_compare(r, rl, b, int(b[W] & 0xff), W) <= 0) { d = p; break; }
  }
  if (d == kDropDest) atomicOr(err, 2u);  // no entry extends this tail anywhere: closure violated
  dest[i] = d;
  idx[i] = uint32_t(i);
}

// rows whose last word names their destination (bits 10..16; 127 = nowhere): tails routed by bg_seqset_tail_rows2
__global__ void k_dest_from_row(const uint64_t* __restrict__ rows, uint64_t n, int row_words, uint32_t nparts,
                                uint64_t* __restrict__ dest, uint32_t* __restrict__ idx) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t d = (rows[i * row_words + row_words - 1] >> 10) & 0x7f;
  dest[i] = d < nparts ? d : kDropDest;
  idx[i] = uint32_t(i);
}

// destinations take few distinct values: lanes with equal dest are grouped with ballots and one lane
// adds the group's size (no serialised same-address LDS atomics)
__global__ __launch_bounds__(256) void k_dest_counts(const uint64_t* __restrict__ dest, uint64_t n,
                                                     unsigned long long* __restrict__ counts) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t lane_lt = (1ull << (threadIdx.x & 63)) - 1;
  const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
  const uint64_t rounds = (n + stride - 1) / stride;
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  for (uint64_t t = 0; t < rounds; ++t, i += stride) {  // every lane runs every round (ballots below)
    bool valid = i < n;
    uint32_t d = valid ? uint32_t(dest[i] & 0xff) : 0u;
    uint64_t m = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      bool bit = (d >> b) & 1;
      uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    uint64_t bv = __ballot(valid);
    m &= valid ? bv : ~bv;
    if (valid && (m & lane_lt) == 0) atomicAdd(&h[d], uint32_t(__popcll(m)));
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// ---- stable multi-way split of rows by destination (<= 256 destinations): per-tile counts -> device scan
// of the [dest][tile] table -> direct row writes.  Replaces sort-by-destination + gather.
constexpr int MS_ITEMS = 8;
constexpr int MS_TILE = 256 * MS_ITEMS;

__device__ __forceinline__ uint64_t ms_match(uint32_t d, bool valid) {  // lanes with the same destination
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

__global__ __launch_bounds__(256) void k_msplit_count(const uint64_t* __restrict__ dest, uint64_t n, uint32_t nparts,
                                                      uint32_t* __restrict__ counts, uint32_t ntiles) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t base = uint64_t(blockIdx.x) * MS_TILE + uint64_t(w) * (MS_ITEMS * 64);
  for (int it = 0; it < MS_ITEMS; ++it) {
    uint64_t i = base + uint64_t(it) * 64 + lane;
    uint64_t dv = i < n ? dest[i] : uint64_t(kDropDest);
    bool valid = i < n && dv < nparts;
    uint32_t d = valid ? uint32_t(dv) : 0u;
    uint64_t m = ms_match(d, valid);
    if (valid && (m & lane_lt) == 0) atomicAdd(&h[d], uint32_t(__popcll(m)));
  }
  __syncthreads();
  if (threadIdx.x < nparts) counts[uint64_t(threadIdx.x) * ntiles + blockIdx.x] = h[threadIdx.x];
}

// where the rows of a split come from: an array, or the candidate suffixes of a cover round generated on the fly
struct RowsSource {
  const uint64_t* rows;
  int row_words;
  __device__ __forceinline__ uint64_t word(uint64_t i, int w) const { return rows[i * uint64_t(row_words) + w]; }
};
struct CandidateSource {  // row t = suffix j_lo + t % span of pending read t / span (k_query_candidates of shard_cover.hip)
  const uint64_t* owords;
  const uint8_t* olen;
  const uint32_t* pending;
  int W, K, j_lo, span;
  __device__ __forceinline__ uint64_t word(uint64_t t, int w) const {
    const uint64_t slot = t / uint64_t(span);
    const int j = j_lo + int(t - slot * uint64_t(span));
    const uint32_t o = pending[slot];
    if (w == W) return uint64_t(int(olen[o]) - j) | (t << 24);
    return window32(owords + uint64_t(o) * W, W, j + 32 * w);
  }
};

template <class Source>
__global__ __launch_bounds__(256) void k_msplit_scatter(Source src, const uint64_t* __restrict__ dest, uint64_t n, int row_words,
                                                        uint32_t nparts, const uint64_t* __restrict__ offs,
                                                        uint32_t ntiles, uint64_t* __restrict__ out) {
  __shared__ uint32_t whist[4][256];
  __shared__ uint64_t goff[256];
  for (int i = threadIdx.x; i < 4 * 256; i += 256) (&whist[0][0])[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t base = uint64_t(blockIdx.x) * MS_TILE + uint64_t(w) * (MS_ITEMS * 64);
  volatile uint32_t* wh = &whist[w][0];
  uint16_t rnk[MS_ITEMS];
  uint8_t dd[MS_ITEMS];
#pragma unroll
  for (int it = 0; it < MS_ITEMS; ++it) {
    uint64_t i = base + uint64_t(it) * 64 + lane;
    uint64_t dv = i < n ? dest[i] : uint64_t(kDropDest);
    bool valid = i < n && dv < nparts;
    uint32_t d = valid ? uint32_t(dv) : 0u;
    uint64_t m = ms_match(d, valid);
    uint32_t before = wh[d];
    uint32_t lt = __popcll(m & lane_lt);
    rnk[it] = uint16_t(before + lt);
    dd[it] = valid ? uint8_t(d) : uint8_t(255);
    if (valid && lt == 0) wh[d] = before + uint32_t(__popcll(m));
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {  // thread = destination: wave bases inside the tile, global offset of the tile's run
    int t = threadIdx.x;
    uint32_t c0 = whist[0][t], c1 = whist[1][t], c2 = whist[2][t];
    whist[0][t] = 0;
    whist[1][t] = c0;
    whist[2][t] = c0 + c1;
    whist[3][t] = c0 + c1 + c2;
    goff[t] = uint32_t(t) < nparts ? offs[uint64_t(t) * ntiles + blockIdx.x] : 0;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < MS_ITEMS; ++it) {
    // the wave's 64 rows of this round are contiguous in the source: copy them word-interleaved (lane =
    // word index), fetching each row's destination from the lane that ranked it
    const uint64_t i0 = base + uint64_t(it) * 64;
    if (i0 >= n) break;
    bool keep = dd[it] != 255;
    uint32_t d = keep ? dd[it] : 0u;
    uint64_t pos = keep ? goff[d] + whist[w][d] + rnk[it] : 0;
    const uint64_t nwords = (n - i0 < 64 ? n - i0 : 64) * uint64_t(row_words);
    for (uint64_t j = lane; j < uint64_t(64) * row_words; j += 64) {  // uniform trip count (shuffles inside)
      int row = int(j / uint64_t(row_words));
      int wd = int(j - uint64_t(row) * row_words);
      uint64_t p = __shfl(pos, row, 64);
      bool k = __shfl(int(keep), row, 64) != 0;
      if (j < nwords && k) out[p * uint64_t(row_words) + wd] = src.word(i0 + uint64_t(row), wd);
    }
  }
}


// ---- rows <-> records / tails

__global__ void k_records_to_rows(const uint64_t* __restrict__ rec, const uint8_t* __restrict__ rlen, uint64_t n, int W,
                                  uint64_t* __restrict__ rows) {
  uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n * uint64_t(W + 1)) return;
  uint64_t i = g / uint64_t(W + 1);
  int w = int(g - i * uint64_t(W + 1));
  rows[g] = w < W ? rec[i * W + w] : uint64_t(rlen[i]);
}
__global__ void k_rows_to_records(const uint64_t* __restrict__ rows, uint64_t n, int W, uint64_t* __restrict__ rec,
                                  uint8_t* __restrict__ rlen) {
  uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n * uint64_t(W + 1)) return;
  uint64_t i = g / uint64_t(W + 1);
  int w = int(g - i * uint64_t(W + 1));
  uint64_t v = rows[g];
  if (w < W) rec[i * W + w] = v; else rlen[i] = uint8_t(v);
}

// A merge input's entry (a received row; last word = length | input << 8) marks the first entry of this shard that
// has it as a prefix: the image of make_mergemap (bio_base/make_mergemap.cpp:150-220) as one byte per (input, entry).
// An entry whose extension is not in this shard (it was dropped at the boundary as a prefix of the next shard's first
// entry) raises the input's carry flag instead.
__global__ __launch_bounds__(128) void k_mark_inputs(const uint64_t* __restrict__ rows, uint64_t n, int W,
                                                     const uint64_t* __restrict__ ekeys,
                                                     const uint8_t* __restrict__ esizes, uint64_t n_ent,
                                                     const uint32_t* __restrict__ dir, int bases, uint32_t n_inputs,
                                                     uint8_t* __restrict__ marks, uint32_t* __restrict__ carry,
                                                     uint32_t* __restrict__ err) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* key = rows + i * (W + 1);
  const int L = int(key[W] & 0xff);
  const uint32_t input = uint32_t((key[W] >> 8) & 0xff);
  if (!L) return;
  if (input >= n_inputs) { atomicOr(err, 1u); return; }
  uint64_t lo = 0, hi = n_ent;
  if (n_ent) {
    uint32_t p = uint32_t(key[0] >> (64 - 2 * bases));
    lo = dir[p];
    hi = dir[p + 1];
  }
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (key_compare(ekeys + mid * W, esizes[mid], key, L, W) < 0) lo = mid + 1; else hi = mid;
  }
  if (lo >= n_ent) { atomicOr(carry + input, 1u); return; }
  if (key_lcp(ekeys + lo * W, esizes[lo], key, L, W) < L) { atomicOr(err, 2u); return; }
  marks[uint64_t(input) * n_ent + lo] = 1;
}

// tail row of entry i: s[1:] zero padded, last word = (len-1) | first base << 8
__global__ void k_tail_rows(const uint64_t* __restrict__ ekeys, const uint8_t* __restrict__ sizes, uint64_t n, int W,
                            uint64_t* __restrict__ rows) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* s = ekeys + i * W;
  uint64_t* r = rows + i * (W + 1);
  for (int w = 0; w < W; ++w) r[w] = window32(s, W, 1 + 32 * w);
  r[W] = uint64_t(sizes[i] - 1) | ((s[0] >> 62) << 8);
}

// Short tail row of entry i: the tail's two leading words and meta = (len-1) | first base << 8 | destination
// shard << 10 | i << 17.  The destination (first shard whose last entry is >= the tail) is settled here, where
// the whole key is at hand: the two leading words decide it unless they tie with a shard's last entry.
__global__ void k_tail_rows2(const uint64_t* __restrict__ ekeys, const uint8_t* __restrict__ sizes, uint64_t n, int W,
                             const uint64_t* __restrict__ bounds, uint32_t nparts, uint64_t* __restrict__ rows,
                             uint32_t* __restrict__ err) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* s = ekeys + i * W;
  const uint64_t t0 = window32(s, W, 1), t1 = W > 1 ? window32(s, W, 33) : 0;
  const int tl = int(sizes[i]) - 1;
  uint64_t d = 127;
  for (uint32_t p = 0; p < nparts; ++p) {
    const uint64_t* b = bounds + uint64_t(p) * (W + 1);
    if (b[W] == ~0ull) continue;
    const uint64_t b0 = b[0], b1 = W > 1 ? b[1] : 0;
    if (t0 < b0 || (t0 == b0 && t1 < b1)) { d = p; break; }
    if (t0 == b0 && t1 == b1) {
      uint64_t tail[kMaxWords];
      for (int w = 0; w < W; ++w) tail[w] = window32(s, W, 1 + 32 * w);
      if (key_compare(tail, tl, b, int(b[W] & 0xff), W) <= 0) { d = p; break; }
    }
  }
  if (d == 127) atomicOr(err, 2u);  // no entry extends this tail anywhere: closure violated
  uint64_t* r = rows + i * 3;
  r[0] = t0;
  r[1] = t1;
  r[2] = uint64_t(tl) | ((s[0] >> 62) << 8) | (d << 10) | (i << 17);
}

// k_tail_rows2 for the entries pick[0 .. m) only (the survivors whose tail still has to travel)
__global__ void k_tail_rows2_picked(const uint64_t* __restrict__ ekeys, const uint8_t* __restrict__ sizes, const uint32_t* __restrict__ pick,
                                    uint64_t m, int W, const uint64_t* __restrict__ bounds, uint32_t nparts, uint64_t* __restrict__ rows,
                                    uint32_t* __restrict__ err) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const uint64_t i = pick[t];
  const uint64_t* s = ekeys + i * W;
  const uint64_t t0 = window32(s, W, 1), t1 = W > 1 ? window32(s, W, 33) : 0;
  const int tl = int(sizes[i]) - 1;
  uint64_t d = 127;
  for (uint32_t p = 0; p < nparts; ++p) {
    const uint64_t* b = bounds + uint64_t(p) * (W + 1);
    if (b[W] == ~0ull) continue;
    const uint64_t b0 = b[0], b1 = W > 1 ? b[1] : 0;
    if (t0 < b0 || (t0 == b0 && t1 < b1)) { d = p; break; }
    if (t0 == b0 && t1 == b1) {
      uint64_t tail[kMaxWords];
      for (int w = 0; w < W; ++w) tail[w] = window32(s, W, 1 + 32 * w);
      if (key_compare(tail, tl, b, int(b[W] & 0xff), W) <= 0) { d = p; break; }
    }
  }
  if (d == 127) atomicOr(err, 2u);
  uint64_t* r = rows + t * 3;
  r[0] = t0;
  r[1] = t1;
  r[2] = uint64_t(tl) | ((s[0] >> 62) << 8) | (d << 10) | (i << 17);
}

// rows (W + 1 words: a key and its length) looked up among the shard's entries: an entry that IS the row must send its tail
__global__ __launch_bounds__(128) void k_mark_need_exact(const uint64_t* __restrict__ rows, uint64_t n_rows, int W,
                                                         const uint64_t* __restrict__ ekeys, const uint8_t* __restrict__ sizes, uint64_t n,
                                                         const uint32_t* __restrict__ dir, int bases, uint8_t* __restrict__ need) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_rows) return;
  const uint64_t* key = rows + t * uint64_t(W + 1);
  const int L = int(key[W] & 0xff);
  const uint32_t p = uint32_t(key[0] >> (64 - 2 * bases));
  uint64_t lo = dir[p], hi = dir[p + 1];
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (key_compare(ekeys + mid * W, sizes[mid], key, L, W) < 0) lo = mid + 1; else hi = mid;
  }
  if (lo < n && int(sizes[lo]) == L && key_lcp(ekeys + lo * W, L, key, L, W) >= L) need[lo] = 1;
}

// full tail rows (W+1 words, meta as in the short row) of the entries named by the short rows flagged in `need`
__global__ void k_tail_rows_full(const uint64_t* __restrict__ ekeys, uint64_t n, int W, const uint64_t* __restrict__ short_rows,
                                 const uint32_t* __restrict__ pick, uint64_t m, uint64_t* __restrict__ rows) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const uint64_t meta = short_rows[uint64_t(pick[t]) * 3 + 2];
  const uint64_t i = meta >> 17;
  const uint64_t* s = ekeys + i * W;
  uint64_t* r = rows + t * uint64_t(W + 1);
  for (int w = 0; w < W; ++w) r[w] = window32(s, W, 1 + 32 * w);
  r[W] = meta;
  (void)n;
}

// local bit words of one slice OR-ed into the global bit vector at bit offset `off`
__global__ void k_place_bits(const uint64_t* __restrict__ local, uint64_t n_local_bits, uint64_t off,
                             unsigned long long* __restrict__ global) {
  uint64_t w = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  uint64_t nw = (n_local_bits + 63) / 64;
  if (w >= nw) return;
  uint64_t v = local[w];
  uint64_t valid = n_local_bits - w * 64;
  if (valid < 64) v &= (1ull << valid) - 1;
  if (!v) return;
  uint64_t bit = off + w * 64;
  uint64_t gw = bit >> 6;
  int sh = int(bit & 63);
  atomicOr(&global[gw], (unsigned long long)(v << sh));
  if (sh && (v >> (64 - sh))) atomicOr(&global[gw + 1], (unsigned long long)(v >> (64 - sh)));
}

__global__ void k_restride_words(const uint64_t* __restrict__ in, uint64_t n, int W_in, int W_out,
                                 uint64_t* __restrict__ out) {
  uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n * uint64_t(W_out)) return;
  uint64_t i = g / uint64_t(W_out);
  int w = int(g - i * uint64_t(W_out));
  out[g] = w < W_in ? in[i * W_in + w] : 0;
}

}  // namespace bg

using namespace bg;

// ---- generic row partition ----------------------------------------------------------------------

extern "C" int bg_partition_rows(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int row_words, int mode,
                                 uint32_t nparts, const uint64_t* h_params, uint64_t* d_out_rows,
                                 uint64_t* h_counts) {
  if (!ctx || !h_counts || (n && (!d_rows || !d_out_rows))) { set_error("bg_partition_rows: null argument"); return BG_EINVAL; }
  if (nparts < 1 || nparts > 128 || row_words < 1 || row_words > kMaxWords + 1) { set_error("bg_partition_rows: bad geometry"); return BG_EINVAL; }
  if (n >= (1ull << 32)) { set_error("bg_partition_rows: %llu rows exceed one call", (unsigned long long)n); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  for (uint32_t p = 0; p < nparts; ++p) h_counts[p] = 0;
  if (!n) return BG_OK;
  DevBuf<uint64_t> da;
  DevBuf<uint32_t> ia, err;
  DevBuf<unsigned long long> counts;
  BG_TRY(da.alloc(ctx, n));
  BG_TRY(ia.alloc(ctx, n));
  BG_TRY(err.alloc(ctx, 1));
  BG_TRY(counts.alloc(ctx, 256));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  BG_HIP(hipMemsetAsync(counts.p, 0, 256 * 8, ctx->stream));
  ctx->tic("partition_dest");
  if (mode == BG_PART_KMER_HASH) {
    if (row_words != 1) { set_error("bg_partition_rows: KMER_HASH takes 1-word rows"); return BG_EINVAL; }
    k_dest_kmer_hash<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows, n, nparts, da.p, ia.p);
  } else if (mode == BG_PART_PREFIX) {
    if (!h_params) { set_error("bg_partition_rows: PREFIX needs params"); return BG_EINVAL; }
    int hb = int(h_params[0]);
    if (hb < 1 || hb > 16) { set_error("bg_partition_rows: prefix bases %d outside 1..16", hb); return BG_EINVAL; }
    std::vector<uint32_t> sp(nparts ? nparts : 1, 0);
    for (uint32_t s = 0; s + 1 < nparts; ++s) sp[s] = uint32_t(h_params[1 + s]);
    DevBuf<uint32_t> dsp;
    BG_TRY(dsp.alloc(ctx, nparts));
    BG_HIP(hipMemcpyAsync(dsp.p, sp.data(), nparts * 4, hipMemcpyHostToDevice, ctx->stream));
    k_dest_prefix<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows, n, row_words, hb, dsp.p, nparts, da.p, ia.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipStreamSynchronize(ctx->stream));
  } else if (mode == BG_PART_KEY_BOUNDS) {
    if (!h_params) { set_error("bg_partition_rows: KEY_BOUNDS needs params"); return BG_EINVAL; }
    DevBuf<uint64_t> dbounds;
    BG_TRY(dbounds.alloc(ctx, uint64_t(nparts) * row_words));
    BG_HIP(hipMemcpyAsync(dbounds.p, h_params, uint64_t(nparts) * row_words * 8, hipMemcpyHostToDevice, ctx->stream));
    k_dest_key_bounds<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows, n, row_words - 1, dbounds.p, nparts, da.p, ia.p,
                                                             err.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipStreamSynchronize(ctx->stream));
  } else if (mode == BG_PART_ROW_DEST) {
    k_dest_from_row<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows, n, row_words, nparts, da.p, ia.p);
  } else {
    set_error("bg_partition_rows: unknown mode %d", mode);
    return BG_EINVAL;
  }
  BG_LAUNCH_CHECK();
  unsigned grid = cdiv(n, 256) < 1024u ? cdiv(n, 256) : 1024u;
  k_dest_counts<<<grid, 256, 0, ctx->stream>>>(da.p, n, counts.p);
  BG_LAUNCH_CHECK();
  ctx->toc(n * (uint64_t(row_words) * 8 + 12));
  const uint32_t ntiles = uint32_t((n + MS_TILE - 1) / MS_TILE);
  DevBuf<uint32_t> tcounts;
  DevBuf<uint64_t> toffs;
  BG_TRY(tcounts.alloc(ctx, uint64_t(nparts) * ntiles));
  BG_TRY(toffs.alloc(ctx, uint64_t(nparts) * ntiles));
  ctx->tic("partition_split");
  k_msplit_count<<<ntiles, 256, 0, ctx->stream>>>(da.p, n, nparts, tcounts.p, ntiles);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32_to_u64(ctx, tcounts.p, toffs.p, uint64_t(nparts) * ntiles, nullptr));
  unsigned long long hc[256];
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(hc, counts.p, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_partition_rows: a tail is not extended by any entry (missing expansion)"); return BG_EINTERNAL; }
  uint64_t n_out = 0;
  for (uint32_t p = 0; p < nparts; ++p) { h_counts[p] = hc[p]; n_out += hc[p]; }
  if (n_out) {
    k_msplit_scatter<<<ntiles, 256, 0, ctx->stream>>>(RowsSource{d_rows, row_words}, da.p, n, row_words, nparts, toffs.p, ntiles,
                                                      d_out_rows);
    BG_LAUNCH_CHECK();
  }
  ctx->toc(n * (uint64_t(row_words) * 16 + 16));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// rows[i] for the ascending positions i in pick: lane = word, so reads and writes are contiguous runs
__global__ void k_select_rows(const uint64_t* __restrict__ rows, const uint32_t* __restrict__ pick, uint64_t m, int row_words,
                              uint64_t* __restrict__ out) {
  const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= m * uint64_t(row_words)) return;
  const uint64_t i = g / uint64_t(row_words);
  const int w = int(g - i * uint64_t(row_words));
  out[g] = rows[uint64_t(pick[i]) * row_words + w];
}

// The rows whose flag byte is non-zero, in order, into d_out (room for cap rows): stream compaction of rows.
extern "C" int bg_rows_select(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int row_words, const uint8_t* d_flags,
                              uint64_t* d_out, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !n_out || (n && (!d_rows || !d_flags))) { set_error("bg_rows_select: null argument"); return BG_EINVAL; }
  if (n >= (1ull << 32) || row_words < 1) { set_error("bg_rows_select: bad geometry"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  *n_out = 0;
  if (!n) return BG_OK;
  DevBuf<uint32_t> pick;
  uint64_t m = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, d_flags, n, &pick, &m));
  if (m > cap || (m && !d_out)) { set_error("bg_rows_select: room for %llu rows, %llu selected", (unsigned long long)cap, (unsigned long long)m); return BG_EINVAL; }
  if (m) {
    ctx->tic("rows_select");
    k_select_rows<<<cdiv(m * uint64_t(row_words), 256), 256, 0, ctx->stream>>>(d_rows, pick.p, m, row_words, d_out);
    BG_LAUNCH_CHECK();
    ctx->toc(m * uint64_t(row_words) * 16 + n);
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = m;
  return BG_OK;
}

// destination of candidate t of a cover round (see CandidateSource): owner of its leading bases, or nowhere when
// the read has no suffix j with at least K bases left
__global__ void k_dest_candidates(CandidateSource cs, uint64_t total, int hb, const uint32_t* __restrict__ splitters,
                                  uint32_t nparts, uint64_t* __restrict__ dest) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint64_t slot = t / uint64_t(cs.span);
  const int j = cs.j_lo + int(t - slot * uint64_t(cs.span));
  const uint32_t o = cs.pending[slot];
  const int rem = int(cs.olen[o]) - j;
  if (rem < cs.K || rem <= 0) { dest[t] = kDropDest; return; }
  const uint32_t bin = uint32_t(window32(cs.owords + uint64_t(o) * cs.W, cs.W, j) >> (64 - 2 * hb));
  uint32_t d = 0;
  for (uint32_t q = 0; q + 1 < nparts; ++q) d += (splitters[q] <= bin) ? 1u : 0u;
  dest[t] = d;
}

// bg_cover_query_candidates + bg_partition_rows(BG_PART_PREFIX) in one: the candidate rows of the round are written
// once, at their place in the partitioned order (the unpartitioned rows — 12 GB for a chr20-sized rank's first
// rounds — are never materialised).  h_params = {prefix bases, nparts - 1 splitters}; h_counts[nparts].
extern "C" int bg_cover_query_candidates_split(bg_ctx* ctx, bg_cover_query* q, int j_lo, int j_hi, uint32_t nparts,
                                               const uint64_t* h_params, uint64_t* d_out_rows, uint64_t cap, uint64_t* h_counts,
                                               uint64_t* n_out) {
  if (!ctx || !q || !h_params || !h_counts || !n_out) { set_error("bg_cover_query_candidates_split: null argument"); return BG_EINVAL; }
  if (j_lo < 1 || j_hi <= j_lo || j_hi > 256) { set_error("bg_cover_query_candidates_split: bad suffix range [%d, %d)", j_lo, j_hi); return BG_EINVAL; }
  if (nparts < 1 || nparts > 128) { set_error("bg_cover_query_candidates_split: bad geometry"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const int span = j_hi - j_lo, rw = q->W + 1;
  const uint64_t total = q->n_pending * uint64_t(span);
  for (uint32_t p = 0; p < nparts; ++p) h_counts[p] = 0;
  *n_out = 0;
  q->j_lo = j_lo;
  q->span = span;
  if (!total) return BG_OK;
  if (total >= (1ull << 32)) { set_error("bg_cover_query_candidates_split: round too large"); return BG_EINVAL; }
  const int hb = int(h_params[0]);
  if (hb < 1 || hb > 16) { set_error("bg_cover_query_candidates_split: prefix bases %d outside 1..16", hb); return BG_EINVAL; }
  std::vector<uint32_t> sp(nparts, 0);
  for (uint32_t x = 0; x + 1 < nparts; ++x) sp[x] = uint32_t(h_params[1 + x]);
  DevBuf<uint32_t> dsp, tcounts;
  DevBuf<uint64_t> da, toffs;
  DevBuf<unsigned long long> counts;
  const uint32_t ntiles = uint32_t((total + MS_TILE - 1) / MS_TILE);
  BG_TRY(dsp.alloc(ctx, nparts));
  BG_TRY(da.alloc(ctx, total));
  BG_TRY(counts.alloc(ctx, 256));
  BG_TRY(tcounts.alloc(ctx, uint64_t(nparts) * ntiles));
  BG_TRY(toffs.alloc(ctx, uint64_t(nparts) * ntiles));
  BG_HIP(hipMemcpyAsync(dsp.p, sp.data(), nparts * 4, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemsetAsync(counts.p, 0, 256 * 8, ctx->stream));
  const CandidateSource cs{q->owords.p, q->olen.p, q->pending.p, q->W, q->K, j_lo, span};
  ctx->tic("partition_dest");
  k_dest_candidates<<<cdiv(total, 256), 256, 0, ctx->stream>>>(cs, total, hb, dsp.p, nparts, da.p);
  BG_LAUNCH_CHECK();
  BG_HIP(hipStreamSynchronize(ctx->stream));  // sp goes out of scope with the function, dsp is read by then
  unsigned grid = cdiv(total, 256) < 1024u ? cdiv(total, 256) : 1024u;
  k_dest_counts<<<grid, 256, 0, ctx->stream>>>(da.p, total, counts.p);
  BG_LAUNCH_CHECK();
  ctx->toc(total * 24);
  ctx->tic("partition_split");
  k_msplit_count<<<ntiles, 256, 0, ctx->stream>>>(da.p, total, nparts, tcounts.p, ntiles);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32_to_u64(ctx, tcounts.p, toffs.p, uint64_t(nparts) * ntiles, nullptr));
  unsigned long long hc[256];
  BG_HIP(hipMemcpyAsync(hc, counts.p, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  uint64_t n_rows = 0;
  for (uint32_t p = 0; p < nparts; ++p) { h_counts[p] = hc[p]; n_rows += hc[p]; }
  if (n_rows > cap || (n_rows && !d_out_rows)) { set_error("bg_cover_query_candidates_split: room for %llu rows, %llu needed",
                                                           (unsigned long long)cap, (unsigned long long)n_rows); return BG_EINVAL; }
  if (n_rows) {
    k_msplit_scatter<<<ntiles, 256, 0, ctx->stream>>>(cs, da.p, total, rw, nparts, toffs.p, ntiles, d_out_rows);
    BG_LAUNCH_CHECK();
  }
  ctx->toc(n_rows * uint64_t(rw) * 8 + total * 16);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = n_rows;
  return BG_OK;
}


// ---- filtered cover rounds -------------------------------------------------------------------------------------------------
// The rounds above ship EVERY suffix j of a window to its owner to learn which is the first covered one.  A suffix can
// only be covered by a read that starts with its first K bases, and the owners know those K-mers: every owner marks the
// first K-mers of the reads it indexes in a bit filter of its own (two bits of one 64-bit word per key), the filters are
// all-gathered (they are disjoint by owner: no reduction), and a sender looks a suffix's K-mer up in its owner's filter
// before it sends anything.  A miss is exact -- the suffix is uncovered and simply becomes a record; only hits (the
// covered suffixes plus a false-positive rate of ~0.5 %) travel as candidates to be verified with bg_cover_test.  Same
// records as the window rounds (the first VERIFIED cover ends a read), about one candidate per read instead of nine.
__device__ __forceinline__ uint64_t filter_mix(uint64_t key) {
  uint64_t h = key * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  return h ^ (h >> 32);
}
__device__ __forceinline__ uint64_t filter_mask(uint64_t h) { return (1ull << ((h >> 8) & 63)) | (1ull << ((h >> 16) & 63)); }

__global__ void k_filter_add(const uint64_t* __restrict__ owords, const uint8_t* __restrict__ olen, uint64_t n, int W, int K,
                             int log2w, unsigned long long* __restrict__ bits) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= n || int(olen[o]) < K) return;  // a read shorter than K covers nothing that is looked up
  const uint64_t h = filter_mix(owords[o * uint64_t(W)] >> (64 - 2 * K));
  atomicOr(&bits[h >> (64 - log2w)], (unsigned long long)filter_mask(h));
}

// one lane per pending read: from its cursor, the next suffixes whose K-mer the owner's filter knows (up to max_hits);
// the cursor moves past what was looked at; a read that runs out of suffixes with K bases is settled (no cover).
__global__ void k_query_scan(const uint64_t* __restrict__ owords, const uint8_t* __restrict__ olen, const uint32_t* __restrict__ pending,
                             uint64_t n_pending, int W, int K, const uint64_t* __restrict__ filter, int log2w, int hb,
                             const uint32_t* __restrict__ splitters, uint32_t nparts, uint32_t max_hits, uint8_t* __restrict__ cursor,
                             uint16_t* __restrict__ end_j, uint8_t* __restrict__ hit_j, uint64_t* __restrict__ dest) {
  const uint64_t slot = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (slot >= n_pending) return;
  const uint32_t o = pending[slot];
  const int L = olen[o];
  const uint64_t* s = owords + uint64_t(o) * W;
  int j = cursor[o];
  uint32_t hits = 0;
  while (j < L && hits < max_hits) {
    if (L - j < K) { j = L; break; }  // shorter suffixes are never looked up
    const uint64_t w0 = window32(s, W, j);
    const uint32_t bin = uint32_t(w0 >> (64 - 2 * hb));
    uint32_t d = 0;
    for (uint32_t q = 0; q + 1 < nparts; ++q) d += (splitters[q] <= bin) ? 1u : 0u;
    const uint64_t h = filter_mix(w0 >> (64 - 2 * K));
    const uint64_t m = filter_mask(h);
    if ((filter[(uint64_t(d) << log2w) + (h >> (64 - log2w))] & m) == m) {
      hit_j[slot * max_hits + hits] = uint8_t(j);
      dest[slot * max_hits + hits] = d;
      ++hits;
    }
    ++j;
  }
  cursor[o] = uint8_t(j);
  for (uint32_t x = hits; x < max_hits; ++x) dest[slot * max_hits + x] = kDropDest;
  if (!hits) end_j[o] = uint16_t(L);  // ran to the end without a candidate: uncovered
}

__global__ void k_gather_pending_slots(const uint32_t* __restrict__ pending, const uint32_t* __restrict__ keep_slots, uint64_t m,
                                       uint32_t* __restrict__ out) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < m) out[i] = pending[keep_slots[i]];
}

struct HitSource {  // row t = suffix hit_j[t] of pending read t / max_hits
  const uint64_t* owords;
  const uint8_t* olen;
  const uint32_t* pending;
  const uint8_t* hit_j;
  int W;
  uint32_t max_hits;
  __device__ __forceinline__ uint64_t word(uint64_t t, int w) const {
    const uint32_t o = pending[t / max_hits];
    const int j = hit_j[t];
    if (w == W) return uint64_t(int(olen[o]) - j) | (t << 24);
    return window32(owords + uint64_t(o) * W, W, j + 32 * w);
  }
};

__global__ void k_scan_first_cover(const uint64_t* __restrict__ rows_sent, const uint8_t* __restrict__ covered, uint64_t n, int W,
                                   uint32_t max_hits, const uint8_t* __restrict__ hit_j, uint32_t* __restrict__ first_cov) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n || !covered[i]) return;
  const uint64_t t = rows_sent[i * uint64_t(W + 1) + W] >> 24;
  atomicMin(&first_cov[t / max_hits], uint32_t(hit_j[t]));
}

__global__ void k_scan_settle(const uint32_t* __restrict__ pending, uint64_t n_pending, const uint32_t* __restrict__ first_cov,
                              uint16_t* __restrict__ end_j, uint8_t* __restrict__ flags) {
  const uint64_t slot = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (slot >= n_pending) return;
  const uint32_t o = pending[slot];
  if (first_cov[slot] != 0xffffffffu) end_j[o] = uint16_t(first_cov[slot]);
  flags[slot] = end_j[o] == 0xffffu ? 1 : 0;  // still unsettled: the scan goes on behind the candidates that failed
}

// records of the window [j_lo, j_hi): suffix j of read o is a record iff 1 <= j < end_j[o] (suffixes with fewer than K
// bases included)
struct RecordSource {
  const uint64_t* owords;
  const uint8_t* olen;
  const uint32_t* pending;
  int W, j_lo, span;
  const uint16_t* end_j;  // the flags of k_emit_records ride above the length (j >= 1 here: a base precedes every record)
  __device__ __forceinline__ uint64_t word(uint64_t t, int w) const {
    const uint64_t slot = t / uint64_t(span);
    const int j = j_lo + int(t - slot * uint64_t(span));
    const uint32_t o = pending[slot];
    if (w == W) {
      const uint64_t* s = owords + uint64_t(o) * W;
      uint64_t meta = uint64_t(int(olen[o]) - j) | kRecHasPred | (((s[(j - 1) >> 5] >> (62 - 2 * ((j - 1) & 31))) & 3) << kRecPredShift);
      if (j + 1 == int(end_j[o])) meta |= kRecLast;
      return meta;
    }
    return window32(owords + uint64_t(o) * W, W, j + 32 * w);
  }
};

__global__ void k_dest_records(RecordSource rs, const uint16_t* __restrict__ end_j, uint64_t total, int hb,
                               const uint32_t* __restrict__ splitters, uint32_t nparts, uint64_t* __restrict__ dest) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint64_t slot = t / uint64_t(rs.span);
  const int j = rs.j_lo + int(t - slot * uint64_t(rs.span));
  const uint32_t o = rs.pending[slot];
  if (j >= int(end_j[o])) { dest[t] = kDropDest; return; }
  const uint32_t bin = uint32_t(window32(rs.owords + uint64_t(o) * rs.W, rs.W, j) >> (64 - 2 * hb));
  uint32_t d = 0;
  for (uint32_t q = 0; q + 1 < nparts; ++q) d += (splitters[q] <= bin) ? 1u : 0u;
  dest[t] = d;
}

__global__ void k_flag_end_beyond(const uint16_t* __restrict__ end_j, uint64_t n, int j_lo, uint8_t* __restrict__ flags) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o < n) flags[o] = (end_j[o] != 0xffffu && int(end_j[o]) > j_lo) ? 1 : 0;
}
__global__ void k_init_scan_state(const uint8_t* __restrict__ olen, uint64_t n, uint8_t* __restrict__ cursor, uint16_t* __restrict__ end_j) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= n) return;
  cursor[o] = 1;
  end_j[o] = olen[o] >= 2 ? 0xffffu : olen[o];  // 0xffff = not settled
}

// rows of `src` (one per slot; dest[t] = destination or kDropDest) written in partitioned order
template <class Source>
static int msplit_generate(bg_ctx* ctx, const Source& src, const uint64_t* d_dest, uint64_t total, int rw, uint32_t nparts,
                           uint64_t* d_out_rows, uint64_t cap, uint64_t* h_counts, uint64_t* n_out, const char* who) {
  DevBuf<uint32_t> tcounts;
  DevBuf<uint64_t> toffs;
  DevBuf<unsigned long long> counts;
  const uint32_t ntiles = uint32_t((total + MS_TILE - 1) / MS_TILE);
  BG_TRY(counts.alloc(ctx, 256));
  BG_TRY(tcounts.alloc(ctx, uint64_t(nparts) * ntiles));
  BG_TRY(toffs.alloc(ctx, uint64_t(nparts) * ntiles));
  BG_HIP(hipMemsetAsync(counts.p, 0, 256 * 8, ctx->stream));
  unsigned grid = cdiv(total, 256) < 1024u ? cdiv(total, 256) : 1024u;
  k_dest_counts<<<grid, 256, 0, ctx->stream>>>(d_dest, total, counts.p);
  BG_LAUNCH_CHECK();
  ctx->tic("partition_split");
  k_msplit_count<<<ntiles, 256, 0, ctx->stream>>>(d_dest, total, nparts, tcounts.p, ntiles);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32_to_u64(ctx, tcounts.p, toffs.p, uint64_t(nparts) * ntiles, nullptr));
  unsigned long long hc[256];
  BG_HIP(hipMemcpyAsync(hc, counts.p, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  uint64_t n_rows = 0;
  for (uint32_t p = 0; p < nparts; ++p) { h_counts[p] = hc[p]; n_rows += hc[p]; }
  if (n_rows > cap || (n_rows && !d_out_rows)) { set_error("%s: room for %llu rows, %llu needed", who, (unsigned long long)cap, (unsigned long long)n_rows); return BG_EINVAL; }
  if (n_rows) {
    k_msplit_scatter<<<ntiles, 256, 0, ctx->stream>>>(src, d_dest, total, rw, nparts, toffs.p, ntiles, d_out_rows);
    BG_LAUNCH_CHECK();
  }
  ctx->toc(n_rows * uint64_t(rw) * 8 + total * 16);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = n_rows;
  return BG_OK;
}

static int split_params(const uint64_t* h_params, uint32_t nparts, int* hb, std::vector<uint32_t>* sp, const char* who) {
  *hb = int(h_params[0]);
  if (*hb < 1 || *hb > 16) { set_error("%s: prefix bases %d outside 1..16", who, *hb); return BG_EINVAL; }
  sp->assign(nparts, 0);
  for (uint32_t x = 0; x + 1 < nparts; ++x) (*sp)[x] = uint32_t(h_params[1 + x]);
  return BG_OK;
}

extern "C" int bg_cover_filter_build(bg_ctx* ctx, const bg_cover* c, int log2_words, uint64_t* d_bits) {
  if (!ctx || !c || !d_bits) { set_error("bg_cover_filter_build: null argument"); return BG_EINVAL; }
  if (log2_words < 4 || log2_words > 36) { set_error("bg_cover_filter_build: 2^%d words outside 2^4..2^36", log2_words); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  BG_HIP(hipMemsetAsync(d_bits, 0, (size_t(1) << log2_words) * 8, ctx->stream));
  if (c->n) {
    ctx->tic("cover_filter");
    k_filter_add<<<cdiv(c->n, 256), 256, 0, ctx->stream>>>(c->owords.p, c->olen.p, c->n, c->W, c->K, log2_words,
                                                          reinterpret_cast<unsigned long long*>(d_bits));
    BG_LAUNCH_CHECK();
    ctx->toc(c->n * 24);
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_cover_query_scan(bg_ctx* ctx, bg_cover_query* q, const uint64_t* d_filter, int log2_words, uint32_t max_hits,
                                   uint32_t nparts, const uint64_t* h_params, uint64_t* d_out_rows, uint64_t cap, uint64_t* h_counts,
                                   uint64_t* n_out) {
  if (!ctx || !q || !d_filter || !h_params || !h_counts || !n_out) { set_error("bg_cover_query_scan: null argument"); return BG_EINVAL; }
  if (nparts < 1 || nparts > 128 || max_hits < 1 || max_hits > 64) { set_error("bg_cover_query_scan: bad geometry"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  for (uint32_t p = 0; p < nparts; ++p) h_counts[p] = 0;
  *n_out = 0;
  if (!q->cursor.p) {  // first scan: every read starts at suffix 1
    BG_TRY(q->cursor.alloc(ctx, q->n));
    BG_TRY(q->end_j.alloc(ctx, q->n));
    if (q->n) {
      k_init_scan_state<<<cdiv(q->n, 256), 256, 0, ctx->stream>>>(q->olen.p, q->n, q->cursor.p, q->end_j.p);
      BG_LAUNCH_CHECK();
    }
  }
  q->max_hits = max_hits;
  const uint64_t total = q->n_pending * uint64_t(max_hits);
  if (!total) return BG_OK;
  if (total >= (1ull << 32)) { set_error("bg_cover_query_scan: round too large"); return BG_EINVAL; }
  int hb = 0;
  std::vector<uint32_t> sp;
  BG_TRY(split_params(h_params, nparts, &hb, &sp, "bg_cover_query_scan"));
  DevBuf<uint32_t> dsp;
  DevBuf<uint64_t> da;
  BG_TRY(dsp.alloc(ctx, nparts));
  BG_TRY(da.alloc(ctx, total));
  BG_TRY(q->hit_j.alloc(ctx, total));
  BG_HIP(hipMemcpyAsync(dsp.p, sp.data(), nparts * 4, hipMemcpyHostToDevice, ctx->stream));
  ctx->tic("cover_scan");
  k_query_scan<<<cdiv(q->n_pending, 128), 128, 0, ctx->stream>>>(q->owords.p, q->olen.p, q->pending.p, q->n_pending, q->W, q->K, d_filter,
                                                                log2_words, hb, dsp.p, nparts, max_hits, q->cursor.p, q->end_j.p,
                                                                q->hit_j.p, da.p);
  BG_LAUNCH_CHECK();
  ctx->toc(q->n_pending * (uint64_t(q->W) * 8 + 64));
  BG_HIP(hipStreamSynchronize(ctx->stream));  // sp goes out of scope with the function
  const HitSource src{q->owords.p, q->olen.p, q->pending.p, q->hit_j.p, q->W, max_hits};
  return msplit_generate(ctx, src, da.p, total, q->W + 1, nparts, d_out_rows, cap, h_counts, n_out, "bg_cover_query_scan");
}

extern "C" int bg_cover_query_scan_answers(bg_ctx* ctx, bg_cover_query* q, const uint64_t* d_rows_sent, const uint8_t* d_covered,
                                           uint64_t n) {
  if (!ctx || !q || (n && (!d_rows_sent || !d_covered))) { set_error("bg_cover_query_scan_answers: null argument"); return BG_EINVAL; }
  if (!q->max_hits || !q->end_j.p) { set_error("bg_cover_query_scan_answers: no scan in flight"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  DevBuf<uint32_t> fc, keep_slots, np;
  DevBuf<uint8_t> flags;
  BG_TRY(fc.alloc(ctx, q->n_pending));
  BG_TRY(flags.alloc(ctx, q->n_pending));
  if (q->n_pending) BG_HIP(hipMemsetAsync(fc.p, 0xff, q->n_pending * 4, ctx->stream));
  if (n) {
    k_scan_first_cover<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows_sent, d_covered, n, q->W, q->max_hits, q->hit_j.p, fc.p);
    BG_LAUNCH_CHECK();
  }
  uint64_t m = 0;
  if (q->n_pending) {
    k_scan_settle<<<cdiv(q->n_pending, 256), 256, 0, ctx->stream>>>(q->pending.p, q->n_pending, fc.p, q->end_j.p, flags.p);
    BG_LAUNCH_CHECK();
  }
  BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, q->n_pending, &keep_slots, &m));
  BG_TRY(np.alloc(ctx, m));
  if (m) {
    k_gather_pending_slots<<<cdiv(m, 256), 256, 0, ctx->stream>>>(q->pending.p, keep_slots.p, m, np.p);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  q->pending = std::move(np);
  q->n_pending = m;
  q->max_hits = 0;
  return BG_OK;
}

// The records of the suffix window [j_lo, j_hi) of all settled reads (every read must be: no scan pending), written in
// partitioned order.  n_reads_out (may be NULL) = reads with records at or beyond j_hi (0: this was the last window).
extern "C" int bg_cover_query_records_split(bg_ctx* ctx, bg_cover_query* q, int j_lo, int j_hi, uint32_t nparts, const uint64_t* h_params,
                                            uint64_t* d_out_rows, uint64_t cap, uint64_t* h_counts, uint64_t* n_out) {
  if (!ctx || !q || !h_params || !h_counts || !n_out) { set_error("bg_cover_query_records_split: null argument"); return BG_EINVAL; }
  if (j_lo < 1 || j_hi <= j_lo || j_hi > 256 || nparts < 1 || nparts > 128) { set_error("bg_cover_query_records_split: bad geometry"); return BG_EINVAL; }
  if (q->n_pending) { set_error("bg_cover_query_records_split: %llu reads are not settled yet", (unsigned long long)q->n_pending); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  for (uint32_t p = 0; p < nparts; ++p) h_counts[p] = 0;
  *n_out = 0;
  if (!q->n || !q->end_j.p) return BG_OK;
  DevBuf<uint8_t> flags;
  DevBuf<uint32_t> live;
  BG_TRY(flags.alloc(ctx, q->n));
  k_flag_end_beyond<<<cdiv(q->n, 256), 256, 0, ctx->stream>>>(q->end_j.p, q->n, j_lo, flags.p);
  BG_LAUNCH_CHECK();
  uint64_t n_live = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, q->n, &live, &n_live));
  const int span = j_hi - j_lo;
  const uint64_t total = n_live * uint64_t(span);
  if (!total) return BG_OK;
  if (total >= (1ull << 32)) { set_error("bg_cover_query_records_split: window too large"); return BG_EINVAL; }
  int hb = 0;
  std::vector<uint32_t> sp;
  BG_TRY(split_params(h_params, nparts, &hb, &sp, "bg_cover_query_records_split"));
  DevBuf<uint32_t> dsp;
  DevBuf<uint64_t> da;
  BG_TRY(dsp.alloc(ctx, nparts));
  BG_TRY(da.alloc(ctx, total));
  BG_HIP(hipMemcpyAsync(dsp.p, sp.data(), nparts * 4, hipMemcpyHostToDevice, ctx->stream));
  const RecordSource rs{q->owords.p, q->olen.p, live.p, q->W, j_lo, span, q->end_j.p};
  ctx->tic("partition_dest");
  k_dest_records<<<cdiv(total, 256), 256, 0, ctx->stream>>>(rs, q->end_j.p, total, hb, dsp.p, nparts, da.p);
  BG_LAUNCH_CHECK();
  ctx->toc(total * 24);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return msplit_generate(ctx, rs, da.p, total, q->W + 1, nparts, d_out_rows, cap, h_counts, n_out, "bg_cover_query_records_split");
}

// The oriented reads whose suffix 1 is covered already (their whole-read record is their last one: its tail was not emitted),
// as rows of W + 1 words (words, length): sent to the shard the read falls into, where -- if the read is an entry --
// bg_seqset_mark_need makes that entry send its tail.  (The whole-read records themselves left before the cover rounds and do
// not know.)  d_rows NULL: the count alone.
__global__ void k_flag_first_covered(const uint16_t* __restrict__ end_j, uint64_t n, uint8_t* __restrict__ flags) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o < n) flags[o] = end_j[o] == 1 ? 1 : 0;
}
__global__ void k_rows_of_picked(const uint64_t* __restrict__ owords, const uint8_t* __restrict__ olen, const uint32_t* __restrict__ pick,
                                 uint64_t m, int W, uint64_t* __restrict__ rows) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const uint32_t o = pick[t];
  uint64_t* r = rows + t * uint64_t(W + 1);
  for (int w = 0; w < W; ++w) r[w] = owords[uint64_t(o) * W + w];
  r[W] = olen[o];
}
extern "C" int bg_cover_query_first_covered_rows(bg_ctx* ctx, const bg_cover_query* q, uint64_t* d_rows, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !q || !n_out) { set_error("bg_cover_query_first_covered_rows: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  *n_out = 0;
  if (!q->n || !q->end_j.p) return BG_OK;
  DevBuf<uint8_t> flags;
  DevBuf<uint32_t> pick;
  uint64_t m = 0;
  BG_TRY(flags.alloc(ctx, q->n));
  k_flag_first_covered<<<cdiv(q->n, 256), 256, 0, ctx->stream>>>(q->end_j.p, q->n, flags.p);
  BG_LAUNCH_CHECK();
  BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, q->n, &pick, &m));
  *n_out = m;
  if (!d_rows || !m) return BG_OK;
  if (m > cap) { set_error("bg_cover_query_first_covered_rows: room for %llu rows, %llu needed", (unsigned long long)cap, (unsigned long long)m); return BG_EINVAL; }
  k_rows_of_picked<<<cdiv(m, 256), 256, 0, ctx->stream>>>(q->owords.p, q->olen.p, pick.p, m, q->W, d_rows);
  BG_LAUNCH_CHECK();
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// reads with records at or beyond suffix j_lo (the size of a records window: x span rows at most)
extern "C" int bg_cover_query_records_live(bg_ctx* ctx, const bg_cover_query* q, int j_lo, uint64_t* n_out) {
  if (!ctx || !q || !n_out) { set_error("bg_cover_query_records_live: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  *n_out = 0;
  if (!q->n || !q->end_j.p) return BG_OK;
  DevBuf<uint8_t> flags;
  DevBuf<uint32_t> live;
  BG_TRY(flags.alloc(ctx, q->n));
  k_flag_end_beyond<<<cdiv(q->n, 256), 256, 0, ctx->stream>>>(q->end_j.p, q->n, j_lo, flags.p);
  BG_LAUNCH_CHECK();
  return compact_flagged<uint32_t>(ctx, flags.p, q->n, &live, n_out);
}

// ---- reads export / import ----------------------------------------------------------------------

extern "C" int bg_reads_words(const bg_reads* r) { return r ? r->W : 0; }

extern "C" int bg_reads_export(bg_ctx* ctx, const bg_reads* r, int W_out, uint64_t* d_words, uint16_t* d_len) {
  if (!ctx || !r || (r->n && (!d_words || !d_len))) { set_error("bg_reads_export: null argument"); return BG_EINVAL; }
  // narrower than the stored stride is fine as long as every read fits (corrected reads keep their input's
  // stride, and truncation can leave the last word of all of them empty)
  if (W_out < 1 || W_out > kMaxWords || uint32_t(W_out) * 32 < r->max_len) {
    set_error("bg_reads_export: %d words do not hold reads of up to %u bases", W_out, r->max_len);
    return BG_EINVAL;
  }
  BG_HIP(hipSetDevice(ctx->device));
  if (r->n) {
    k_restride_words<<<cdiv(r->n * W_out, 256), 256, 0, ctx->stream>>>(r->words.p, r->n, r->W, W_out, d_words);
    BG_LAUNCH_CHECK();
    BG_HIP(hipMemcpyAsync(d_len, r->len.p, r->n * 2, hipMemcpyDeviceToDevice, ctx->stream));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_reads_import(bg_ctx* ctx, uint64_t n, int W, uint32_t max_len, const uint64_t* d_words,
                               const uint16_t* d_len, bg_reads** out) {
  if (!ctx || !out || (n && (!d_words || !d_len))) { set_error("bg_reads_import: null argument"); return BG_EINVAL; }
  if (W < 1 || W > kMaxWords || max_len > uint32_t(32 * W)) { set_error("bg_reads_import: bad geometry"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_reads> r(new bg_reads);
  r->ctx = ctx;
  r->n = n;
  r->W = W;
  r->max_len = max_len;
  BG_TRY(r->words.alloc(ctx, n * W));
  BG_TRY(r->nmask.alloc(ctx, n * W));
  BG_TRY(r->len.alloc(ctx, n));
  if (n) {
    BG_HIP(hipMemcpyAsync(r->words.p, d_words, n * W * 8, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemsetAsync(r->nmask.p, 0, n * W * 4, ctx->stream));
    BG_HIP(hipMemcpyAsync(r->len.p, d_len, n * 2, hipMemcpyDeviceToDevice, ctx->stream));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *out = r.release();
  return BG_OK;
}

// ---- records ------------------------------------------------------------------------------------

extern "C" int bg_suffix_records(bg_ctx* ctx, const bg_reads* reads, uint64_t read_lo, uint64_t read_hi, int cover_k,
                                 bg_records** out) {
  if (!ctx || !reads || !out) { set_error("bg_suffix_records: null argument"); return BG_EINVAL; }
  if (cover_k < 1 || cover_k > 31) { set_error("bg_suffix_records: cover_k %d outside 1..31", cover_k); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_records> rec(new bg_records);
  rec->ctx = ctx;
  rec->W = reads->W;
  BG_TRY(suffix_records_impl(ctx, reads, read_lo, read_hi, cover_k, &rec->rec, &rec->rlen, &rec->n));
  *out = rec.release();
  return BG_OK;
}
extern "C" uint64_t bg_records_count(const bg_records* r) { return r ? r->n : 0; }
// The rows came from bg_cover_query_records_split / bg_oriented_rows of a CREATE (their last word carries the emission's flags
// above the length): bg_entries_from_records may set prev bits where the records meet.  Rows of a merge carry the input's
// number there instead and must not be marked.
extern "C" void bg_records_set_meta(bg_records* r, int has_meta) { if (r) r->has_meta = has_meta != 0; }
extern "C" int bg_records_words(const bg_records* r) { return r ? r->W : 0; }
extern "C" void bg_records_free(bg_records* r) { delete r; }

extern "C" int bg_records_rows(bg_ctx* ctx, const bg_records* r, uint64_t* d_rows) {
  if (!ctx || !r || (r->n && !d_rows)) { set_error("bg_records_rows: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (r->n && r->RS == r->W + 1) {  // records that came as rows are rows
    BG_HIP(hipMemcpyAsync(d_rows, r->rec.p, r->n * uint64_t(r->W + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
  } else if (r->n) {
    k_records_to_rows<<<cdiv(r->n * (r->W + 1), 256), 256, 0, ctx->stream>>>(r->rec.p, r->rlen.p, r->n, r->W, d_rows);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// histogram of the leading `bases` bases of every `every`-th record (the splitter histogram of the records
// exchange; splitters only have to balance the shards, and 24 G global atomics/s is all the L2 gives: counting
// all 159 M records of a chr20-sized rank took 6.7 ms, one in eight takes under one).  4^bases 64-bit bins.
__global__ void k_records_prefix_hist(const uint64_t* __restrict__ rec, uint64_t n, int W, int bases, uint64_t every,
                                      unsigned long long* __restrict__ hist) {
  uint64_t i = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * every;
  if (i < n) atomicAdd(&hist[rec[i * W] >> (64 - 2 * bases)], 1ull);
}

extern "C" int bg_records_prefix_hist(bg_ctx* ctx, const bg_records* r, int bases, uint64_t every, uint64_t* d_hist) {
  if (!ctx || !r || !d_hist) { set_error("bg_records_prefix_hist: null argument"); return BG_EINVAL; }
  if (bases < 1 || bases > 12) { set_error("bg_records_prefix_hist: %d bases outside 1..12", bases); return BG_EINVAL; }
  if (every < 1) { set_error("bg_records_prefix_hist: sampling step 0"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  BG_HIP(hipMemsetAsync(d_hist, 0, (size_t(1) << (2 * bases)) * 8, ctx->stream));
  if (r->n) {
    k_records_prefix_hist<<<cdiv(cdiv(r->n, every), 256), 256, 0, ctx->stream>>>(
        r->rec.p, r->n, r->RS ? r->RS : r->W, bases, every, reinterpret_cast<unsigned long long*>(d_hist));  // (row stride)
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_records_from_rows(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int W, bg_records** out) {
  if (!ctx || !out || (n && !d_rows)) { set_error("bg_records_from_rows: null argument"); return BG_EINVAL; }
  if (W < 1 || W > kMaxWords) { set_error("bg_records_from_rows: bad W"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_records> r(new bg_records);
  r->ctx = ctx;
  r->n = n;
  r->W = W;
  // the rows ARE the records (W key words, then the length): one copy, no split into words and lengths
  r->RS = W + 1;
  BG_TRY(r->rec.alloc(ctx, n * (W + 1)));
  if (n) BG_HIP(hipMemcpyAsync(r->rec.p, d_rows, n * (W + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *out = r.release();
  return BG_OK;
}

extern "C" int bg_records_from_row_parts(bg_ctx* ctx, uint32_t n_parts, const uint64_t* const* d_rows, const uint64_t* h_counts,
                                         int W, bg_records** out) {
  if (!ctx || !out || (n_parts && (!d_rows || !h_counts))) { set_error("bg_records_from_row_parts: null argument"); return BG_EINVAL; }
  if (W < 1 || W > kMaxWords) { set_error("bg_records_from_row_parts: bad W"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  uint64_t n = 0;
  for (uint32_t p = 0; p < n_parts; ++p) n += h_counts[p];
  std::unique_ptr<bg_records> r(new bg_records);
  r->ctx = ctx;
  r->n = n;
  r->W = W;
  r->RS = W + 1;  // the rows are the records (see bg_records_from_rows)
  BG_TRY(r->rec.alloc(ctx, n * (W + 1)));
  uint64_t off = 0;
  for (uint32_t p = 0; p < n_parts; ++p) {
    const uint64_t m = h_counts[p];
    if (!m) continue;
    if (!d_rows[p]) { set_error("bg_records_from_row_parts: null part"); return BG_EINVAL; }
    BG_HIP(hipMemcpyAsync(r->rec.p + off * (W + 1), d_rows[p], m * (W + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    off += m;
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *out = r.release();
  return BG_OK;
}

// sorted, prefix-free entries of one shard (keys/sizes/shared; no prev bits yet).  Consumes records.
extern "C" int bg_entries_from_records(bg_ctx* ctx, bg_records* rec, bg_seqset** out, bg_build_stats* stats) {
  if (!ctx || !rec || !out) { set_error("bg_entries_from_records: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_seqset> ss(new bg_seqset);
  ss->ctx = ctx;
  ss->W = rec->W;
  ss->uuid = make_uuid();
  uint64_t n_rec = rec->n, n_tied = 0;
  // rows that carry the emission's flags: prev bits are set where the records meet (k_gather_dedup_build), the shard keeps
  // which survivors still have to send their tail and which look-ups decide the rest (BG_SHARD_TAILS_ALL=1: every tail travels)
  LocalPrev lp;
  lp.shard = true;
  const bool local = rec->has_meta && rec->RS == rec->W + 1 && !getenv("BG_SHARD_TAILS_ALL");
  BG_TRY(entries_from_records_impl(ctx, rec->rec, rec->rlen, n_rec, rec->W, ss.get(), &n_tied, nullptr, rec->RS, local ? &lp : nullptr));
  if (local && lp.valid) {
    ss->emask = std::move(lp.emask);
    ss->need = std::move(lp.need);
    ss->n_cond = lp.n_items;
    // (room for the four rows bg_seqset_fix_boundary may add)
    BG_TRY(ss->cond.alloc(ctx, (lp.n_items + 4) * uint64_t(rec->W + 1)));
    if (lp.n_items) BG_HIP(hipMemcpyAsync(ss->cond.p, lp.list.p, lp.n_items * uint64_t(rec->W + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  rec->n = 0;
  if (stats) { stats->n_records = n_rec; stats->n_tied_records = n_tied; stats->n_entries = ss->n; }
  *out = ss.release();
  return BG_OK;
}

// first / last entry as rows of W+1 words (last word = length, or ~0 when the shard is empty)
extern "C" int bg_seqset_boundary_rows(bg_ctx* ctx, const bg_seqset* ss, uint64_t* h_first_row, uint64_t* h_last_row) {
  if (!ctx || !ss || !h_first_row || !h_last_row) { set_error("bg_seqset_boundary_rows: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const int W = ss->W;
  for (int w = 0; w <= W; ++w) { h_first_row[w] = 0; h_last_row[w] = 0; }
  h_first_row[W] = h_last_row[W] = ~0ull;
  if (!ss->n) return BG_OK;
  BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_boundary_rows"));
  uint8_t l0 = 0, l1 = 0;
  BG_HIP(hipMemcpyAsync(h_first_row, ss->keys.p, W * 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(h_last_row, ss->keys.p + (ss->n - 1) * W, W * 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(&l0, ss->sizes.p, 1, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(&l1, ss->sizes.p + (ss->n - 1), 1, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  h_first_row[W] = l0;
  h_last_row[W] = l1;
  return BG_OK;
}

// next_first: first entry of the next non-empty shard (or NULL): drop our last entry if it is a prefix
// of it (part_repo.cpp:294-303 min_next_entry).  prev_last: last entry of the previous non-empty
// shard (or NULL): shared[0] = their common prefix length (builder.cpp:225).
extern "C" int bg_seqset_fix_boundary(bg_ctx* ctx, bg_seqset* ss, const uint64_t* h_prev_last_row,
                                      const uint64_t* h_next_first_row) {
  if (!ctx || !ss) { set_error("bg_seqset_fix_boundary: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const int W = ss->W;
  if (!ss->n) return BG_OK;
  std::vector<uint64_t> first(W + 1), last(W + 1);
  BG_TRY(bg_seqset_boundary_rows(ctx, ss, first.data(), last.data()));
  if (h_next_first_row && h_next_first_row[W] != ~0ull) {
    int ll = int(last[W]), nl = int(h_next_first_row[W] & 0xff);
    if (key_lcp(last.data(), ll, h_next_first_row, nl, W) == ll) {
      ss->n -= 1;
      if (ss->emask.p) {
        // the dropped entry t is not an entry: what its records set on it (prev_b[t], b.t a suffix) holds for the next shard's
        // first entry only if b.t is an entry -- looked up like the records dropped as proper prefixes inside the shard
        uint8_t m = 0;
        BG_HIP(copy_d2h(ctx, &m, ss->emask.p + ss->n, 1));
        for (int b = 0; b < 4; ++b) {
          if (!((m >> b) & 1)) continue;
          std::vector<uint64_t> row(W + 1);
          for (int w = 0; w < W; ++w) row[w] = ((w ? last[w - 1] : uint64_t(b)) << 62) | (last[w] >> 2);
          row[W] = uint64_t(ll + 1);
          if (ll + 1 > 32 * W) continue;  // (cannot be a suffix of a read of this width)
          BG_HIP(copy_h2d(ctx, ss->cond.p + ss->n_cond * uint64_t(W + 1), row.data(), (W + 1) * 8));
          ++ss->n_cond;
        }
      }
    }
  }
  if (ss->n && h_prev_last_row && h_prev_last_row[W] != ~0ull) {
    uint8_t sh = uint8_t(key_lcp(h_prev_last_row, int(h_prev_last_row[W] & 0xff), first.data(), int(first[W]), W));
    BG_HIP(hipMemcpyAsync(ss->shared.p, &sh, 1, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return BG_OK;
}

extern "C" int bg_seqset_words(const bg_seqset* ss) { return ss ? ss->W : 0; }

// Sharded merge: d_rows = the rows this shard received (n rows of W+1 words, last word = length | input << 8),
// ss = the shard's entries after bg_seqset_fix_boundary.  d_marks[input * n_entries + e] = 1 when entry e is the
// image of one of that input's entries under the merge map; h_carry[input] = 1 when an entry of the input maps past
// the end of this shard (= to the first entry of the next non-empty shard).
extern "C" int bg_seqset_mark_inputs(bg_ctx* ctx, const bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows,
                                     uint32_t n_inputs, uint8_t* d_marks, uint32_t* h_carry) {
  if (!ctx || !ss || !h_carry || (n_rows && !d_rows) || (ss->n && !d_marks)) { set_error("bg_seqset_mark_inputs: null argument"); return BG_EINVAL; }
  if (!n_inputs || n_inputs > 256) { set_error("bg_seqset_mark_inputs: %u inputs outside 1..256", n_inputs); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  for (uint32_t i = 0; i < n_inputs; ++i) h_carry[i] = 0;
  if (ss->n) BG_HIP(hipMemsetAsync(d_marks, 0, uint64_t(n_inputs) * ss->n, ctx->stream));
  if (!n_rows) { BG_HIP(hipStreamSynchronize(ctx->stream)); return BG_OK; }
  DevBuf<uint32_t> dir, flags;
  BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_mark_inputs"));
  int bases = 0;
  BG_TRY(flags.alloc(ctx, n_inputs + 1));
  BG_HIP(hipMemsetAsync(flags.p, 0, (n_inputs + 1) * 4, ctx->stream));
  if (ss->n) BG_TRY(build_prefix_dir_keys(ctx, ss->keys.p, ss->n, ss->W, &dir, &bases));
  ctx->tic("merge_mark_inputs");
  k_mark_inputs<<<cdiv(n_rows, 128), 128, 0, ctx->stream>>>(d_rows, n_rows, ss->W, ss->keys.p, ss->sizes.p, ss->n, dir.p, bases,
                                                          n_inputs, d_marks, flags.p, flags.p + n_inputs);
  BG_LAUNCH_CHECK();
  ctx->toc(n_rows * ((ss->W + 1) * 8 + 64));
  std::vector<uint32_t> h(n_inputs + 1);
  BG_HIP(hipMemcpyAsync(h.data(), flags.p, (n_inputs + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h[n_inputs] & 1) { set_error("bg_seqset_mark_inputs: a row names an input >= %u", n_inputs); return BG_EINVAL; }
  if (h[n_inputs] & 2) { set_error("bg_seqset_mark_inputs: an input entry has no extension among the merged entries"); return BG_EINTERNAL; }
  for (uint32_t i = 0; i < n_inputs; ++i) h_carry[i] = h[i];
  return BG_OK;
}

// The entries as rows of W+1 words (key words, then the length): the input of a sharded merge, which sends the
// entries of several seqsets through the records exchange (biograph_merge.cpp:189-290 merges by entry).  Works on a
// finished seqset as well as on a shard that has entries only.
extern "C" int bg_seqset_entry_rows(bg_ctx* ctx, const bg_seqset* ss, uint64_t* d_rows) {
  if (!ctx || !ss || (ss->n && !d_rows)) { set_error("bg_seqset_entry_rows: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (ss->n) {
    BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_entry_rows"));
    if (!ss->sizes.p) { set_error("bg_seqset_entry_rows: the seqset holds no sizes"); return BG_EINVAL; }
    k_records_to_rows<<<cdiv(ss->n * (ss->W + 1), 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->sizes.p, ss->n, ss->W, d_rows);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_seqset_tail_rows(bg_ctx* ctx, const bg_seqset* ss, uint64_t* d_rows) {
  if (!ctx || !ss || (ss->n && !d_rows)) { set_error("bg_seqset_tail_rows: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (ss->n) {
    BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_tail_rows"));
    k_tail_rows<<<cdiv(ss->n, 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->sizes.p, ss->n, ss->W, d_rows);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// Short tails (3 words: see k_tail_rows2) of all entries, each with its destination shard in the row;
// h_bounds = nparts rows of W+1 words: the last entry of every shard (last word ~0 = empty shard).
extern "C" int bg_seqset_tail_rows2(bg_ctx* ctx, const bg_seqset* ss, uint32_t nparts, const uint64_t* h_bounds,
                                    uint64_t* d_rows) {
  if (!ctx || !ss || !h_bounds || (ss->n && !d_rows)) { set_error("bg_seqset_tail_rows2: null argument"); return BG_EINVAL; }
  if (nparts < 1 || nparts > 126) { set_error("bg_seqset_tail_rows2: %u shards outside 1..126", nparts); return BG_EINVAL; }
  if (ss->n >= (1ull << 47)) { set_error("bg_seqset_tail_rows2: too many entries"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!ss->n) return BG_OK;
  BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_tail_rows2"));
  const int W = ss->W;
  DevBuf<uint64_t> db;
  DevBuf<uint32_t> err;
  BG_TRY(db.alloc(ctx, uint64_t(nparts) * (W + 1)));
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemcpyAsync(db.p, h_bounds, uint64_t(nparts) * (W + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  ctx->tic("tails_rows");
  k_tail_rows2<<<cdiv(ss->n, 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->sizes.p, ss->n, W, db.p, nparts, d_rows, err.p);
  BG_LAUNCH_CHECK();
  ctx->toc(ss->n * (uint64_t(W) * 8 + 25));
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_seqset_tail_rows2: a tail is not extended by any entry (missing expansion)"); return BG_EINTERNAL; }
  return BG_OK;
}

// ---- a shard whose entry build set prev bits where its records met (bg_records_set_meta): only some tails still travel.
// bg_seqset_local_prev: 1 when the shard is one.  bg_seqset_cond_rows: the look-ups that decide a few more tails (rows of
// W + 1 words: base + record; they go to the shard their key falls into, bg_seqset_mark_need there).  After those:
// bg_seqset_tails_needed = how many entries send their tail, bg_seqset_tail_rows2_needed = their short rows.
extern "C" int bg_seqset_local_prev(const bg_seqset* ss) { return ss && ss->need.p ? 1 : 0; }
extern "C" int bg_seqset_cond_rows(const bg_seqset* ss, const uint64_t** d_rows, uint64_t* n_rows) {
  if (!ss || !d_rows || !n_rows) { set_error("bg_seqset_cond_rows: null argument"); return BG_EINVAL; }
  *d_rows = ss->cond.p;
  *n_rows = ss->n_cond;
  return BG_OK;
}
extern "C" int bg_seqset_mark_need(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows) {
  if (!ctx || !ss || (n_rows && !d_rows)) { set_error("bg_seqset_mark_need: null argument"); return BG_EINVAL; }
  if (!ss->need.p) { set_error("bg_seqset_mark_need: the shard keeps no tail flags"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!n_rows || !ss->n) return BG_OK;
  BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_mark_need"));
  DevBuf<uint32_t> dir;
  int bases = 0;
  BG_TRY(build_prefix_dir_keys(ctx, ss->keys.p, ss->n, ss->W, &dir, &bases));
  k_mark_need_exact<<<cdiv(n_rows, 128), 128, 0, ctx->stream>>>(d_rows, n_rows, ss->W, ss->keys.p, ss->sizes.p, ss->n, dir.p, bases, ss->need.p);
  BG_LAUNCH_CHECK();
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}
extern "C" int bg_seqset_tail_rows2_needed(bg_ctx* ctx, const bg_seqset* ss, uint32_t nparts, const uint64_t* h_bounds,
                                           uint64_t* d_rows, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !ss || !h_bounds || !n_out) { set_error("bg_seqset_tail_rows2_needed: null argument"); return BG_EINVAL; }
  if (!ss->need.p) { set_error("bg_seqset_tail_rows2_needed: the shard keeps no tail flags"); return BG_EINVAL; }
  if (nparts < 1 || nparts > 126) { set_error("bg_seqset_tail_rows2_needed: %u shards outside 1..126", nparts); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  *n_out = 0;
  if (!ss->n) return BG_OK;
  BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_tail_rows2_needed"));
  const int W = ss->W;
  DevBuf<uint32_t> pick, err;
  uint64_t m = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, ss->need.p, ss->n, &pick, &m));
  *n_out = m;
  if (!d_rows) return BG_OK;  // (the count alone: the caller sizes its buffer)
  if (m > cap) { set_error("bg_seqset_tail_rows2_needed: room for %llu rows, %llu needed", (unsigned long long)cap, (unsigned long long)m); return BG_EINVAL; }
  if (!m) return BG_OK;
  DevBuf<uint64_t> db;
  BG_TRY(db.alloc(ctx, uint64_t(nparts) * (W + 1)));
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemcpyAsync(db.p, h_bounds, uint64_t(nparts) * (W + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  ctx->tic("tails_rows");
  k_tail_rows2_picked<<<cdiv(m, 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->sizes.p, pick.p, m, W, db.p, nparts, d_rows, err.p);
  BG_LAUNCH_CHECK();
  ctx->toc(m * (uint64_t(W) * 8 + 25));
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_seqset_tail_rows2_needed: a tail is not extended by any entry (missing expansion)"); return BG_EINTERNAL; }
  return BG_OK;
}

// Full tails of the short rows i with d_need[i] != 0 (d_short_rows: this shard's short rows in the order the
// flags refer to): *n_out rows of W+1 words into d_rows_out (room for n rows).
extern "C" int bg_seqset_tail_rows_full(bg_ctx* ctx, const bg_seqset* ss, const uint64_t* d_short_rows, const uint8_t* d_need,
                                        uint64_t n, uint64_t* d_rows_out, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !ss || !n_out || (n && (!d_short_rows || !d_need))) { set_error("bg_seqset_tail_rows_full: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  *n_out = 0;
  if (!n) return BG_OK;
  BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_tail_rows_full"));
  DevBuf<uint32_t> pick;
  uint64_t m = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, d_need, n, &pick, &m));
  if (m > cap || (m && !d_rows_out)) { set_error("bg_seqset_tail_rows_full: room for %llu rows, %llu needed", (unsigned long long)cap, (unsigned long long)m); return BG_EINVAL; }
  if (m) {
    k_tail_rows_full<<<cdiv(m, 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->n, ss->W, d_short_rows, pick.p, m, d_rows_out);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = m;
  return BG_OK;
}

// Short tails received from all shards: those the two-word prefixes of this shard's entries settle set their prev
// bit; d_need_full[i] = 1 asks the sender for the whole tail (bg_seqset_tail_rows_full -> bg_seqset_apply_tails).
extern "C" int bg_seqset_apply_tails2(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows, uint8_t* d_need_full) {
  if (!ctx || !ss || (n_rows && (!d_rows || !d_need_full))) { set_error("bg_seqset_apply_tails2: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!ss->bits[0].p) BG_TRY(alloc_prev_bits(ctx, ss));
  if (!n_rows) return BG_OK;
  BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_apply_tails2"));
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  ctx->tic("prev_bits");
  BG_TRY(apply_tails2_impl(ctx, ss, d_rows, n_rows, d_need_full, err.p));
  ctx->toc(n_rows * (24 + 16 + 1));
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_seqset_apply_tails2: missing expansion (entry set not closed under pop_front)"); return BG_EINTERNAL; }
  return BG_OK;
}

extern "C" int bg_seqset_apply_tails(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows) {
  if (!ctx || !ss || (n_rows && !d_rows)) { set_error("bg_seqset_apply_tails: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!ss->bits[0].p) BG_TRY(alloc_prev_bits(ctx, ss));
  if (!n_rows) return BG_OK;
  BG_TRY(seqset_require_keys(ctx, ss, "bg_seqset_apply_tails"));
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  ctx->tic("prev_bits");
  BG_TRY(apply_tails_impl(ctx, ss, d_rows, n_rows, err.p));
  ctx->toc(n_rows * (ss->W * 8 + 1 + 8));
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_seqset_apply_tails: missing expansion (entry set not closed under pop_front)"); return BG_EINTERNAL; }
  return BG_OK;
}

// raw slice of one shard: sizes[n], shared[n] (bytes) and 4 x ceil(n/64) bit words (A,C,G,T)
extern "C" int bg_seqset_slice_export(bg_ctx* ctx, const bg_seqset* ss, uint8_t* d_sizes, uint8_t* d_shared,
                                      uint64_t* d_bits) {
  if (!ctx || !ss) { set_error("bg_seqset_slice_export: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n = ss->n, nw = (n + 63) / 64;
  if (n) {
    if (!ss->bits[0].p) { set_error("bg_seqset_slice_export: prev bits not built"); return BG_EINVAL; }
    BG_HIP(hipMemcpyAsync(d_sizes, ss->sizes.p, n, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(d_shared, ss->shared.p, n, hipMemcpyDeviceToDevice, ctx->stream));
    for (int b = 0; b < 4; ++b)
      BG_HIP(hipMemcpyAsync(d_bits + b * nw, ss->bits[b].p, nw * 8, hipMemcpyDeviceToDevice, ctx->stream));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// Final merge: concatenation of the shards' slices in rank order (shards are lexicographic ranges, so
// no k-way merge), then seqset::finalize.  d_bits holds, shard after shard, 4 x ceil(n_p/64) words.
extern "C" int bg_seqset_assemble(bg_ctx* ctx, uint32_t nparts, const uint64_t* h_part_n, const uint8_t* d_sizes,
                                  const uint8_t* d_shared, const uint64_t* d_bits, uint32_t max_len, bg_seqset** out) {
  if (!ctx || !out || !h_part_n) { set_error("bg_seqset_assemble: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  uint64_t n = 0;
  for (uint32_t p = 0; p < nparts; ++p) n += h_part_n[p];
  std::unique_ptr<bg_seqset> ss(new bg_seqset);
  ss->ctx = ctx;
  ss->n = n;
  ss->W = 0;
  ss->max_len = n ? max_len : 0;
  ss->uuid = make_uuid();
  BG_TRY(ss->sizes.alloc(ctx, n));
  BG_TRY(ss->shared.alloc(ctx, n));
  if (n) {
    BG_HIP(hipMemcpyAsync(ss->sizes.p, d_sizes, n, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(ss->shared.p, d_shared, n, hipMemcpyDeviceToDevice, ctx->stream));
  }
  BG_TRY(alloc_prev_bits(ctx, ss.get()));
  uint64_t off = 0, woff = 0;
  for (uint32_t p = 0; p < nparts; ++p) {
    uint64_t np = h_part_n[p], nw = (np + 63) / 64;
    for (int b = 0; b < 4 && np; ++b) {
      k_place_bits<<<cdiv(nw, 256), 256, 0, ctx->stream>>>(d_bits + woff + b * nw, np, off,
                                                           reinterpret_cast<unsigned long long*>(ss->bits[b].p));
      BG_LAUNCH_CHECK();
    }
    off += np;
    woff += 4 * nw;
  }
  ctx->tic("seqset_finalize");
  BG_TRY(seqset_finalize_device(ctx, ss.get()));
  ctx->toc(n * 3);
  *out = ss.release();
  return BG_OK;
}
