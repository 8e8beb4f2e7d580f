// build_kmer.hip — S1 (2-bit packing), S2 (canonical k-mer count + filter + hash set) and S3 (read
// correction) of the seqset build, for gfx950.
//
// Reference rows: A2 dna_base.h:38-56 / dna_sequence.h:97; B1 kmer_counter.h:275-326;
// B3 kmer_count_table.h:54-103; B4 kmerize_bf.cpp:290-318; C1 fast_read_correct.cpp:16-182;
// C2 correct_reads.cpp:154-178.
#include "bg_device_funcs.h"
#include "bg_internal.h"
#include "compact.h"

#include <stdlib.h>

#include <vector>

namespace bg {

// ------------------------------------------------------------------------------------------------
// S1: lengths, then one thread per (read, 32-base word).

__global__ __launch_bounds__(256) void k_read_lengths(const uint64_t* __restrict__ offs, uint64_t n,
                                                      uint16_t* __restrict__ len, uint32_t* __restrict__ max_len,
                                                      uint32_t* __restrict__ err) {
  __shared__ uint32_t sm[4];
  uint32_t m = 0;
  bool bad = false;
  for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) {
    uint64_t l = offs[i + 1] - offs[i];
    if (offs[i + 1] < offs[i] || l > 255) {  // biograph_create.cpp:134-139 (no --allow-long-reads)
      bad = true;
      l = 0;
    }
    len[i] = uint16_t(l);
    m = uint32_t(l) > m ? uint32_t(l) : m;
  }
  if (bad) atomicOr(err, 1u);
  // block reduction -> one atomic per block (every lane takes part: no early return above)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    uint32_t t = __shfl_xor(m, d, 64);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
    atomicMax(max_len, a > b ? a : b);
  }
}

// The characters of a workgroup's 256 words are one stretch of the input (reads follow each other: a length is the
// difference of two offsets), at most 8 KB: it is staged in LDS with aligned 16-byte loads, and a thread takes its
// 32 characters from there.  (Four unaligned 8-byte loads a thread, 32 bytes apart from lane to lane, kept the
// kernel at a third of the memory rate.)
constexpr int kPackStageDwords = 2048 + 8 + 12;  // 8 KB + alignment slack + what a thread reads past its own bytes
__global__ __launch_bounds__(256) void k_pack_reads(const uint8_t* __restrict__ seq, const uint64_t* __restrict__ offs,
                             const uint16_t* __restrict__ len, uint64_t n, int W, uint64_t* __restrict__ words,
                             uint32_t* __restrict__ nmask, uint32_t* __restrict__ err) {
  __shared__ uint32_t stage[kPackStageDwords];
  const uint64_t total = n * uint64_t(W);
  const uint64_t g0 = uint64_t(blockIdx.x) * blockDim.x;
  const uint64_t g = g0 + threadIdx.x;
  // g / W as one multiply-high with ceil(2^32 / W) (exact for g < 2^32 / 8) where it can: the 64-bit division is
  // a hundred instructions a thread
  const uint32_t rcp_w = W > 1 ? uint32_t((0x100000000ull + uint32_t(W) - 1) / uint32_t(W)) : 0u;
  auto read_of = [&](uint64_t x) -> uint64_t {
    if (W == 1) return x;
    if (x < (1ull << 29)) return __umulhi(uint32_t(x), rcp_w);
    return x / uint64_t(W);
  };
  // the stretch of the input this workgroup needs (block-uniform)
  const uint64_t gl = (g0 + blockDim.x < total ? g0 + blockDim.x : total) - 1;  // last word (g0 < total: the grid is sized so)
  const uint64_t i_first = read_of(g0), i_last = read_of(gl);
  const int w_first = int(g0 - i_first * uint64_t(W)), w_last = int(gl - i_last * uint64_t(W));
  const int l_first = len[i_first], l_last = len[i_last];
  const uint8_t* b_lo = seq + offs[i_first] + (32 * w_first < l_first ? 32 * w_first : l_first);
  const uint8_t* b_hi = seq + offs[i_last] + (32 * (w_last + 1) < l_last ? 32 * (w_last + 1) : l_last);
  const uint8_t* a_lo = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(b_lo) & ~uintptr_t(15));
  if (b_hi > b_lo) {
    // whole 16-byte pieces: one that holds a valid byte lies in that byte's page, whatever else it covers
    const uint32_t n16 = uint32_t((b_hi - a_lo + 15) >> 4);
    for (uint32_t v = threadIdx.x; v < n16; v += blockDim.x)
      reinterpret_cast<uint4*>(stage)[v] = reinterpret_cast<const uint4*>(a_lo)[v];
  }
  __syncthreads();
  if (g >= total) return;
  const uint64_t i = read_of(g);
  int w = int(g - i * uint64_t(W));
  int L = len[i];
  int lo = w * 32;
  uint64_t word = 0;
  uint32_t nm = 0;
  if (lo < L) {
    const uint32_t lb = uint32_t(seq + offs[i] + lo - a_lo);  // this word's first character in the staged stretch
    const int cnt = L - lo < 32 ? L - lo : 32;
    uint32_t q[8];
    {
      uint32_t d[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) d[j] = stage[(lb >> 2) + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = __builtin_amdgcn_alignbyte(d[j + 1], d[j], lb & 3);
    }
    bool bad = false;
    // eight characters at a time (SWAR): 2-bit codes A0 C1 G2 T3 (dna_base.h:38-56) from bits 1..2 of the
    // letter, exact byte compares for the ACGT / acgt test and for 'N' (only upper-case N is special,
    // kmer_counter.h:306; anything else makes dna_base.h:54 throw)
    constexpr uint64_t kOnes = 0x0101010101010101ull, kLow7 = 0x7f7f7f7f7f7f7f7full;
    auto eq = [&](uint64_t v, uint64_t c) {  // 0x80 in every byte of v that equals c
      const uint64_t z = v ^ (c * kOnes);
      return ~(((z & kLow7) + kLow7) | z | kLow7);
    };
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      const int m = cnt - 8 * c8;  // characters of this chunk
      if (m <= 0) break;
      uint64_t x = uint64_t(q[2 * c8]) | (uint64_t(q[2 * c8 + 1]) << 32);  // characters 8 c8 .. 8 c8 + 7, first one lowest
      if (m < 8) x &= (1ull << (8 * m)) - 1;  // (what follows belongs to the next read)
      const uint64_t y = __builtin_bswap64(x);  // first character in the top byte
      const uint64_t t = (y >> 1) & 0x0303030303030303ull;
      uint64_t code = t ^ ((t >> 1) & kOnes);
      const uint64_t u = y & 0xDFDFDFDFDFDFDFDFull;
      const uint64_t acgt = eq(u, 'A') | eq(u, 'C') | eq(u, 'G') | eq(u, 'T');
      const uint64_t isn = eq(y, 'N');
      code &= (acgt >> 7) * 3;
      uint64_t c1 = (code | (code >> 6)) & 0x000F000F000F000Full;
      c1 = (c1 | (c1 >> 12)) & 0x000000FF000000FFull;
      c1 = (c1 | (c1 >> 24)) & 0xFFFFull;
      const uint32_t valid8 = uint32_t((((acgt >> 7) & kOnes) * 0x0102040810204080ull) >> 56);
      const uint32_t n8 = uint32_t((((isn >> 7) & kOnes) * 0x0102040810204080ull) >> 56);
      const uint32_t in8 = m >= 8 ? 0xFFu : ((0xFFu << (8 - m)) & 0xFFu);  // characters that exist
      if ((~valid8 & ~n8 & in8) != 0) bad = true;
      word |= c1 << (48 - 16 * c8);
      nm |= (n8 & in8) << (24 - 8 * c8);
    }
    if (bad) atomicOr(err, 2u);
    if (nm && !(__atomic_load_n(err, __ATOMIC_RELAXED) & 4u)) atomicOr(err, 4u);  // some read holds an N
  }
  words[g] = word;
  nmask[g] = nm;
}

// ------------------------------------------------------------------------------------------------
// S2: k-mer extraction — one wave per read, one lane per window.

__global__ void k_kmer_counts_per_read(const uint16_t* __restrict__ len, uint64_t n, int k,
                                       uint32_t* __restrict__ nk) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int L = len[i];
  nk[i] = L >= k ? uint32_t(L - k + 1) : 0u;
}

__global__ __launch_bounds__(256) void k_extract_kmers(const uint64_t* __restrict__ words,
                                                       const uint32_t* __restrict__ nmask,
                                                       const uint16_t* __restrict__ len,
                                                       const uint64_t* __restrict__ koff, uint64_t n, int W, int k,
                                                       uint64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  uint64_t i = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (i >= n) return;  // wave-uniform
  int L = len[i];
  if (L < k) return;
  int nk = L - k + 1;
  uint64_t myw = lane < W ? words[i * W + lane] : 0;
  uint32_t myn = lane < W ? nmask[i * W + lane] : 0;
  uint64_t base = koff[i];
  for (int j0 = 0; j0 < nk; j0 += 64) {
    int j = j0 + lane;
    int q = j >> 5, r = j & 31;
    int q0 = q < 63 ? q : 63, q1 = q + 1 < 63 ? q + 1 : 63;
    uint64_t w0 = __shfl(myw, q0, 64), w1 = __shfl(myw, q1, 64);
    uint32_t n0 = __shfl(myn, q0, 64), n1 = __shfl(myn, q1, 64);
    if (j < nk) {
      uint64_t x = r ? ((w0 << (2 * r)) | (w1 >> (64 - 2 * r))) : w0;
      uint32_t nm = r ? ((n0 << r) | (n1 >> (32 - r))) : n0;
      uint64_t kmer = x >> (64 - 2 * k);
      bool invalid = (nm >> (32 - k)) != 0;  // any N inside the window (kmer_counter.h:306-311)
      bool fl;
      uint64_t canon = kmer_canon(kmer, k, &fl);
      out[base + j] = invalid ? ~0ull : canon;
    }
  }
}

// heads of equal-key runs in the sorted instance array (the all-ones sentinel run is a head too)
__global__ void k_flag_heads(const uint64_t* __restrict__ keys, uint64_t n, uint8_t* __restrict__ flags) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// keep iff run length >= min_count (kmerize_bf.cpp:291-294) and not the sentinel
__global__ void k_flag_kept(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ head_pos, uint64_t nheads,
                            uint64_t n, uint32_t min_count, uint8_t* __restrict__ flags) {
  uint64_t r = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= nheads) return;
  uint64_t s = head_pos[r];
  uint64_t e = r + 1 < nheads ? head_pos[r + 1] : n;
  flags[r] = (keys[s] != ~0ull && (e - s) >= uint64_t(min_count)) ? 1 : 0;
}

__global__ void k_gather_kept(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ head_pos,
                              uint64_t nheads, uint64_t n, const uint64_t* __restrict__ kept_r, uint64_t nkept,
                              uint64_t* __restrict__ kmers, uint32_t* __restrict__ counts) {
  uint64_t j = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= nkept) return;
  uint64_t r = kept_r[j];
  uint64_t s = head_pos[r];
  uint64_t e = r + 1 < nheads ? head_pos[r + 1] : n;
  kmers[j] = keys[s];
  uint64_t c = e - s;
  counts[j] = c > 0xffffffffull ? 0xffffffffu : uint32_t(c);
}

__global__ void k_hash_insert(const uint64_t* __restrict__ kmers, uint64_t n, uint64_t* __restrict__ slots,
                              int log2cap) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t km = kmers[i];
  uint64_t mask = (1ull << log2cap) - 1;
  uint64_t s = hash_slot(km, log2cap);
  const int off = hash_slot_offset(km, log2cap);
  for (;;) {
    for (int u = 0; u < kBucketSlots; ++u) {
      uint64_t t = s + ((off + u) & (kBucketSlots - 1));
      if (slots[t] != kEmptySlot) continue;
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&slots[t]),
                                         (unsigned long long)kEmptySlot, (unsigned long long)km);
      if (old == kEmptySlot) return;
    }
    s = (s + kBucketSlots) & mask;
  }
}

// The table region by region.  Global atomics write a line through to HBM each (k_hash_insert: 64 M compare-and-swaps
// = 4.5 GB fetched + 4.2 GB written for a 1 GB table), so big tables are built without them: the keys are partitioned
// by the top 16 bits of their hash — the top 16 bits of their home slot — and one workgroup per region builds its
// 2^(log2cap - 16) slots in LDS and writes them out as whole lines.  A key whose chain runs past the end of its region
// (its buckets there are full) is listed and inserted afterwards the old way, behind the finished regions: every bucket
// before a key's place along its chain is full either way, which is all a look-up relies on.
constexpr int kRegionMaxSlots = 4096;
__global__ __launch_bounds__(256) void k_hash_build_regions(const uint64_t* __restrict__ keys,
                                                            const uint64_t* __restrict__ bounds, int log2cap,
                                                            uint64_t* __restrict__ slots, uint64_t* __restrict__ spill,
                                                            unsigned long long* __restrict__ n_spill) {
  __shared__ unsigned long long tab[kRegionMaxSlots];
  const uint32_t region = blockIdx.x;
  const uint32_t rs = 1u << (log2cap - 16);
  const uint64_t base = uint64_t(region) * rs;
  for (uint32_t i = threadIdx.x; i < rs; i += 256) tab[i] = (unsigned long long)kEmptySlot;
  __syncthreads();
  const uint64_t lo = bounds[region], hi = bounds[region + 1];
  for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const uint64_t km = keys[i];
    uint32_t s = uint32_t(hash_slot(km, log2cap) - base);
    const int off = hash_slot_offset(km, log2cap);
    bool placed = false;
    while (!placed && s < rs) {
      for (int u = 0; u < kBucketSlots && !placed; ++u) {
        const uint32_t t = s + uint32_t((off + u) & (kBucketSlots - 1));
        if (tab[t] != (unsigned long long)kEmptySlot) continue;
        placed = atomicCAS(&tab[t], (unsigned long long)kEmptySlot, (unsigned long long)km) == (unsigned long long)kEmptySlot;
      }
      if (!placed) s += kBucketSlots;
    }
    if (!placed) spill[atomicAdd(n_spill, 1ull)] = km;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < rs; i += 256) slots[base + i] = tab[i];
}

// fwd_starts_read / rev_starts_read (kmer_counter.h:316-320, kmer_count_table.h:82-88): first
// window of a read (position 0, no N) sets fwd unless flipped; last window sets rev unless flipped.
__global__ void k_set_start_flags(const uint64_t* __restrict__ words, const uint32_t* __restrict__ nmask,
                                  const uint16_t* __restrict__ len, uint64_t n, int W, int k,
                                  uint64_t* __restrict__ slots, int log2cap) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int L = len[i];
  if (L < k) return;
  KmerTable tab{slots, log2cap, k};
  const uint64_t* w = words + i * W;
  const uint32_t* nm = nmask + i * W;
  for (int which = 0; which < 2; ++which) {
    if (which == 1 && L == k) break;  // the single window was handled as first AND last
    int j = which ? L - k : 0;
    if ((nwindow32(nm, W, j) >> (32 - k)) != 0) continue;
    uint64_t kmer = window32(w, W, j) >> (64 - 2 * k);
    bool fl;
    uint64_t canon = kmer_canon(kmer, k, &fl);
    int64_t s = tab.find(canon);
    if (s < 0) continue;
    bool set_fwd = (which == 0);              // is_first_kmer
    bool set_rev = (which == 1) || (L == k);  // is_last_kmer
    if (fl) { bool t = set_fwd; set_fwd = set_rev; set_rev = t; }
    uint64_t bits = (set_fwd ? kFwdStartsBit : 0) | (set_rev ? kRevStartsBit : 0);
    atomicOr(reinterpret_cast<unsigned long long*>(&slots[s]), (unsigned long long)bits);
  }
}

__global__ void k_gather_flags(const uint64_t* __restrict__ kmers, uint64_t n, const uint64_t* __restrict__ slots,
                               int log2cap, int k, uint8_t* __restrict__ flags) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  KmerTable tab{slots, log2cap, k};
  int64_t s = tab.find(kmers[i]);
  uint8_t f = 0;
  if (s >= 0) {
    uint64_t v = slots[s];
    if (v & kFwdStartsBit) f |= 1;
    if (v & kRevStartsBit) f |= 2;
  }
  flags[i] = f;
}

// ---- counting by hash partition + LDS tables (the default path) ---------------------------------
// After radix_partition_hash16 all instances of a k-mer sit in one of 65536 buckets (avg N/65536
// instances, a few thousand distinct keys).  One workgroup streams a bucket once from HBM and counts it
// in an LDS open-addressing table (64-bit CAS + 32-bit add); keys seen >= min_count times leave through
// one global cursor.  A bucket whose distinct keys do not fit sets its overflow flag and is recounted
// by the sort-based path.  Role of kmer_count_table<uint8_t>::increment (kmer_count_table.h:54-103).

__global__ void k_bucket_bounds(const uint64_t* __restrict__ keys, uint64_t n, int bits,
                                uint64_t* __restrict__ bounds) {
  uint64_t b = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // 0 .. 2^bits
  if (b > (1ull << bits)) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {  // first position whose hash prefix is >= b
    uint64_t mid = (lo + hi) >> 1;
    if (((keys[mid] * kPartitionMult) >> (64 - bits)) < b) lo = mid + 1; else hi = mid;
  }
  bounds[b] = lo;
}

constexpr int KC_THREADS = 512;

__global__ __launch_bounds__(KC_THREADS) void k_kmer_count_lds(const uint64_t* __restrict__ keys,
                                                               const uint64_t* __restrict__ bounds, int log2cap,
                                                               int bucket_bits, uint32_t min_count, uint64_t out_cap,
                                                               uint64_t* __restrict__ out_kmers,
                                                               uint32_t* __restrict__ out_counts,
                                                               unsigned long long* __restrict__ out_n,
                                                               unsigned long long* __restrict__ distinct_total,
                                                               uint8_t* __restrict__ overflow) {
  extern __shared__ __align__(16) unsigned char kc_smem[];
  const uint32_t cap = 1u << log2cap;
  unsigned long long* tkeys = reinterpret_cast<unsigned long long*>(kc_smem);
  uint32_t* tcnt = reinterpret_cast<uint32_t*>(kc_smem + size_t(cap) * 8);
  __shared__ uint32_t s_over, s_kept, s_distinct, s_cursor;
  __shared__ unsigned long long s_base;
  const uint32_t bucket = blockIdx.x;
  const uint64_t lo = bounds[bucket], hi = bounds[bucket + 1];
  if (lo == hi) return;  // block-uniform
  for (uint32_t i = threadIdx.x; i < cap; i += KC_THREADS) { tkeys[i] = ~0ull; tcnt[i] = 0; }
  if (threadIdx.x == 0) { s_over = 0; s_kept = 0; s_distinct = 0; s_cursor = 0; }
  __syncthreads();
  const uint32_t mask = cap - 1;
  const uint32_t max_probe = cap < 256u ? cap : 256u;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += KC_THREADS) {
    unsigned long long k = keys[i];
    if (k == ~0ull) continue;  // window held an N (kmer_counter.h:306-311)
    uint32_t slot = uint32_t((k * kPartitionMult) >> (64 - bucket_bits - log2cap)) & mask;  // hash bits below the bucket's
    uint32_t probes = 0;
    for (;;) {
      // LDS atomics run at about a lane per clock: most occurrences find their key already in place, and a
      // plain read (full rate) spares them the compare-and-swap
      unsigned long long cur = __atomic_load_n(&tkeys[slot], __ATOMIC_RELAXED);
      if (cur == ~0ull) {
        cur = atomicCAS(&tkeys[slot], ~0ull, k);
        if (cur == ~0ull) cur = k;
      }
      if (cur == k) { atomicAdd(&tcnt[slot], 1u); break; }
      slot = (slot + 1) & mask;
      if (++probes >= max_probe) { s_over = 1; break; }
    }
  }
  __syncthreads();
  if (s_over) {
    if (threadIdx.x == 0) overflow[bucket] = 1;
    return;
  }
  uint32_t my_kept = 0, my_distinct = 0;
  for (uint32_t i = threadIdx.x; i < cap; i += KC_THREADS) {
    if (tkeys[i] != ~0ull) {
      ++my_distinct;
      if (tcnt[i] >= min_count) ++my_kept;
    }
  }
  if (my_distinct) atomicAdd(&s_distinct, my_distinct);
  if (my_kept) atomicAdd(&s_kept, my_kept);
  __syncthreads();
  if (threadIdx.x == 0) {
    s_base = s_kept ? atomicAdd(out_n, (unsigned long long)s_kept) : 0ull;
    if (s_distinct) atomicAdd(distinct_total, (unsigned long long)s_distinct);
  }
  __syncthreads();
  if (!s_kept) return;
  for (uint32_t i = threadIdx.x; i < cap; i += KC_THREADS) {
    if (tkeys[i] != ~0ull && tcnt[i] >= min_count) {
      uint64_t o = s_base + atomicAdd(&s_cursor, 1u);
      if (o < out_cap) { out_kmers[o] = tkeys[i]; out_counts[o] = tcnt[i]; }
    }
  }
}

// ---- locus order ---------------------------------------------------------------------------------
// Reads are processed in the order of a per-read minimiser (the smallest hash over the read's canonical
// 20-mers): reads covering the same genome position become neighbours, so the ~coverage look-ups of
// each k-mer (hash-set probes of the screen / search, cover-index probes) happen close together in time
// and are served by L2 / Infinity Cache instead of HBM.  Purely a schedule: every output is written at
// the read's own index.
constexpr int kLocusM = 20;

__global__ __launch_bounds__(256) void k_read_locus_key(const uint64_t* __restrict__ words,
                                                        const uint16_t* __restrict__ len, uint64_t first,
                                                        uint64_t count, int W, uint64_t* __restrict__ key,
                                                        uint32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  uint64_t t = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (t >= count) return;  // wave-uniform
  uint64_t i = first + t;
  int L = len[i];
  uint64_t myw = lane < W ? words[i * W + lane] : 0;
  uint64_t best = ~0ull;
  int nw = L - kLocusM + 1;
  for (int j0 = 0; j0 < nw; j0 += 64) {
    int j = j0 + lane;
    int q = j >> 5, r = j & 31;
    int q0 = q < 63 ? q : 63, q1 = q + 1 < 63 ? q + 1 : 63;
    uint64_t w0 = __shfl(myw, q0, 64), w1 = __shfl(myw, q1, 64);
    if (j < nw) {
      uint64_t x = r ? ((w0 << (2 * r)) | (w1 >> (64 - 2 * r))) : w0;
      bool fl;
      uint64_t c = kmer_canon(x >> (64 - 2 * kLocusM), kLocusM, &fl);
      uint64_t h = c * 0xD6E8FEB86659FD93ull;
      h ^= h >> 32;
      best = h < best ? h : best;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    uint64_t o = __shfl_xor(best, d, 64);
    best = o < best ? o : best;
  }
  if (lane == 0) { key[t] = best; idx[t] = uint32_t(i); }
}

// perm[t] = index of the t-th read of [first, first+count) in locus order
int read_locus_order(bg_ctx* ctx, const bg_reads* reads, uint64_t first, uint64_t count, DevBuf<uint32_t>* perm) {
  DevBuf<uint64_t> ka, kb;
  DevBuf<uint32_t> ia, ib;
  BG_TRY(ka.alloc(ctx, count));
  BG_TRY(kb.alloc(ctx, count));
  BG_TRY(ia.alloc(ctx, count));
  BG_TRY(ib.alloc(ctx, count));
  if (count) {
    ctx->tic("locus_order");
    k_read_locus_key<<<cdiv(count, 4), 256, 0, ctx->stream>>>(reads->words.p, reads->len.p, first, count, reads->W,
                                                              ka.p, ia.p);
    BG_LAUNCH_CHECK();
    ctx->toc(count * (reads->W * 8 + 12));
  }
  bool in_b = false;
  BG_TRY(radix_sort_u64(ctx, ka.p, kb.p, ia.p, ib.p, count, 0, 64, &in_b));
  *perm = std::move(in_b ? ib : ia);
  return BG_OK;
}

// ------------------------------------------------------------------------------------------------
// S3: correction.  Screen (wave per read, all windows looked up in parallel), then DFS for the
// reads that need it.

__global__ __launch_bounds__(256) void k_correct_screen(const uint64_t* __restrict__ words,
                                                        const uint32_t* __restrict__ nmask,
                                                        const uint16_t* __restrict__ len, uint64_t n, int W, int k,
                                                        const uint64_t* __restrict__ slots, int log2cap,
                                                        uint64_t* __restrict__ out_words,
                                                        uint16_t* __restrict__ out_len,
                                                        uint8_t* __restrict__ need_dfs,
                                                        uint64_t* __restrict__ present,
                                                        const uint32_t* __restrict__ perm) {
  const int lane = threadIdx.x & 63;
  // slot in locus order; XCD-aware: workgroups b, b+8, ... share an L2, so each XCD takes one contiguous
  // eighth of the order (placement is a speed matter only)
  uint64_t slot = xcd_chunk(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (slot >= n) return;
  uint64_t i = perm[slot];
  int L = len[i];
  uint64_t myw = lane < W ? words[i * W + lane] : 0;
  uint32_t myn = lane < W ? nmask[i * W + lane] : 0;
  if (L < k) {  // correct_reads.cpp:155
    if (lane < W) out_words[i * W + lane] = 0;
    if (lane == 0) { out_len[i] = 0; need_dfs[i] = 0; }
    return;
  }
  KmerTable tab{slots, log2cap, k};
  int nk = L - k + 1;
  bool all_ok = true;
  for (int j0 = 0; j0 < nk; j0 += 64) {
    int j = j0 + lane;
    int q = j >> 5, r = j & 31;
    int q0 = q < 63 ? q : 63, q1 = q + 1 < 63 ? q + 1 : 63;
    uint64_t w0 = __shfl(myw, q0, 64), w1 = __shfl(myw, q1, 64);
    uint32_t n0 = __shfl(myn, q0, 64), n1 = __shfl(myn, q1, 64);
    bool ok = true;
    if (j < nk) {
      uint64_t x = r ? ((w0 << (2 * r)) | (w1 >> (64 - 2 * r))) : w0;
      uint32_t nm = r ? ((n0 << r) | (n1 >> (32 - r))) : n0;
      ok = ((nm >> (32 - k)) == 0) && tab.has(x >> (64 - 2 * k));
    }
    uint64_t bal = __ballot(ok && j < nk);
    if (lane == 0) present[i * 4 + (j0 >> 6)] = bal;  // presence bits of the unmodified windows
    if (!__all(ok)) all_ok = false;
  }
  if (all_ok) {  // nothing to correct: output == input
    if (lane < W) out_words[i * W + lane] = myw;
    if (lane == 0) { out_len[i] = uint16_t(L); need_dfs[i] = 0; }
  } else if (lane == 0) {
    need_dfs[i] = 1;
  }
}

// The same decision from presence bits that came with the k-mer count (bg_kmerset::present): no look-ups.
__global__ __launch_bounds__(256) void k_screen_from_present(const uint64_t* __restrict__ words,
                                                             const uint16_t* __restrict__ len, uint64_t n, int W, int k,
                                                             const uint64_t* __restrict__ present, int pstride,
                                                             uint64_t* __restrict__ out_words,
                                                             uint16_t* __restrict__ out_len,
                                                             uint8_t* __restrict__ need_dfs) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int L = len[i];
  bool all_ok = L >= k;
  if (all_ok) {
    const int nk = L - k + 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cnt = nk - c * 64;  // windows of this word
      if (cnt > 0) {
        const uint64_t want = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1);
        if ((present[i * pstride + c] & want) != want) all_ok = false;
      }
    }
  }
  if (L < k) {  // correct_reads.cpp:155
    for (int w = 0; w < W; ++w) out_words[i * W + w] = 0;
    out_len[i] = 0;
    need_dfs[i] = 0;
  } else if (all_ok) {  // nothing to correct: output == input
    for (int w = 0; w < W; ++w) out_words[i * W + w] = words[i * W + w];
    out_len[i] = uint16_t(L);
    need_dfs[i] = 0;
  } else {
    need_dfs[i] = 1;
  }
}

struct CorrectCounters {
  unsigned long long modified, modified_bases;
};

__global__ __launch_bounds__(64) void k_correct_dfs(const uint64_t* __restrict__ words,
                                                    const uint32_t* __restrict__ nmask,
                                                    const uint16_t* __restrict__ len,
                                                    const uint64_t* __restrict__ list, uint64_t nlist, int W, int k,
                                                    const uint64_t* __restrict__ slots, int log2cap,
                                                    int max_corrections, int min_good_run,
                                                    const uint16_t* __restrict__ needed_by_len,
                                                    uint64_t* __restrict__ out_words, uint16_t* __restrict__ out_len,
                                                    CorrectCounters* __restrict__ ctr,
                                                    const uint64_t* __restrict__ present, int pstride) {
  extern __shared__ uint64_t dfs_smem[];  // 64 rows of W + 1 words, 64 of 5 presence words, 64 of W + 1 N masks
  uint64_t* const own_w = dfs_smem;
  uint64_t* const own_p = own_w + 64 * (W + 1);
  uint32_t* const own_n = reinterpret_cast<uint32_t*>(own_p + 64 * 5);
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  int clen = 0, ncorr = 0;
  if (t < nlist) {
    // read-index order on purpose: in locus order the expensive reads (shared SNP / error loci) pile up in
    // the same waves and the divergent search loses more to imbalance than it gains in locality
    uint64_t i = list[t];
    KmerTable tab{slots, log2cap, k};
    int L = len[i];
    // the read, its N masks and its presence words in LDS: the search fetches a base, a window, a presence bit at a
    // time, and from global memory every one of those is a request of its own (rows padded against bank conflicts)
    uint64_t* const my_w = own_w + threadIdx.x * (W + 1);
    uint32_t* const my_n = own_n + threadIdx.x * (W + 1);
    uint64_t* const my_p = own_p + threadIdx.x * 5;
    for (int w = 0; w < W; ++w) { my_w[w] = words[i * W + w]; my_n[w] = nmask[i * W + w]; }
    for (int w = 0; w < pstride && w < 4; ++w) my_p[w] = present[i * pstride + w];
    uint64_t ow[kMaxWords];
    clen = frc_correct_read(tab, my_w, my_n, L, W, k, max_corrections, min_good_run,
                            int(needed_by_len[L]), ow, &ncorr, my_p);
    for (int w = 0; w < W; ++w) out_words[i * W + w] = clen ? ow[w] : 0;
    out_len[i] = uint16_t(clen);
  }
  // one pair of atomics per wave (the counters are single addresses, and a global atomic writes its line through:
  // three million of them, one per modified read, were 0.4 GB of the kernel's writes); ncorr <= 8
  const bool mod = clen && ncorr;
  const unsigned long long wave_mod = __popcll(__ballot(mod));
  unsigned long long wave_bases = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) wave_bases += (unsigned long long)__popcll(__ballot(mod && ((ncorr >> b) & 1))) << b;
  if ((threadIdx.x & 63) == 0 && wave_mod) {
    atomicAdd(&ctr->modified, wave_mod);
    atomicAdd(&ctr->modified_bases, wave_bases);
  }
}

struct CorrectTotals {
  unsigned long long corrected_reads, corrected_bases, failed, truncated, dropped_bases, max_len;
};

__global__ __launch_bounds__(256) void k_correct_stats(const uint16_t* __restrict__ len,
                                                       const uint16_t* __restrict__ out_len, uint64_t n, int k,
                                                       CorrectTotals* __restrict__ tot) {
  __shared__ unsigned long long sm[4][6];
  unsigned long long v[6] = {0, 0, 0, 0, 0, 0};  // [5] = max corrected length
  auto one = [&](int L, int C) {
    if (C > 0) {
      v[0] += 1;
      v[1] += C;
      if (C < L) { v[3] += 1; v[4] += L - C; }
      v[5] = (unsigned long long)C > v[5] ? C : v[5];
    } else if (L >= k) {
      v[2] += 1;
    }
  };
  // eight reads per thread and step, two 16-byte loads (the arrays are 16-byte aligned); one workgroup per CU: the six
  // totals are single addresses, and 2048 workgroups' atomics on them took longer than the sums
  const uint64_t n8 = n / 8;
  for (uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; g < n8; g += uint64_t(gridDim.x) * blockDim.x) {
    const uint4 a = reinterpret_cast<const uint4*>(len)[g], b = reinterpret_cast<const uint4*>(out_len)[g];
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      one(int(aw[j] & 0xffff), int(bw[j] & 0xffff));
      one(int(aw[j] >> 16), int(bw[j] >> 16));
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n8 * 8) one(len[n8 * 8 + threadIdx.x], out_len[n8 * 8 + threadIdx.x]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] += __shfl_xor(v[q], d, 64);
    unsigned long long t = __shfl_xor(v[5], d, 64);
    v[5] = t > v[5] ? t : v[5];
  }
  if ((threadIdx.x & 63) == 0)
    for (int q = 0; q < 6; ++q) sm[threadIdx.x >> 6][q] = v[q];
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long r[6];
    for (int q = 0; q < 5; ++q) r[q] = sm[0][q] + sm[1][q] + sm[2][q] + sm[3][q];
    r[5] = 0;
    for (int w = 0; w < 4; ++w) r[5] = sm[w][5] > r[5] ? sm[w][5] : r[5];
    if (r[0]) atomicAdd(&tot->corrected_reads, r[0]);
    if (r[1]) atomicAdd(&tot->corrected_bases, r[1]);
    if (r[2]) atomicAdd(&tot->failed, r[2]);
    if (r[3]) atomicAdd(&tot->truncated, r[3]);
    if (r[4]) atomicAdd(&tot->dropped_bases, r[4]);
    if (r[5]) atomicMax(&tot->max_len, r[5]);
  }
}

__global__ void k_unpack_reads(const uint64_t* __restrict__ words, const uint16_t* __restrict__ len,
                               const uint64_t* __restrict__ offs, uint64_t n, int W, uint8_t* __restrict__ seq) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int L = len[i];
  uint8_t* p = seq + offs[i];
  for (int j = 0; j < L; ++j) p[j] = "ACGT"[read_code(words + i * W, j)];
}

__global__ void k_len_to_u32(const uint16_t* __restrict__ len, uint64_t n, uint32_t* __restrict__ out) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = len[i];
}

}  // namespace bg

using namespace bg;

// ================================================================================================
// C ABI: S1

static int pack_reads_impl(bg_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n, bg_reads** out) {
  if (!ctx || !out || (n && (!d_offs))) { set_error("bg_pack_reads: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_reads> r(new bg_reads);
  r->ctx = ctx;
  r->n = n;
  BG_TRY(r->len.alloc(ctx, n));
  DevBuf<uint32_t> scal;  // [0]=max_len [1]=err
  BG_TRY(scal.alloc(ctx, 2));
  BG_HIP(hipMemsetAsync(scal.p, 0, 8, ctx->stream));
  uint64_t h_offs[2] = {0, 0};
  if (n) {
    ctx->tic("pack_lengths");
    k_read_lengths<<<(cdiv(n, 256) < 2048u ? cdiv(n, 256) : 2048u), 256, 0, ctx->stream>>>(d_offs, n, r->len.p, scal.p, scal.p + 1);
    BG_LAUNCH_CHECK();
    ctx->toc(n * 10);
    BG_HIP(hipMemcpyAsync(&h_offs[0], d_offs, 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipMemcpyAsync(&h_offs[1], d_offs + n, 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  uint32_t h_scal[2] = {0, 0};
  BG_HIP(hipMemcpyAsync(h_scal, scal.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_scal[1]) {
    set_error("bg_pack_reads: read longer than 255 bases or unsorted offsets");
    return BG_EINVAL;
  }
  r->max_len = h_scal[0];
  r->total_bases = h_offs[1] - h_offs[0];
  r->W = int((r->max_len + 31) / 32);
  if (r->W == 0) r->W = 1;
  // (measured: W = 8 for 150-base reads -- every read, oriented read and row a 64-byte line of its own: the kernels that fetch single
  // rows got SLOWER with it, k_items_tie_small 2.89 -> 3.19 ms, k_prev_list_items 3.35 -> 3.70, k_cover_count_sm 5.05 -> 5.13, and
  // the streaming ones pay the 1.6 x: 48.6 -> 51.6 ms for the step)
  BG_TRY(r->words.alloc(ctx, n * r->W));
  BG_TRY(r->nmask.alloc(ctx, n * r->W));
  if (n) {
    ctx->tic("pack_reads");
    k_pack_reads<<<cdiv(n * r->W, 256), 256, 0, ctx->stream>>>(d_seq, d_offs, r->len.p, n, r->W, r->words.p,
                                                                r->nmask.p, scal.p + 1);
    BG_LAUNCH_CHECK();
    ctx->toc(r->total_bases + n * r->W * 12);
    BG_HIP(hipMemcpyAsync(h_scal, scal.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    if (h_scal[1] & 2) {
      set_error("bg_pack_reads: read contains a character outside ACGTN");
      return BG_EBADBASE;
    }
    r->any_n = (h_scal[1] & 4) != 0;
  } else {
    r->any_n = false;
  }
  *out = r.release();
  return BG_OK;
}

extern "C" int bg_pack_reads(bg_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                             bg_reads** out) {
  return pack_reads_impl(ctx, d_seq, d_offs, n_reads, out);
}

extern "C" int bg_pack_reads_host(bg_ctx* ctx, const uint8_t* h_seq, const uint64_t* h_offs, uint64_t n,
                                  bg_reads** out) {
  if (!ctx || !out || !h_offs) { set_error("bg_pack_reads_host: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  uint64_t total = h_offs[n];
  DevBuf<uint8_t> dseq;
  DevBuf<uint64_t> doffs;
  BG_TRY(dseq.alloc(ctx, total));
  BG_TRY(doffs.alloc(ctx, n + 1));
  if (total) BG_HIP(hipMemcpyAsync(dseq.p, h_seq, total, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipMemcpyAsync(doffs.p, h_offs, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  int rc = pack_reads_impl(ctx, dseq.p, doffs.p, n, out);
  (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

extern "C" uint64_t bg_reads_count(const bg_reads* r) { return r ? r->n : 0; }
extern "C" uint64_t bg_reads_total_bases(const bg_reads* r) { return r ? r->total_bases : 0; }
extern "C" uint32_t bg_reads_max_len(const bg_reads* r) { return r ? r->max_len : 0; }
extern "C" void bg_reads_free(bg_reads* r) { delete r; }

extern "C" int bg_reads_download(bg_ctx* ctx, const bg_reads* r, uint8_t* h_seq, uint64_t* h_offs) {
  if (!ctx || !r || !h_offs) { set_error("bg_reads_download: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  uint64_t n = r->n;
  DevBuf<uint32_t> l32;
  DevBuf<uint64_t> offs, total;
  DevBuf<uint8_t> seq;
  BG_TRY(l32.alloc(ctx, n));
  BG_TRY(offs.alloc(ctx, n + 1));
  BG_TRY(total.alloc(ctx, 1));
  if (n) {
    k_len_to_u32<<<cdiv(n, 256), 256, 0, ctx->stream>>>(r->len.p, n, l32.p);
    BG_LAUNCH_CHECK();
  }
  BG_TRY(exclusive_scan_u32_to_u64(ctx, l32.p, offs.p, n, total.p));
  BG_HIP(hipMemcpyAsync(offs.p + n, total.p, 8, hipMemcpyDeviceToDevice, ctx->stream));
  uint64_t h_total = 0;
  BG_HIP(hipMemcpyAsync(&h_total, total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  BG_TRY(seq.alloc(ctx, h_total));
  if (n) {
    k_unpack_reads<<<cdiv(n, 256), 256, 0, ctx->stream>>>(r->words.p, r->len.p, offs.p, n, r->W, seq.p);
    BG_LAUNCH_CHECK();
  }
  if (h_total && h_seq) BG_HIP(hipMemcpyAsync(h_seq, seq.p, h_total, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(h_offs, offs.p, (n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// ================================================================================================
// C ABI: S2

namespace bg {

// number of k-mer windows of every read and their total
static int kmer_offsets(bg_ctx* ctx, const bg_reads* reads, int k, DevBuf<uint64_t>* koff, uint64_t* N) {
  const uint64_t n = reads->n;
  DevBuf<uint32_t> nk;
  DevBuf<uint64_t> total;
  BG_TRY(nk.alloc(ctx, n));
  BG_TRY(koff->alloc(ctx, n));
  BG_TRY(total.alloc(ctx, 1));
  if (n) {
    k_kmer_counts_per_read<<<cdiv(n, 256), 256, 0, ctx->stream>>>(reads->len.p, n, k, nk.p);
    BG_LAUNCH_CHECK();
  }
  BG_TRY(exclusive_scan_u32_to_u64(ctx, nk.p, koff->p, n, total.p));
  BG_HIP(hipMemcpyAsync(N, total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// canonical k-mer of every window (all-ones for windows holding an N) into d_keys[N]
int kmer_extract_impl(bg_ctx* ctx, const bg_reads* reads, int k, const uint64_t* d_koff, uint64_t N,
                      uint64_t* d_keys) {
  const uint64_t n = reads->n;
  if (!N) return BG_OK;
  ctx->tic("kmer_extract");
  k_extract_kmers<<<cdiv(n, 4), 256, 0, ctx->stream>>>(reads->words.p, reads->nmask.p, reads->len.p, d_koff, n,
                                                       reads->W, k, d_keys);
  BG_LAUNCH_CHECK();
  ctx->toc(n * reads->W * 12 + N * 8);
  return BG_OK;
}

// Sort-based counting (fallback for buckets whose distinct keys overflow the LDS table, and the
// reference point of the tests): full radix sort of the instances (ka = data, kb = scratch) + run
// lengths; keeps canonical k-mers seen >= min_count times: fills ks->kmers / counts / size / distinct.
int kmer_count_keys_sort_impl(bg_ctx* ctx, uint64_t* ka, uint64_t* kb, uint64_t N, uint32_t min_count,
                              bg_kmerset* ks) {
  uint64_t nkept = 0, nheads = 0;
  ks->distinct = 0;
  if (N) {
    bool in_b = false;
    // sort on the full key so that the all-ones sentinel of N-windows lands at the end; the census
    // skips the constant high bytes when there is none.
    BG_TRY(radix_sort_u64(ctx, ka, kb, nullptr, nullptr, N, 0, 64, &in_b));
    uint64_t* sorted = in_b ? kb : ka;
    ctx->tic("kmer_runs");
    DevBuf<uint64_t> head_pos, kept_r;
    {
      DevBuf<uint8_t> flags;
      BG_TRY(flags.alloc(ctx, N));
      k_flag_heads<<<cdiv(N, 256), 256, 0, ctx->stream>>>(sorted, N, flags.p);
      BG_LAUNCH_CHECK();
      BG_TRY(compact_flagged<uint64_t>(ctx, flags.p, N, &head_pos, &nheads));
    }
    {
      DevBuf<uint8_t> flags;
      BG_TRY(flags.alloc(ctx, nheads));
      k_flag_kept<<<cdiv(nheads, 256), 256, 0, ctx->stream>>>(sorted, head_pos.p, nheads, N, min_count, flags.p);
      BG_LAUNCH_CHECK();
      BG_TRY(compact_flagged<uint64_t>(ctx, flags.p, nheads, &kept_r, &nkept));
    }
    BG_TRY(ks->kmers.alloc(ctx, nkept));
    BG_TRY(ks->counts.alloc(ctx, nkept));
    if (nkept) {
      k_gather_kept<<<cdiv(nkept, 256), 256, 0, ctx->stream>>>(sorted, head_pos.p, nheads, N, kept_r.p, nkept,
                                                              ks->kmers.p, ks->counts.p);
      BG_LAUNCH_CHECK();
    }
    // distinct = heads minus the sentinel run (if any): the last key tells
    uint64_t last_key = 0;
    BG_HIP(hipMemcpyAsync(&last_key, sorted + (N - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    ks->distinct = nheads - (last_key == ~0ull ? 1 : 0);
    ctx->toc(N * 10 + nheads * 9 + nkept * 12);
  } else {
    BG_TRY(ks->kmers.alloc(ctx, 0));
    BG_TRY(ks->counts.alloc(ctx, 0));
  }
  ks->size = nkept;
  return BG_OK;
}

static int kmer_lds_log2cap() {
  // BG_KMER_LDS_LOG2CAP shrinks the table (tests force the overflow path with it)
  const char* e = getenv("BG_KMER_LDS_LOG2CAP");
  int v = e ? atoi(e) : 12;
  return v < 1 ? 1 : (v > 13 ? 13 : v);
}

// Default counting path: hash partition (2 radix passes on 16 hash bits) + one LDS counting pass.
// ka = instances (clobbered), kb = scratch, both N.  Output k-mers are sorted ascending.
// partitioned = the instances already sit in kb grouped by 16 hash bits (radix_partition_hash16_reads)
int kmer_count_skm(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t N, uint32_t min_count, bg_kmerset* ks, bool* done);
int kmer_skm_finish_present(bg_ctx* ctx, bg_kmerset* ks);

int kmer_count_keys_impl(bg_ctx* ctx, uint64_t* ka, uint64_t* kb, uint64_t N, uint32_t min_count, bg_kmerset* ks,
                         bool partitioned = false) {
  ks->distinct = 0;
  ks->size = 0;
  if (!N) {
    BG_TRY(ks->kmers.alloc(ctx, 0));
    BG_TRY(ks->counts.alloc(ctx, 0));
    return BG_OK;
  }
  bool in_b = partitioned;
  if (!partitioned) BG_TRY(radix_partition_hash16(ctx, ka, kb, N, &in_b));
  // One LDS table holds a few thousand distinct k-mers: 65536 buckets carry about 1.6 G instances at 30x
  // (chr20-sized input).  Bigger inputs get one more partition level so that a bucket stays around 24 k
  // instances (BG_KMER_EXTRA_BITS overrides, for tests).
  // A rank of a sharded build owns a slice of the hash space: its N keys fill only that slice's buckets, so the
  // level is chosen from the average over the non-empty buckets, not from N / 65536.
  int extra = 0;
  {
    DevBuf<uint64_t> b16;
    BG_TRY(b16.alloc(ctx, 65537));
    k_bucket_bounds<<<cdiv(65537, 256), 256, 0, ctx->stream>>>(in_b ? kb : ka, N, 16, b16.p);
    BG_LAUNCH_CHECK();
    std::vector<uint64_t> hb(65537);
    BG_HIP(hipMemcpyAsync(hb.data(), b16.p, 65537 * 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t used = 0;
    for (int b = 0; b < 65536; ++b) used += hb[b + 1] > hb[b];
    const uint64_t avg = N / (used ? used : 1);
    while (extra < 8 && (avg >> extra) > 24000) ++extra;
  }
  if (const char* e = getenv("BG_KMER_EXTRA_BITS")) extra = std::max(0, std::min(8, atoi(e)));
  if (extra) {
    BG_TRY(radix_partition_hash_refine(ctx, in_b ? kb : ka, in_b ? ka : kb, N, extra));
    in_b = !in_b;
  }
  uint64_t* part = in_b ? kb : ka;
  uint64_t* scratch = in_b ? ka : kb;
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
  DevBuf<uint64_t> ok_a, ok_b;
  DevBuf<uint32_t> oc_a, oc_b;
  BG_TRY(ok_a.alloc(ctx, out_cap));
  BG_TRY(oc_a.alloc(ctx, out_cap));
  ctx->tic("kmer_count_lds");
  k_bucket_bounds<<<cdiv(uint64_t(NB) + 1, 256), 256, 0, ctx->stream>>>(part, N, bucket_bits, bounds.p);
  BG_LAUNCH_CHECK();
  const int lg = kmer_lds_log2cap();
  const size_t smem = (size_t(1) << lg) * 12;
  BG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_kmer_count_lds),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
  k_kmer_count_lds<<<NB, KC_THREADS, smem, ctx->stream>>>(part, bounds.p, lg, bucket_bits, min_count, out_cap, ok_a.p, oc_a.p,
                                                          ctr.p, ctr.p + 1, overflow.p);
  BG_LAUNCH_CHECK();
  ctx->toc(N * 8);
  unsigned long long h_ctr[2] = {0, 0};
  std::vector<uint8_t> h_over(NB);
  std::vector<uint64_t> h_bounds(NB + 1);
  BG_HIP(hipMemcpyAsync(h_ctr, ctr.p, 16, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(h_over.data(), overflow.p, NB, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  uint64_t nkept = h_ctr[0], distinct = h_ctr[1];
  // overflowed buckets: recount them together with the sort-based path
  uint64_t n_over = 0;
  for (uint32_t b = 0; b < NB; ++b) n_over += h_over[b];
  if (n_over) {
    BG_HIP(hipMemcpyAsync(h_bounds.data(), bounds.p, (NB + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t tot = 0;
    for (uint32_t b = 0; b < NB; ++b)
      if (h_over[b]) tot += h_bounds[b + 1] - h_bounds[b];
    if (tot >= (1ull << 32)) {  // the recount below is sized for stragglers, not for a whole input
      set_error("k-mer counting: %llu buckets of %u overflowed their tables (%llu instances); raise BG_KMER_EXTRA_BITS",
                (unsigned long long)n_over, NB, (unsigned long long)tot);
      return BG_EINTERNAL;
    }
    DevBuf<uint64_t> ta, tb;
    BG_TRY(ta.alloc(ctx, tot));
    BG_TRY(tb.alloc(ctx, tot));
    uint64_t pos = 0;
    for (uint32_t b = 0; b < NB; ++b) {
      if (!h_over[b]) continue;
      uint64_t len = h_bounds[b + 1] - h_bounds[b];
      BG_HIP(hipMemcpyAsync(ta.p + pos, part + h_bounds[b], len * 8, hipMemcpyDeviceToDevice, ctx->stream));
      pos += len;
    }
    bg_kmerset tmp;
    tmp.ctx = ctx;
    tmp.k = ks->k;
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
  (void)scratch;
  if (nkept > out_cap) { set_error("k-mer output overflow (internal error)"); return BG_EINTERNAL; }
  // blocks finish in any order: the set is exact but unordered; kmerset_sort() orders it when a caller
  // wants the sorted view (download), the build itself only needs the hash set.
  BG_TRY(ks->kmers.alloc(ctx, nkept));
  BG_TRY(ks->counts.alloc(ctx, nkept));
  if (nkept) {
    BG_HIP(hipMemcpyAsync(ks->kmers.p, ok_a.p, nkept * 8, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(ks->counts.p, oc_a.p, nkept * 4, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  ks->size = nkept;
  ks->distinct = distinct;
  ks->sorted = false;
  return BG_OK;
}

// ascending k-mers (+ their counts)
int kmerset_sort(bg_ctx* ctx, bg_kmerset* ks) {
  if (ks->sorted || ks->size < 2) { ks->sorted = true; return BG_OK; }
  DevBuf<uint64_t> kb;
  DevBuf<uint32_t> cb;
  BG_TRY(kb.alloc(ctx, ks->size));
  BG_TRY(cb.alloc(ctx, ks->size));
  bool sb = false;
  BG_TRY(radix_sort_u64(ctx, ks->kmers.p, kb.p, ks->counts.p, cb.p, ks->size, 0, 64, &sb));
  if (sb) {
    BG_HIP(hipMemcpyAsync(ks->kmers.p, kb.p, ks->size * 8, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(ks->counts.p, cb.p, ks->size * 4, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  ks->sorted = true;
  return BG_OK;
}

// hash set over ks->kmers; start flags from `reads` when given
int kmer_hash_build_impl(bg_ctx* ctx, bg_kmerset* ks, const bg_reads* reads) {
  const uint64_t nkept = ks->size;
  int lg = 10;
  // load factor <= 0.7 (8-slot buckets keep chains short) rather than <= 0.5: half the table, which matters
  // beyond 2 GiB where random probes fall off the TLB reach (tools/probe_rate.hip: 55 -> 22 G probes/s at 4 GiB)
  while ((1ull << lg) * 7 < nkept * 10) ++lg;
  if (const char* e = getenv("BG_HASH_MIN_LOG2CAP")) lg = std::max(lg, std::min(30, atoi(e)));  // tests
  ks->log2cap = lg;
  BG_TRY(ks->slots.alloc(ctx, 1ull << lg));
  int region_min_lg = 24;  // below that the table is small enough for the atomics not to matter (tests lower it to 19)
  if (const char* e = getenv("BG_HASH_REGION_MIN_LOG2CAP")) region_min_lg = std::max(19, atoi(e));
  if (lg >= region_min_lg && lg - 16 <= 12 && !getenv("BG_HASH_BUILD_ATOMIC")) {  // region by region (k_hash_build_regions)
    DevBuf<uint64_t> ta, tb, bounds, spill;
    DevBuf<unsigned long long> n_spill;
    BG_TRY(ta.alloc(ctx, nkept));
    BG_TRY(tb.alloc(ctx, nkept));
    BG_TRY(bounds.alloc(ctx, 65537));
    BG_TRY(spill.alloc(ctx, nkept));
    BG_TRY(n_spill.alloc(ctx, 1));
    BG_HIP(hipMemsetAsync(n_spill.p, 0, 8, ctx->stream));
    BG_TRY(radix_partition_hash16_from(ctx, ks->kmers.p, ta.p, tb.p, nkept));
    BG_TRY(radix_partition_bounds(ctx, ta.p, nkept, 16, bounds.p));
    ctx->tic("kmer_hash_build");
    k_hash_build_regions<<<65536, 256, 0, ctx->stream>>>(ta.p, bounds.p, lg, ks->slots.p, spill.p, n_spill.p);
    BG_LAUNCH_CHECK();
    unsigned long long h_spill = 0;
    BG_HIP(hipMemcpyAsync(&h_spill, n_spill.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    if (h_spill) {
      k_hash_insert<<<cdiv(h_spill, 256), 256, 0, ctx->stream>>>(spill.p, h_spill, ks->slots.p, lg);
      BG_LAUNCH_CHECK();
    }
  } else {
    BG_HIP(hipMemsetAsync(ks->slots.p, 0xff, (1ull << lg) * 8, ctx->stream));
    ctx->tic("kmer_hash_build");
    if (nkept) {
      k_hash_insert<<<cdiv(nkept, 256), 256, 0, ctx->stream>>>(ks->kmers.p, nkept, ks->slots.p, lg);
      BG_LAUNCH_CHECK();
    }
  }
  if (reads && reads->n) {
    k_set_start_flags<<<cdiv(reads->n, 256), 256, 0, ctx->stream>>>(reads->words.p, reads->nmask.p, reads->len.p,
                                                                    reads->n, reads->W, ks->k, ks->slots.p, lg);
    BG_LAUNCH_CHECK();
  }
  ctx->toc((1ull << lg) * 8 + nkept * 16 + (reads ? reads->n : 0) * 2 * 64);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

}  // namespace bg

extern "C" int bg_kmer_count(bg_ctx* ctx, const bg_reads* reads, int k, uint32_t min_count, bg_kmerset** out) {
  return bg::kmer_count_impl(ctx, reads, k, min_count, true, out);
}

// with_flags = false: the fwd/rev_starts_read flags (kmer_set.h:20-21) are not set — the whole-create path never
// reads them (the entry set does not depend on them), only bg_kmerset_download reports them
int bg::kmer_count_impl(bg_ctx* ctx, const bg_reads* reads, int k, uint32_t min_count, bool with_flags, bg_kmerset** out) {
  if (!ctx || !reads || !out) { set_error("bg_kmer_count: null argument"); return BG_EINVAL; }
  if (k < 1 || k > 31) { set_error("bg_kmer_count: kmer size %d outside 1..31", k); return BG_EINVAL; }
  if (min_count < 1) min_count = 1;
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_kmerset> ks(new bg_kmerset);
  ks->ctx = ctx;
  ks->k = k;
  DevBuf<uint64_t> koff;
  uint64_t N = 0;
  BG_TRY(kmer_offsets(ctx, reads, k, &koff, &N));
  ks->instances = N;
  // default: super-k-mer path (kmer_skm.hip); BG_KMER_SKM=0 keeps the instance path below (A/B measurements,
  // and what k < 14 falls back to)
  bool done = false;
  {
    const char* skm = getenv("BG_KMER_SKM");
    if (!(skm && skm[0] == '0')) BG_TRY(kmer_count_skm(ctx, reads, k, N, min_count, ks.get(), &done));
  }
  if (!done) {
    DevBuf<uint64_t> ka, kb;
    BG_TRY(ka.alloc(ctx, N));
    BG_TRY(kb.alloc(ctx, N));
    // extraction fused into the first partition pass (instances written once, never re-read by a histogram);
    // BG_KMER_UNFUSED=1 keeps the separate extraction kernel (A/B measurements)
    const char* unfused = getenv("BG_KMER_UNFUSED");
    if (unfused && unfused[0] == '1') {
      BG_TRY(kmer_extract_impl(ctx, reads, k, koff.p, N, ka.p));
      BG_TRY(kmer_count_keys_impl(ctx, ka.p, kb.p, N, min_count, ks.get()));
    } else {
      BG_TRY(radix_partition_hash16_reads(ctx, reads, k, N, ka.p, kb.p));
      BG_TRY(kmer_count_keys_impl(ctx, ka.p, kb.p, N, min_count, ks.get(), true));
    }
  }
  BG_TRY(kmer_hash_build_impl(ctx, ks.get(), with_flags ? reads : nullptr));
  BG_TRY(kmer_skm_finish_present(ctx, ks.get()));
  *out = ks.release();
  return BG_OK;
}

// ---- sharded build: the same phases with the exchange points exposed

extern "C" int bg_kmer_instances(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t* n_out) {
  if (!ctx || !reads || !n_out) { set_error("bg_kmer_instances: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  DevBuf<uint64_t> koff;
  return kmer_offsets(ctx, reads, k, &koff, n_out);
}

extern "C" int bg_kmer_extract(bg_ctx* ctx, const bg_reads* reads, int k, uint64_t* d_keys, uint64_t cap) {
  if (!ctx || !reads || (!d_keys && cap)) { set_error("bg_kmer_extract: null argument"); return BG_EINVAL; }
  if (k < 1 || k > 31) { set_error("bg_kmer_extract: kmer size %d outside 1..31", k); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  DevBuf<uint64_t> koff;
  uint64_t N = 0;
  BG_TRY(kmer_offsets(ctx, reads, k, &koff, &N));
  if (N > cap) { set_error("bg_kmer_extract: buffer holds %llu of %llu keys", (unsigned long long)cap, (unsigned long long)N); return BG_EINVAL; }
  BG_TRY(kmer_extract_impl(ctx, reads, k, koff.p, N, d_keys));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_kmer_extract_by_owner(bg_ctx* ctx, const bg_reads* reads, int k, uint32_t nparts, uint64_t* d_keys,
                                        uint64_t cap, uint64_t* h_counts) {
  if (!ctx || !reads || !h_counts || (!d_keys && cap)) { set_error("bg_kmer_extract_by_owner: null argument"); return BG_EINVAL; }
  if (k < 1 || k > 31) { set_error("bg_kmer_extract_by_owner: kmer size %d outside 1..31", k); return BG_EINVAL; }
  if (nparts < 1 || nparts > 256) { set_error("bg_kmer_extract_by_owner: nparts %u outside 1..256", nparts); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  DevBuf<uint64_t> koff;
  uint64_t N = 0;
  BG_TRY(kmer_offsets(ctx, reads, k, &koff, &N));
  if (N > cap) { set_error("bg_kmer_extract_by_owner: buffer holds %llu of %llu keys", (unsigned long long)cap, (unsigned long long)N); return BG_EINVAL; }
  uint64_t starts[257];
  BG_TRY(radix_partition_first_reads(ctx, reads, k, N, d_keys, starts));
  // owner of hash byte b = b * nparts / 256: contiguous bucket ranges, so the rows of a rank are contiguous
  for (uint32_t p = 0; p < nparts; ++p) {
    uint32_t lo = (p * 256u + nparts - 1) / nparts, hi = ((p + 1) * 256u + nparts - 1) / nparts;
    h_counts[p] = (N ? starts[hi] : 0) - (N ? starts[lo] : 0);
  }
  return BG_OK;
}

extern "C" int bg_kmer_count_keys(bg_ctx* ctx, uint64_t* d_keys, uint64_t n, int k, uint32_t min_count,
                                  bg_kmerset** out) {
  if (!ctx || !out || (n && !d_keys)) { set_error("bg_kmer_count_keys: null argument"); return BG_EINVAL; }
  if (k < 1 || k > 31) { set_error("bg_kmer_count_keys: kmer size %d outside 1..31", k); return BG_EINVAL; }
  if (min_count < 1) min_count = 1;
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_kmerset> ks(new bg_kmerset);
  ks->ctx = ctx;
  ks->k = k;
  ks->instances = n;
  DevBuf<uint64_t> kb;
  BG_TRY(kb.alloc(ctx, n));
  BG_TRY(kmer_count_keys_impl(ctx, d_keys, kb.p, n, min_count, ks.get()));
  *out = ks.release();
  return BG_OK;
}

extern "C" int bg_kmerset_export(bg_ctx* ctx, const bg_kmerset* ks, uint64_t* d_kmers_out) {
  if (!ctx || !ks) { set_error("bg_kmerset_export: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (ks->size) {
    BG_HIP(hipMemcpyAsync(d_kmers_out, ks->kmers.p, ks->size * 8, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return BG_OK;
}

extern "C" int bg_kmerset_from_kmers(bg_ctx* ctx, const uint64_t* d_kmers, uint64_t n, int k,
                                     const bg_reads* reads_for_flags, bg_kmerset** out) {
  if (!ctx || !out || (n && !d_kmers)) { set_error("bg_kmerset_from_kmers: null argument"); return BG_EINVAL; }
  if (k < 1 || k > 31) { set_error("bg_kmerset_from_kmers: kmer size %d outside 1..31", k); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_kmerset> ks(new bg_kmerset);
  ks->ctx = ctx;
  ks->k = k;
  ks->size = n;
  ks->distinct = n;
  BG_TRY(ks->kmers.alloc(ctx, n));
  BG_TRY(ks->counts.alloc(ctx, n));
  if (n) {
    BG_HIP(hipMemcpyAsync(ks->kmers.p, d_kmers, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemsetAsync(ks->counts.p, 0, n * 4, ctx->stream));
  }
  BG_TRY(kmer_hash_build_impl(ctx, ks.get(), reads_for_flags));
  *out = ks.release();
  return BG_OK;
}

extern "C" uint64_t bg_kmerset_size(const bg_kmerset* ks) { return ks ? ks->size : 0; }
extern "C" uint64_t bg_kmerset_distinct(const bg_kmerset* ks) { return ks ? ks->distinct : 0; }
extern "C" uint64_t bg_kmerset_instances(const bg_kmerset* ks) { return ks ? ks->instances : 0; }
extern "C" void bg_kmerset_free(bg_kmerset* ks) { delete ks; }

extern "C" int bg_kmerset_download(bg_ctx* ctx, const bg_kmerset* ks, uint64_t* h_kmers, uint8_t* h_flags,
                                   uint32_t* h_counts) {
  if (!ctx || !ks) { set_error("bg_kmerset_download: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  uint64_t n = ks->size;
  if (!n) return BG_OK;
  BG_TRY(kmerset_sort(ctx, const_cast<bg_kmerset*>(ks)));
  if (h_kmers) BG_HIP(hipMemcpyAsync(h_kmers, ks->kmers.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (h_counts) BG_HIP(hipMemcpyAsync(h_counts, ks->counts.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (h_flags) {
    DevBuf<uint8_t> fl;
    BG_TRY(fl.alloc(ctx, n));
    k_gather_flags<<<cdiv(n, 256), 256, 0, ctx->stream>>>(ks->kmers.p, n, ks->slots.p, ks->log2cap, ks->k, fl.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipMemcpyAsync(h_flags, fl.p, n, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// ================================================================================================
// C ABI: S3

extern "C" int bg_correct_reads(bg_ctx* ctx, const bg_reads* reads, const bg_kmerset* ks,
                                const bg_correct_params* params, bg_reads** out, bg_correct_stats* stats) {
  if (!ctx || !reads || !ks || !params || !out) { set_error("bg_correct_reads: null argument"); return BG_EINVAL; }
  if (params->max_corrections > uint32_t(kMaxCorrections)) {
    set_error("bg_correct_reads: max_corrections %u > %d", params->max_corrections, kMaxCorrections);
    return BG_EINVAL;
  }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n = reads->n;
  const int W = reads->W, k = ks->k;
  std::unique_ptr<bg_reads> r(new bg_reads);
  r->ctx = ctx;
  r->n = n;
  r->W = W;
  BG_TRY(r->words.alloc(ctx, n * W));
  BG_TRY(r->nmask.alloc(ctx, n * W));
  BG_TRY(r->len.alloc(ctx, n));
  if (n) BG_HIP(hipMemsetAsync(r->nmask.p, 0, n * W * 4, ctx->stream));

  // needed_good_bases per length: unsigned(double(float trim) * double(len)), correct_reads.cpp:174
  uint16_t needed[256];
  for (int L = 0; L < 256; ++L) needed[L] = uint16_t(unsigned(double(params->trim_after_portion) * double(L)));
  DevBuf<uint16_t> d_needed;
  BG_TRY(d_needed.alloc(ctx, 256));
  BG_HIP(hipMemcpyAsync(d_needed.p, needed, sizeof needed, hipMemcpyHostToDevice, ctx->stream));

  DevBuf<uint8_t> need;
  DevBuf<uint64_t> present_own;  // 4 x 64 window bits per read (reads are <= 255 bases)
  BG_TRY(need.alloc(ctx, n));
  // presence bits that came with the count of these very reads (super-k-mer path), else look every window up
  const bool have_present = ks->present.p && ks->present_of == reads && ks->present_n == n && !getenv("BG_SCREEN_LOOKUP");
  if (!have_present) BG_TRY(present_own.alloc(ctx, n * 4));
  const uint64_t* present = have_present ? ks->present.p : present_own.p;
  const int pstride = have_present ? ks->present_stride : 4;
  DevBuf<CorrectCounters> ctr;
  DevBuf<CorrectTotals> tot;
  BG_TRY(ctr.alloc(ctx, 1));
  BG_TRY(tot.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(ctr.p, 0, sizeof(CorrectCounters), ctx->stream));
  BG_HIP(hipMemsetAsync(tot.p, 0, sizeof(CorrectTotals), ctx->stream));
  uint64_t nlist = 0;
  if (n >= (1ull << 32)) { set_error("bg_correct_reads: too many reads for one shard"); return BG_EINVAL; }
  if (n) {
    if (have_present) {
      ctx->tic("correct_screen_bits");
      k_screen_from_present<<<cdiv(n, 256), 256, 0, ctx->stream>>>(reads->words.p, reads->len.p, n, W, k, present,
                                                                    pstride, r->words.p, r->len.p, need.p);
      BG_LAUNCH_CHECK();
      ctx->toc(n * (W * 16 + 32 + 5));
    } else {
      DevBuf<uint32_t> perm;
      BG_TRY(read_locus_order(ctx, reads, 0, n, &perm));
      ctx->tic("correct_screen");
      k_correct_screen<<<cdiv(n, 4), 256, 0, ctx->stream>>>(reads->words.p, reads->nmask.p, reads->len.p, n, W, k,
                                                            ks->slots.p, ks->log2cap, r->words.p, r->len.p, need.p,
                                                            present_own.p, perm.p);
      BG_LAUNCH_CHECK();
      ctx->toc(n * W * 20 + ks->instances * 8);
    }
    DevBuf<uint64_t> list;
    BG_TRY(compact_flagged<uint64_t>(ctx, need.p, n, &list, &nlist));
    if (nlist) {
      ctx->tic("correct_dfs");
      k_correct_dfs<<<cdiv(nlist, 64), 64, size_t(64) * ((W + 1) * 12 + 40), ctx->stream>>>(
          reads->words.p, reads->nmask.p, reads->len.p, list.p, nlist, W, k, ks->slots.p, ks->log2cap,
          int(params->max_corrections), int(params->min_good_run), d_needed.p, r->words.p, r->len.p, ctr.p,
          present, pstride);
      BG_LAUNCH_CHECK();
      ctx->toc(nlist * W * 20);
    }
    k_correct_stats<<<(cdiv(n, 2048) < 256u ? cdiv(n, 2048) : 256u), 256, 0, ctx->stream>>>(reads->len.p, r->len.p, n, k, tot.p);
    BG_LAUNCH_CHECK();
  }
  CorrectCounters hc;
  CorrectTotals ht;
  BG_HIP(hipMemcpyAsync(&hc, ctr.p, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(&ht, tot.p, sizeof ht, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  r->total_bases = ht.corrected_bases;
  r->max_len = uint32_t(ht.max_len);
  if (stats) {
    stats->corrected_reads = ht.corrected_reads;
    stats->corrected_bases = ht.corrected_bases;
    stats->failed = ht.failed;
    stats->modified = hc.modified;
    stats->modified_bases = hc.modified_bases;
    stats->truncated = ht.truncated;
    stats->dropped_bases = ht.dropped_bases;
  }
  *out = r.release();
  return BG_OK;
}

// ================================================================================================
// Routed correction of the sharded build: the kept k-mers stay with their owner ranks; the search of a
// read runs where the read is and fetches what it needs beyond the read's own windows as probes
// (RoutedTable, bg_device_funcs.h).  One step = the search of every pending read with the probes it has;
// reads that missed a look-up hand out new probes (strings whose windows are the probe's k-mers: they
// travel as super-k-mers like the reads themselves, csrc/kmer_skm.hip) and run again next step.

namespace bg {

int skm_presence_from_bits(bg_ctx* ctx, const bg_reads* reads, int k, const uint64_t* d_recs, const uint64_t* d_org,
                           const uint32_t* d_bits, uint64_t n, DevBuf<uint64_t>* present, int* stride);
int skm_probe_answers(bg_ctx* ctx, const uint64_t* d_org, const uint32_t* d_bits, uint64_t n, ProbeDesc* pool, uint64_t base,
                      uint64_t n_probes);

__global__ __launch_bounds__(64) void k_correct_dfs_routed(const uint64_t* __restrict__ words,
                                                           const uint32_t* __restrict__ nmask,
                                                           const uint16_t* __restrict__ len,
                                                           const uint64_t* __restrict__ list,
                                                           const uint32_t* __restrict__ pend, uint64_t n_pend, int W, int k,
                                                           int max_corrections, int min_good_run,
                                                           const uint16_t* __restrict__ needed_by_len,
                                                           uint64_t* __restrict__ out_words, uint16_t* __restrict__ out_len,
                                                           CorrectCounters* __restrict__ ctr,
                                                           const uint64_t* __restrict__ present, int pstride,
                                                           ProbeDesc* __restrict__ pool, uint32_t* __restrict__ head,
                                                           unsigned long long* __restrict__ cursor, uint64_t round_base,
                                                           uint32_t* __restrict__ desc_slot, uint8_t* __restrict__ still) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool valid = t < n_pend;
  const uint32_t slot = valid ? pend[t] : 0;
  RoutedTable rt;
  rt.pool = pool;
  rt.head = 0;
  rt.k = k;
  rt.n_new = 0;
  rt.missed = false;
  int ncorr = 0, clen = 0;
  extern __shared__ uint64_t dfs_smem[];  // as in k_correct_dfs: the read, its presence words and its N masks
  uint64_t* const own_w = dfs_smem;
  uint64_t* const own_p = own_w + 64 * (W + 1);
  uint32_t* const own_n = reinterpret_cast<uint32_t*>(own_p + 64 * 5);
  if (valid) {
    const uint64_t i = list[slot];
    rt.head = head[slot];
    const int L = len[i];
    uint64_t* const my_w = own_w + threadIdx.x * (W + 1);
    uint32_t* const my_n = own_n + threadIdx.x * (W + 1);
    uint64_t* const my_p = own_p + threadIdx.x * 5;
    for (int w = 0; w < W; ++w) { my_w[w] = words[i * W + w]; my_n[w] = nmask[i * W + w]; }
    for (int w = 0; w < pstride && w < 4; ++w) my_p[w] = present[i * pstride + w];
    uint64_t ow[kMaxWords];
    clen = frc_correct_read(rt, my_w, my_n, L, W, k, max_corrections, min_good_run,
                            int(needed_by_len[L]), ow, &ncorr, my_p);
    if (!rt.missed) {
      for (int w = 0; w < W; ++w) out_words[i * W + w] = clen ? ow[w] : 0;
      out_len[i] = uint16_t(clen);
      still[t] = 0;
    }
  }
  // The counters and the probe pool's cursor are single addresses: one atomic per wave each instead of one per read
  // (in the first round every read of the list asks for pool places).
  const int lane = threadIdx.x & 63;
  const uint64_t below = (1ull << lane) - 1;
  const bool mod = valid && !rt.missed && clen && ncorr;
  const unsigned long long wave_mod = __popcll(__ballot(mod));
  unsigned long long wave_bases = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) wave_bases += (unsigned long long)__popcll(__ballot(mod && ((ncorr >> b) & 1))) << b;  // ncorr <= 8
  const bool miss = valid && rt.missed;
  uint32_t before = 0, wave_new = 0;  // pool places of the lanes below this one, of the whole wave
#pragma unroll
  for (int b = 0; b < 3; ++b) {  // n_new <= kMaxNewProbes = 4
    const uint64_t mb = __ballot(miss && ((rt.n_new >> b) & 1));
    before += uint32_t(__popcll(mb & below)) << b;
    wave_new += uint32_t(__popcll(mb)) << b;
  }
  unsigned long long wave_base = 0;
  if (lane == 0) {
    if (wave_mod) {
      atomicAdd(&ctr->modified, wave_mod);
      atomicAdd(&ctr->modified_bases, wave_bases);
    }
    if (wave_new) wave_base = atomicAdd(cursor, (unsigned long long)wave_new);
  }
  wave_base = __shfl(wave_base, 0, 64);
  if (!miss) return;
  // rt.n_new >= 1: the first miss of a run is never one of this run's own new probes
  const uint64_t base = wave_base + before;
  uint32_t prev = rt.head;
  for (int q = 0; q < rt.n_new; ++q) {
    ProbeDesc d;
    d.q = rt.newq[q];
    d.mask[0] = d.mask[1] = d.mask[2] = d.mask[3] = 0;
    d.meta = rt.newmeta[q];  // the run is filled in by k_probe_strings
    d.next = prev;
    pool[base + q] = d;
    desc_slot[base + q - round_base] = slot;
    prev = uint32_t(base + q);
  }
  head[slot] = prev;
  still[t] = 1;
}

// the 4 strings of every new probe (W = 2 words each) and the probe's run
__global__ void k_probe_strings(const uint64_t* __restrict__ words, const uint32_t* __restrict__ nmask,
                                const uint16_t* __restrict__ len, const uint64_t* __restrict__ list, int W, int k,
                                const uint64_t* __restrict__ present, int pstride, ProbeDesc* __restrict__ pool,
                                uint64_t round_base, uint64_t n_new, const uint32_t* __restrict__ desc_slot,
                                uint64_t* __restrict__ pwords, uint16_t* __restrict__ plen) {
  const uint64_t d = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (d >= n_new) return;
  const uint64_t i = list[desc_slot[d]];
  ProbeDesc& pd = pool[round_base + d];
  const int L = len[i];
  const int anchor = frc_anchor(present + i * pstride, L, k);  // >= 0: the search got as far as a look-up
  const int view = int((pd.meta >> 16) & 1u), pos = int(pd.meta & 255u);
  const ReadView in = frc_view(words + i * W, nmask + i * W, L, W, k, present + i * pstride, anchor, view);
  const int run = probe_run(in, pos, k);
  pd.meta |= uint32_t(run) << 8;
  for (int b = 0; b < 4; ++b) {
    uint64_t s[2];
    probe_string(in, pd.q, pos, b, run, k, s);
    pwords[(4 * d + b) * 2] = s[0];
    pwords[(4 * d + b) * 2 + 1] = s[1];
    plen[4 * d + b] = uint16_t(k + run);
  }
}

__global__ void k_iota_u32_corr(uint32_t* __restrict__ dst, uint64_t n) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = uint32_t(i);
}
__global__ void k_fill_u32_corr(uint32_t* __restrict__ dst, uint64_t n, uint32_t v) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}
__global__ void k_gather_u32_corr(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n,
                                  uint32_t* __restrict__ dst) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

}  // namespace bg

struct bg_correction {
  bg_ctx* ctx = nullptr;
  const bg_reads* reads = nullptr;
  int k = 0;
  bg_correct_params prm{};
  std::unique_ptr<bg_reads> out;
  bg::DevBuf<uint64_t> present;
  int pstride = 1;
  bg::DevBuf<uint16_t> needed;
  bg::DevBuf<uint64_t> list;       // reads that need the search
  uint64_t nlist = 0;
  bg::DevBuf<uint32_t> head;       // newest probe of every list slot
  bg::DevBuf<uint32_t> pend;       // list slots still searching
  uint64_t n_pend = 0;
  bg::DevBuf<bg::ProbeDesc> pool;
  uint64_t pool_used = 0;
  bg::DevBuf<uint32_t> desc_slot;  // list slot of every probe of the round in flight
  uint64_t round_base = 0, round_new = 0;
  bg::DevBuf<bg::CorrectCounters> ctr;
  uint64_t steps = 0, probes = 0;
};

extern "C" int bg_corr_begin(bg_ctx* ctx, const bg_reads* reads, int k, const bg_correct_params* params,
                             const uint64_t* d_recs, const uint64_t* d_org, const uint32_t* d_bits, uint64_t n_recs,
                             bg_correction** out) {
  if (!ctx || !reads || !params || !out || (n_recs && (!d_recs || !d_org || !d_bits))) { set_error("bg_corr_begin: null argument"); return BG_EINVAL; }
  if (k < 1 || k > 31) { set_error("bg_corr_begin: kmer size %d outside 1..31", k); return BG_EINVAL; }
  if (params->max_corrections > uint32_t(kMaxCorrections)) { set_error("bg_corr_begin: max_corrections %u > %d", params->max_corrections, kMaxCorrections); return BG_EINVAL; }
  const uint64_t n = reads->n;
  if (n >= (1ull << 32)) { set_error("bg_corr_begin: too many reads for one shard"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_correction> c(new bg_correction);
  c->ctx = ctx;
  c->reads = reads;
  c->k = k;
  c->prm = *params;
  const int W = reads->W;
  c->out.reset(new bg_reads);
  bg_reads* r = c->out.get();
  r->ctx = ctx;
  r->n = n;
  r->W = W;
  BG_TRY(r->words.alloc(ctx, n * W));
  BG_TRY(r->nmask.alloc(ctx, n * W));
  BG_TRY(r->len.alloc(ctx, n));
  if (n) BG_HIP(hipMemsetAsync(r->nmask.p, 0, n * W * 4, ctx->stream));
  uint16_t needed[256];
  for (int L = 0; L < 256; ++L) needed[L] = uint16_t(unsigned(double(params->trim_after_portion) * double(L)));
  BG_TRY(c->needed.alloc(ctx, 256));
  BG_HIP(hipMemcpyAsync(c->needed.p, needed, sizeof needed, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));  // `needed` is on this frame
  BG_TRY(skm_presence_from_bits(ctx, reads, k, d_recs, d_org, d_bits, n_recs, &c->present, &c->pstride));
  BG_TRY(c->ctr.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(c->ctr.p, 0, sizeof(CorrectCounters), ctx->stream));
  DevBuf<uint8_t> need;
  BG_TRY(need.alloc(ctx, n));
  if (n) {
    ctx->tic("correct_screen_bits");
    k_screen_from_present<<<cdiv(n, 256), 256, 0, ctx->stream>>>(reads->words.p, reads->len.p, n, W, k, c->present.p, c->pstride,
                                                                  r->words.p, r->len.p, need.p);
    BG_LAUNCH_CHECK();
    ctx->toc(n * (W * 16 + 32 + 5));
  }
  BG_TRY(compact_flagged<uint64_t>(ctx, need.p, n, &c->list, &c->nlist));
  BG_TRY(c->head.alloc(ctx, c->nlist));
  BG_TRY(c->pend.alloc(ctx, c->nlist));
  if (c->nlist) {
    k_fill_u32_corr<<<cdiv(c->nlist, 256), 256, 0, ctx->stream>>>(c->head.p, c->nlist, kNoProbe);
    BG_LAUNCH_CHECK();
    k_iota_u32_corr<<<cdiv(c->nlist, 256), 256, 0, ctx->stream>>>(c->pend.p, c->nlist);
    BG_LAUNCH_CHECK();
  }
  c->n_pend = c->nlist;
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *out = c.release();
  return BG_OK;
}

extern "C" void bg_corr_free(bg_correction* c) { delete c; }

// One step.  *n_pending_out = reads that still miss look-ups after it; *probes_out = the strings of their new
// probes as a reads object (4 per probe; NULL when there are none) — the caller ships their super-k-mers to
// the owners (bg_skm_extract_by_owner), gets the windows' presence back (bg_skm_lookup on the owner) and hands
// it to bg_corr_answers before the next step.  Ranks without pending reads still take part in the exchange.
extern "C" int bg_corr_step(bg_ctx* ctx, bg_correction* c, uint64_t* n_pending_out, bg_reads** probes_out) {
  if (!ctx || !c || !n_pending_out || !probes_out) { set_error("bg_corr_step: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  *probes_out = nullptr;
  *n_pending_out = 0;
  c->round_new = 0;
  if (!c->n_pend) return BG_OK;
  const bg_reads* reads = c->reads;
  const int W = reads->W, k = c->k;
  const uint64_t room = c->pool_used + uint64_t(kMaxNewProbes) * c->n_pend;
  if (room >= (1ull << 32)) { set_error("bg_corr_step: probe pool exceeds 2^32 entries"); return BG_ENOMEM; }
  if (c->pool.n < room) {
    DevBuf<ProbeDesc> np;
    BG_TRY(np.alloc(ctx, room + room / 2));
    if (c->pool_used) BG_HIP(hipMemcpyAsync(np.p, c->pool.p, c->pool_used * sizeof(ProbeDesc), hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    c->pool = std::move(np);
  }
  BG_TRY(c->desc_slot.alloc(ctx, uint64_t(kMaxNewProbes) * c->n_pend));
  DevBuf<unsigned long long> cursor;
  DevBuf<uint8_t> still;
  BG_TRY(cursor.alloc(ctx, 1));
  BG_TRY(still.alloc(ctx, c->n_pend));
  unsigned long long h_cur = c->pool_used;
  BG_HIP(hipMemcpyAsync(cursor.p, &h_cur, 8, hipMemcpyHostToDevice, ctx->stream));
  ctx->tic("correct_dfs");
  k_correct_dfs_routed<<<cdiv(c->n_pend, 64), 64, size_t(64) * ((W + 1) * 12 + 40), ctx->stream>>>(
      reads->words.p, reads->nmask.p, reads->len.p, c->list.p, c->pend.p, c->n_pend, W, k, int(c->prm.max_corrections),
      int(c->prm.min_good_run), c->needed.p, c->out->words.p, c->out->len.p, c->ctr.p, c->present.p, c->pstride, c->pool.p,
      c->head.p, cursor.p, c->pool_used, c->desc_slot.p, still.p);
  BG_LAUNCH_CHECK();
  ctx->toc(c->n_pend * W * 20);
  BG_HIP(hipMemcpyAsync(&h_cur, cursor.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  DevBuf<uint32_t> keep_t, np;
  uint64_t m = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, still.p, c->n_pend, &keep_t, &m));  // synchronises
  BG_TRY(np.alloc(ctx, m));
  if (m) {
    k_gather_u32_corr<<<cdiv(m, 256), 256, 0, ctx->stream>>>(c->pend.p, keep_t.p, m, np.p);
    BG_LAUNCH_CHECK();
  }
  c->round_base = c->pool_used;
  c->round_new = h_cur - c->pool_used;
  c->pool_used = h_cur;
  c->steps += 1;
  c->probes += c->round_new;
  if (c->round_new) {
    std::unique_ptr<bg_reads> pr(new bg_reads);
    pr->ctx = ctx;
    pr->n = 4 * c->round_new;
    pr->W = 2;
    pr->max_len = uint32_t(2 * k - 1);
    BG_TRY(pr->words.alloc(ctx, pr->n * 2));
    BG_TRY(pr->nmask.alloc(ctx, pr->n * 2));
    BG_TRY(pr->len.alloc(ctx, pr->n));
    BG_HIP(hipMemsetAsync(pr->nmask.p, 0, pr->n * 2 * 4, ctx->stream));
    k_probe_strings<<<cdiv(c->round_new, 256), 256, 0, ctx->stream>>>(reads->words.p, reads->nmask.p, reads->len.p, c->list.p, W, k,
                                                                    c->present.p, c->pstride, c->pool.p, c->round_base,
                                                                    c->round_new, c->desc_slot.p, pr->words.p, pr->len.p);
    BG_LAUNCH_CHECK();
    *probes_out = pr.release();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  c->pend = std::move(np);
  c->n_pend = m;
  *n_pending_out = m;
  return BG_OK;
}

// d_org / d_bits: origins (bg_skm_extract_by_owner on the probes of the last step) and the owners' answers in
// the same order.
extern "C" int bg_corr_answers(bg_ctx* ctx, bg_correction* c, const uint64_t* d_org, const uint32_t* d_bits, uint64_t n) {
  if (!ctx || !c || (n && (!d_org || !d_bits))) { set_error("bg_corr_answers: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  return skm_probe_answers(ctx, d_org, d_bits, n, c->pool.p, c->round_base, c->round_new);
}

extern "C" int bg_corr_finish(bg_ctx* ctx, bg_correction* c, bg_reads** out, bg_correct_stats* stats) {
  if (!ctx || !c || !out) { set_error("bg_corr_finish: null argument"); return BG_EINVAL; }
  if (c->n_pend) { set_error("bg_corr_finish: %llu reads still miss look-ups", (unsigned long long)c->n_pend); return BG_EINVAL; }
  if (!c->out) { set_error("bg_corr_finish: already finished"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n = c->reads->n;
  DevBuf<CorrectTotals> tot;
  BG_TRY(tot.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(tot.p, 0, sizeof(CorrectTotals), ctx->stream));
  if (n) {
    k_correct_stats<<<(cdiv(n, 2048) < 256u ? cdiv(n, 2048) : 256u), 256, 0, ctx->stream>>>(c->reads->len.p, c->out->len.p, n, c->k, tot.p);
    BG_LAUNCH_CHECK();
  }
  CorrectCounters hc;
  CorrectTotals ht;
  BG_HIP(hipMemcpyAsync(&hc, c->ctr.p, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipMemcpyAsync(&ht, tot.p, sizeof ht, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  c->out->total_bases = ht.corrected_bases;
  c->out->max_len = uint32_t(ht.max_len);
  if (stats) {
    stats->corrected_reads = ht.corrected_reads;
    stats->corrected_bases = ht.corrected_bases;
    stats->failed = ht.failed;
    stats->modified = hc.modified;
    stats->modified_bases = hc.modified_bases;
    stats->truncated = ht.truncated;
    stats->dropped_bases = ht.dropped_bases;
  }
  *out = c->out.release();
  return BG_OK;
}

extern "C" uint64_t bg_corr_probes(const bg_correction* c) { return c ? c->probes : 0; }
