// build_seqset.hip — S4 (suffix records: exact-cover emission, sort, prefix-dedup) and S5 (seqset
// arrays: sizes/shared/prev_{ACGT} rank vectors/fixed) for gfx950.
//
// Reference rows: D1-D5 repo_seq.h:85-183, part_repo.cpp:32-126, expand.cpp:180-380,
// 581-661 — replaced by their net semantics, row E: the entry set is the prefix-free maxima of
// every suffix of every corrected read and of its reverse complement; F1 builder.cpp:8-164;
// F2 builder.cpp:207-263, seqset.cpp:19-44,113-129, bitcount.cpp:84-124.
//
// Emission rule: for an oriented read s, suffix j>=1 and every later one are skipped
// as soon as s[j:] is a prefix of some oriented read.  Covered suffixes are prefixes of suffixes of
// the covering read at strictly smaller offsets, so the prefix-free maxima are unchanged and no
// expansion passes (expand.cpp:581) are needed.
#include "bg_device_funcs.h"
#include "bg_internal.h"
#include "compact.h"

#include <stdlib.h>

namespace bg {

constexpr int kSortSkipBits = 32;  // the sort key keeps the leading 16 bases of a record (see k_first_words)


// ------------------------------------------------------------------------------------------------
// oriented reads (+ their lengths as bytes and their sort keys: leading 16 bases | oriented id)

// k0: one sort key a read (leading bases | read) for sort_records_impl, or -- items_K > 0 -- two words a read: the 16-byte item
// of items_cover_order (first word, read << 24 | length << 8; a read shorter than items_K gets a scattered first word)
__global__ void k_oriented_words(const uint64_t* __restrict__ words, const uint16_t* __restrict__ len, uint64_t n,
                                 int W, uint64_t* __restrict__ owords, uint8_t* __restrict__ olen,
                                 uint64_t* __restrict__ k0, int items_K) {
  uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= 2 * n) return;
  uint64_t i = o >> 1;
  int L = len[i];
  uint64_t tmp[kMaxWords];
  oriented_words(words + i * W, L, W, (o & 1) != 0, tmp);
  for (int w = 0; w < W; ++w) owords[o * W + w] = L ? tmp[w] : 0;
  olen[o] = uint8_t(L);
  if (items_K > 0) {
    k0[2 * o] = L >= items_K ? tmp[0] : cover_item_scatter(o);
    k0[2 * o + 1] = (o << 24) | (uint64_t(L) << 8);
  } else {
    k0[o] = ((L ? tmp[0] : 0ull) & ~((1ull << kSortSkipBits) - 1)) | o;
  }
}

// reads at least K bases long, in sorted order (shorter ones cannot act as a cover)
__global__ void k_flag_long_enough(const uint32_t* __restrict__ sidx, const uint8_t* __restrict__ olen, uint64_t n, int K,
                                   uint8_t* __restrict__ flags) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = olen[sidx[i]] >= K ? 1 : 0;
}
__global__ void k_cover_keys(const uint32_t* __restrict__ sidx, const uint32_t* __restrict__ keep, uint64_t m,
                             const uint64_t* __restrict__ owords, int W, int K, uint64_t* __restrict__ skeys,
                             uint32_t* __restrict__ soid) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t o = sidx[keep[i]];
  soid[i] = o;
  skeys[i] = owords[uint64_t(o) * W] >> (64 - 2 * K);
}

__global__ void k_flag_key_heads(const uint64_t* __restrict__ keys, uint64_t n, uint8_t* __restrict__ flags) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t k = keys[i];
  flags[i] = (k != ~0ull && (i == 0 || k != keys[i - 1])) ? 1 : 0;
}

// cover index: hash from first-K-mer to the start of its run in the sorted (key, oriented id) list, one word per slot
// (cover_slot, bg_device_funcs.h).  A run of one candidate (the usual case) stores the candidate's oriented id and length
// in the slot itself: a hit goes from the slot straight to the candidate's words.
__global__ void k_cover_insert(const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ soid,
                               const uint8_t* __restrict__ olen, uint64_t n_o, const uint32_t* __restrict__ head_pos,
                               uint64_t nheads, uint64_t* __restrict__ slots, int log2cap) {
  uint64_t r = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= nheads) return;
  const uint32_t pos = head_pos[r];
  const uint64_t key = skeys[pos];
  const uint64_t run = (r + 1 < nheads ? uint64_t(head_pos[r + 1]) : n_o) - pos;
  uint64_t word;
  if (run == 1) {
    const uint32_t c = soid[pos];
    word = cover_slot(key, olen[c], true, c);
  } else {
    word = cover_slot(key, uint32_t(run < 255 ? run : 255), false, pos);
  }
  uint64_t mask = (1ull << log2cap) - 1;
  uint64_t s = hash_slot(key, log2cap);
  const int off = hash_slot_offset(key, log2cap);
  for (;;) {
    for (int u = 0; u < kBucketSlots; ++u) {
      uint64_t t = s + ((off + u) & (kBucketSlots - 1));
      if (slots[t] != kEmptySlot) continue;
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&slots[t]),
                                         (unsigned long long)kEmptySlot, (unsigned long long)word);
      if (old == kEmptySlot) return;
    }
    s = (s + kBucketSlots) & mask;
  }
}

// One suffix look-up per lane and round; a lane whose read is finished takes the next read of its wave's
// chunk at once (the reads need 1..L rounds each — without the refill a wave runs as long as its slowest
// read with most lanes idle).
constexpr int kCoverChunk = 1024;  // oriented reads per wave

// The reads of a wave's chunk come in through LDS kCoverBatch at a time, as one run of consecutive words (a lane fetching its own
// 40-byte row asked for three sectors of its own, one more for its length byte, and wrote its count as a fifth request:
// half of the ten requests a read cost, in a kernel that runs at the rate requests are served), and the chunk's counts
// leave as one run at the end.  Dynamic LDS: 256 own rows of `stride` words (W, made odd against bank conflicts), then per
// wave kCoverBatch staged rows of W words, their lengths and the chunk's counts as bytes.
__host__ __device__ inline int cover_row_stride(int W) { return W | 1; }
constexpr int kCoverBatch = 32;  // reads staged at a time per wave
__host__ __device__ inline size_t cover_count_lds(int W) {
  return size_t(256) * cover_row_stride(W) * 8 + size_t(4) * (kCoverBatch * W * 8 + kCoverBatch + kCoverChunk);
}
__global__ __launch_bounds__(256, 8) __attribute__((amdgpu_num_sgpr(80))) void k_cover_count(CoverIndex ci, const uint64_t* __restrict__ owords,
                                                     uint64_t o_base, uint64_t n_emit, int W,
                                                     uint32_t* __restrict__ cnt) {
  extern __shared__ uint64_t cc_lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t chunk0 = (uint64_t(blockIdx.x) * 4 + wv) * kCoverChunk;
  if (chunk0 >= n_emit) return;  // (wave-uniform)
  const uint32_t end = uint32_t(n_emit - chunk0 < uint64_t(kCoverChunk) ? n_emit - chunk0 : uint64_t(kCoverChunk));  // reads of this chunk
  const int stride = cover_row_stride(W);
  uint64_t* const s = cc_lds + threadIdx.x * stride;                   // the lane's own read
  uint64_t* const st = cc_lds + 256 * stride + wv * kCoverBatch * W;   // the wave's staged batch
  uint8_t* const bytes = reinterpret_cast<uint8_t*>(cc_lds + 256 * stride + 4 * kCoverBatch * W);
  uint8_t* const st_len = bytes + wv * kCoverBatch;
  uint8_t* const counts = bytes + 4 * kCoverBatch + wv * kCoverChunk;
  const uint64_t* const chunk_words = owords + (o_base + chunk0) * uint64_t(W);  // (16-byte aligned: o_base even, chunk0 and the batches' starts even)
  uint32_t next = 0, sbase = ~0u, t = 0;  // reads of the chunk by their number in it
  bool active = false;
  int L = 0, j = 0, c = 0;
  for (;;) {
    const uint64_t need = __ballot(!active);
    if (need) {
      const uint32_t mine = next + uint32_t(__popcll(need & lane_lt));
      bool want = !active && mine < end;
      next += uint32_t(__popcll(need));
      for (;;) {
        const uint64_t wm = __ballot(want);
        if (!wm) break;
        // the batch that holds the smallest wanted read (the wanting lanes' reads ascend with the lane)
        const uint32_t bb = uint32_t(__shfl(int(mine), __ffsll((unsigned long long)wm) - 1, 64)) & ~uint32_t(kCoverBatch - 1);
        if (bb != sbase) {
          const uint32_t rows = end - bb < uint32_t(kCoverBatch) ? end - bb : uint32_t(kCoverBatch);
          const uint32_t nw = rows * uint32_t(W);
          const uint64_t* src = chunk_words + uint64_t(bb) * uint64_t(W);
          const ulonglong2* src2 = reinterpret_cast<const ulonglong2*>(src);
          for (uint32_t x = uint32_t(lane); x * 2 + 1 < nw; x += 64) {
            const ulonglong2 v = src2[x];
            st[2 * x] = v.x;
            st[2 * x + 1] = v.y;
          }
          if ((nw & 1u) && lane == 0) st[nw - 1] = src[nw - 1];
          if (uint32_t(lane) < rows) st_len[lane] = ci.olen[o_base + chunk0 + bb + uint32_t(lane)];
          sbase = bb;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (want && mine < sbase + uint32_t(kCoverBatch)) {
          t = mine;
          const int q = int(mine - sbase);
          L = st_len[q];
          for (int w = 0; w < W; ++w) s[w] = st[q * W + w];
          j = 1;
          c = 1;
          if (L <= 1) counts[t] = uint8_t(L);  // dropped read: nothing; one base: itself
          else active = true;
          want = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the stage is free again once every lane that wanted one of its rows has it)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (!__ballot(active)) {
      if (next >= end) break;
      continue;
    }
    if (active) {
      const int rem = L - j;
      bool fin;
      if (rem < ci.K) { c += rem; fin = true; }
      else if (cover_probe(ci, owords, s, L, j, W)) fin = true;
      else { ++c; ++j; fin = j >= L; }
      if (fin) { counts[t] = uint8_t(c); active = false; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (uint32_t x = uint32_t(lane); x < end; x += 64) cnt[chunk0 + x] = counts[x];
}

// The same with one memory round per loop pass whatever a lane is doing.  k_cover_count's pass is a whole look-up: bucket,
// then -- for the lanes that hit -- the candidate's words behind it, and a wave's pass lasts as long as its slowest lane's
// chain of dependent loads (nearly every pass has a lane that hits).  Here a lane is in one of two states, PROBE (fetch the
// bucket of suffix j's K-mer) or VERIFY (fetch the one candidate the slot named): both fetch 64 bytes from an address of the
// lane's own, so every pass is ONE round trip, and the 5.5 misses + 1 hit of an average read are 7 of them instead of 5.5 +
// 2 behind each other.  What is rare goes the old way, lane by lane: runs of several candidates (cover_check_slot), slots
// whose fingerprint belongs to another K-mer (cover_probe from the chain's start).  owords holds 8 spare words at its end
// (a candidate's 64 bytes may reach past the last read).  chr20: 5.05 ms with the runs parked (5.4 without) against 5.9 (and 6.1 before the staged reads); 90
// registers, five waves per SIMD -- with the rare paths stubbed out (71 registers, seven waves) it took the same 5.3 ms, and
// held to 64 registers by a waves-per-EU hint it spilled inside the loop and took 12.3.
struct __attribute__((aligned(8))) CoverPair { uint64_t x, y; };
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_cover_count_sm(CoverIndex ci, const uint64_t* __restrict__ owords,
                                                     uint64_t o_base, uint64_t n_emit, int W,
                                                     uint32_t* __restrict__ cnt) {
  extern __shared__ uint64_t cc_lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t chunk0 = (uint64_t(blockIdx.x) * 4 + wv) * kCoverChunk;
  if (chunk0 >= n_emit) return;  // (wave-uniform)
  const uint32_t end = uint32_t(n_emit - chunk0 < uint64_t(kCoverChunk) ? n_emit - chunk0 : uint64_t(kCoverChunk));
  const int stride = cover_row_stride(W);
  uint64_t* const s = cc_lds + threadIdx.x * stride;
  uint64_t* const st = cc_lds + 256 * stride + wv * kCoverBatch * W;
  uint8_t* const bytes = reinterpret_cast<uint8_t*>(cc_lds + 256 * stride + 4 * kCoverBatch * W);
  uint8_t* const st_len = bytes + wv * kCoverBatch;
  uint8_t* const counts = bytes + 4 * kCoverBatch + wv * kCoverChunk;
  const uint64_t* const chunk_words = owords + (o_base + chunk0) * uint64_t(W);
  const uint64_t slot_mask = (1ull << ci.log2cap) - 1;
  uint32_t next = 0, sbase = ~0u, t = 0;
  bool active = false, verify = false, fresh = true, parked = false;
  int L = 0, j = 0, c = 0;
  uint32_t vc = 0, vlen = 0;
  uint64_t key = 0, pb = 0, psv = 0;
  for (;;) {
    const uint64_t need = __ballot(!active);
    if (need) {
      const uint32_t mine = next + uint32_t(__popcll(need & lane_lt));
      bool want = !active && mine < end;
      next += uint32_t(__popcll(need));
      for (;;) {
        const uint64_t wm = __ballot(want);
        if (!wm) break;
        const uint32_t bb = uint32_t(__shfl(int(mine), __ffsll((unsigned long long)wm) - 1, 64)) & ~uint32_t(kCoverBatch - 1);
        if (bb != sbase) {
          const uint32_t rows = end - bb < uint32_t(kCoverBatch) ? end - bb : uint32_t(kCoverBatch);
          const uint32_t nw = rows * uint32_t(W);
          const uint64_t* src = chunk_words + uint64_t(bb) * uint64_t(W);
          const ulonglong2* src2 = reinterpret_cast<const ulonglong2*>(src);
          for (uint32_t x = uint32_t(lane); x * 2 + 1 < nw; x += 64) {
            const ulonglong2 v = src2[x];
            st[2 * x] = v.x;
            st[2 * x + 1] = v.y;
          }
          if ((nw & 1u) && lane == 0) st[nw - 1] = src[nw - 1];
          if (uint32_t(lane) < rows) st_len[lane] = ci.olen[o_base + chunk0 + bb + uint32_t(lane)];
          sbase = bb;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (want && mine < sbase + uint32_t(kCoverBatch)) {
          t = mine;
          const int q = int(mine - sbase);
          L = st_len[q];
          for (int w = 0; w < W; ++w) s[w] = st[q * W + w];
          j = 1;
          c = 1;
          verify = false;
          fresh = true;
          parked = false;
          if (L <= 1) counts[t] = uint8_t(L);
          else active = true;
          want = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (!__ballot(active)) {
      if (next >= end) break;
      continue;
    }
    // this pass's 64 bytes
    int res = -1;  // 1: suffix j is covered, 0: it is not, -1: not known yet
    const int rem = L - j;
    const CoverPair* p = nullptr;
    // Lanes whose slot named a RUN of candidates (mates, duplicates, a repeat family) wait, parked, until a quarter of the
    // wave is -- then they walk their runs together (cover_check_slot: a chain of dependent loads per lane).  Taken as they
    // came, one or two lanes did that in nearly every pass and the other sixty waited: a genome with 7.5 % of it in 5000
    // copies of one element took 10.0 ms here against 5.9 in k_cover_count.
    {
      const uint64_t pk = __ballot(active && parked);
      if (pk && (__popcll(pk) >= 16 || !__ballot(active && !parked))) {
        if (active && parked) {
          const int r = cover_check_slot(ci, owords, s, j, rem, W, key, psv);
          res = r >= 0 ? r : (cover_probe(ci, owords, s, L, j, W) ? 1 : 0);
          parked = false;
        }
      }
    }
    if (active && !parked && res < 0) {
      if (verify) {
        p = reinterpret_cast<const CoverPair*>(owords + uint64_t(vc) * uint64_t(W));
      } else if (rem < ci.K) {
        c += rem - 1;  // (suffixes shorter than K are never looked up: all emitted; the pass below adds the last one)
        j = L - 1;
        res = 0;
      } else {
        if (fresh) {
          key = window32(s, W, j) >> (64 - 2 * ci.K);
          pb = hash_slot(key, ci.log2cap);
          fresh = false;
        }
        p = reinterpret_cast<const CoverPair*>(ci.slots + pb);
      }
    }
    uint64_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = 0;
    if (p) {
      const CoverPair a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
      v[0] = a0.x; v[1] = a0.y; v[2] = a1.x; v[3] = a1.y; v[4] = a2.x; v[5] = a2.y; v[6] = a3.x; v[7] = a3.y;
    }
    if (p && verify) {
      // the candidate's words against the suffix (suffix_is_prefix with the words in registers)
      const int full = rem >> 5, tail = rem & 31;
      const uint64_t tail_mask = tail ? ~0ull << (2 * (32 - tail)) : ~0ull;
      bool same = int(vlen) >= rem;
#pragma unroll
      for (int w = 0; w < kMaxWords; ++w) {
        if (w < W && (w < full || (w == full && tail))) {
          const uint64_t m = w == full ? tail_mask : ~0ull;
          same = same && ((window32(s, W, j + 32 * w) & m) == (v[w] & m));
        }
      }
      verify = false;
      if (same) res = 1;
      else if ((v[0] >> (64 - 2 * ci.K)) == key) res = 0;
      else res = cover_probe(ci, owords, s, L, j, W) ? 1 : 0;  // the slot was another K-mer's: the chain, slot by slot
    } else if (p) {
      const uint64_t fp = cover_fingerprint(key);
      uint32_t hit = 0, empty = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        hit |= uint32_t((v[u] >> 40) == fp && v[u] != kEmptySlot) << u;
        empty |= uint32_t(v[u] == kEmptySlot) << u;
      }
      if (hit) {
        uint64_t sv = 0;
#pragma unroll
        for (int u = 7; u >= 0; --u)
          if ((hit >> u) & 1u) sv = v[u];
        if (uint32_t(sv) & kCoverSingle) {
          vc = uint32_t(sv) & ~kCoverSingle;
          vlen = uint32_t(sv >> 32) & 0xffu;
          verify = true;  // the next pass fetches the candidate
        } else {
          psv = sv;
          parked = true;
        }
      } else if (empty) {
        res = 0;
      } else {
        pb = (pb + kBucketSlots) & slot_mask;  // a full bucket without the K-mer: on along the chain
      }
    }
    if (active && res >= 0) {
      bool fin = res == 1;
      if (!fin) { ++c; ++j; fresh = true; fin = j >= L; }
      if (fin) { counts[t] = uint8_t(c); active = false; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (uint32_t x = uint32_t(lane); x < end; x += 64) cnt[chunk0 + x] = counts[x];
}

// records: rows of RS words, rec[r*RS + w] = zero padded key words (w < W) and, when RS = W + 1, the length in word W;
// rlen[r] = the length as a byte when the rows do not carry it (RS = W)
// One wave expands 64 oriented reads: its records are contiguous in the output, so the lanes take the wave's
// RECORDS in turn (lane = consecutive record), find the read a record belongs to by a 6-step search over the
// 64 record offsets held one per lane, and the wave writes them out fully coalesced through LDS — instead of one
// lane looping over all the records of its read.
__global__ __launch_bounds__(256) void k_emit_records(const uint64_t* __restrict__ owords,
                                                      const uint16_t* __restrict__ len,
                                                      const uint32_t* __restrict__ cnt,
                                                      const uint64_t* __restrict__ roff, uint64_t o_base,
                                                      uint64_t n_emit, int W, int RS, uint64_t* __restrict__ rec,
                                                      uint8_t* __restrict__ rlen, uint64_t* __restrict__ k0) {
  const int lane = threadIdx.x & 63;
  const uint64_t t0 = (uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 64;
  if (t0 >= n_emit) return;
  const uint64_t t = t0 + lane;
  const bool live = t < n_emit;
  const uint64_t my_off = live ? roff[t] : ~0ull;  // ascending over the lanes (exclusive scan), ~0 past the end
  const uint32_t my_cnt = live ? cnt[t] : 0u;
  const int my_len = live ? int(len[(o_base + t) >> 1]) : 0;
  const uint64_t r0 = __shfl(my_off, 0, 64);
  uint64_t r1 = my_off + my_cnt;  // end of the wave's records = max over live lanes
  {
    uint64_t e = live ? r1 : 0;
#pragma unroll
    for (int d = 32; d; d >>= 1) {
      uint64_t o = __shfl_xor(e, d, 64);
      e = o > e ? o : e;
    }
    r1 = e;
  }
  // A lane builds one record a round — the read it belongs to found by a 6-step search over the 64 record offsets
  // held one per lane, its words cut out of the read — into the wave's LDS rows, which the wave then writes out as
  // one run of consecutive words.  (Lanes walking the output words did the search and the division once per word:
  // 3.3 ms for chr20's 154 M records.)
  // RS words per row: the W key words and, when RS = W + 1, the record's length behind them (k_gather_dedup_build gets
  // it with the row instead of through a request of its own)
  extern __shared__ uint64_t s_emit[];  // 4 waves x 64 rows of RS words
  uint64_t* const rows = s_emit + (threadIdx.x >> 6) * 64 * RS;
  const uint32_t nrec = uint32_t(r1 - r0);                       // (at most 64 x 255)
  const uint32_t my_rel = live ? uint32_t(my_off - r0) : ~0u;
  for (uint32_t x0 = 0; x0 < nrec; x0 += 64) {
    const uint32_t xq = x0 + uint32_t(lane);
    int lo = 0;  // last lane whose first record is <= xq
    // (offsets relative to the wave's first record, 32 bits: a shuffle of a 64-bit value is two)
#pragma unroll
    for (int step = 32; step; step >>= 1) {
      const uint32_t v = uint32_t(__shfl(int(my_rel), lo + step, 64));
      if (v <= xq) lo += step;
    }
    // reads without records share their offset with the next read: the search lands on the last of them,
    // which is the one that owns the record
    const uint32_t off = uint32_t(__shfl(int(my_rel), lo, 64));
    const int L = __shfl(my_len, lo, 64);
    const uint32_t rcnt = uint32_t(__shfl(int(my_cnt), lo, 64));
    if (xq < nrec) {
      const int j = int(xq - off);
      const uint64_t q = r0 + xq;
      const uint64_t* s = owords + (o_base + t0 + uint64_t(lo)) * uint64_t(W);
      uint64_t* row = rows + lane * RS;
      // the record's words = the read's from base j on: every word of the read is fetched once
      const int sq = j >> 5, sr = j & 31;
      uint64_t word0 = 0, cur = sq < W ? s[sq] : 0;
      for (int w = 0; w < W; ++w) {
        const uint64_t next = sq + w + 1 < W ? s[sq + w + 1] : 0;
        const uint64_t word = sr ? (cur << (2 * sr)) | (next >> (64 - 2 * sr)) : cur;
        row[w] = word;
        if (w == 0) word0 = word;
        cur = next;
      }
      if (RS > W) {
        // behind the length (low byte, all that the sort and the tie kernels read): what k_gather_dedup_build needs to set
        // the prev bits where the records meet (kRecHasPred: the suffix one base earlier exists, kRecPredShift: the base in
        // front of this one, kRecLast: the read's last record — its tail was not emitted)
        uint64_t meta = uint64_t(L - j);
        if (j > 0) meta |= kRecHasPred | (((s[(j - 1) >> 5] >> (62 - 2 * ((j - 1) & 31))) & 3) << kRecPredShift);
        if (uint32_t(j) + 1 == rcnt) meta |= kRecLast;
        row[W] = meta;
      }
      if (rlen) rlen[q] = uint8_t(L - j);
      if (k0) k0[q] = (word0 & ~((1ull << kSortSkipBits) - 1)) | q;  // the sort key (k_first_words), for free
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t cntw = (nrec - x0 < 64u ? nrec - x0 : 64u) * uint32_t(RS);
    uint64_t* dst = rec + (r0 + x0) * uint64_t(RS);
    // written once and next read by the gather after the sort: stores that go around the caches (3.29 -> 2.35 ms; 16-byte
    // stores were slower than these 8-byte ones)
    for (uint32_t k = uint32_t(lane); k < cntw; k += 64) __builtin_nontemporal_store(rows[k], &dst[k]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Sort input: the leading 16 bases of a record in the high half of the key, the record's index in the low
// half — a keys-only sort on bits 32..63 (4 radix passes moving 8 bytes per record instead of 5-8 passes
// moving 12).  Records that agree on their leading 16 bases are almost always suffixes at the same genome
// position, which the tie handling orders by full-key compares anyway.
__global__ void k_first_words(const uint64_t* __restrict__ rec, uint64_t n, int RS, uint64_t* __restrict__ k0) {
  uint64_t r = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n) return;
  k0[r] = (rec[r * RS] & ~((1ull << kSortSkipBits) - 1)) | r;
}


// ------------------------------------------------------------------------------------------------
// tie refinement after the first-word sort


// Tie groups (runs of equal first word) of at most kSmallGroup records are ordered right here by
// one lane with full-key comparisons (insertion sort over the members' rows, fetched into LDS).  Only larger
// groups are flagged for the workgroup-per-group kernels and the word-by-word radix rounds below.  With `defer`
// (the entry build) the groups of at most kTileGroup records are not ordered here but in k_gather_dedup_build.
constexpr int kSmallGroup = 32;
constexpr int kTileGroup = 8;  // defer: groups up to this size are left to k_gather_dedup_build's tile
constexpr int kSmallHalo = kSmallGroup + 1;  // how far a member looks for the ends of its group
constexpr int kSmallSlots = 256 + kSmallGroup;  // a group headed in a workgroup ends before its position 256 + 32
// skeys = the sorted packed keys (leading bases in the high half, record index in the low half): this kernel
// also splits them, i.e. writes sidx for every position.
// A workgroup stages the leading halves of its 256 positions (+ a halo) and their record indices in LDS, finds the
// groups there, and then every MEMBER fetches its own row into LDS — one round trip for all the rows of the
// workgroup — before the heads order their groups without touching global memory again; sidx goes out as one
// coalesced store per position.  (A head that compared rows in global memory paid a dependent round trip per
// compare, and one per step of the end search and per index before that: the kernel was 4.8 ms of latency.)
// Dynamic LDS: kSmallSlots rows of W + 1 words (the last one = the record's length).
__global__ __launch_bounds__(256) void k_sort_small_groups(const uint64_t* __restrict__ skeys, uint64_t n,
                                                           const uint64_t* __restrict__ rec,
                                                           const uint8_t* __restrict__ rlen, int W, int RS,
                                                           uint32_t* __restrict__ sidx, uint8_t* __restrict__ flags,
                                                           int defer) {
  extern __shared__ uint64_t s_rows[];
  __shared__ uint32_t s_hi[256 + 2 * kSmallHalo];  // positions b0 - kSmallHalo ...
  __shared__ uint32_t s_idx[kSmallSlots];          // positions b0 ...
  __shared__ uint16_t s_slot[kSmallSlots];         // position -> whose index ends up there
  __shared__ uint32_t s_tail;                      // positions past the workgroup's own that its last group holds
  const int tid = threadIdx.x;
  const int LW = W + 1;
  const uint64_t b0 = uint64_t(blockIdx.x) * 256;
  if (tid == 0) s_tail = 0;
  for (int t = tid; t < 256 + 2 * kSmallHalo; t += 256) {
    const uint64_t p = b0 + uint64_t(t) - uint64_t(kSmallHalo);  // (wraps before position 0: not < n)
    const int q = t - kSmallHalo;
    const bool slot = q >= 0 && q < kSmallSlots;
    if (slot) s_slot[q] = uint16_t(q);
    if (p < n) {
      const uint64_t kv = skeys[p];
      s_hi[t] = uint32_t(kv >> kSortSkipBits);
      if (slot) s_idx[q] = uint32_t(kv);
    }
  }
  __syncthreads();
  auto hi = [&](uint64_t p) { return s_hi[int(int64_t(p) - int64_t(b0)) + kSmallHalo]; };
  auto fetch_row = [&](int q) {
    const uint32_t r = s_idx[q];
    const uint64_t* src = rec + uint64_t(r) * RS;
    uint64_t* dst = s_rows + q * LW;
    for (int w = 0; w < W; ++w) dst[w] = src[w];
    dst[W] = RS > W ? (src[W] & 0xff) : uint64_t(rlen[r]);
  };
  const uint64_t i = b0 + uint64_t(tid);
  uint8_t big = 0;
  bool small_member = false;
  uint64_t s = i, e = i + 1;
  if (i < n) {
    const uint32_t k = hi(i);
    const bool tied = (i > 0 && hi(i - 1) == k) || (i + 1 < n && hi(i + 1) == k);
    if (tied) {
      while (s > 0 && i - s <= uint64_t(kSmallGroup) && hi(s - 1) == k) --s;
      while (e < n && e - s <= uint64_t(kSmallGroup) && hi(e) == k) ++e;
      if (e - s > uint64_t(kSmallGroup)) big = 1;
      else small_member = true;
    }
  }
  // (a group whose head sits in the previous workgroup is that one's; defer: groups of at most kTileGroup records —
  // nearly all of them: two or three reads' suffixes at one genome position — are left as they are for
  // k_gather_dedup_build, which has every row in LDS anyway and orders them there)
  const bool here = small_member && (!defer || e - s > uint64_t(kTileGroup));
  const bool mine = here && s >= b0;
  if (mine) {
    fetch_row(tid);
    if (s == i && e > b0 + 256) s_tail = uint32_t(e - (b0 + 256));  // (one group at most reaches past the end)
  }
  __syncthreads();
  const uint32_t tail = s_tail;
  if (uint32_t(tid) < tail) fetch_row(256 + tid);
  __syncthreads();
  if (mine && s == i) {  // head of a small group: insertion sort of its members' places
    const int g = int(e - s);
    uint16_t* sl = s_slot + tid;
    for (int a = 1; a < g; ++a) {
      const uint16_t xs = sl[a];
      const uint64_t* xk = s_rows + int(xs) * LW;
      const int xl = int(xk[W]);
      int b = a - 1;
      while (b >= 0) {
        const uint16_t ys = sl[b];
        const uint64_t* yk = s_rows + int(ys) * LW;
        if (key_compare(yk, int(yk[W]), xk, xl, W) <= 0) break;
        sl[b + 1] = ys;
        --b;
      }
      sl[b + 1] = xs;
    }
  }
  __syncthreads();
  if (i < n) {
    if (!(here && s < b0)) sidx[i] = s_idx[s_slot[tid]];
    flags[i] = big;
  }
  if (uint32_t(tid) < tail) sidx[b0 + 256 + uint64_t(tid)] = s_idx[s_slot[256 + tid]];
}

// Medium tie groups (kSmallGroup < size <= kMediumGroup): one workgroup per group stages the group's keys in
// LDS and orders them with a bitonic network of full-key compares.  Only larger groups go to the radix rounds
// (dozens of tiny launches per round, whatever the number of records).  What the first launch leaves (groups
// over kMediumGroup) gets a second one with a CU's whole LDS per group (3072 members of 150-base reads).
constexpr int kMediumGroup = 1024;

__global__ void k_tied_run_heads(const uint32_t* __restrict__ tpos, const uint64_t* __restrict__ tkey, uint64_t T,
                                 uint8_t* __restrict__ head) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= T) return;
  head[t] = (t == 0 || tpos[t] != tpos[t - 1] + 1 || tkey[t] != tkey[t - 1]) ? 1 : 0;
}

// Handles the groups with min_size < size <= cap (cap members fit the dynamic LDS).  Bigger ones: still = 1 for
// their members, unless check_equal is set and all their keys turn out to be one and the same (the oriented reads
// that correction emptied, both strands of each, are such a group: tens of thousands of equal keys that the
// radix rounds would carry through every word) — equal keys are in place as they are.
__global__ __launch_bounds__(1024) void k_sort_medium_groups(const uint32_t* __restrict__ hpos, uint64_t G, uint64_t T,
                                                             const uint32_t* __restrict__ tpos,
                                                             const uint64_t* __restrict__ rec,
                                                             const uint8_t* __restrict__ rlen, int W, int RS,
                                                             uint32_t* __restrict__ sidx, uint8_t* __restrict__ still,
                                                             uint32_t min_size, uint32_t cap, int check_equal) {
  extern __shared__ uint64_t mg_smem[];
  uint64_t* skey = mg_smem;                                            // cap * W
  uint32_t* sid = reinterpret_cast<uint32_t*>(skey + size_t(cap) * W);  // record of each member
  uint16_t* perm = reinterpret_cast<uint16_t*>(sid + cap);             // sorted order -> member (cap rounded up to 2^x)
  __shared__ uint32_t s_differ;
  const uint64_t g = blockIdx.x;
  const uint32_t t0 = hpos[g];
  const uint32_t t1 = g + 1 < G ? hpos[g + 1] : uint32_t(T);
  const uint32_t size = t1 - t0;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  auto len_of = [&](uint32_t r) { return RS > W ? uint8_t(rec[uint64_t(r) * RS + W]) : rlen[r]; };
  if (size <= min_size) return;  // block-uniform; done by an earlier launch
  const uint32_t p0 = tpos[t0];  // the group occupies sorted positions p0 .. p0 + size
  if (size > cap) {  // block-uniform
    uint32_t differ = 1;
    if (check_equal) {
      if (tid == 0) s_differ = 0;
      __syncthreads();
      const uint32_t r0 = sidx[p0];
      bool d = false;
      for (uint32_t i = tid; i < size && !d; i += nt) {
        const uint32_t r = sidx[p0 + i];
        d = len_of(r) != len_of(r0);
        for (int w = 0; w < W && !d; ++w) d = rec[uint64_t(r) * RS + w] != rec[uint64_t(r0) * RS + w];
      }
      if (d) s_differ = 1;
      __syncthreads();
      differ = s_differ;
    }
    for (uint32_t i = tid; i < size; i += nt) still[t0 + i] = uint8_t(differ);
    return;
  }
  uint32_t P = 1;
  while (P < size) P <<= 1;
  uint8_t* slen = reinterpret_cast<uint8_t*>(perm + (cap > P ? cap : P));
  for (uint32_t i = tid; i < size; i += nt) {
    const uint32_t r = sidx[p0 + i];
    sid[i] = r;
    slen[i] = len_of(r);
    still[t0 + i] = 0;
  }
  for (uint32_t i = tid; i < P; i += nt) perm[i] = i < size ? uint16_t(i) : uint16_t(0xffff);
  __syncthreads();
  for (uint32_t x = tid; x < size * uint32_t(W); x += nt) {
    const uint32_t i = x / uint32_t(W), w = x - i * uint32_t(W);
    skey[x] = rec[uint64_t(sid[i]) * RS + w];
  }
  __syncthreads();
  for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
    for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < P; i += nt) {
        const uint32_t x = i ^ j;
        if (x > i) {
          const uint16_t a = perm[i], b = perm[x];
          // padding (0xffff) sorts after every member
          const bool gt = a == 0xffff ? b != 0xffff
                                      : (b != 0xffff && key_compare(skey + size_t(a) * W, slen[a], skey + size_t(b) * W,
                                                                    slen[b], W) > 0);
          const bool up = (i & k2) == 0;
          if (gt == up) { perm[i] = b; perm[x] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = tid; i < size; i += nt) sidx[p0 + i] = sid[perm[i]];
}

// subset element t sits at sorted position tpos[t]; it starts a group unless it continues the run of
// its predecessor (adjacent position, same previous-round group and same key).
__global__ void k_group_heads(const uint32_t* __restrict__ tpos, const uint64_t* __restrict__ tkey,
                              const uint32_t* __restrict__ prev_gid, uint64_t T, uint32_t* __restrict__ head) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= T) return;
  bool cont = t > 0 && tpos[t] == tpos[t - 1] + 1 && tkey[t] == tkey[t - 1] &&
              (prev_gid == nullptr || prev_gid[t] == prev_gid[t - 1]);
  head[t] = cont ? 0u : 1u;
}

__global__ void k_gid_from_scan(const uint32_t* __restrict__ excl, const uint32_t* __restrict__ head, uint64_t T,
                                uint32_t* __restrict__ gid) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= T) return;
  gid[t] = excl[t] + head[t] - 1;
}

__global__ void k_gather_u64_by_u32(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n,
                                    uint64_t* __restrict__ dst) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// leading-bases part of the packed sort keys at the given positions
__global__ void k_gather_key_prefix(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n,
                                    uint64_t* __restrict__ dst) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]] & ~((1ull << kSortSkipBits) - 1);
}
__global__ void k_gather_u32_by_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n,
                                    uint32_t* __restrict__ dst) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_iota_u32(uint32_t* __restrict__ dst, uint64_t n) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = uint32_t(i);
}

// next-round key of subset element t: word `w` of its record (or the length when w == W)
__global__ void k_round_keys(const uint64_t* __restrict__ rec, const uint8_t* __restrict__ rlen,
                             const uint32_t* __restrict__ sidx, const uint32_t* __restrict__ tpos, uint64_t T, int W,
                             int RS, int w, uint64_t* __restrict__ key) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= T) return;
  uint32_t r = sidx[tpos[t]];
  key[t] = (w < W) ? rec[uint64_t(r) * RS + w] : uint64_t(RS > W ? uint8_t(rec[uint64_t(r) * RS + W]) : rlen[r]);
}

__global__ void k_u32_to_u64(const uint32_t* __restrict__ src, uint64_t n, uint64_t* __restrict__ dst) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// after the two stable sorts: perm[u] = subset element that belongs at subset slot u.
// new_sidx[u] = record of that element; written back to sidx[tpos[u]].
__global__ void k_apply_perm(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ tpos,
                             const uint32_t* __restrict__ old_rec_of_t, uint64_t T, uint32_t* __restrict__ sidx) {
  uint64_t u = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= T) return;
  sidx[tpos[u]] = old_rec_of_t[perm[u]];
}

// still tied: same group and same round key as a neighbour (in the re-sorted subset order)
__global__ void k_flag_subset_ties(const uint32_t* __restrict__ gid, const uint64_t* __restrict__ key, uint64_t T,
                                   uint8_t* __restrict__ flags) {
  uint64_t u = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= T) return;
  bool t = (u > 0 && gid[u - 1] == gid[u] && key[u - 1] == key[u]) ||
           (u + 1 < T && gid[u + 1] == gid[u] && key[u + 1] == key[u]);
  flags[u] = t ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// dedup + arrays

// Sorted order -> contiguous keys/len, and in the same pass (the tile is in LDS anyway) the lcp with the
// predecessor and "is a prefix of (or equal to) the successor" -> dropped (expand.cpp:19-48 skip_dups).
// One block = 256 sorted positions + one halo record on each side.
constexpr int kGatherTile = 256;
__global__ __launch_bounds__(kGatherTile) void k_gather_lcp_keep(
    const uint64_t* __restrict__ rec, const uint8_t* __restrict__ rlen, const uint32_t* __restrict__ sidx, uint64_t n,
    int W, int RS, uint64_t* __restrict__ skeys, uint8_t* __restrict__ slen, uint8_t* __restrict__ lcp_prev,
    uint8_t* __restrict__ keep, uint32_t* __restrict__ err) {
  extern __shared__ uint64_t tile[];  // (kGatherTile + 2) * W words, then the lengths
  uint8_t* tlen = reinterpret_cast<uint8_t*>(tile + (kGatherTile + 2) * W);
  const uint64_t b0 = uint64_t(blockIdx.x) * kGatherTile;
  for (int g = threadIdx.x; g < (kGatherTile + 2) * W; g += kGatherTile) {
    int i = g / W, w = g - i * W;
    uint64_t pos = b0 + uint64_t(i) - 1;  // wraps for the missing predecessor of position 0
    if (pos < n) {
      uint32_t r = sidx[pos];
      uint64_t v = rec[uint64_t(r) * RS + w];
      tile[g] = v;
      bool own = i >= 1 && i <= kGatherTile;
      if (own) skeys[pos * W + w] = v;
      if (w == 0) {
        uint8_t l = RS > W ? uint8_t(rec[uint64_t(r) * RS + W]) : rlen[r];
        tlen[i] = l;
        if (own) slen[pos] = l;
      }
    }
  }
  __syncthreads();
  const uint64_t pos = b0 + threadIdx.x;
  if (pos >= n) return;
  const uint64_t* a = tile + (threadIdx.x + 1) * W;
  const int alen = tlen[threadIdx.x + 1];
  int lp = 0;
  if (pos > 0) {
    lp = key_lcp(a - W, tlen[threadIdx.x], a, alen, W);
    if (key_compare(a - W, tlen[threadIdx.x], a, alen, W) > 0) atomicOr(err, 1u);  // not sorted
  }
  lcp_prev[pos] = uint8_t(lp);
  bool drop = false;
  if (pos + 1 < n) drop = key_lcp(a, alen, a + W, tlen[threadIdx.x + 2], W) == alen;
  keep[pos] = drop ? 0 : 1;
}

// The three steps above and below in one pass (gather, lcp / keep, compaction + entry build): the gathered tile
// never leaves LDS.  The position of a tile's first surviving record in the output is the number of survivors
// in all earlier tiles: a single-pass scan with decoupled look-back — a tile publishes its own count at once
// (state = AGG | count), adds up its predecessors' states until it meets one that already carries its inclusive
// prefix, then publishes its own (INCL | prefix).  Tiles take their number from a ticket counter, so every
// predecessor a tile waits for is already running.
// shared[e] = LCP of consecutive survivors (= min of the LCPs over the dropped run between them, LCP being
// transitive through prefixes); a tile's first survivor gets it from k_fix_tile_first_shared.
constexpr uint64_t kScanAgg = 1ull << 62, kScanIncl = 2ull << 62, kScanVal = (1ull << 62) - 1;

#ifdef BG_EXP_FUSEDPOS
constexpr int kFusedPos = BG_EXP_FUSEDPOS;             // (timing experiment)
#else
constexpr int kFusedPos = 2;                           // sorted positions per thread
#endif
constexpr int kFusedTile = kGatherTile * kFusedPos;    // positions per tile: twice the data in flight per wave and
                                                       // half the look-backs of a 256-position tile
// sort_small: the tie groups of at most kTileGroup records come unordered (k_sort_small_groups, defer) and are put
// in order here, in LDS — every row is fetched for the gather anyway, and ordering them beforehand fetched more than
// half of them a first time (3.5 ms).  A tile then brings a halo of kFusedHalo positions on each side instead of
// one: a group of at most kTileGroup records that touches the tile's own positions or their two neighbours lies
// wholly inside it, a group with more members in sight is in order already (k_sort_small_groups up to kSmallGroup
// records, the medium-group kernels beyond), and what the halo shows of a group that does not reach the tile's
// positions may be ordered or not — it is not used.  A group that straddles two tiles is ordered by both, alike.
// (kTileGroup = 8: groups of two or three are what a genome makes; bigger ones come from repeats, where ranking a
// group of g costs every member g compares — with kTileGroup = kSmallGroup a repeat-heavy genome built 1.2 ms slower
// than before — and a halo of 10 rows per side instead of 34 is 4 % more rows instead of 13 %.)
constexpr int kFusedHalo = kTileGroup + 2;
constexpr int kFusedSlots = kFusedTile + 2 * kFusedHalo;
__global__ __launch_bounds__(kGatherTile) void k_gather_dedup_build(
    const uint64_t* __restrict__ rec, const uint8_t* __restrict__ rlen, const uint32_t* __restrict__ sidx, uint64_t n,
    int W, int RS, unsigned long long* __restrict__ state, uint32_t* __restrict__ ticket, uint64_t* __restrict__ ekeys,
    uint8_t* __restrict__ sizes, uint8_t* __restrict__ shared, uint64_t* __restrict__ pre2,
    uint64_t* __restrict__ tile_first, uint32_t* __restrict__ err, int sort_small, uint8_t* __restrict__ emask,
    uint64_t* __restrict__ list, unsigned long long* __restrict__ list_n, uint64_t list_cap,
    unsigned long long* __restrict__ bstate, uint8_t* __restrict__ need) {
  const int H = sort_small ? kFusedHalo : 1;  // slot i of the tile = sorted position b0 + i - H
  const int NS = kFusedTile + 2 * H;
  extern __shared__ uint64_t tile[];  // NS * W words, then lengths, per-position lcp, keep flags, record flags
  uint8_t* tlen = reinterpret_cast<uint8_t*>(tile + NS * W);
  uint8_t* tlcp = tlen + NS;
  uint8_t* tkeep = tlcp + kFusedTile;
  uint8_t* tmeta = tkeep + kFusedTile;  // NS: bits 8.. of word W of every row (kRecHasPred ...), shifted down
  // prev bits where the records meet (emask != nullptr; rows of W + 1 words with the flags of k_emit_records).  A record
  // (r, j), j >= 1, is the tail of the suffix one base earlier, b.s with b = the read's base j - 1.  If the record's string
  // s is an entry (it survives, or it equals the survivor that follows it), b.s is an entry as well (a suffix b.s.z would
  // make s.z a suffix, and s would not be maximal), and it sets prev_b[s] (builder.cpp:85-108): no search.
  // What is left goes to a list of tails for k_prev_list: the tails of survivors that are the last record of their read
  // (the tail was not emitted: one record in six), and the records that are dropped as PROPER prefixes (b.s sets
  // prev_b[first entry that extends s] only if b.s is an entry itself: checked there), or whose survivor is out of sight.
  __shared__ uint32_t tmk[kFusedTile];  // survivor rank -> prev bases (bit b)
  __shared__ uint32_t s_nitems;
  __shared__ unsigned long long s_lbase;
  __shared__ uint32_t tidx[kFusedSlots];
  __shared__ uint16_t pmap[kFusedSlots];
  __shared__ uint16_t tsrc[kFusedTile];  // survivor rank -> slot of its row
  __shared__ uint8_t tsh[kFusedTile];    // survivor rank -> shared
  __shared__ uint32_t s_tile, s_wave[kFusedTile / 64];
  __shared__ unsigned long long s_excl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { s_tile = atomicAdd(ticket, 1u); s_nitems = 0; }
  if (emask)
    for (int i = tid; i < kFusedTile; i += kGatherTile) tmk[i] = 0;
  __syncthreads();
  const uint64_t t = s_tile;
  const uint64_t b0 = t * kFusedTile;
  // record of every position of the tile (+ halo) first, so that the word loads below are independent of
  // each other and all in flight at once (a dependent sidx -> rec chain per word made a tile ten memory
  // round trips long)
  for (int i = tid; i < NS; i += kGatherTile) {
    const uint64_t pos = b0 + uint64_t(i) - uint64_t(H);  // wraps for the missing predecessors of position 0
    uint32_t r = 0xffffffffu;
    if (pos < n) {
      r = sidx[pos];
      if (RS == W) tlen[i] = rlen[r];  // (RS = W + 1: the length comes with the row, below)
    }
    tidx[i] = r;
  }
  __syncthreads();
  if ((RS & 1) == 0 && (reinterpret_cast<uintptr_t>(rec) & 15) == 0) {
    // rows of an even number of words are 16-byte aligned: two words a load, half the requests (8.61 -> 8.51 ms)
    const int RP = RS >> 1;
    const uint32_t rcp = (65536u + uint32_t(RP) - 1) / uint32_t(RP);
    const int total = NS * RP;
    constexpr int kIters = (kFusedSlots * ((kMaxWords + 2) / 2) + kGatherTile - 1) / kGatherTile;
    ulonglong2 v[kIters];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int g = tid + it * kGatherTile;
      v[it] = make_ulonglong2(0, 0);
      if (g < total) {
        const int i = int((uint32_t(g) * rcp) >> 16), h = g - i * RP;
        const uint32_t r = tidx[i];
        if (r != 0xffffffffu) v[it] = *reinterpret_cast<const ulonglong2*>(rec + uint64_t(r) * RS + 2 * h);
      }
    }
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int g = tid + it * kGatherTile;
      if (g < total) {
        const int i = int((uint32_t(g) * rcp) >> 16), h = g - i * RP;
        const uint64_t xs[2] = {v[it].x, v[it].y};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int w = 2 * h + u;
          if (w < W) tile[i * W + w] = xs[u];
          else if (w == W) { tlen[i] = uint8_t(xs[u]); tmeta[i] = uint8_t(xs[u] >> 8); }
        }
      }
    }
  } else {
    // RS words of every row: its W key words and, when RS = W + 1, its length (a request of its own otherwise: the
    // lengths' lines were a third of what the kernel fetched)
    const uint32_t rcp = (65536u + uint32_t(RS) - 1) / uint32_t(RS);  // g / RS = (g * rcp) >> 16 for g < 9000, RS <= 9
    const int total = NS * RS;
    constexpr int kIters = (kFusedSlots * (kMaxWords + 1) + kGatherTile - 1) / kGatherTile;
    uint64_t v[kIters];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int g = tid + it * kGatherTile;
      v[it] = 0;
      if (g < total) {
        const int i = int((uint32_t(g) * rcp) >> 16), w = g - i * RS;
        const uint32_t r = tidx[i];
        if (r != 0xffffffffu) v[it] = rec[uint64_t(r) * RS + w];
      }
    }
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const int g = tid + it * kGatherTile;
      if (g < total) {
        const int i = int((uint32_t(g) * rcp) >> 16), w = g - i * RS;
        if (w < W) tile[i * W + w] = v[it];
        else if (w == W) { tlen[i] = uint8_t(v[it]); tmeta[i] = uint8_t(v[it] >> 8); }  // (words beyond: padding of wider rows)
      }
    }
  }
  __syncthreads();
#ifdef BG_EXP_GATHER_ONLY  // timing experiment (tools/build_variant.sh): what the gather alone costs -- the tile's rows straight out again
  for (int x = tid; x < kFusedTile * W; x += kGatherTile)
    if (b0 * W + x < n * uint64_t(W)) ekeys[b0 * W + x] = tile[H * W + x];
  if (tid == 0) tile_first[t] = ~0ull;
  if (tid == 0 && b0 + kFusedTile >= n) state[(n + kFusedTile - 1) / kFusedTile - 1] = kScanIncl | n;
  for (int x = tid; x < kFusedTile; x += kGatherTile)
    if (b0 + x < n) { sizes[b0 + x] = tlen[H + x]; shared[b0 + x] = 0; if (emask) emask[b0 + x] = 0; }
  return;
#endif
  // pmap[i] = the slot whose row belongs at slot i.  Every member of a small group finds its rank in the group by
  // comparing its row with the others' (all members at once, a few LDS reads each: one lane per group moving rows
  // about made the kernel 5 ms longer); the rows stay where they are.
  for (int j = tid; j < NS; j += kGatherTile) {
    int at = j;
    if (sort_small && tidx[j] != 0xffffffffu) {
      const uint32_t k = uint32_t(tile[j * W] >> kSortSkipBits);
      auto same = [&](int q) {
        return q >= 0 && q < NS && tidx[q] != 0xffffffffu && uint32_t(tile[q * W] >> kSortSkipBits) == k;
      };
      if (same(j - 1) || same(j + 1)) {
        int s = j, e = j + 1;
        while (j - s <= kTileGroup && same(s - 1)) --s;
        while (e - s <= kTileGroup && same(e)) ++e;
        if (e - s <= kTileGroup) {
          uint64_t xr[kMaxWords];
#pragma unroll
          for (int w = 0; w < kMaxWords; ++w) xr[w] = w < W ? tile[j * W + w] : 0;
          const int xl = tlen[j];
          int rank = 0;
          for (int q = s; q < e; ++q) {
            if (q == j) continue;
            const uint64_t* y = tile + q * W;
            int c = 0;  // row q against this one
#pragma unroll
            for (int w = 0; w < kMaxWords; ++w)
              if (w < W && !c) {
                const uint64_t yw = y[w];
                if (yw != xr[w]) c = yw < xr[w] ? -1 : 1;
              }
            if (!c) {
              const int yl = tlen[q];
              c = yl < xl ? -1 : (yl > xl ? 1 : 0);
            }
            if (c < 0 || (c == 0 && q < j)) ++rank;
          }
          at = s + rank;
        }
      }
    }
    pmap[at] = uint16_t(j);
  }
  __syncthreads();
  // position p of the tile = thread (p % 256), slice (p / 256)
  bool keep[kFusedPos];
  uint64_t km[kFusedPos];
#pragma unroll
  for (int sl = 0; sl < kFusedPos; ++sl) {
    const int p = sl * kGatherTile + tid;
    const uint64_t pos = b0 + p;
    const int sp = pmap[p + H], sprev = pmap[p + H - 1], snext = pmap[p + H + 1];
    const uint64_t* a = tile + sp * W;
    const int alen = tlen[sp];
    keep[sl] = false;
    if (pos < n) {
      int lp = 0;
      if (pos > 0) {
        lp = key_lcp(tile + sprev * W, tlen[sprev], a, alen, W);
        if (key_compare(tile + sprev * W, tlen[sprev], a, alen, W) > 0) atomicOr(err, 1u);  // not sorted
      }
      tlcp[p] = uint8_t(lp);
      // a record that is a prefix of (or equal to) its successor is dropped (expand.cpp:19-48 skip_dups)
      keep[sl] = !(pos + 1 < n && key_lcp(a, alen, tile + snext * W, tlen[snext], W) == alen);
    }
    tkeep[p] = keep[sl] ? 1 : 0;
    km[sl] = __ballot(keep[sl]);
    if (lane == 0) s_wave[sl * (kGatherTile / 64) + wave] = uint32_t(__popcll(km[sl]));
  }
  __syncthreads();
  // rank of the survivors inside the tile
  uint32_t before[kFusedPos], count = 0;
#pragma unroll
  for (int sl = 0; sl < kFusedPos; ++sl) before[sl] = uint32_t(__popcll(km[sl] & ((1ull << lane) - 1)));
  for (int v2 = 0; v2 < kFusedTile / 64; ++v2) {
    const uint32_t c = s_wave[v2];
#pragma unroll
    for (int sl = 0; sl < kFusedPos; ++sl)
      if (v2 < sl * (kGatherTile / 64) + wave) before[sl] += c;
    count += c;
  }
#ifdef BG_EXP_NO_LOOKBACK  // timing experiment: what the kernel costs without waiting for its predecessors
  if (tid == 0) { s_excl = t * kFusedTile; tile_first[t] = ~0ull; }
  if (false) {
#else
  // decoupled look-back (wave 0)
  if (wave == 0) {
#endif
    unsigned long long excl = 0;
    if (bstate) {
      // Two levels.  With every tile of the chip at the same stage at once, the nearest predecessor that already knows its
      // prefix is as far back as there are tiles in flight (1280): twenty rounds of 64 states, each a round trip to the L2,
      // one behind the other.  Here a tile adds up the counts of the tiles before it in its BLOCK of 64 tiles (one round),
      // the block's last tile publishes the block's sum, and the blocks before it are one more look-back of the same kind
      // over block states -- 64 blocks = 4096 tiles a round.
      const uint64_t ntiles = (n + kFusedTile - 1) / kFusedTile;
      const uint64_t bk = t >> 6;
      const int k = int(t & 63);
      const bool closes = k == 63 || t + 1 == ntiles;
      if (lane == 0) __hip_atomic_store(&state[t], kScanAgg | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long in_block = 0;
      for (;;) {
        const unsigned long long v =
            lane < k ? __hip_atomic_load(&state[(bk << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kScanAgg;
        const uint64_t inval = __ballot((v >> 62) == 0);
        if (inval) {
          const uint64_t widx = (bk << 6) + uint64_t(__ffsll((unsigned long long)inval) - 1);
          if (lane == 0)
            while ((__hip_atomic_load(&state[widx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 62) == 0)
              __builtin_amdgcn_s_sleep(32);
          continue;
        }
        unsigned long long part = lane < k ? (v & kScanVal) : 0ull;
#pragma unroll
        for (int d = 32; d; d >>= 1) part += __shfl_xor(part, d, 64);
        in_block = part;
        break;
      }
      if (closes && lane == 0)
        __hip_atomic_store(&bstate[bk], (bk == 0 ? kScanIncl : kScanAgg) | (in_block + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long blocks = 0;
      if (bk > 0) {
        int64_t j = int64_t(bk) - 1;
        for (;;) {
          const int64_t idx = j - lane;
          const unsigned long long v =
              idx >= 0 ? __hip_atomic_load(&bstate[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kScanIncl;
          const int flag = int(v >> 62);
          const uint64_t incl = __ballot(flag == 2), inval = __ballot(flag == 0);
          const int first = incl ? __ffsll((unsigned long long)incl) - 1 : 64;
          const uint64_t upto = first >= 63 ? ~0ull : ((2ull << first) - 1);
          if (inval & upto) {
            const int64_t widx = j - (__ffsll((unsigned long long)(inval & upto)) - 1);
            if (lane == 0)
              while ((__hip_atomic_load(&bstate[widx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 62) == 0)
                __builtin_amdgcn_s_sleep(32);
            continue;
          }
          unsigned long long part = (lane <= first) ? (v & kScanVal) : 0ull;
#pragma unroll
          for (int d = 32; d; d >>= 1) part += __shfl_xor(part, d, 64);
          blocks += part;
          if (incl) break;
          j -= 64;
        }
        if (closes && lane == 0)
          __hip_atomic_store(&bstate[bk], kScanIncl | (blocks + in_block + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      excl = blocks + in_block;
      if (t + 1 == ntiles && lane == 0)  // the host reads the number of entries here
        __hip_atomic_store(&state[t], kScanIncl | (excl + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
    if (lane == 0)
      __hip_atomic_store(&state[t], (t == 0 ? kScanIncl : kScanAgg) | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t > 0) {
      int64_t j = int64_t(t) - 1;
      for (;;) {
        const int64_t idx = j - lane;
        // (the state words are self-contained: relaxed accesses are enough)
        const unsigned long long v =
            idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kScanIncl;
        const int flag = int(v >> 62);
        const uint64_t incl = __ballot(flag == 2), inval = __ballot(flag == 0);
        const int first = incl ? __ffsll((unsigned long long)incl) - 1 : 64;  // nearest predecessor with a prefix
        const uint64_t upto = first >= 63 ? ~0ull : ((2ull << first) - 1);     // lanes 0..first
        if (inval & upto) {
          // someone in between has not published yet: one lane polls that one state, with a pause between polls
          // (thousands of tiles polling 64 states each at full speed take the L2 away from the gathers)
          const int64_t widx = j - (__ffsll((unsigned long long)(inval & upto)) - 1);
          if (lane == 0)
            while ((__hip_atomic_load(&state[widx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 62) == 0)
              __builtin_amdgcn_s_sleep(32);
          continue;
        }
        unsigned long long part = (lane <= first) ? (v & kScanVal) : 0ull;
#pragma unroll
        for (int d = 32; d; d >>= 1) part += __shfl_xor(part, d, 64);
        excl += part;
        if (incl) break;
        j -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&state[t], kScanIncl | (excl + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }
    if (lane == 0) {
      s_excl = excl;
      tile_first[t] = count ? excl : ~0ull;
    }
  }
  // prev bits of the records that meet their survivor in this tile; the rest become list items
  int item_kind[kFusedPos], item_slot[kFusedPos];  // 0 none, 1 the tail of this survivor, 2 this record as a tail, checked
  uint32_t item_at[kFusedPos];
#pragma unroll
  for (int sl = 0; sl < kFusedPos; ++sl) {
    item_kind[sl] = 0;
    item_slot[sl] = 0;
    item_at[sl] = 0;
    const int p = sl * kGatherTile + tid;
    if (emask && b0 + p < n) {
      const int sp = pmap[p + H];
      const uint32_t m = tmeta[sp];
      const bool has_pred = (m & uint32_t(kRecHasPred >> 8)) != 0;
      const uint32_t pb = (m >> (kRecPredShift - 8)) & 3u;
      if (keep[sl]) {
        if (has_pred) atomicOr(&tmk[before[sl]], 1u << pb);
        // (need: a shard of a sharded build -- the survivor's tail is asked for through a flag, not a list row: the tails
        // exchange makes the rows.  Its whole-read records arrived before their cover was known and carry no flag: the reads
        // whose suffix 1 is covered come as look-up rows from their senders, bg_cover_query_first_covered_rows)
        const bool last = (m & uint32_t(kRecLast >> 8)) != 0;
        if (last) {
          if (need) atomicOr(&tmk[before[sl]], 0x100u);
          else item_kind[sl] = 1;
        }
      } else if (has_pred) {
        // the survivor this record is a prefix of: the first one after it (a run of dropped records is a chain of prefixes)
        int q = p + 1;
        const int qe = p + 16 < kFusedTile ? p + 16 : kFusedTile;
        while (q < qe && !tkeep[q]) ++q;
        if (q < qe && tlen[pmap[q + H]] == tlen[sp]) atomicOr(&tmk[before[sl]], 1u << pb);  // equal strings; its rank = survivors before p
        else item_kind[sl] = 2;
      }
      if (item_kind[sl]) { item_slot[sl] = sp; item_at[sl] = atomicAdd(&s_nitems, 1u); }
    }
  }
  // survivors compacted through LDS so that every output array is written in whole lines (a thread writing
  // its own 40-byte entry leaves the lanes of a store 40 bytes apart)
#pragma unroll
  for (int sl = 0; sl < kFusedPos; ++sl) {
    if (keep[sl]) {
      const int p = sl * kGatherTile + tid;
      tsrc[before[sl]] = pmap[p + H];
      int sh = 0;
      if (before[sl]) {  // min lcp over the dropped run back to the previous survivor (which is in this tile)
        sh = tlcp[p];
        for (int q = p - 1; !tkeep[q]; --q) {
          const int v = tlcp[q];
          sh = v < sh ? v : sh;
        }
      }
      tsh[before[sl]] = uint8_t(sh);  // slot 0 (the tile's first survivor) is set by k_fix_tile_first_shared
    }
  }
  __syncthreads();
  if (emask && tid == 0 && s_nitems) s_lbase = atomicAdd(list_n, (unsigned long long)s_nitems);
  const uint64_t e0 = s_excl;
  const uint32_t rcpw = (65536u + uint32_t(W) - 1) / uint32_t(W);
  for (uint32_t x = tid; x < count * uint32_t(W); x += kGatherTile) {
    const uint32_t slot = (x * rcpw) >> 16, w = x - slot * uint32_t(W);
    __builtin_nontemporal_store(tile[uint32_t(tsrc[slot]) * W + w], &ekeys[e0 * W + x]);
  }
  for (uint32_t x = tid; x < count * 2; x += kGatherTile) {
    const uint32_t slot = x >> 1, w = x & 1;
    pre2[e0 * 2 + x] = int(w) < W ? tile[uint32_t(tsrc[slot]) * W + w] : 0ull;
  }
  for (uint32_t x = tid; x < count; x += kGatherTile) {
    sizes[e0 + x] = tlen[uint32_t(tsrc[x])];
    shared[e0 + x] = tsh[x];
    if (emask) emask[e0 + x] = uint8_t(tmk[x]);
    if (need) need[e0 + x] = uint8_t(tmk[x] >> 8);
  }
  if (!emask) return;
  __syncthreads();
  if (!s_nitems) return;
  const unsigned long long lbase = s_lbase;
#pragma unroll
  for (int sl = 0; sl < kFusedPos; ++sl) {
    if (!item_kind[sl]) continue;
    const unsigned long long at = lbase + item_at[sl];
    if (at >= list_cap) continue;  // (the host sees the count and falls back to the search for every entry)
    const uint64_t* a = tile + item_slot[sl] * W;
    uint64_t* row = list + at * uint64_t(W + 1);
    const uint64_t len = tlen[item_slot[sl]];
    if (item_kind[sl] == 1) {  // s[1:], its length, s[0]
      for (int w = 0; w < W; ++w) row[w] = (a[w] << 2) | (w + 1 < W ? a[w + 1] >> 62 : 0ull);
      row[W] = (len - 1) | ((a[0] >> 62) << 8);
    } else if (!need) {  // s, its length, the base in front of it; bit 16: only if that base + s is an entry
      for (int w = 0; w < W; ++w) row[w] = a[w];
      row[W] = len | (uint64_t((tmeta[item_slot[sl]] >> (kRecPredShift - 8)) & 3u) << 8) | (1ull << 16);
    } else {  // a shard: b.s itself, to be looked up where it sorts (if it is an entry there, that entry sends its tail)
      const uint64_t b = uint64_t((tmeta[item_slot[sl]] >> (kRecPredShift - 8)) & 3u);
      for (int w = 0; w < W; ++w) row[w] = ((w ? a[w - 1] : b) << 62) | (a[w] >> 2);
      row[W] = len + 1;
    }
  }
}

// shared of every tile's first survivor: LCP with the survivor before it (in an earlier tile)
__global__ void k_fix_tile_first_shared(const uint64_t* __restrict__ tile_first, uint64_t ntiles,
                                        const uint64_t* __restrict__ ekeys, const uint8_t* __restrict__ sizes, int W,
                                        uint8_t* __restrict__ shared) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const uint64_t e = tile_first[t];
  if (e == ~0ull) return;
  shared[e] = e ? uint8_t(key_lcp(ekeys + (e - 1) * W, sizes[e - 1], ekeys + e * W, sizes[e], W)) : uint8_t(0);
}

// entry e = kept position p: sizes, shared = min lcp over the dropped run before p (lcp is
// "transitive" through prefixes), keys.
__global__ void k_build_entries(const uint64_t* __restrict__ skeys, const uint8_t* __restrict__ slen,
                                const uint8_t* __restrict__ lcp_prev, const uint32_t* __restrict__ kept_pos,
                                uint64_t n_ent, int W, uint64_t* __restrict__ ekeys, uint8_t* __restrict__ sizes,
                                uint8_t* __restrict__ shared, uint64_t* __restrict__ pre2) {
  uint64_t e = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= n_ent) return;
  uint32_t p = kept_pos[e];
  uint32_t lo = e ? kept_pos[e - 1] + 1 : 0;  // first position after the previous kept one
  int sh = lcp_prev[p];
  for (uint32_t q = p; q > lo; --q) {
    int v = lcp_prev[q - 1];
    sh = v < sh ? v : sh;
  }
  if (e == 0) sh = 0;
  sizes[e] = slen[p];
  shared[e] = uint8_t(sh);
  uint64_t k0 = 0, k1 = 0;
  for (int w = 0; w < W; ++w) {
    const uint64_t v = skeys[uint64_t(p) * W + w];
    ekeys[e * W + w] = v;
    if (w == 0) k0 = v;
    if (w == 1) k1 = v;
  }
  pre2[e * 2] = k0;
  pre2[e * 2 + 1] = k1;
}

// Directory over the leading kPrefixBases bases of the sorted entries: dir[p] = first entry whose
// (zero padded) leading bases are >= p, dir[4^kPrefixBases] = n.  Narrows every lower_bound below to
// the handful of entries sharing the query's prefix.
// bases of the directory, chosen per build: 4^B <= n/4, 8..12 (measured on 144 M entries: 11 and 12 bases
// 14.0 ms for the whole prev-bit stage, 13 bases 14.8 — the bigger table falls out of cache)
int prefix_dir_bases(uint64_t n) {
  int b = 8;
  while (b < 12 && (1ull << (2 * (b + 1))) <= n / 4 + 1) ++b;
  return b;
}
// Directory over the leading kPrefixBases bases: dir[p] = first entry whose prefix is >= p, dir[4^B] = n.
// Built without serial fills (a shard of a sharded build, or a small seqset, leaves most prefixes empty):
// heads of non-empty prefixes, then a suffix-min over the table in 2048-prefix chunks.
constexpr int kDirChunk = 2048;

__global__ void k_dir_heads(const uint64_t* __restrict__ ekeys, uint64_t n, int W, int bases,
                            uint32_t* __restrict__ dir) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t cur = uint32_t(ekeys[i * W] >> (64 - 2 * bases));
  if (i == 0 || uint32_t(ekeys[(i - 1) * W] >> (64 - 2 * bases)) != cur) dir[cur] = uint32_t(i);
}

__global__ __launch_bounds__(256) void k_dir_chunk_min(const uint32_t* __restrict__ dir, uint32_t* __restrict__ cmin) {
  __shared__ uint32_t sm[4];
  uint64_t base = uint64_t(blockIdx.x) * kDirChunk + threadIdx.x * 8;
  uint32_t m = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < 8; ++j) m = min(m, dir[base + j]);
  for (int d = 32; d; d >>= 1) m = min(m, uint32_t(__shfl_xor(int(m), d, 64)));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) cmin[blockIdx.x] = min(min(sm[0], sm[1]), min(sm[2], sm[3]));
}

// carry[c] = first entry of the first non-empty prefix after chunk c (n if none): a suffix minimum over the chunk
// minima by one workgroup, every thread a run of consecutive chunks (one thread walking all of them took 0.4 ms)
__global__ __launch_bounds__(1024) void k_dir_chunk_carry(const uint32_t* __restrict__ cmin, uint32_t nchunks, uint32_t n,
                                                          uint32_t* __restrict__ carry) {
  __shared__ uint32_t tmin[1024];
  const uint32_t per = (nchunks + 1023) / 1024;
  const uint32_t lo = min(nchunks, threadIdx.x * per), hi = min(nchunks, lo + per);
  uint32_t m = 0xffffffffu;
  for (uint32_t c = lo; c < hi; ++c) m = min(m, cmin[c]);
  tmin[threadIdx.x] = m;
  __syncthreads();
  uint32_t run = n;
  for (uint32_t t = threadIdx.x + 1; t < 1024; ++t) run = min(run, tmin[t]);
  for (uint32_t c = hi; c > lo; --c) {
    carry[c - 1] = run;
    run = min(run, cmin[c - 1]);
  }
}

__global__ __launch_bounds__(256) void k_dir_fill(uint32_t* __restrict__ dir, const uint32_t* __restrict__ carry) {
  __shared__ uint32_t tmin[256];
  uint64_t base = uint64_t(blockIdx.x) * kDirChunk + threadIdx.x * 8;
  uint32_t v[8];
  uint32_t m = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = dir[base + j]; m = min(m, v[j]); }
  tmin[threadIdx.x] = m;
  __syncthreads();
  uint32_t run = carry[blockIdx.x];
  for (int t = 255; t > int(threadIdx.x); --t) run = min(run, tmin[t]);  // values only grow with the prefix
#pragma unroll
  for (int j = 7; j >= 0; --j) {
    run = min(run, v[j]);
    dir[base + j] = run;
  }
}

// The directory in two halves for a caller that writes the heads itself (k_items_entries: dir[first bases of an entry's first
// word] = the entry, for every entry whose first bases differ from the entry's before): room, 0xff and the end mark; then
// the fill of the empty prefixes.
int prefix_dir_prepare(bg_ctx* ctx, uint64_t n, DevBuf<uint32_t>* dir, int* bases_out) {
  const int bases = prefix_dir_bases(n);
  *bases_out = bases;
  const uint64_t nb = 1ull << (2 * bases);  // >= 4^8, a multiple of kDirChunk
  if (n >= 0xffffffffull) { set_error("shard of %llu entries exceeds the prefix directory", (unsigned long long)n); return BG_EINVAL; }
  BG_TRY(dir->alloc(ctx, nb + 1));
  BG_HIP(hipMemsetAsync(dir->p, 0xff, nb * 4, ctx->stream));
  const uint32_t n32 = uint32_t(n);
  BG_HIP(hipMemcpyAsync(dir->p + nb, &n32, 4, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));  // n32 is read by the async copy above
  return BG_OK;
}
int prefix_dir_finish(bg_ctx* ctx, uint64_t n, int bases, DevBuf<uint32_t>* dir) {
  const uint64_t nb = 1ull << (2 * bases);
  const uint32_t nchunks = uint32_t(nb / kDirChunk);
  DevBuf<uint32_t> cmin, carry;
  BG_TRY(cmin.alloc(ctx, nchunks));
  BG_TRY(carry.alloc(ctx, nchunks));
  k_dir_chunk_min<<<nchunks, 256, 0, ctx->stream>>>(dir->p, cmin.p);
  BG_LAUNCH_CHECK();
  k_dir_chunk_carry<<<1, 1024, 0, ctx->stream>>>(cmin.p, nchunks, uint32_t(n), carry.p);
  BG_LAUNCH_CHECK();
  k_dir_fill<<<nchunks, 256, 0, ctx->stream>>>(dir->p, carry.p);
  BG_LAUNCH_CHECK();
  return BG_OK;
}
int build_prefix_dir_keys(bg_ctx* ctx, const uint64_t* keys, uint64_t n, int W, DevBuf<uint32_t>* dir, int* bases_out) {
  BG_TRY(prefix_dir_prepare(ctx, n, dir, bases_out));
  if (n) {
    k_dir_heads<<<cdiv(n, 256), 256, 0, ctx->stream>>>(keys, n, W, *bases_out, dir->p);
    BG_LAUNCH_CHECK();
  }
  return prefix_dir_finish(ctx, n, *bases_out, dir);
}

// lower_bound of (tail, tl) in the entries, searched inside its prefix bucket only
__device__ __forceinline__ uint64_t entries_lower_bound(const uint64_t* __restrict__ ekeys,
                                                        const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                        const uint32_t* __restrict__ dir, int bases,
                                                        const uint64_t* tail, int tl) {
  uint32_t p = uint32_t(tail[0] >> (64 - 2 * bases));
  uint64_t lo = dir[p], hi = dir[p + 1];
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (key_compare(ekeys + mid * W, sizes[mid], tail, tl, W) < 0) lo = mid + 1; else hi = mid;
  }
  (void)n;
  return lo;
}

// marks: one byte per (base, entry); the four planes start a multiple of 64 bytes apart so that k_pack_marks can read
// them 16 bytes a lane
__host__ __device__ inline uint64_t marks_stride(uint64_t n) { return (n + 63) & ~63ull; }

// prev bits (builder.cpp:85-108, closed form of SURVEY row F1): entry s sets
// prev_{s[0]}[lower_bound(entries, s[1:])]; the target must extend s[1:] ("Missing expansion?").
__global__ __launch_bounds__(256) void k_prev_bits(const uint64_t* __restrict__ ekeys,
                                                   const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                   const uint32_t* __restrict__ dir, int bases,
                                                   uint8_t* __restrict__ marks, uint32_t* __restrict__ err) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* s = ekeys + i * W;
  int b = int(s[0] >> 62);
  int tl = int(sizes[i]) - 1;
  uint64_t tail[kMaxWords];
  for (int w = 0; w < W; ++w) tail[w] = window32(s, W, 1 + 32 * w);
  uint64_t lo = entries_lower_bound(ekeys, sizes, n, W, dir, bases, tail, tl);
  if (lo >= n || key_lcp(ekeys + lo * W, sizes[lo], tail, tl, W) < tl) {
    atomicOr(err, 2u);
    return;
  }
  // one byte per (base, target): neighbouring entries hit neighbouring targets, so a shared 64-bit word
  // would serialise 64 same-address atomics per wave; bytes are plain coalesced stores, packed below
  marks[uint64_t(b) * marks_stride(n) + lo] = 1;
}

// target of a tail (s[1:], tl bases) among the entries: the search on the dense two-word prefixes, the
// full-key search where that is not conclusive (see k_prev_bits_pre2).  false = no entry extends the tail.
__device__ __forceinline__ bool prev_target_pre2(const uint64_t* __restrict__ ekeys, const uint64_t* __restrict__ pre2,
                                                 const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                 const uint32_t* __restrict__ dir, int bases, const uint64_t* tail, int tl,
                                                 int verify, uint32_t* __restrict__ err, uint64_t* out) {
  const uint64_t t0 = tail[0], t1 = W > 1 ? tail[1] : 0;
  uint64_t lo = 0;
  bool sure = false;
  if (tl >= 64) {
    const uint32_t p = uint32_t(t0 >> (64 - 2 * bases));
    // these look-ups run into the memory pipeline's address rate: one 16-byte load per prefix, not two of 8
    const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(pre2);
    uint64_t a = dir[p], z = dir[p + 1];
    while (a < z) {  // first entry whose two leading words are >= the tail's
      const uint64_t mid = (a + z) >> 1;
      const ulonglong2 m = p2[mid];
      if (m.x < t0 || (m.x == t0 && m.y < t1)) a = mid + 1; else z = mid;
    }
    lo = a;
    if (lo < n) {
      const ulonglong2 c0 = p2[lo];
      sure = c0.x == t0 && c0.y == t1;
      if (sure && lo + 1 < n) {
        const ulonglong2 c1 = p2[lo + 1];
        sure = c1.x != t0 || c1.y != t1;
      }
    }
  }
  if (!sure || verify) {
    const uint64_t full = entries_lower_bound(ekeys, sizes, n, W, dir, bases, tail, tl);
    if (sure && full != lo) atomicOr(err, 4u);  // the shortcut disagrees with the full search
    lo = full;
    if (lo >= n || key_lcp(ekeys + lo * W, sizes[lo], tail, tl, W) < tl) {
      atomicOr(err, 2u);
      return false;
    }
  }
  *out = lo;
  return true;
}

// The same with the search run on the dense two-word prefixes (16 bytes per entry instead of W*8: the
// candidates of one search share a line or two, and neighbouring threads' windows overlap).  A tail of at
// least 64 bases whose two leading words occur in exactly one entry has found its target; anything else
// (short tails, entries that agree on their first 64 bases) takes the full-key search.  `verify` re-checks
// the unambiguous targets against the full keys as well (BG_VERIFY_CLOSURE=1; the tests run both ways).
__global__ __launch_bounds__(256) void k_prev_bits_pre2(const uint64_t* __restrict__ ekeys,
                                                        const uint64_t* __restrict__ pre2,
                                                        const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                        const uint32_t* __restrict__ dir, int bases,
                                                        uint8_t* __restrict__ marks, uint32_t* __restrict__ err,
                                                        int verify) {
  // XCD-aware: workgroups b, b + 8, ... share an L2, so each XCD takes one contiguous eighth of the entries —
  // neighbouring entries search neighbouring targets
  uint64_t i = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* s = ekeys + i * W;
  const int tl = int(sizes[i]) - 1;
  // the key one base on: every word of the key is fetched once (window32 per word fetched each twice, and these
  // look-ups are bound by the number of requests)
  uint64_t key[kMaxWords + 1];
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w) key[w] = w < W ? s[w] : 0;
  key[kMaxWords] = 0;
  const int b = int(key[0] >> 62);
  uint64_t tail[kMaxWords];
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w) tail[w] = (key[w] << 2) | (key[w + 1] >> 62);
  uint64_t lo;
  // (fetching only the three words the two-word search needs and the rest on demand: 5.6 -> 6.4 ms)
  if (!prev_target_pre2(ekeys, pre2, sizes, n, W, dir, bases, tail, tl, verify, err, &lo)) return;
  marks[uint64_t(b) * marks_stride(n) + lo] = 1;
}

// ---- prev bits as ONE merge (round 4).  The entries that start with base b are a contiguous stretch of the sorted entries
// and their tails s[1:] ascend with them, so their targets lower_bound(entries, s[1:]) ascend too (and are pairwise distinct):
// instead of n directory + binary searches of dependent random probes (the 5.6 ms of k_prev_bits_pre2), a workgroup takes
// kPmChunk consecutive ENTRIES as targets, stages their two-word prefixes in LDS once, and streams through the four stretches
// of tails -- one per base -- whose targets fall among them.  Where those stretches begin is settled beforehand by a full-key
// binary search per (chunk, base) (k_pm_bounds: tails of base b that are greater than the last entry before the chunk), so
// everything a tail needs is in the chunk: its rows sit in LDS (kPmChunk x W words) and the tail's target is a full-key
// binary search there, checked to extend the tail ("Missing expansion?", builder.cpp:95-96) -- every target, not only the
// ones BG_VERIFY_CLOSURE used to ask for.  (First version: the two-word test of k_prev_bits_pre2 on prefixes staged in
// LDS, the full-key search of the 9 % it does not settle in the chunk's rows in global memory -- 5.95 ms against 6.46: one
// lane in eleven walking ten dependent loads holds its whole wave.)  The chunk's bits are put together in LDS and leave
// as whole words: no marks array, no directory.  builder.cpp:85-108.
constexpr int kPmChunk = 1024;

__global__ void k_pm_first_base(const uint64_t* __restrict__ ekeys, uint64_t n, int W, uint64_t* __restrict__ F) {
  const int b = threadIdx.x;
  if (b > 4) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {  // first entry whose first base is >= b
    const uint64_t mid = (lo + hi) >> 1;
    if (int(ekeys[mid * W] >> 62) < b) lo = mid + 1; else hi = mid;
  }
  F[b] = b == 4 ? n : lo;
}

__device__ __forceinline__ void pm_tail(const uint64_t* __restrict__ s, int W, uint64_t* tail) {
  uint64_t key[kMaxWords + 1];
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w) key[w] = w < W ? s[w] : 0;
  key[kMaxWords] = 0;
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w) tail[w] = (key[w] << 2) | (key[w + 1] >> 62);
}

// bounds[g * 4 + b], g = 0 .. G: the first entry of base b whose target lies at or beyond entry g * kPmChunk
__global__ void k_pm_bounds(const uint64_t* __restrict__ ekeys, const uint8_t* __restrict__ sizes, uint64_t n, int W,
                            const uint64_t* __restrict__ F, uint64_t G, uint64_t* __restrict__ bounds) {
  const uint64_t idx = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= (G + 1) * 4) return;
  const uint64_t g = idx >> 2;
  const int b = int(idx & 3);
  if (g == 0) { bounds[idx] = F[b]; return; }
  if (g == G) { bounds[idx] = F[b + 1]; return; }
  const uint64_t e = g * kPmChunk - 1;  // target >= g * kPmChunk  <=>  tail > entry e
  uint64_t ref[kMaxWords];
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w) ref[w] = w < W ? ekeys[e * W + w] : 0;
  const int rl = sizes[e];
  uint64_t lo = F[b], hi = F[b + 1];
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    uint64_t tail[kMaxWords];
    pm_tail(ekeys + mid * W, W, tail);
    if (key_compare(tail, int(sizes[mid]) - 1, ref, rl, W) <= 0) lo = mid + 1; else hi = mid;
  }
  bounds[idx] = lo;
}

__global__ __launch_bounds__(256) void k_prev_merge(const uint64_t* __restrict__ ekeys, const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                    const uint64_t* __restrict__ bounds, unsigned long long* __restrict__ bA,
                                                    unsigned long long* __restrict__ bC, unsigned long long* __restrict__ bG,
                                                    unsigned long long* __restrict__ bT, uint32_t* __restrict__ err) {
  extern __shared__ uint64_t pm_keys[];  // kPmChunk rows of W words: the chunk's entries, then their sizes
  uint8_t* pm_len = reinterpret_cast<uint8_t*>(pm_keys + kPmChunk * W);
  __shared__ unsigned long long sbits[4][kPmChunk / 64];
  const uint64_t g = blockIdx.x;
  const uint64_t e0 = g * kPmChunk;
  const int m = int(n - e0 < uint64_t(kPmChunk) ? n - e0 : uint64_t(kPmChunk));
  const int tid = threadIdx.x;
  // where this chunk's four stretches of tails begin and end (eight loads in flight, not one behind the other)
  uint64_t blo[4], bhi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) { blo[b] = bounds[g * 4 + b]; bhi[b] = bounds[(g + 1) * 4 + b]; }
  // the chunk's rows are consecutive: one coalesced stream, eight 16-byte loads of a thread in flight at a time (a loop
  // of load -> store waits a memory round trip per iteration: 20 of them made the kernel 7 ms long)
  {
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(ekeys + e0 * W);
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(pm_keys);
    const int pairs = (m * W) >> 1;
    for (int x0 = 0; x0 < pairs; x0 += 8 * 256) {
      ulonglong2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = x0 + u * 256 + tid;
        if (x < pairs) v[u] = src[x];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = x0 + u * 256 + tid;
        if (x < pairs) dst[x] = v[u];
      }
    }
    if (((m * W) & 1) && tid == 0) pm_keys[m * W - 1] = ekeys[e0 * W + uint64_t(m * W - 1)];
  }
  for (int x = tid; x < m; x += 256) pm_len[x] = sizes[e0 + x];
  if (tid < 4 * (kPmChunk / 64)) (&sbits[0][0])[tid] = 0;
  __syncthreads();
  // The tails: rounds of up to 256 consecutive entries of one base.  Their rows are one contiguous stretch of words, which
  // the workgroup fetches word by word (thread t: words t, t + 256, ... -- whole lines per instruction; a thread fetching
  // its own 40-byte row asks for 20 lines per instruction, five times over, and these kernels are bound by requests) into a
  // staging tile in LDS; the next round's words travel while this round is searched.
  uint64_t* stage = reinterpret_cast<uint64_t*>(pm_len + kPmChunk);  // 256 rows of W words, then their sizes
  uint8_t* stage_len = reinterpret_cast<uint8_t*>(stage + 256 * W);
  // the rounds in order: (base, first row, rows)
  int rb = 0;
  uint64_t ri = blo[0];
  auto skip_empty = [&]() { while (rb < 4 && ri >= bhi[rb]) { ++rb; if (rb < 4) ri = blo[rb]; } };
  skip_empty();
  uint64_t pre[kMaxWords + 1];
  int pre_len = 0;
  auto fetch = [&](uint64_t row0, int cnt) {  // this thread's words of the round [row0, row0 + cnt)
    const int words = cnt * W;
#pragma unroll
    for (int u = 0; u < kMaxWords + 1; ++u) {
      const int x = u * 256 + tid;
      pre[u] = x < words ? ekeys[row0 * W + x] : 0;
    }
    pre_len = tid < cnt ? sizes[row0 + tid] : 0;
  };
  int cur_b = rb, cur_cnt = 0;
  uint64_t cur_row = ri;
  if (rb < 4) {
    cur_cnt = int(bhi[rb] - ri < 256 ? bhi[rb] - ri : 256);
    fetch(cur_row, cur_cnt);
  }
  while (cur_cnt) {
    // what was fetched goes to the tile
    const int words = cur_cnt * W;
#pragma unroll
    for (int u = 0; u < kMaxWords + 1; ++u) {
      const int x = u * 256 + tid;
      if (x < words) stage[x] = pre[u];
    }
    if (tid < cur_cnt) stage_len[tid] = uint8_t(pre_len);
    // the next round
    ri = cur_row + cur_cnt;
    rb = cur_b;
    skip_empty();
    const int nb = rb;
    const uint64_t nrow = ri;
    const int ncnt = rb < 4 ? int(bhi[rb] - ri < 256 ? bhi[rb] - ri : 256) : 0;
    __syncthreads();
    if (ncnt) fetch(nrow, ncnt);
    if (tid < cur_cnt) {
      uint64_t key[kMaxWords + 1];
#pragma unroll
      for (int w = 0; w < kMaxWords; ++w) key[w] = w < W ? stage[tid * W + w] : 0;
      key[kMaxWords] = 0;
      const int tl = int(stage_len[tid]) - 1;
      uint64_t tail[kMaxWords];
#pragma unroll
      for (int w = 0; w < kMaxWords; ++w) tail[w] = (key[w] << 2) | (key[w + 1] >> 62);
      int a = 0, z = m;
      while (a < z) {  // first entry of the chunk that is >= the tail: its target (k_pm_bounds put it inside the chunk)
        const int mid = (a + z) >> 1;
        const uint64_t* row = pm_keys + mid * W;
        // (words, length) order with every index a constant: a loop over W indexes the tail's registers by a variable and
        // sends them to scratch memory -- with key_compare here the kernel took 8.6 ms
        int c = 0;
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
          if (w < W && c == 0) {
            const uint64_t x = row[w];
            c = x < tail[w] ? -1 : (x > tail[w] ? 1 : 0);
          }
        if (c == 0) c = int(pm_len[mid]) < tl ? -1 : (int(pm_len[mid]) > tl ? 1 : 0);
        if (c < 0) a = mid + 1; else z = mid;
      }
      bool ok = a < m;
      if (ok) {  // the target must extend the tail: lcp >= tl
        const uint64_t* row = pm_keys + a * W;
        int l = 0;
        bool more = true;
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
          if (w < W && more) {
            const int c = lcp_word(row[w], tail[w]);
            l += c;
            more = c == 32;
          }
        const int mn = int(pm_len[a]) < tl ? int(pm_len[a]) : tl;
        ok = (l < mn ? l : mn) >= tl;
      }
      if (!ok)
        atomicOr(err, 2u);  // "Missing expansion?" (builder.cpp:95-96): every target is checked, as BG_VERIFY_CLOSURE used to ask for
      else
        atomicOr(&sbits[cur_b][a >> 6], 1ull << (a & 63));
    }
    __syncthreads();  // the tile is free again
    cur_b = nb;
    cur_row = nrow;
    cur_cnt = ncnt;
  }
  __syncthreads();
  const uint64_t nwords = (n + 63) / 64;
  if (tid < 4 * (kPmChunk / 64)) {
    const int b = tid / (kPmChunk / 64), w = tid % (kPmChunk / 64);
    const uint64_t word = e0 / 64 + w;
    unsigned long long* bits = b == 0 ? bA : b == 1 ? bC : b == 2 ? bG : bT;
    if (word < nwords) bits[word] = sbits[b][w];
  }
}

// marks[b*n + t] -> bit t of prev_b (OR-ed into the existing words)
constexpr int kPackWordsPerWave = 16;

__global__ __launch_bounds__(256) void k_pack_marks(const uint8_t* __restrict__ marks, uint64_t n,
                                                    unsigned long long* __restrict__ bA,
                                                    unsigned long long* __restrict__ bC,
                                                    unsigned long long* __restrict__ bG,
                                                    unsigned long long* __restrict__ bT) {
  const uint64_t nwords = (n + 63) / 64;
  const uint64_t chunks = (nwords + kPackWordsPerWave - 1) / kPackWordsPerWave;  // per base
  uint64_t g = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);  // one wave per 16 output words of one base
  if (g >= 4 * chunks) return;
  const int b = int(g / chunks);
  const uint64_t w0 = (g - uint64_t(b) * chunks) * kPackWordsPerWave;
  const int lane = threadIdx.x & 63;
  unsigned long long* bits = b == 0 ? bA : b == 1 ? bC : b == 2 ? bG : bT;
  const uint64_t stride = marks_stride(n);
  // a lane takes 16 marks (bytes 0 / 1) with one 16-byte load — a byte a lane moved 64 bytes per load instruction —
  // and four lanes put one output word together
  const uint64_t t0 = w0 * 64 + uint64_t(lane) * 16;
  uint4 x = make_uint4(0, 0, 0, 0);
  if (t0 < stride) x = *reinterpret_cast<const uint4*>(marks + uint64_t(b) * stride + t0);
  auto nib = [](uint32_t v) { v &= 0x01010101u; return (v | (v >> 7) | (v >> 14) | (v >> 21)) & 0xfu; };
  unsigned long long v = nib(x.x) | (nib(x.y) << 4) | (nib(x.z) << 8) | (nib(x.w) << 12);
  v |= __shfl_down(v, 1, 64) << 16;
  v |= __shfl_down(v, 2, 64) << 32;
  const uint64_t word = w0 + uint64_t(lane >> 2);
  if ((lane & 3) == 0 && word < nwords && v) bits[word] |= v;
}

// The tails k_gather_dedup_build could not settle where the records meet (rows of W + 1 words: the tail, then
// length | base << 8 | checked << 16): prev_base[first entry that extends the tail], and for a checked one only if
// base + tail is an entry itself.  emask: one byte per entry, bit b = prev_b.
__global__ __launch_bounds__(256) void k_prev_list(const uint64_t* __restrict__ rows, uint64_t nt,
                                                   const uint64_t* __restrict__ ekeys, const uint64_t* __restrict__ pre2,
                                                   const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                   const uint32_t* __restrict__ dir, int bases, uint8_t* __restrict__ emask,
                                                   uint32_t* __restrict__ err, int verify) {
  const uint64_t t = xcd_chunk(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const uint64_t* row = rows + t * uint64_t(W + 1);
  uint64_t tail[kMaxWords];
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w) tail[w] = w < W ? row[w] : 0;
  const uint64_t meta = row[W];
  const int tl = int(meta & 0xff), b = int((meta >> 8) & 3);
  uint64_t lo;
  if (!prev_target_pre2(ekeys, pre2, sizes, n, W, dir, bases, tail, tl, verify, err, &lo)) return;
  if (meta & (1ull << 16)) {
    uint64_t key[kMaxWords];
#pragma unroll
    for (int w = 0; w < kMaxWords; ++w) key[w] = w < W ? ((w ? tail[w - 1] << 62 : uint64_t(b) << 62) | (tail[w] >> 2)) : 0;
    const int kl = tl + 1;
    const uint64_t at = entries_lower_bound(ekeys, sizes, n, W, dir, bases, key, kl);
    if (at >= n || int(sizes[at]) != kl || key_lcp(ekeys + at * W, kl, key, kl, W) < kl) return;
  }
  atomicOr(reinterpret_cast<unsigned int*>(emask + (lo & ~3ull)), (1u << b) << (8 * unsigned(lo & 3)));
}

// emask -> the four prev bit vectors (one wave per 16 words of each)
__global__ __launch_bounds__(256) void k_pack_emask(const uint8_t* __restrict__ emask, uint64_t n,
                                                    unsigned long long* __restrict__ bA, unsigned long long* __restrict__ bC,
                                                    unsigned long long* __restrict__ bG, unsigned long long* __restrict__ bT) {
  const uint64_t nwords = (n + 63) / 64;
  const uint64_t g = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < kPackWordsPerWave; ++k) {
    const uint64_t word = g * kPackWordsPerWave + k;
    if (word >= nwords) return;
    const uint64_t e = word * 64 + lane;
    const uint32_t m = e < n ? emask[e] : 0u;
    const unsigned long long a = __ballot(m & 1u), c = __ballot(m & 2u), gg = __ballot(m & 4u), tt = __ballot(m & 8u);
    if (lane == 0) { bA[word] = a; bC[word] = c; bG[word] = gg; bT[word] = tt; }
  }
}

// tails received from other shards (rows of W+1 words: s[1:], (len-1) | base << 8) -> prev bits
__global__ __launch_bounds__(256) void k_apply_tails(const uint64_t* __restrict__ rows, uint64_t nt,
                                                     const uint64_t* __restrict__ ekeys,
                                                     const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                     const uint32_t* __restrict__ dir, int bases,
                                                     uint8_t* __restrict__ marks, uint32_t* __restrict__ err) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const uint64_t* tail = rows + t * (W + 1);
  int tl = int(tail[W] & 0xff), b = int((tail[W] >> 8) & 3);
  uint64_t lo = entries_lower_bound(ekeys, sizes, n, W, dir, bases, tail, tl);
  if (lo >= n || key_lcp(ekeys + lo * W, sizes[lo], tail, tl, W) < tl) { atomicOr(err, 2u); return; }
  marks[uint64_t(b) * marks_stride(n) + lo] = 1;
}

// bitcount::finalize (io/bitcount.cpp:84-124): one thread per 512-bit group.
__global__ void k_bitcount_groups(const uint64_t* __restrict__ bits, uint64_t nwords, uint64_t ngroups,
                                  uint64_t* __restrict__ subaccum, uint32_t* __restrict__ group_pop) {
  uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  uint64_t sub = 0;
  uint32_t tot = 0;
  for (int j = 0; j < 8; ++j) {
    uint64_t wi = g * 8 + j;
    uint32_t pc = wi < nwords ? uint32_t(__popcll(bits[wi])) : 0u;
    sub = (sub << 8) | pc;
    tot += pc;
  }
  subaccum[g] = sub;
  group_pop[g] = tot;
}

// less_than_search index (io/int_map_interface.cpp:87-126): min of shared[] per 64 and per 4096
__global__ void k_min_index1(const uint8_t* __restrict__ shared, uint64_t n, uint64_t n1, uint8_t* __restrict__ min1) {
  uint64_t f = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (f >= n1) return;
  uint64_t lo = f * 64, hi = lo + 64 < n ? lo + 64 : n;
  unsigned m = 255;
  if (hi - lo == 64) {  // four 16-byte loads (the array is 64-byte aligned), minimum of the 64 bytes
    const uint4* q = reinterpret_cast<const uint4*>(shared + lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 v = q[j];
      for (uint32_t x : {v.x, v.y, v.z, v.w}) {
        m = min(m, x & 0xffu);
        m = min(m, (x >> 8) & 0xffu);
        m = min(m, (x >> 16) & 0xffu);
        m = min(m, x >> 24);
      }
    }
  } else {
    for (uint64_t p = lo; p < hi; ++p) m = shared[p] < m ? shared[p] : m;
  }
  min1[f] = uint8_t(m);
}
__global__ void k_min_index2(const uint8_t* __restrict__ min1, uint64_t n1, uint64_t n2, uint8_t* __restrict__ min2) {
  uint64_t f = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (f >= n2) return;
  uint64_t lo = f * 64, hi = lo + 64 < n1 ? lo + 64 : n1;
  unsigned m = 255;
  for (uint64_t p = lo; p < hi; ++p) m = min1[p] < m ? min1[p] : m;
  min2[f] = uint8_t(m);
}

// ------------------------------------------------------------------------------------------------

// Allocates and fills subaccum/accum/fixed and the min index from bits/shared (device).
// slice: the entries are a stretch of a larger seqset (prev bits point anywhere): fixed[] = the slice's own
// cumulative popcounts, no sum check.
// rank blocks of SeqsetView::count: block g of base b covers entries [384 g, 384 g + 384)
__global__ void k_rank64_build(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ subaccum,
                               const uint64_t* __restrict__ accum, uint64_t n, uint64_t nblocks, uint64_t* __restrict__ out) {
  const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= nblocks) return;
  const uint64_t nw = (n + 63) >> 6, ngroups = (n + 511) >> 9;
  const uint64_t i0 = g * 384;
  // set bits before entry i0, through the finalized bitcount (bitcount.h:218-270; i0 is a multiple of 64: no partial word)
  uint64_t before = 0;
  {
    const uint64_t group = i0 >> 9, sub = (i0 >> 6) & 7;
    uint64_t sa = group < ngroups ? subaccum[group] : 0;
    sa >>= 56 - 8 * sub;
    sa &= 0xffffffffffffff00ull;
    sa = (sa >> 32) + (sa & 0xffffffffull);
    sa = (sa >> 16) + (sa & 0xffff);
    sa = (sa >> 8) + (sa & 0xff);
    before = accum[group] + sa;
  }
  uint64_t w[6], packed = 0;
  unsigned run = 0;
  for (int k = 0; k < 6; ++k) {
    const uint64_t idx = g * 6 + k;
    w[k] = idx < nw ? bits[idx] : 0;
    run += unsigned(__popcll(w[k]));
    if (k < 5) packed |= uint64_t(run) << (9 * k);  // set bits of the block before word k + 1
  }
  uint64_t* o = out + g * 8;
  o[0] = before;
  o[1] = packed;
  for (int k = 0; k < 6; ++k) o[2 + k] = w[k];
}

// rank4 of SeqsetView (bg_device_funcs.h): group counts first, then the blocks
struct RankSrc {  // the finalized bitcount of the four bases
  const uint64_t* bits[4];
  const uint64_t* subaccum[4];
  const uint64_t* accum[4];
  uint64_t n;
  __device__ uint64_t count(int b, uint64_t i) const {  // bitcount.h:218-270 (i a multiple of 32)
    const uint64_t nw = (n + 63) >> 6, ngroups = (n + 511) >> 9;
    uint64_t x = (i >> 6) < nw ? bits[b][i >> 6] : 0;
    x &= (1ull << (i & 63)) - 1;
    const uint64_t group = i >> 9, sub = (i >> 6) & 7;
    uint64_t sa = group < ngroups ? subaccum[b][group] : 0;
    sa >>= 56 - 8 * sub;
    sa &= 0xffffffffffffff00ull;
    sa = (sa >> 32) + (sa & 0xffffffffull);
    sa = (sa >> 16) + (sa & 0xffff);
    sa = (sa >> 8) + (sa & 0xff);
    return accum[b][group] + sa + uint64_t(__popcll(x));
  }
};
__global__ void k_rank4_super(RankSrc src, uint64_t ngroups16, uint64_t* __restrict__ super) {
  const uint64_t x = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (x >= ngroups16 * 4) return;
  super[x] = src.count(int(x & 3), (x >> 2) * 65536 * 96);
}
__global__ void k_rank4_build(RankSrc src, uint64_t nblocks, const uint64_t* __restrict__ super, uint32_t* __restrict__ out) {
  const uint64_t x = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // (block, base)
  if (x >= nblocks * 4) return;
  const uint64_t g = x >> 2;
  const int b = int(x & 3);
  const uint64_t i0 = g * 96, nw = (src.n + 63) >> 6;
  const uint64_t wi = i0 >> 6;
  const uint64_t a = wi < nw ? src.bits[b][wi] : 0, c = wi + 1 < nw ? src.bits[b][wi + 1] : 0;
  uint32_t w0, w1, w2;
  if (i0 & 32) { w0 = uint32_t(a >> 32); w1 = uint32_t(c); w2 = uint32_t(c >> 32); }
  else { w0 = uint32_t(a); w1 = uint32_t(a >> 32); w2 = uint32_t(c); }
  uint4 q;
  q.x = uint32_t(src.count(b, i0) - super[(g >> 16) * 4 + b]);
  q.y = w0; q.z = w1; q.w = w2;
  *reinterpret_cast<uint4*>(out + x * 4) = q;
}

// walk blocks of SeqsetView (bg_device_funcs.h): group counts first, then a thread a word of a block
__global__ void k_wblk_super(RankSrc src, uint64_t ngroups16, uint64_t* __restrict__ super) {
  const uint64_t x = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (x >= ngroups16 * 4) return;
  super[x] = src.count(int(x & 3), (x >> 2) * 65536 * 40);
}
__global__ void k_wblk_build(RankSrc src, const uint8_t* __restrict__ sizes, const uint8_t* __restrict__ shared, uint64_t nblocks,
                             const uint64_t* __restrict__ super, uint64_t* __restrict__ out) {
  const uint64_t x = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // (block, word)
  if (x >= nblocks * 16) return;
  const uint64_t g = x >> 4, i0 = g * 40;
  const int t = int(x & 15);
  uint64_t v = 0;
  if (t < 4) {  // the block's 40 prev bits of base t (i0 is a multiple of 8: they lie in at most two words) under their running count
    const uint64_t nw = (src.n + 63) >> 6, wi = i0 >> 6;
    const unsigned sh = unsigned(i0 & 63);
    const uint64_t a = wi < nw ? src.bits[t][wi] : 0, c = wi + 1 < nw ? src.bits[t][wi + 1] : 0;
    const uint64_t bits40 = ((a >> sh) | (sh ? c << (64 - sh) : 0)) & ((1ull << 40) - 1);
    v = ((src.count(t, i0) - super[(g >> 16) * 4 + t]) << 40) | bits40;
  } else if (t < 14) {  // five words of sizes, five of shared (the arrays are 8-byte aligned, their allocations multiples of 512 bytes)
    const uint8_t* a = t < 9 ? sizes : shared;
    const uint64_t wi = (i0 >> 3) + uint64_t(t < 9 ? t - 4 : t - 9);
    if (wi * 8 < src.n) v = reinterpret_cast<const uint64_t*>(a)[wi];
  }
  out[x] = v;
}

void seqset_ensure_walk_blocks(const bg_seqset* css) {
  bg_seqset* ss = const_cast<bg_seqset*>(css);
  if (!ss || ss->wblk_state.load(std::memory_order_acquire) != 0) return;
  std::lock_guard<std::mutex> lk(ss->keys_mutex);
  if (ss->wblk_state.load(std::memory_order_relaxed) != 0) return;
  bg_ctx* ctx = ss->ctx;
  const uint64_t n = ss->n;
  int state = 2;
  if (ctx && n && n < (1ull << 35) && ss->sizes.p && ss->shared.p && ss->bits[0].p && ss->accum[0].p && ss->subaccum[0].p) {
    const uint64_t nblocks = n / 40 + 1, ngroups16 = (nblocks + 65535) / 65536;  // position n itself is asked for
    RankSrc src;
    for (int b = 0; b < 4; ++b) { src.bits[b] = ss->bits[b].p; src.subaccum[b] = ss->subaccum[b].p; src.accum[b] = ss->accum[b].p; }
    src.n = n;
    DevBuf<uint64_t> blk, super;
    if (hipSetDevice(ctx->device) == hipSuccess && super.alloc(ctx, ngroups16 * 4) == BG_OK && blk.alloc(ctx, nblocks * 16) == BG_OK) {
      k_wblk_super<<<cdiv(ngroups16 * 4, 64), 64, 0, ctx->stream>>>(src, ngroups16, super.p);
      k_wblk_build<<<cdiv(nblocks * 16, 256), 256, 0, ctx->stream>>>(src, ss->sizes.p, ss->shared.p, nblocks, super.p, blk.p);
      if (hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess) {
        ss->wblk = std::move(blk);
        ss->wblk_super = std::move(super);
        state = 1;
      }
    }
  }
  ss->wblk_state.store(state, std::memory_order_release);
}

// sel256 of SeqsetView from the rank blocks: block g holds the set bits number [before, before + ones) of its base; for every
// multiple of 256 among them, the entry that carries it
__global__ void k_sel256_build(const uint64_t* __restrict__ rank64, uint64_t nblocks, uint64_t n, uint64_t n_slots, uint64_t* __restrict__ out) {
  const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= nblocks) return;
  const uint64_t* blk = rank64 + g * 8;
  const uint64_t before = blk[0];
  unsigned cum[7];
  cum[0] = 0;
  for (int k = 0; k < 5; ++k) cum[k + 1] = unsigned((blk[1] >> (9 * k)) & 0x1ff);
  cum[6] = cum[5] + unsigned(__popcll(blk[7]));
  const uint64_t ones = cum[6];
  for (uint64_t m = (before + 255) & ~uint64_t(255); m < before + ones; m += 256) {
    unsigned need = unsigned(m - before);  // the need-th set bit of the block (0-based)
    int w = 0;
    while (w < 5 && cum[w + 1] <= need) ++w;
    need -= cum[w];
    uint64_t x = blk[2 + w];
    for (unsigned k = 0; k < need; ++k) x &= x - 1;
    const uint64_t pos = g * 384 + uint64_t(w) * 64 + uint64_t(__ffsll((long long)x) - 1);
    if (m / 256 < n_slots && pos < n) out[m / 256] = pos;
  }
}
__global__ void k_fill_u64(uint64_t* __restrict__ p, uint64_t n, uint64_t v) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

int seqset_finalize_device(bg_ctx* ctx, bg_seqset* ss, bool slice) {
  const uint64_t n = ss->n;
  ss->wblk.reset();  // (walk blocks of arrays that are being made again)
  ss->wblk_super.reset();
  ss->wblk_state.store(slice ? 2 : 0);
  const uint64_t nwords = (n + 63) / 64, ngroups = (n + 511) / 512, naccum = (n + 1 + 511) / 512;
  DevBuf<uint32_t> gp;
  DevBuf<uint64_t> total;
  BG_TRY(gp.alloc(ctx, ngroups));
  BG_TRY(total.alloc(ctx, 4));
  uint64_t h_total[4] = {0, 0, 0, 0};
  for (int b = 0; b < 4; ++b) {
    BG_TRY(ss->subaccum[b].alloc(ctx, ngroups));
    BG_TRY(ss->accum[b].alloc(ctx, naccum));
    BG_HIP(hipMemsetAsync(ss->accum[b].p, 0, naccum * 8, ctx->stream));
    if (ngroups) {
      k_bitcount_groups<<<cdiv(ngroups, 256), 256, 0, ctx->stream>>>(ss->bits[b].p, nwords, ngroups,
                                                                    ss->subaccum[b].p, gp.p);
      BG_LAUNCH_CHECK();
    }
    BG_TRY(exclusive_scan_u32_to_u64(ctx, gp.p, ss->accum[b].p, ngroups, total.p + b));
    if (naccum > ngroups)  // n % 512 == 0: the extra slot holds the grand total (bitcount.cpp:118-121)
      BG_HIP(hipMemcpyAsync(ss->accum[b].p + ngroups, total.p + b, 8, hipMemcpyDeviceToDevice, ctx->stream));
  }
  BG_HIP(hipMemcpyAsync(h_total, total.p, 32, hipMemcpyDeviceToHost, ctx->stream));
  uint64_t n1 = (n + 63) / 64, n2 = (n1 + 63) / 64;
  BG_TRY(ss->min1.alloc(ctx, n1));
  BG_TRY(ss->min2.alloc(ctx, n2));
  if (n) {
    k_min_index1<<<cdiv(n1, 256), 256, 0, ctx->stream>>>(ss->shared.p, n, n1, ss->min1.p);
    BG_LAUNCH_CHECK();
    k_min_index2<<<cdiv(n2, 256), 256, 0, ctx->stream>>>(ss->min1.p, n1, n2, ss->min2.p);
    BG_LAUNCH_CHECK();
  }
  if (!slice) {  // (a slice of a sharded seqset is written, not walked)
    const uint64_t nblocks = n / 384 + 1;  // position n itself is asked for (the end of the whole range)
    for (int b = 0; b < 4; ++b) {
      BG_TRY(ss->rank64[b].alloc(ctx, nblocks * 8));
      k_rank64_build<<<cdiv(nblocks, 256), 256, 0, ctx->stream>>>(ss->bits[b].p, ss->subaccum[b].p, ss->accum[b].p, n, nblocks,
                                                                ss->rank64[b].p);
      BG_LAUNCH_CHECK();
    }
  }
  if (!slice && !getenv("BG_NO_RANK4")) {
    const uint64_t nblocks = n / 96 + 1, ngroups16 = (nblocks + 65535) / 65536;
    RankSrc src;
    for (int b = 0; b < 4; ++b) { src.bits[b] = ss->bits[b].p; src.subaccum[b] = ss->subaccum[b].p; src.accum[b] = ss->accum[b].p; }
    src.n = n;
    BG_TRY(ss->rank4_super.alloc(ctx, ngroups16 * 4));
    BG_TRY(ss->rank4.alloc(ctx, nblocks * 16));
    k_rank4_super<<<cdiv(ngroups16 * 4, 64), 64, 0, ctx->stream>>>(src, ngroups16, ss->rank4_super.p);
    BG_LAUNCH_CHECK();
    k_rank4_build<<<cdiv(nblocks * 4, 256), 256, 0, ctx->stream>>>(src, nblocks, ss->rank4_super.p, ss->rank4.p);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (!slice) {  // (the totals are on the host now: one slot per 256 set bits, one more for the sentinel)
    const uint64_t nblocks = n / 384 + 1;
    for (int b = 0; b < 4; ++b) {
      const uint64_t n_slots = h_total[b] / 256 + 2;
      BG_TRY(ss->sel256[b].alloc(ctx, n_slots));
      k_fill_u64<<<cdiv(n_slots, 256), 256, 0, ctx->stream>>>(ss->sel256[b].p, n_slots, n);
      BG_LAUNCH_CHECK();
      k_sel256_build<<<cdiv(nblocks, 256), 256, 0, ctx->stream>>>(ss->rank64[b].p, nblocks, n, n_slots, ss->sel256[b].p);
      BG_LAUNCH_CHECK();
    }
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  uint64_t off = 0;
  for (int b = 0; b < 4; ++b) { ss->fixed[b] = off; off += h_total[b]; }  // seqset.cpp:113-129
  ss->fixed[4] = off;
  if (!slice && off != n) {
    set_error("Invalid seqset in finalize: %llu != %llu", (unsigned long long)off, (unsigned long long)n);
    return BG_EINTERNAL;
  }
  return BG_OK;
}

}  // namespace bg


namespace bg {

// Cover index over n_o oriented reads (owords: n_o x W zero-padded words, olen: their lengths).  The reads are
// sorted by their full key (the machinery of the record sort): reads with the same first K-mer are then
// consecutive AND in key order, so a look-up among the candidates of one K-mer is a binary search — a repeat
// family puts thousands of reads behind one K-mer, and scanning them made the look-ups quadratic
// (tools/stress_repeats.py: 3.5 s for a 20 Mb genome).  k0o (optional, consumed): sort keys written by whoever
// made the oriented words.  The store keeps pointers into owords / olen: they must outlive it.
static bool cover_by_items(uint64_t n_o, int cover_k) {
  const char* e = getenv("BG_COVER_ITEMS");
  return !(e && e[0] == '0') && n_o && n_o < (1ull << 31) && cover_k <= 32;
}
int cover_index_build(bg_ctx* ctx, const uint64_t* owords, const uint8_t* olen, uint64_t n_o, int W, int cover_k,
                      DevBuf<uint64_t>* k0o, CoverStore* cs) {
  if (n_o >= (1ull << 31)) { set_error("too many reads for one cover index (%llu oriented)", (unsigned long long)n_o); return BG_EINVAL; }
  DevBuf<uint32_t> head_pos;
  uint64_t m = 0, nheads = 0;  // reads that can act as a cover, distinct first K-mers
  {
    DevBuf<uint32_t> sidx_all, keep;
    DevBuf<uint8_t> flags;
    // the reads in order through 16-byte items (items_build.hip; BG_COVER_ITEMS=0: through sort_records_impl's rows order and
    // two gathers, as before)
    bool by_items = false;
    if (cover_by_items(n_o, cover_k)) BG_TRY(bg::items_cover_order(ctx, owords, olen, n_o, W, cover_k, k0o, &cs->skeys, &cs->soid, &m, &by_items));
    BG_TRY(flags.alloc(ctx, n_o));
    if (!by_items) {
      BG_TRY(sort_records_impl(ctx, owords, olen, n_o, W, k0o, &sidx_all, nullptr));
      if (n_o) {
        k_flag_long_enough<<<cdiv(n_o, 256), 256, 0, ctx->stream>>>(sidx_all.p, olen, n_o, cover_k, flags.p);
        BG_LAUNCH_CHECK();
      }
      BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, n_o, &keep, &m));
      BG_TRY(cs->skeys.alloc(ctx, m));
      BG_TRY(cs->soid.alloc(ctx, m));
      if (m) {
        k_cover_keys<<<cdiv(m, 256), 256, 0, ctx->stream>>>(sidx_all.p, keep.p, m, owords, W, cover_k, cs->skeys.p, cs->soid.p);
        BG_LAUNCH_CHECK();
      }
    }
    if (m) {
      k_flag_key_heads<<<cdiv(m, 256), 256, 0, ctx->stream>>>(cs->skeys.p, m, flags.p);
      BG_LAUNCH_CHECK();
    }
    BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, m, &head_pos, &nheads));
  }
  int lg = 10;
  // load factor <= 0.7 (8-slot buckets keep chains short) rather than <= 0.5: half the table, which matters
  // beyond 2 GiB where random probes fall off the TLB reach (tools/probe_rate.hip: 55 -> 22 G probes/s at 4 GiB)
  while ((1ull << lg) * 7 < nheads * 10) ++lg;
  BG_TRY(cs->slots.alloc(ctx, 1ull << lg));
  BG_HIP(hipMemsetAsync(cs->slots.p, 0xff, (1ull << lg) * 8, ctx->stream));
  if (nheads) {
    k_cover_insert<<<cdiv(nheads, 256), 256, 0, ctx->stream>>>(cs->skeys.p, cs->soid.p, olen, m, head_pos.p, nheads, cs->slots.p, lg);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));  // head_pos goes out of scope
  cs->ci = CoverIndex{cs->slots.p, lg, cs->skeys.p, cs->soid.p, m, cover_k, olen};
  cs->owords = owords;
  return BG_OK;
}

// Phase A, first half: the oriented reads, the cover index over ALL reads of `reads`, and for the oriented reads of [lo, hi) how
// many suffix records each emits under the exact-cover rule (cnt), where they start (roff) and how many there are in all.
int cover_counts_impl(bg_ctx* ctx, const bg_reads* reads, uint64_t lo, uint64_t hi, int cover_k, CoverCounts* cc) {
  const uint64_t n = reads->n;
  const uint64_t n_o = 2 * n;
  const int W = reads->W;
  if (n_o >= (1ull << 31)) { set_error("too many reads for one cover index (%llu)", (unsigned long long)n); return BG_EINVAL; }
  if (lo > hi || hi > n) { set_error("bad read range"); return BG_EINVAL; }
  cc->o_base = 2 * lo;
  cc->n_emit = 2 * (hi - lo);
  cc->n_rec = 0;
  const uint64_t o_base = cc->o_base, n_emit = cc->n_emit;
  DevBuf<uint64_t>& owords = cc->owords;
  DevBuf<uint8_t>& olen = cc->olen;
  BG_TRY(owords.alloc(ctx, n_o * W + 8));  // (8 spare words: k_cover_count_sm fetches 64 bytes from a candidate's first word)
  BG_TRY(olen.alloc(ctx, n_o));
  CoverStore cs;
  {
    DevBuf<uint64_t> k0o;
    const bool as_items = cover_by_items(n_o, cover_k);  // (the reads' sort items beside the reads: cover_index_build)
    BG_TRY(k0o.alloc(ctx, as_items ? 2 * n_o : n_o));
    if (n_o) {
      ctx->tic("suffix_oriented");
      k_oriented_words<<<cdiv(n_o, 256), 256, 0, ctx->stream>>>(reads->words.p, reads->len.p, n, W, owords.p, olen.p, k0o.p, as_items ? cover_k : 0);
      BG_LAUNCH_CHECK();
      ctx->toc(n * W * 8 + n_o * W * 8 + n_o * 9);
    }
    BG_TRY(cover_index_build(ctx, owords.p, olen.p, n_o, W, cover_k, &k0o, &cs));
  }
  const CoverIndex ci = cs.ci;
  DevBuf<uint64_t> total;
  BG_TRY(cc->cnt.alloc(ctx, n_emit));
  BG_TRY(cc->roff.alloc(ctx, n_emit));
  BG_TRY(total.alloc(ctx, 1));
  if (n_emit) {
    ctx->tic("suffix_cover_count");
    if (getenv("BG_COVER_COUNT_ROUNDS"))  // (a whole look-up per pass, as before round 5)
      k_cover_count<<<cdiv(cdiv(n_emit, kCoverChunk), 4), 256, cover_count_lds(W), ctx->stream>>>(ci, owords.p, o_base, n_emit, W, cc->cnt.p);
    else
      k_cover_count_sm<<<cdiv(cdiv(n_emit, kCoverChunk), 4), 256, cover_count_lds(W), ctx->stream>>>(ci, owords.p, o_base, n_emit, W, cc->cnt.p);
    BG_LAUNCH_CHECK();
    ctx->toc(n_emit * W * 8 * 2);
  }
  BG_TRY(exclusive_scan_u32_to_u64(ctx, cc->cnt.p, cc->roff.p, n_emit, total.p));
  BG_HIP(copy_d2h(ctx, &cc->n_rec, total.p, 8));
  return BG_OK;  // (the cover index goes with cs: only the counts and the oriented reads are needed from here on)
}

// Phase A: suffix records of reads [lo, hi) under the exact-cover rule; the cover index spans ALL
// reads of `reads`.
int suffix_rows_from_counts(bg_ctx* ctx, const bg_reads* reads, CoverCounts* cc, DevBuf<uint64_t>* rec, DevBuf<uint8_t>* rlen,
                            DevBuf<uint64_t>* k0, int RS) {
  if (RS <= 0) RS = reads->W;
  const int W = reads->W;
  const uint64_t n_rec = cc->n_rec, n_emit = cc->n_emit, o_base = cc->o_base;
  BG_TRY(rec->alloc(ctx, n_rec * RS));
  if (RS == W) BG_TRY(rlen->alloc(ctx, n_rec));  // (rows of W + 1 words carry the lengths themselves)
  const bool want_k0 = k0 && n_rec < (1ull << 32);
  if (want_k0) BG_TRY(k0->alloc(ctx, n_rec));
  if (n_rec) {
    ctx->tic("suffix_emit");
    k_emit_records<<<cdiv(cdiv(n_emit, 64), 4), 256, size_t(256) * RS * 8, ctx->stream>>>(cc->owords.p, reads->len.p, cc->cnt.p, cc->roff.p, o_base, n_emit, W,
                                                               RS, rec->p, RS == W ? rlen->p : nullptr, want_k0 ? k0->p : nullptr);
    BG_LAUNCH_CHECK();
    ctx->toc(n_emit * W * 8 + n_rec * (RS * 8 + 1 + (want_k0 ? 8 : 0)));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));  // (the caller may let cc go)
  return BG_OK;
}

int suffix_records_impl(bg_ctx* ctx, const bg_reads* reads, uint64_t lo, uint64_t hi, int cover_k,
                        DevBuf<uint64_t>* rec, DevBuf<uint8_t>* rlen, uint64_t* n_rec_out, DevBuf<uint64_t>* k0, int RS) {
  CoverCounts cc;
  BG_TRY(cover_counts_impl(ctx, reads, lo, hi, cover_k, &cc));
  if (getenv("BG_ITEMS_PROBE")) BG_TRY(items_probe(ctx, cc.owords.p, reads->len.p, cc.cnt.p, cc.roff.p, cc.o_base, cc.n_emit, reads->W, cc.n_rec));  // (items_build.hip)
  BG_TRY(suffix_rows_from_counts(ctx, reads, &cc, rec, rlen, k0, RS));
  *n_rec_out = cc.n_rec;
  return BG_OK;
}

// Order of records by full key (words, length): sidx[i] = record (row of rec / rlen) at sorted position i.
// The n_keys sort keys are (leading 16 bases | record), taken from *k0 when it holds them (consumed), else made
// from rec rows 0 .. n_keys-1.  *n_tied_out = records in tie groups larger than kSmallGroup.
// defer_small: the tie groups of at most kSmallGroup records are left in the order the radix sort gave them (the
// caller's k_gather_dedup_build orders them in LDS).
int sort_records_impl(bg_ctx* ctx, const uint64_t* rec_p, const uint8_t* rlen_p, uint64_t n_rec, int W,
                      DevBuf<uint64_t>* k0, DevBuf<uint32_t>* sidx_out, uint64_t* n_tied_out, int RS, bool defer_small) {
  if (RS <= 0) RS = W;  // words from one row to the next (W + 1: the length sits behind the key words as well)
  struct { const uint64_t* p; } rec{rec_p};
  struct { const uint8_t* p; } rlen{rlen_p};
  DevBuf<uint32_t>& sidx = *sidx_out;
  uint64_t n_tied = 0;
  {
    // ---- sort by the leading 16 bases (record index packed into the key)
    DevBuf<uint64_t> k0a, k0b;
    if (k0 && k0->p && k0->n == n_rec) {  // the emission wrote the sort keys along with the records
      k0a = std::move(*k0);
    } else {
      BG_TRY(k0a.alloc(ctx, n_rec));
      if (n_rec) {
        k_first_words<<<cdiv(n_rec, 256), 256, 0, ctx->stream>>>(rec.p, n_rec, RS, k0a.p);
        BG_LAUNCH_CHECK();
      }
    }
    BG_TRY(k0b.alloc(ctx, n_rec));
    bool in_b = false;
    BG_TRY(radix_sort_u64(ctx, k0a.p, k0b.p, nullptr, nullptr, n_rec, kSortSkipBits, 64, &in_b));
    DevBuf<uint64_t>& ks = in_b ? k0b : k0a;
    (in_b ? k0a : k0b).reset();
    BG_TRY(sidx.alloc(ctx, n_rec));  // written by k_sort_small_groups

    // ---- small tie groups in-lane; larger ones word by word, then by length
    DevBuf<uint32_t> tpos;   // subset slot -> sorted position
    DevBuf<uint64_t> tkey;   // current-round key per subset slot
    DevBuf<uint32_t> gid;    // group ordinal per subset slot
    uint64_t T = 0;
    {
      DevBuf<uint8_t> flags;
      BG_TRY(flags.alloc(ctx, n_rec));
      if (n_rec) {
        ctx->tic("suffix_small_ties");
        k_sort_small_groups<<<cdiv(n_rec, 256), 256, size_t(kSmallSlots) * (W + 1) * 8, ctx->stream>>>(ks.p, n_rec, rec.p, rlen.p, W, RS, sidx.p, flags.p,
                                                                                                        defer_small ? 1 : 0);
        BG_LAUNCH_CHECK();
        ctx->toc(n_rec * 13);
      }
      BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, n_rec, &tpos, &T));
      n_tied = T;  // records in tie groups larger than kSmallGroup (refined by the radix rounds)
    }
    if (T) {
      BG_TRY(tkey.alloc(ctx, T));
      k_gather_key_prefix<<<cdiv(T, 256), 256, 0, ctx->stream>>>(ks.p, tpos.p, T, tkey.p);
      BG_LAUNCH_CHECK();
    }
    ks.reset();
    if (T && !getenv("BG_NO_MEDIUM_GROUPS")) {
      // groups up to kMediumGroup are finished here; what is left for the rounds below is usually nothing
      DevBuf<uint8_t> head, still;
      DevBuf<uint32_t> hpos, keep_t;
      uint64_t G = 0, T2 = 0;
      BG_TRY(head.alloc(ctx, T));
      BG_TRY(still.alloc(ctx, T));
      BG_HIP(hipMemsetAsync(still.p, 0, T, ctx->stream));
      k_tied_run_heads<<<cdiv(T, 256), 256, 0, ctx->stream>>>(tpos.p, tkey.p, T, head.p);
      BG_LAUNCH_CHECK();
      BG_TRY(compact_flagged<uint32_t>(ctx, head.p, T, &hpos, &G));
      auto group_smem = [&](uint32_t cap) {  // keys, records, order (a power of two of 16-bit places), lengths
        uint32_t p2 = 1;
        while (p2 < cap) p2 <<= 1;
        return size_t(cap) * (size_t(W) * 8 + 4 + 1) + size_t(p2) * 2;
      };
      uint32_t large = 0;  // the biggest group a CU's LDS takes at this read length
      for (uint32_t cap : {3072u, 2560u, 2048u, 1536u})
        if (!large && group_smem(cap) <= 156 * 1024) large = cap;
      ctx->tic("suffix_medium_ties");
      BG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_medium_groups),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, int(group_smem(large ? large : kMediumGroup))));
      k_sort_medium_groups<<<unsigned(G), 256, group_smem(kMediumGroup), ctx->stream>>>(
          hpos.p, G, T, tpos.p, rec.p, rlen.p, W, RS, sidx.p, still.p, 0u, uint32_t(kMediumGroup), 0);
      BG_LAUNCH_CHECK();
      ctx->toc(T * (uint64_t(W) * 8 + 9));
      BG_TRY(compact_flagged<uint32_t>(ctx, still.p, T, &keep_t, &T2));
      if (T2 && large && !getenv("BG_NO_LARGE_GROUPS")) {
        k_sort_medium_groups<<<unsigned(G), 1024, group_smem(large), ctx->stream>>>(
            hpos.p, G, T, tpos.p, rec.p, rlen.p, W, RS, sidx.p, still.p, uint32_t(kMediumGroup), large, 1);
        BG_LAUNCH_CHECK();
        BG_TRY(compact_flagged<uint32_t>(ctx, still.p, T, &keep_t, &T2));
      }
      DevBuf<uint32_t> tpos2;
      DevBuf<uint64_t> tkey2;
      BG_TRY(tpos2.alloc(ctx, T2));
      BG_TRY(tkey2.alloc(ctx, T2));
      if (T2) {
        k_gather_u32_by_u32<<<cdiv(T2, 256), 256, 0, ctx->stream>>>(tpos.p, keep_t.p, T2, tpos2.p);
        BG_LAUNCH_CHECK();
        k_gather_u64_by_u32<<<cdiv(T2, 256), 256, 0, ctx->stream>>>(tkey.p, keep_t.p, T2, tkey2.p);
        BG_LAUNCH_CHECK();
      }
      BG_HIP(hipStreamSynchronize(ctx->stream));
      if (getenv("BG_DEBUG_TIES")) fprintf(stderr, "ties: n_rec %llu, in groups > %d: %llu in %llu groups, left after the medium groups: %llu\n", (unsigned long long)n_rec, kSmallGroup, (unsigned long long)T, (unsigned long long)G, (unsigned long long)T2);
      tpos = std::move(tpos2);
      tkey = std::move(tkey2);
      T = T2;
    }
  bool have_gid = false;
  for (int w = 0; w <= W && T; ++w) {  // w = 0: the bits of the first word the sort skipped
    // groups of this round = runs of equal (previous group, previous key) at adjacent positions
    DevBuf<uint32_t> head, excl, ngid;
    BG_TRY(head.alloc(ctx, T));
    BG_TRY(excl.alloc(ctx, T));
    BG_TRY(ngid.alloc(ctx, T));
    k_group_heads<<<cdiv(T, 256), 256, 0, ctx->stream>>>(tpos.p, tkey.p, have_gid ? gid.p : nullptr, T, head.p);
    BG_LAUNCH_CHECK();
    BG_TRY(exclusive_scan_u32(ctx, head.p, excl.p, T, nullptr));
    k_gid_from_scan<<<cdiv(T, 256), 256, 0, ctx->stream>>>(excl.p, head.p, T, ngid.p);
    BG_LAUNCH_CHECK();
    gid = std::move(ngid);
    have_gid = true;
    head.reset(); excl.reset();

    // sort subset by (gid, key_w): LSD = key_w first, then gid (stable)
    DevBuf<uint64_t> ka, kb;
    DevBuf<uint32_t> va, vb, rec_of_t;
    BG_TRY(ka.alloc(ctx, T));
    BG_TRY(kb.alloc(ctx, T));
    BG_TRY(va.alloc(ctx, T));
    BG_TRY(vb.alloc(ctx, T));
    BG_TRY(rec_of_t.alloc(ctx, T));
    k_round_keys<<<cdiv(T, 256), 256, 0, ctx->stream>>>(rec.p, rlen.p, sidx.p, tpos.p, T, W, RS, w, ka.p);
    BG_LAUNCH_CHECK();
    k_gather_u32_by_u32<<<cdiv(T, 256), 256, 0, ctx->stream>>>(sidx.p, tpos.p, T, rec_of_t.p);
    BG_LAUNCH_CHECK();
    k_iota_u32<<<cdiv(T, 256), 256, 0, ctx->stream>>>(va.p, T);
    BG_LAUNCH_CHECK();
    bool b1 = false;
    BG_TRY(radix_sort_u64(ctx, ka.p, kb.p, va.p, vb.p, T, 0, 64, &b1));
    uint64_t* k1 = b1 ? kb.p : ka.p;   // keys sorted by word
    uint32_t* p1 = b1 ? vb.p : va.p;   // subset element per rank
    uint64_t* k1o = b1 ? ka.p : kb.p;  // scratch
    uint32_t* p1o = b1 ? va.p : vb.p;
    // second key = gid of the element; carry the element id
    DevBuf<uint32_t> g32;
    DevBuf<uint64_t> g64a, g64b;
    BG_TRY(g32.alloc(ctx, T));
    BG_TRY(g64a.alloc(ctx, T));
    BG_TRY(g64b.alloc(ctx, T));
    k_gather_u32_by_u32<<<cdiv(T, 256), 256, 0, ctx->stream>>>(gid.p, p1, T, g32.p);
    BG_LAUNCH_CHECK();
    k_u32_to_u64<<<cdiv(T, 256), 256, 0, ctx->stream>>>(g32.p, T, g64a.p);
    BG_LAUNCH_CHECK();
    bool b2 = false;
    BG_TRY(radix_sort_u64(ctx, g64a.p, g64b.p, p1, p1o, T, 0, 32, &b2));
    uint32_t* perm = b2 ? p1o : p1;        // subset element per final slot
    uint64_t* gsorted = b2 ? g64b.p : g64a.p;
    // write the new order back and fetch this round's keys in the new order
    k_apply_perm<<<cdiv(T, 256), 256, 0, ctx->stream>>>(perm, tpos.p, rec_of_t.p, T, sidx.p);
    BG_LAUNCH_CHECK();
    // keys in new order: re-read from the records through the updated sidx
    k_round_keys<<<cdiv(T, 256), 256, 0, ctx->stream>>>(rec.p, rlen.p, sidx.p, tpos.p, T, W, RS, w, k1o);
    BG_LAUNCH_CHECK();
    // gid in new order (as u32)
    DevBuf<uint32_t> gnew;
    BG_TRY(gnew.alloc(ctx, T));
    k_gather_u32_by_u32<<<cdiv(T, 256), 256, 0, ctx->stream>>>(gid.p, perm, T, gnew.p);
    BG_LAUNCH_CHECK();
    (void)gsorted; (void)k1;
    // which subset slots are still tied after this word?
    DevBuf<uint8_t> flags;
    DevBuf<uint32_t> keep_t;
    uint64_t T2 = 0;
    BG_TRY(flags.alloc(ctx, T));
    k_flag_subset_ties<<<cdiv(T, 256), 256, 0, ctx->stream>>>(gnew.p, k1o, T, flags.p);
    BG_LAUNCH_CHECK();
    BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, T, &keep_t, &T2));
    DevBuf<uint32_t> tpos2, gid2;
    DevBuf<uint64_t> tkey2;
    BG_TRY(tpos2.alloc(ctx, T2));
    BG_TRY(gid2.alloc(ctx, T2));
    BG_TRY(tkey2.alloc(ctx, T2));
    if (T2) {
      k_gather_u32_by_u32<<<cdiv(T2, 256), 256, 0, ctx->stream>>>(tpos.p, keep_t.p, T2, tpos2.p);
      BG_LAUNCH_CHECK();
      k_gather_u32_by_u32<<<cdiv(T2, 256), 256, 0, ctx->stream>>>(gnew.p, keep_t.p, T2, gid2.p);
      BG_LAUNCH_CHECK();
      k_gather_u64_by_u32<<<cdiv(T2, 256), 256, 0, ctx->stream>>>(k1o, keep_t.p, T2, tkey2.p);
      BG_LAUNCH_CHECK();
    }
    BG_HIP(hipStreamSynchronize(ctx->stream));
    tpos = std::move(tpos2);
    gid = std::move(gid2);
    tkey = std::move(tkey2);
    T = T2;
  }
  }

  if (n_tied_out) *n_tied_out = n_tied;
  return BG_OK;
}

// Phase B: sort the records, drop prefixes/duplicates, build keys/sizes/shared of the surviving
// entries into `ss`.  Consumes rec/rlen.
int entries_from_records_impl(bg_ctx* ctx, DevBuf<uint64_t>& rec, DevBuf<uint8_t>& rlen, uint64_t n_rec, int W,
                              bg_seqset* ss, uint64_t* n_tied_out, DevBuf<uint64_t>* k0, int RS, LocalPrev* lp) {
  if (RS <= 0) RS = W;
  if (n_rec >= (1ull << 32)) {
    set_error("%llu suffix records exceed one shard", (unsigned long long)n_rec);
    return BG_EINVAL;
  }
  uint64_t n_tied = 0, n_ent = 0;
  DevBuf<uint32_t> sidx;  // sorted order -> record
  // (the fused pass below orders the small tie groups itself; BG_SMALL_TIES_EARLY: as before, for comparisons)
  const bool fused = !getenv("BG_ENTRIES_UNFUSED");
  const bool defer_small = fused && !getenv("BG_SMALL_TIES_EARLY");
  BG_TRY(sort_records_impl(ctx, rec.p, rlen.p, n_rec, W, k0, &sidx, &n_tied, RS, defer_small));

  // ---- dedup + entries
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 2));
  BG_HIP(hipMemsetAsync(err.p, 0, 8, ctx->stream));
  if (fused) {
    // one pass: gather + dedup + entry build (the arrays are sized for n_rec entries, ~10 % more than survive)
    ss->W = W;
    BG_TRY(ss->keys.alloc(ctx, n_rec * W));
    BG_TRY(ss->pre2.alloc(ctx, n_rec * 2));
    BG_TRY(ss->sizes.alloc(ctx, n_rec));
    BG_TRY(ss->shared.alloc(ctx, n_rec));
    if (n_rec) {
      const uint64_t ntiles = (n_rec + kFusedTile - 1) / kFusedTile;
      DevBuf<unsigned long long> state;
      DevBuf<uint64_t> tile_first;
      DevBuf<uint32_t> ticket;
      const uint64_t nblocks = (ntiles + 63) / 64;
      const bool two_level = !getenv("BG_LOOKBACK_FLAT");  // (the look-back over tile states alone, as before round 5)
      BG_TRY(state.alloc(ctx, ntiles + nblocks));
      BG_TRY(tile_first.alloc(ctx, ntiles));
      BG_TRY(ticket.alloc(ctx, 1));
      BG_HIP(hipMemsetAsync(state.p, 0, (ntiles + nblocks) * 8, ctx->stream));
      BG_HIP(hipMemsetAsync(ticket.p, 0, 4, ctx->stream));
      // prev bits where the records meet (rows that carry the flags of k_emit_records): see the kernel
      const bool local_prev = lp && RS > W && !getenv("BG_PREV_SEARCH_ALL");
      if (local_prev) {
        lp->cap = n_rec / 3 + 1024;
        BG_TRY(lp->emask.alloc(ctx, (n_rec + 3) & ~3ull));
        BG_TRY(lp->list.alloc(ctx, lp->cap * uint64_t(W + 1)));
        BG_TRY(lp->count.alloc(ctx, 1));
        BG_HIP(hipMemsetAsync(lp->count.p, 0, 8, ctx->stream));
        if (lp->shard) BG_TRY(lp->need.alloc(ctx, (n_rec + 3) & ~3ull));
      }
      ctx->tic("entries_dedup_build");
      const size_t slots = size_t(kFusedTile) + 2 * (defer_small ? kFusedHalo : 1);
      const size_t lds = slots * W * 8 + 2 * slots + 2 * kFusedTile;
      k_gather_dedup_build<<<unsigned(ntiles), kGatherTile, lds, ctx->stream>>>(
          rec.p, rlen.p, sidx.p, n_rec, W, RS, state.p, ticket.p, ss->keys.p, ss->sizes.p, ss->shared.p, ss->pre2.p, tile_first.p,
          err.p, defer_small ? 1 : 0, local_prev ? lp->emask.p : nullptr, local_prev ? lp->list.p : nullptr,
          local_prev ? reinterpret_cast<unsigned long long*>(lp->count.p) : nullptr, local_prev ? lp->cap : 0,
          two_level ? state.p + ntiles : nullptr, (local_prev && lp->shard) ? lp->need.p : nullptr);
      BG_LAUNCH_CHECK();
      k_fix_tile_first_shared<<<cdiv(ntiles, 256), 256, 0, ctx->stream>>>(tile_first.p, ntiles, ss->keys.p, ss->sizes.p, W,
                                                                          ss->shared.p);
      BG_LAUNCH_CHECK();
      ctx->toc(n_rec * (W * 8 * 2 + 8 + 16));
#if defined(BG_EXP_GATHER_ONLY) || defined(BG_EXP_NO_LOOKBACK)
      BG_HIP(hipStreamSynchronize(ctx->stream));
      set_error("timing experiment: the entry build's result is not a seqset");
      return BG_EINTERNAL;
#endif
      unsigned long long last = 0;
      BG_HIP(hipMemcpyAsync(&last, state.p + (ntiles - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
      if (local_prev) BG_HIP(hipMemcpyAsync(&lp->n_items, lp->count.p, 8, hipMemcpyDeviceToHost, ctx->stream));
      BG_HIP(hipStreamSynchronize(ctx->stream));
      n_ent = last & kScanVal;
      if (local_prev) lp->valid = lp->n_items <= lp->cap;
    }
    rec.reset(); rlen.reset(); sidx.reset();
    ss->n = n_ent;
  } else {
    DevBuf<uint64_t> skeys;
    DevBuf<uint8_t> slen, lcp_prev, keep;
    BG_TRY(skeys.alloc(ctx, n_rec * W));
    BG_TRY(slen.alloc(ctx, n_rec));
    BG_TRY(lcp_prev.alloc(ctx, n_rec));
    BG_TRY(keep.alloc(ctx, n_rec));
    DevBuf<uint32_t> kept_pos;
    if (n_rec) {
      ctx->tic("entries_dedup");
      const size_t lds = size_t(kGatherTile + 2) * W * 8 + kGatherTile + 2;
      k_gather_lcp_keep<<<cdiv(n_rec, kGatherTile), kGatherTile, lds, ctx->stream>>>(rec.p, rlen.p, sidx.p, n_rec, W, RS, skeys.p,
                                                                                  slen.p, lcp_prev.p, keep.p, err.p);
      BG_LAUNCH_CHECK();
      ctx->toc(n_rec * (W * 8 * 2 + 8));
    }
    rec.reset(); rlen.reset(); sidx.reset();
    BG_TRY(compact_flagged<uint32_t>(ctx, keep.p, n_rec, &kept_pos, &n_ent));
    ss->n = n_ent;
    ss->W = W;
    BG_TRY(ss->keys.alloc(ctx, n_ent * W));
    BG_TRY(ss->pre2.alloc(ctx, n_ent * 2));
    BG_TRY(ss->sizes.alloc(ctx, n_ent));
    BG_TRY(ss->shared.alloc(ctx, n_ent));
    if (n_ent) {
      ctx->tic("entries_build");
      k_build_entries<<<cdiv(n_ent, 256), 256, 0, ctx->stream>>>(skeys.p, slen.p, lcp_prev.p, kept_pos.p, n_ent, W,
                                                                ss->keys.p, ss->sizes.p, ss->shared.p, ss->pre2.p);
      BG_LAUNCH_CHECK();
      ctx->toc(n_ent * (W * 16 + 8 + 16));
    }
  }
  uint32_t h_err[2] = {0, 0};
  BG_HIP(copy_d2h(ctx, h_err, err.p, 8));
  if (h_err[0] & 1) { set_error("suffix records not sorted (internal error)"); return BG_EINTERNAL; }
  if (n_tied_out) *n_tied_out = n_tied;
  return BG_OK;
}

// leading-bases directory of a shard's entries (needs resident keys)
int build_prefix_dir(bg_ctx* ctx, const bg_seqset* ss, DevBuf<uint32_t>* dir, int* bases) {
  return build_prefix_dir_keys(ctx, ss->keys.p, ss->n, ss->W, dir, bases);
}

static int pack_marks(bg_ctx* ctx, bg_seqset* ss, const uint8_t* marks) {
  const uint64_t nwords = (ss->n + 63) / 64;
  if (!nwords) return BG_OK;
  k_pack_marks<<<cdiv(4 * cdiv(nwords, kPackWordsPerWave), 4), 256, 0, ctx->stream>>>(
      marks, ss->n, reinterpret_cast<unsigned long long*>(ss->bits[0].p),
      reinterpret_cast<unsigned long long*>(ss->bits[1].p), reinterpret_cast<unsigned long long*>(ss->bits[2].p),
      reinterpret_cast<unsigned long long*>(ss->bits[3].p));
  BG_LAUNCH_CHECK();
  return BG_OK;
}

__global__ __launch_bounds__(256) void k_apply_tails_pre2(const uint64_t* __restrict__ rows, uint64_t nt,
                                                          const uint64_t* __restrict__ ekeys,
                                                          const uint64_t* __restrict__ pre2,
                                                          const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                          const uint32_t* __restrict__ dir, int bases,
                                                          uint8_t* __restrict__ marks, uint32_t* __restrict__ err,
                                                          int verify) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const uint64_t* row = rows + t * (W + 1);
  uint64_t tail[kMaxWords];
  for (int w = 0; w < W; ++w) tail[w] = row[w];
  const int tl = int(row[W] & 0xff), b = int((row[W] >> 8) & 3);
  uint64_t lo;
  if (!prev_target_pre2(ekeys, pre2, sizes, n, W, dir, bases, tail, tl, verify, err, &lo)) return;
  marks[uint64_t(b) * marks_stride(n) + lo] = 1;
}

// short tails (3 words: two leading words of the tail, meta): a tail shorter than 64 bases is all there in its
// two words; a longer one whose leading words occur in exactly one entry has found its target (the test of
// k_prev_bits_pre2); the rest need the whole tail
__global__ __launch_bounds__(256) void k_apply_tails_short(const uint64_t* __restrict__ rows, uint64_t nt,
                                                           const uint64_t* __restrict__ ekeys,
                                                           const uint64_t* __restrict__ pre2,
                                                           const uint8_t* __restrict__ sizes, uint64_t n, int W,
                                                           const uint32_t* __restrict__ dir, int bases,
                                                           uint8_t* __restrict__ marks, uint8_t* __restrict__ need_full,
                                                           uint32_t* __restrict__ err, int verify) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const uint64_t t0 = rows[t * 3], t1 = rows[t * 3 + 1], meta = rows[t * 3 + 2];
  const int tl = int(meta & 0xff), b = int((meta >> 8) & 3);
  need_full[t] = 0;
  if (tl >= 64 && (verify || W <= 2)) { need_full[t] = 1; return; }  // W <= 2 cannot hold 64 + bases: never
  if (tl < 64) {
    uint64_t tail[kMaxWords];
    tail[0] = t0;
    tail[1] = t1;
    for (int w = 2; w < kMaxWords; ++w) tail[w] = 0;
    uint64_t lo;
    if (!prev_target_pre2(ekeys, pre2, sizes, n, W, dir, bases, tail, tl, 0, err, &lo)) return;
    marks[uint64_t(b) * marks_stride(n) + lo] = 1;
    return;
  }
  const uint32_t p = uint32_t(t0 >> (64 - 2 * bases));
  const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(pre2);
  uint64_t a = dir[p], z = dir[p + 1];
  while (a < z) {
    const uint64_t mid = (a + z) >> 1;
    const ulonglong2 m = p2[mid];
    if (m.x < t0 || (m.x == t0 && m.y < t1)) a = mid + 1; else z = mid;
  }
  bool sure = false;
  if (a < n) {
    const ulonglong2 c0 = p2[a];
    sure = c0.x == t0 && c0.y == t1;
    if (sure && a + 1 < n) {
      const ulonglong2 c1 = p2[a + 1];
      sure = c1.x != t0 || c1.y != t1;
    }
  }
  // the one entry with these 64 leading bases: it is the target if it is long enough to extend the tail at all
  // (an entry that agrees on 64 bases but is shorter than the tail cannot be: ask for the whole tail)
  if (sure && int(sizes[a]) >= tl) {
    // NB equal leading words do not prove the entry extends the tail beyond base 64: the closure of the entry set
    // does (some entry extends the tail, and this is the only one that starts like it)
    marks[uint64_t(b) * marks_stride(n) + a] = 1;
  } else {
    need_full[t] = 1;
  }
}

int apply_tails2_impl(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows, uint8_t* d_need_full, uint32_t* d_err) {
  DevBuf<uint32_t> dir;
  DevBuf<uint8_t> marks;
  int bases = 0;
  const bool use_pre2 = ss->pre2.p && ss->pre2.n >= 2 * ss->n;
  if (!use_pre2) { set_error("short tails need the entries' two-word prefixes"); return BG_EINVAL; }
  BG_TRY(build_prefix_dir_keys(ctx, ss->pre2.p, ss->n, 2, &dir, &bases));
  BG_TRY(marks.alloc(ctx, 4 * marks_stride(ss->n)));
  if (ss->n) BG_HIP(hipMemsetAsync(marks.p, 0, 4 * marks_stride(ss->n), ctx->stream));
  const char* v = getenv("BG_VERIFY_CLOSURE");
  const bool full = (v && v[0] == '1') || getenv("BG_PREV_FULL_SEARCH");
  k_apply_tails_short<<<cdiv(n_rows, 256), 256, 0, ctx->stream>>>(d_rows, n_rows, ss->keys.p, ss->pre2.p, ss->sizes.p, ss->n, ss->W,
                                                                 dir.p, bases, marks.p, d_need_full, d_err, full ? 1 : 0);
  BG_LAUNCH_CHECK();
  BG_TRY(pack_marks(ctx, ss, marks.p));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

int apply_tails_impl(bg_ctx* ctx, bg_seqset* ss, const uint64_t* d_rows, uint64_t n_rows, uint32_t* d_err) {
  DevBuf<uint32_t> dir;
  DevBuf<uint8_t> marks;
  int bases = 0;
  const bool use_pre2 = ss->pre2.p && ss->pre2.n >= 2 * ss->n && !getenv("BG_PREV_FULL_SEARCH");
  if (use_pre2) BG_TRY(build_prefix_dir_keys(ctx, ss->pre2.p, ss->n, 2, &dir, &bases));
  else BG_TRY(build_prefix_dir(ctx, ss, &dir, &bases));
  BG_TRY(marks.alloc(ctx, 4 * marks_stride(ss->n)));
  if (ss->n) BG_HIP(hipMemsetAsync(marks.p, 0, 4 * marks_stride(ss->n), ctx->stream));
  if (use_pre2) {
    const char* v = getenv("BG_VERIFY_CLOSURE");
    k_apply_tails_pre2<<<cdiv(n_rows, 256), 256, 0, ctx->stream>>>(d_rows, n_rows, ss->keys.p, ss->pre2.p, ss->sizes.p, ss->n,
                                                                   ss->W, dir.p, bases, marks.p, d_err, v && v[0] == '1');
  } else {
    k_apply_tails<<<cdiv(n_rows, 256), 256, 0, ctx->stream>>>(d_rows, n_rows, ss->keys.p, ss->sizes.p, ss->n, ss->W, dir.p,
                                                              bases, marks.p, d_err);
  }
  BG_LAUNCH_CHECK();
  BG_TRY(pack_marks(ctx, ss, marks.p));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

int alloc_prev_bits(bg_ctx* ctx, bg_seqset* ss) {
  const uint64_t nwords = (ss->n + 63) / 64;
  for (int b = 0; b < 4; ++b) {
    BG_TRY(ss->bits[b].alloc(ctx, nwords));
    if (nwords) BG_HIP(hipMemsetAsync(ss->bits[b].p, 0, nwords * 8, ctx->stream));
  }
  if (nwords && ss->emask.p) {  // a shard whose entry build set prev bits where its records met: the tails that arrive add to them
    k_pack_emask<<<cdiv(cdiv(nwords, kPackWordsPerWave), 4), 256, 0, ctx->stream>>>(
        ss->emask.p, ss->n, reinterpret_cast<unsigned long long*>(ss->bits[0].p), reinterpret_cast<unsigned long long*>(ss->bits[1].p),
        reinterpret_cast<unsigned long long*>(ss->bits[2].p), reinterpret_cast<unsigned long long*>(ss->bits[3].p));
    BG_LAUNCH_CHECK();
  }
  return BG_OK;
}

// Phase C (single shard): prev bits from the shard's own entries.
int prev_bits_local_impl(bg_ctx* ctx, bg_seqset* ss, LocalPrev* lp) {
  const uint64_t n_ent = ss->n;
  BG_TRY(alloc_prev_bits(ctx, ss));
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 2));
  BG_HIP(hipMemsetAsync(err.p, 0, 8, ctx->stream));
  DevBuf<uint32_t> dir;
  DevBuf<uint8_t> marks;
  const bool use_pre2 = ss->pre2.p && ss->pre2.n >= 2 * n_ent && !getenv("BG_PREV_FULL_SEARCH");
  const char* pm = getenv("BG_PREV_MERGE");
  if (n_ent && lp && lp->valid && lp->searched) {
    ctx->tic("prev_bits");
    const uint64_t nwords = (n_ent + 63) / 64;
    k_pack_emask<<<cdiv(cdiv(nwords, kPackWordsPerWave), 4), 256, 0, ctx->stream>>>(
        lp->emask.p, n_ent, reinterpret_cast<unsigned long long*>(ss->bits[0].p), reinterpret_cast<unsigned long long*>(ss->bits[1].p),
        reinterpret_cast<unsigned long long*>(ss->bits[2].p), reinterpret_cast<unsigned long long*>(ss->bits[3].p));
    BG_LAUNCH_CHECK();
    ctx->toc(n_ent * 2);
    BG_HIP(hipStreamSynchronize(ctx->stream));
    return BG_OK;
  }
  if (n_ent && lp && lp->valid && use_pre2 && !(pm && pm[0] == '1')) {
    // most bits are set already (k_gather_dedup_build); the listed tails are searched as every entry's tail used to be
    ctx->tic("prev_bits");
    int bases = 0;
    BG_TRY(build_prefix_dir_keys(ctx, ss->pre2.p, n_ent, 2, &dir, &bases));
    const char* v = getenv("BG_VERIFY_CLOSURE");
    if (lp->n_items) {
      k_prev_list<<<cdiv(lp->n_items, 256), 256, 0, ctx->stream>>>(lp->list.p, lp->n_items, ss->keys.p, ss->pre2.p, ss->sizes.p, n_ent,
                                                                  ss->W, dir.p, bases, lp->emask.p, err.p, v && v[0] == '1');
      BG_LAUNCH_CHECK();
    }
    const uint64_t nwords = (n_ent + 63) / 64;
    k_pack_emask<<<cdiv(cdiv(nwords, kPackWordsPerWave), 4), 256, 0, ctx->stream>>>(
        lp->emask.p, n_ent, reinterpret_cast<unsigned long long*>(ss->bits[0].p), reinterpret_cast<unsigned long long*>(ss->bits[1].p),
        reinterpret_cast<unsigned long long*>(ss->bits[2].p), reinterpret_cast<unsigned long long*>(ss->bits[3].p));
    BG_LAUNCH_CHECK();
    ctx->toc(lp->n_items * (ss->W * 8 + 8 + 16) + n_ent * 2);
    uint32_t h_err[2] = {0, 0};
    BG_HIP(copy_d2h(ctx, h_err, err.p, 8));
    if (h_err[0] & 2) { set_error("missing expansion (entry set not closed under pop_front)"); return BG_EINTERNAL; }
    if (h_err[0] & 4) { set_error("prev bits: the two-word search disagrees with the full-key search (internal error)"); return BG_EINTERNAL; }
    return BG_OK;
  }
  // Off unless asked for (BG_PREV_MERGE=1): the merge gives the same bits (the tests run it) but, as written, not faster --
  // chr20: 9.4 ms against 5.6 + 0.5 (directory).  A workgroup is a chain of dependent steps (bounds -> 40 KB of rows into LDS
  // -> per round: stage 256 tails, ten-step LDS search) at three workgroups per CU: ~50 us each, 184 rounds of them.  The
  // first version (two-word prefixes of the chunk in LDS, 16 KB, the 9 % of tails they do not settle searched in global
  // memory) took 5.95 ms.  What it would need: chunks of 256 entries (7 workgroups per CU) with two-level bounds, and a
  // thread taking a run of consecutive tails and walking forward from the previous target instead of searching afresh.
  if (n_ent && ss->keys.p && !getenv("BG_PREV_FULL_SEARCH") && pm && pm[0] == '1') {
    ctx->tic("prev_bits");
    const uint64_t G = (n_ent + kPmChunk - 1) / kPmChunk;
    DevBuf<uint64_t> F, bounds;
    BG_TRY(F.alloc(ctx, 8));
    BG_TRY(bounds.alloc(ctx, (G + 1) * 4));
    k_pm_first_base<<<1, 64, 0, ctx->stream>>>(ss->keys.p, n_ent, ss->W, F.p);
    BG_LAUNCH_CHECK();
    k_pm_bounds<<<cdiv((G + 1) * 4, 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->sizes.p, n_ent, ss->W, F.p, G, bounds.p);
    BG_LAUNCH_CHECK();
    const size_t lds = size_t(kPmChunk) * size_t(ss->W) * 8 + kPmChunk + 256 * size_t(ss->W) * 8 + 256;
    static bool lds_set = false;
    if (!lds_set) {
      BG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prev_merge), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 kPmChunk * kMaxWords * 8 + kPmChunk + 256 * kMaxWords * 8 + 256));
      lds_set = true;
    }
    k_prev_merge<<<unsigned(G), 256, lds, ctx->stream>>>(ss->keys.p, ss->sizes.p, n_ent, ss->W, bounds.p,
                                                        reinterpret_cast<unsigned long long*>(ss->bits[0].p),
                                                        reinterpret_cast<unsigned long long*>(ss->bits[1].p),
                                                        reinterpret_cast<unsigned long long*>(ss->bits[2].p),
                                                        reinterpret_cast<unsigned long long*>(ss->bits[3].p), err.p);
    BG_LAUNCH_CHECK();
    ctx->toc(n_ent * (ss->W * 8 + 1 + 8));
  } else if (n_ent) {
    ctx->tic("prev_bits");
    int bases = 0;
    if (use_pre2) BG_TRY(build_prefix_dir_keys(ctx, ss->pre2.p, n_ent, 2, &dir, &bases));
    else BG_TRY(build_prefix_dir(ctx, ss, &dir, &bases));
    BG_TRY(marks.alloc(ctx, 4 * marks_stride(n_ent)));
    BG_HIP(hipMemsetAsync(marks.p, 0, 4 * marks_stride(n_ent), ctx->stream));
    if (use_pre2) {
      const char* v = getenv("BG_VERIFY_CLOSURE");
      k_prev_bits_pre2<<<cdiv(n_ent, 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->pre2.p, ss->sizes.p, n_ent, ss->W, dir.p,
                                                                  bases, marks.p, err.p, v && v[0] == '1');
    } else {
      k_prev_bits<<<cdiv(n_ent, 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->sizes.p, n_ent, ss->W, dir.p, bases, marks.p,
                                                             err.p);
    }
    BG_LAUNCH_CHECK();
    BG_TRY(pack_marks(ctx, ss, marks.p));
    ctx->toc(n_ent * (ss->W * 8 + 1 + 8));
  }
  uint32_t h_err[2] = {0, 0};
  BG_HIP(copy_d2h(ctx, h_err, err.p, 8));
  if (h_err[0] & 2) { set_error("missing expansion (entry set not closed under pop_front)"); return BG_EINTERNAL; }
  if (h_err[0] & 4) { set_error("prev bits: the two-word search disagrees with the full-key search (internal error)"); return BG_EINTERNAL; }
  return BG_OK;
}

}  // namespace bg

using namespace bg;

extern "C" int bg_seqset_build(bg_ctx* ctx, const bg_reads* reads, int cover_k, bg_seqset** out,
                               bg_build_stats* stats) {
  if (!ctx || !reads || !out) { set_error("bg_seqset_build: null argument"); return BG_EINVAL; }
  if (cover_k < 1 || cover_k > 31) { set_error("bg_seqset_build: cover_k %d outside 1..31", cover_k); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_seqset> ss(new bg_seqset);
  ss->ctx = ctx;
  ss->W = reads->W;
  ss->uuid = make_uuid();
  uint64_t n_rec = 0, n_tied = 0;
  DevBuf<uint64_t> rec;
  DevBuf<uint8_t> rlen;
  DevBuf<uint64_t> k0;
  // (rows of W + 1 words: every record carries its length behind its key words)
  // (padded to 8 words -- one 64-byte sector a row -- the entry build's gather took 8.4 ms instead of 8.8 and the emission 3.9
  // instead of 3.3: measured and dropped)
  const int RS = reads->W + 1;
  LocalPrev lp;
  bool by_items = false;
  // The entries from 16-byte items (items_build.hip) -- the same arrays without rows, sorted keys or a key
  // table (made when a consumer asks: seqset_ensure_keys).  BG_ITEMS_BUILD=0: through the rows, as before; also what an input
  // takes that the items path declines (*done = false: nothing of ss / lp was touched, the cover stage's counts are reused).
  const char* items_env = getenv("BG_ITEMS_BUILD");
  const bool items_on = !(items_env && items_env[0] == '0');
  CoverCounts cc;
  BG_TRY(cover_counts_impl(ctx, reads, 0, reads->n, cover_k, &cc));
  n_rec = cc.n_rec;
  if (getenv("BG_ITEMS_PROBE")) BG_TRY(items_probe(ctx, cc.owords.p, reads->len.p, cc.cnt.p, cc.roff.p, cc.o_base, cc.n_emit, reads->W, cc.n_rec));
  if (items_on) BG_TRY(items_entries_impl(ctx, reads, &cc, ss.get(), &lp, &by_items));
  if (getenv("BG_DEBUG_PREV")) fprintf(stderr, "entries from items: %s\n", by_items ? "yes" : "no (through the rows)");
  if (!by_items) {
    BG_TRY(suffix_rows_from_counts(ctx, reads, &cc, &rec, &rlen, &k0, RS));
    cc = CoverCounts();
    BG_TRY(entries_from_records_impl(ctx, rec, rlen, n_rec, reads->W, ss.get(), &n_tied, &k0, RS, &lp));
  }
  BG_TRY(prev_bits_local_impl(ctx, ss.get(), &lp));
  if (getenv("BG_DEBUG_PREV")) fprintf(stderr, "prev bits: %llu of %llu entries' worth of tails searched (list %s)\n", (unsigned long long)(lp.valid ? lp.n_items : ss->n), (unsigned long long)ss->n, lp.valid ? "ok" : "not used");
  // the longest read is never a proper prefix of anything longer, so it survives as an entry
  ss->max_len = ss->n ? reads->max_len : 0;
  ctx->tic("seqset_finalize");
  BG_TRY(seqset_finalize_device(ctx, ss.get()));
  ctx->toc(ss->n * 3);
  if (stats) {
    stats->n_records = n_rec;
    stats->n_tied_records = n_tied;
    stats->n_entries = ss->n;
    stats->max_len = ss->max_len;
  }
  *out = ss.release();
  return BG_OK;
}
