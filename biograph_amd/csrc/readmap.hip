// readmap.hip — readmap construction on the GPU (the first "next" row):
// make_readmap::create_from_reads (modules/bio_mapred/make_readmap.cpp:120-392).
//
//   1. entry id of every kept read and of its reverse complement = first seqset entry having it as a
//      prefix (seqset::find_existing_unique, bio_base/seqset.cpp:173-188) — one lower_bound on the
//      resident entry keys through the leading-bases directory;
//   2. mate-loop rows (LOOP_START, RC, MATE, MATE_RC; make_readmap.h:56-90) packed into two sort words
//      hi = entry:37|type:2|read_len:10|mate_len:10 and lo = loop_entry:37, sorted (hi, lo) — the
//      total order of make_readmap.h:187-213;
//   3. sparse_multi bit vectors, read lengths, is_forward;
//   4. the mate loops.  The reference claims rows sequentially "to stay deterministic"
//      (:328-392): the k-th LOOP_START (in row order) that targets a group of identical
//      (entry, type, length) rows takes the k-th row of that group.  That rank is computed here with one
//      stable sort per hop (RC, MATE, MATE_RC), which reproduces the sequential result exactly.
#include <string.h>
#include <time.h>

#include <thread>

#include "bg_device_funcs.h"
#include "bg_internal.h"
#include "compact.h"
#include "items_entries.h"

namespace bg {

constexpr uint64_t kNoLoopEntry = (1ull << 37) - 1;  // make_readmap.h:54

__device__ __forceinline__ uint64_t rm_hi(uint64_t entry, int type, uint32_t len, uint32_t mate_len) {
  return (entry << 22) | (uint64_t(type) << 20) | (uint64_t(len) << 10) | uint64_t(mate_len);
}
__device__ __forceinline__ uint64_t hi_entry(uint64_t hi) { return hi >> 22; }
__device__ __forceinline__ int hi_type(uint64_t hi) { return int((hi >> 20) & 3); }
__device__ __forceinline__ uint32_t hi_len(uint64_t hi) { return uint32_t((hi >> 10) & 1023); }
__device__ __forceinline__ uint32_t hi_mate_len(uint64_t hi) { return uint32_t(hi & 1023); }

// entry id of oriented read o (2i = read i, 2i+1 = its reverse complement); ~0 for dropped reads
__global__ __launch_bounds__(128) void k_rm_entry_ids(const uint64_t* __restrict__ words,
                                                     const uint16_t* __restrict__ len, uint64_t n_o, int W,
                                                     const uint64_t* __restrict__ ekeys,
                                                     const uint8_t* __restrict__ esizes, uint64_t n_ent, int EW,
                                                     const uint32_t* __restrict__ dir, int bases,
                                                     const uint8_t* __restrict__ marks,
                                                     uint64_t* __restrict__ eid, uint32_t* __restrict__ err) {
  uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= n_o) return;
  int L = len[o >> 1];
  if (!L) { eid[o] = ~0ull; return; }
  uint64_t key[kMaxWords];
  oriented_words(words + (o >> 1) * W, L, W, (o & 1) != 0, key);
  for (int w = W; w < EW; ++w) key[w] = 0;
  uint32_t p = uint32_t(key[0] >> (64 - 2 * bases));
  uint64_t lo = dir[p], hi = dir[p + 1];
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (key_compare(ekeys + mid * EW, esizes[mid], key, L, EW) < 0) lo = mid + 1; else hi = mid;
  }
  if (lo >= n_ent || key_lcp(ekeys + lo * EW, esizes[lo], key, L, EW) < L) {
    atomicOr(err, 1u);  // "Read ... was not found in seqset" (make_readmap.cpp:147-151)
    lo = 0;
  } else if (marks) {  // see bg_readmap_build_marked
    while (!marks[lo]) {
      ++lo;
      if (lo >= n_ent || key_lcp(ekeys + lo * EW, esizes[lo], key, L, EW) < L) { atomicOr(err, 4u); lo = 0; break; }
    }
  }
  eid[o] = lo;
}

// The same over the entries of a build from items, before anybody has asked for its key table (items_entries.h): the search the
// prev bits' tails went through -- the first words' directory, the entries' first words, the reads where those tie -- instead of
// 40 bytes of keys an entry made for it (k_items_keys 5.6 ms + this look-up 5.4 on the chr20 bench; this one alone: the table
// is not made at all in `create`).  A read's own record is among the entries' records or absorbed by one: it is always found.
__global__ __launch_bounds__(128) void k_rm_entry_ids_items(const uint64_t* __restrict__ words, const uint16_t* __restrict__ len,
                                                           uint64_t n_o, int W, ItEntries t, uint64_t* __restrict__ eid,
                                                           uint32_t* __restrict__ err) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= n_o) return;
  const int L = len[o >> 1];
  if (!L) { eid[o] = ~0ull; return; }
  uint64_t key[kMaxWords];
  oriented_words(words + (o >> 1) * W, L, W, (o & 1) != 0, key);
  for (int w = W; w < kMaxWords; ++w) key[w] = 0;
  bool alone = false;
  uint64_t lo = it_lower_bound(t, key, L, &alone);
  if (lo >= t.n || (!alone && it_entry_lcp(t, lo, key, L) < L)) {
    atomicOr(err, 1u);  // "Read ... was not found in seqset" (make_readmap.cpp:147-151)
    lo = 0;
  }
  eid[o] = lo;
}

// The same for a seqset opened from a file (no resident keys): seqset::find (bio_base/seqset.cpp:149-153) by
// push_front from the read's last base; the entry id is the start of the range found.
__global__ __launch_bounds__(128) void k_rm_entry_ids_fm(SeqsetView v, const uint64_t* __restrict__ words,
                                                        const uint16_t* __restrict__ len, uint64_t n_o, int W,
                                                        const uint8_t* __restrict__ marks,
                                                        uint64_t* __restrict__ eid, uint32_t* __restrict__ err) {
  uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= n_o) return;
  int L = len[o >> 1];
  if (!L) { eid[o] = ~0ull; return; }
  uint64_t key[kMaxWords];
  oriented_words(words + (o >> 1) * W, L, W, (o & 1) != 0, key);
  uint64_t rb = 0, re = v.n;
  uint32_t rs = 0;
  for (int p = L - 1; p >= 0 && rb < re; --p) {
    const int b = int((key[p >> 5] >> (62 - 2 * (p & 31))) & 3);
    v.push_front(rb, re, rs, b, &rb, &re, &rs);
  }
  if (rb >= re) {
    atomicOr(err, 1u);  // "Read ... was not found in seqset" (make_readmap.cpp:147-151)
    rb = 0;
  } else if (marks) {
    while (rb < re && !marks[rb]) ++rb;
    if (rb >= re) { atomicOr(err, 4u); rb = 0; }
  }
  eid[o] = rb;
}

static SeqsetView rm_seqset_view(const bg_seqset* ss) {
  SeqsetView v;
  v.n = ss->n;
  v.sizes = ss->sizes.p;
  v.shared = ss->shared.p;
  for (int b = 0; b < 4; ++b) { v.bits[b] = ss->bits[b].p; v.subaccum[b] = ss->subaccum[b].p; v.accum[b] = ss->accum[b].p; }
  for (int b = 0; b < 5; ++b) v.fixed[b] = ss->fixed[b];
  v.min1 = ss->min1.p;
  v.min2 = ss->min2.p;
  return v;
}

// rows of record r (pair r = reads 2r, 2r+1 when paired, else read r); unused slots get hi = ~0
__global__ void k_rm_rows(const uint64_t* __restrict__ words, const uint16_t* __restrict__ len, uint64_t n_rec, int W,
                          int paired, const uint64_t* __restrict__ eid, uint64_t* __restrict__ hi,
                          uint64_t* __restrict__ lo) {
  uint64_t r = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n_rec) return;
  const int per = paired ? 4 : 2;
  uint64_t* h = hi + r * per;
  uint64_t* l = lo + r * per;
  for (int t = 0; t < per; ++t) { h[t] = ~0ull; l[t] = 0; }
  uint64_t a = paired ? 2 * r : r, b = a + 1;
  int la = len[a], lb = paired ? int(len[b]) : 0;
  if (!la && !lb) return;
  if (!la) { a = b; la = lb; lb = 0; }  // the surviving mate alone (biograph_create.cpp:880-889)
  uint64_t e = eid[2 * a], erc = eid[2 * a + 1];
  if (lb) {
    uint64_t me = eid[2 * b], merc = eid[2 * b + 1];
    uint32_t rl = uint32_t(la), ml = uint32_t(lb);
    if (key_compare(words + a * W, la, words + b * W, lb, W) > 0) {  // canonical order (:171-176)
      uint64_t t = me; me = e; e = t;
      t = merc; merc = erc; erc = t;
      uint32_t u = ml; ml = rl; rl = u;
    }
    h[0] = rm_hi(e, 0, rl, 0);      l[0] = erc;
    h[1] = rm_hi(erc, 1, rl, ml);   l[1] = me;
    h[2] = rm_hi(me, 2, ml, 0);     l[2] = merc;
    h[3] = rm_hi(merc, 3, ml, 0);   l[3] = kNoLoopEntry;
  } else {
    h[0] = rm_hi(e, 0, uint32_t(la), 0);    l[0] = erc;
    h[1] = rm_hi(erc, 1, uint32_t(la), 0);  l[1] = kNoLoopEntry;
  }
}

__global__ void k_rm_flag_valid(const uint64_t* __restrict__ hi, uint64_t n, uint8_t* __restrict__ f) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) f[i] = hi[i] != ~0ull;
}
__global__ void k_rm_flag_type(const uint64_t* __restrict__ hi, uint64_t n, int type, uint8_t* __restrict__ f) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) f[i] = hi_type(hi[i]) == type;
}

// per sorted row: read length, is_forward, group-head flag, entry marks
__global__ void k_rm_row_fields(const uint64_t* __restrict__ hi, uint64_t n, uint64_t* __restrict__ read_len,
                                uint64_t* __restrict__ is_fwd, uint8_t* __restrict__ head,
                                uint8_t* __restrict__ entry_marks, uint64_t* __restrict__ entry_id) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t h = hi[i];
  read_len[i] = hi_len(h);
  int t = hi_type(h);
  is_fwd[i] = (t == 0 || t == 2) ? 1 : 0;
  head[i] = (i == 0 || hi_entry(hi[i - 1]) != hi_entry(h)) ? 1 : 0;
  entry_marks[hi_entry(h)] = 1;
  entry_id[i] = hi_entry(h);
}

__global__ __launch_bounds__(256) void k_rm_min_len(const uint64_t* __restrict__ read_len, uint64_t n,
                                                    uint32_t* __restrict__ out) {
  uint32_t m = ~0u;
  for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256)
    m = min(m, uint32_t(read_len[i]));
  for (int d = 32; d; d >>= 1) m = min(m, uint32_t(__shfl_xor(int(m), d, 64)));
  if ((threadIdx.x & 63) == 0) atomicMin(out, m);
}

// bytes -> bits (bit i = word i/64 bit i%64), one wave per word
__global__ __launch_bounds__(256) void k_bytes_to_bits(const uint8_t* __restrict__ marks, uint64_t n,
                                                       unsigned long long* __restrict__ bits) {
  uint64_t w = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (w >= (n + 63) / 64) return;
  uint64_t t = w * 64 + (threadIdx.x & 63);
  uint64_t word = __ballot(t < n && marks[t] != 0);
  if ((threadIdx.x & 63) == 0) bits[w] = word;
}

__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t* __restrict__ a, uint64_t n, uint64_t v) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// hop 1: LOOP_START row i -> first RC row of (loop_entry, read_len)                 (:283-302)
__global__ void k_rm_group_rc(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint64_t n,
                              const uint32_t* __restrict__ starts, uint64_t ns, uint64_t* __restrict__ g,
                              uint32_t* __restrict__ ord) {
  uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  uint32_t i = starts[s];
  g[s] = lower_bound_u64(hi, n, rm_hi(lo[i], 1, hi_len(hi[i]), 0));
  ord[s] = uint32_t(s);
}

// claimed row = group start + rank among the claimers of that group (claimers sorted stably by group,
// i.e. in row order inside a group)
__global__ void k_rm_claim(const uint64_t* __restrict__ gs, const uint32_t* __restrict__ ords, uint64_t ns,
                           uint64_t n_rows, uint32_t* __restrict__ claimed) {
  uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= ns) return;
  uint64_t g = gs[p];
  if (g >= n_rows) { claimed[ords[p]] = 0xffffffffu; return; }  // claimer without this hop
  uint64_t first = lower_bound_u64(gs, ns, g);
  claimed[ords[p]] = uint32_t(g + (p - first));
}

// after hop 1: check the RC row, close unpaired loops, open hop 2                     (:352-365)
__global__ void k_rm_after_rc(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint64_t n,
                              const uint32_t* __restrict__ starts, const uint32_t* __restrict__ rc, uint64_t ns,
                              uint64_t* __restrict__ ptr, uint64_t* __restrict__ g, uint32_t* __restrict__ ord,
                              uint32_t* __restrict__ err) {
  uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  uint32_t i = starts[s], r = rc[s];
  ord[s] = uint32_t(s);
  if (r >= n || hi_type(hi[r]) != 1 || hi_entry(hi[r]) != lo[i] || hi_len(hi[r]) != hi_len(hi[i])) {
    atomicOr(err, 2u);
    g[s] = ~0ull;
    return;
  }
  ptr[i] = r;
  if (lo[r] == kNoLoopEntry) {  // no mate: RC points back to the start
    ptr[r] = i;
    g[s] = ~0ull;
    return;
  }
  g[s] = lower_bound_u64(hi, n, rm_hi(lo[r], 2, hi_mate_len(hi[r]), 0));
}

__global__ void k_rm_after_mate(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint64_t n,
                                const uint32_t* __restrict__ rc, const uint32_t* __restrict__ mate, uint64_t ns,
                                uint64_t* __restrict__ ptr, uint64_t* __restrict__ g, uint32_t* __restrict__ ord,
                                uint32_t* __restrict__ err) {
  uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  ord[s] = uint32_t(s);
  uint32_t m = mate[s];
  if (m == 0xffffffffu) { g[s] = ~0ull; return; }
  uint32_t r = rc[s];
  if (m >= n || hi_type(hi[m]) != 2 || hi_entry(hi[m]) != lo[r] || hi_len(hi[m]) != hi_mate_len(hi[r])) {
    atomicOr(err, 2u);
    g[s] = ~0ull;
    return;
  }
  ptr[r] = m;
  g[s] = lower_bound_u64(hi, n, rm_hi(lo[m], 3, hi_mate_len(hi[r]), 0));
}

__global__ void k_rm_after_mate_rc(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint64_t n,
                                   const uint32_t* __restrict__ starts, const uint32_t* __restrict__ rc,
                                   const uint32_t* __restrict__ mate, const uint32_t* __restrict__ mrc, uint64_t ns,
                                   uint64_t* __restrict__ ptr, uint32_t* __restrict__ err) {
  uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  uint32_t x = mrc[s];
  if (x == 0xffffffffu) return;
  uint32_t m = mate[s], r = rc[s];
  if (x >= n || hi_type(hi[x]) != 3 || hi_entry(hi[x]) != lo[m] || hi_len(hi[x]) != hi_mate_len(hi[r])) {
    atomicOr(err, 2u);
    return;
  }
  ptr[m] = x;
  ptr[x] = starts[s];  // loop back to the beginning (:389)
}

// LSB-first packing of n values at `bits` bits each (packed_varbit_vector.cpp:80-147, packed_vector.h)
__global__ void k_pack_bits_u64(const uint64_t* __restrict__ vals, uint64_t n, int bits, uint64_t nwords,
                                uint64_t* __restrict__ out) {
  uint64_t w = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint64_t bit_lo = w * 64, bit_hi = bit_lo + 64;
  uint64_t word = 0;
  for (uint64_t i = bit_lo / bits; i < n; ++i) {
    uint64_t b = i * bits;
    if (b >= bit_hi) break;
    uint64_t v = vals[i];
    if (b >= bit_lo) word |= v << (b - bit_lo);
    else word |= v >> (bit_lo - b);
  }
  out[w] = word;
}

__global__ void k_iota32(uint32_t* __restrict__ a, uint64_t n) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) a[i] = uint32_t(i);
}
__global__ void k_gather64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n,
                           uint64_t* __restrict__ dst) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_bitcount_groups_rm(const uint64_t* __restrict__ bits, uint64_t nwords, uint64_t ngroups,
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

// bitcount::finalize (io/bitcount.cpp:84-124) for a plain bit vector
static int bitcount_finalize(bg_ctx* ctx, const DevBuf<uint64_t>& bits, uint64_t nbits, DevBuf<uint64_t>* subaccum,
                             DevBuf<uint64_t>* accum) {
  const uint64_t nwords = (nbits + 63) / 64, ngroups = (nbits + 511) / 512, naccum = (nbits + 512) / 512;
  DevBuf<uint32_t> gp;
  DevBuf<uint64_t> total;
  BG_TRY(gp.alloc(ctx, ngroups));
  BG_TRY(total.alloc(ctx, 1));
  BG_TRY(subaccum->alloc(ctx, ngroups));
  BG_TRY(accum->alloc(ctx, naccum));
  BG_HIP(hipMemsetAsync(accum->p, 0, naccum * 8, ctx->stream));
  if (ngroups) {
    k_bitcount_groups_rm<<<cdiv(ngroups, 256), 256, 0, ctx->stream>>>(bits.p, nwords, ngroups, subaccum->p, gp.p);
    BG_LAUNCH_CHECK();
  }
  BG_TRY(exclusive_scan_u32_to_u64(ctx, gp.p, accum->p, ngroups, total.p));
  if (naccum > ngroups)
    BG_HIP(hipMemcpyAsync(accum->p + ngroups, total.p, 8, hipMemcpyDeviceToDevice, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

static int readmap_min_len(bg_ctx* ctx, bg_readmap* rm);
int bitcount_finalize_plain(bg_ctx* ctx, const DevBuf<uint64_t>& bits, uint64_t nbits, DevBuf<uint64_t>* subaccum,
                            DevBuf<uint64_t>* accum);

// stable sort of (key, payload) pairs; returns pointers to the sorted arrays
static int sort_pairs(bg_ctx* ctx, DevBuf<uint64_t>& ka, DevBuf<uint64_t>& kb, DevBuf<uint32_t>& va,
                      DevBuf<uint32_t>& vb, uint64_t n, int bit_hi, uint64_t** ks, uint32_t** vs) {
  bool in_b = false;
  BG_TRY(radix_sort_u64(ctx, ka.p, kb.p, va.p, vb.p, n, 0, bit_hi, &in_b));
  *ks = in_b ? kb.p : ka.p;
  *vs = in_b ? vb.p : va.p;
  return BG_OK;
}

// Rows (hi, lo) in n_slots slots (hi = ~0: unused slot) -> the valid rows sorted by (hi, lo): the total order of
// make_readmap.h:187-213.  LSD: by lo (37 bits), then stably by hi; unused slots sort to the end.
// hi_bits: the bits of hi that tell rows apart (22 + the bits of the largest entry id; an unused slot is all ones there as well
// and still sorts behind every row): the digits above them were passes that only moved the unused slots
static int rm_sort_rows(bg_ctx* ctx, DevBuf<uint64_t>& hi, DevBuf<uint64_t>& lo, uint64_t n_slots, DevBuf<uint64_t>* hi_s,
                        DevBuf<uint64_t>* lo_s, uint64_t* n_rows_out, int hi_bits = 64) {
  if (n_slots >= (1ull << 32)) { set_error("readmap: more than 2^32 rows on one device"); return BG_EINVAL; }
  DevBuf<uint64_t> ka, kb;
  DevBuf<uint32_t> va, vb;
  BG_TRY(ka.alloc(ctx, n_slots));
  BG_TRY(kb.alloc(ctx, n_slots));
  BG_TRY(va.alloc(ctx, n_slots));
  BG_TRY(vb.alloc(ctx, n_slots));
  if (n_slots) {
    BG_HIP(hipMemcpyAsync(ka.p, lo.p, n_slots * 8, hipMemcpyDeviceToDevice, ctx->stream));
    k_iota32<<<cdiv(n_slots, 256), 256, 0, ctx->stream>>>(va.p, n_slots);
    BG_LAUNCH_CHECK();
  }
  uint64_t* ks = nullptr;
  uint32_t* vs = nullptr;
  BG_TRY(sort_pairs(ctx, ka, kb, va, vb, n_slots, 37, &ks, &vs));
  uint64_t* k2 = ks == ka.p ? kb.p : ka.p;
  uint32_t* v2 = vs == va.p ? vb.p : va.p;
  if (n_slots) {
    k_gather64<<<cdiv(n_slots, 256), 256, 0, ctx->stream>>>(hi.p, vs, n_slots, k2);
    BG_LAUNCH_CHECK();
  }
  // second sort: keys k2 (hi), payload vs; scratch = the other pair
  bool in_b = false;
  BG_TRY(radix_sort_u64(ctx, k2, ks, vs, v2, n_slots, 0, hi_bits, &in_b));
  uint64_t* hsorted = in_b ? ks : k2;
  uint32_t* perm = in_b ? v2 : vs;
  DevBuf<uint8_t> f;
  DevBuf<uint32_t> valid_pos;
  BG_TRY(f.alloc(ctx, n_slots));
  if (n_slots) {
    k_rm_flag_valid<<<cdiv(n_slots, 256), 256, 0, ctx->stream>>>(hsorted, n_slots, f.p);
    BG_LAUNCH_CHECK();
  }
  uint64_t n_rows = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, f.p, n_slots, &valid_pos, &n_rows));  // valid rows are a prefix
  BG_TRY(hi_s->alloc(ctx, n_rows));
  BG_TRY(lo_s->alloc(ctx, n_rows));
  if (n_rows) {
    BG_HIP(hipMemcpyAsync(hi_s->p, hsorted, n_rows * 8, hipMemcpyDeviceToDevice, ctx->stream));
    k_gather64<<<cdiv(n_rows, 256), 256, 0, ctx->stream>>>(lo.p, perm, n_rows, lo_s->p);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_rows_out = n_rows;
  return BG_OK;
}

}  // namespace bg


int bg::bitcount_finalize_plain(bg_ctx* ctx, const DevBuf<uint64_t>& bits, uint64_t nbits,
                                DevBuf<uint64_t>* subaccum, DevBuf<uint64_t>* accum) {
  return bitcount_finalize(ctx, bits, nbits, subaccum, accum);
}

using namespace bg;

static int bg::readmap_min_len(bg_ctx* ctx, bg_readmap* rm) {
  rm->min_read_len = 0;
  if (!rm->n_reads) return BG_OK;
  DevBuf<uint32_t> m;
  BG_TRY(m.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(m.p, 0xff, 4, ctx->stream));
  k_rm_min_len<<<unsigned(std::min<uint64_t>(cdiv(rm->n_reads, 256), 1024)), 256, 0, ctx->stream>>>(rm->read_len.p, rm->n_reads, m.p);
  BG_LAUNCH_CHECK();
  BG_HIP(hipMemcpyAsync(&rm->min_read_len, m.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_readmap_build(bg_ctx* ctx, const bg_seqset* ss, const bg_reads* reads, int is_paired,
                                bg_readmap** out) {
  return bg_readmap_build_marked(ctx, ss, reads, is_paired, nullptr, out);
}

// d_marks (one byte per entry of ss, or NULL): ss is a MERGED seqset and the bytes mark the images of the sample's own
// entries under the merge map (bg_seqset_mark_inputs).  A read then goes to the first MARKED entry having it as a
// prefix, which is where make_readmap::create_from_fast_migrate (make_readmap.cpp:455-519) puts it: the read's entry
// in its sample's seqset, carried through the map.  (Sample entries with the read R as a prefix map, in order, to
// marked entries that have R as a prefix; a marked entry with prefix R comes from a sample entry x with x and R both
// prefixes of it, and x cannot be a proper prefix of R in a prefix-free set that covers R.)
extern "C" int bg_readmap_build_marked(bg_ctx* ctx, const bg_seqset* ss, const bg_reads* reads, int is_paired,
                                       const uint8_t* d_marks, bg_readmap** out) {
  if (!ctx || !ss || !reads || !out) { set_error("bg_readmap_build: null argument"); return BG_EINVAL; }
  // A build from items makes its key table when asked -- and is not asked here while its entries can be searched as they are
  // (BG_READMAP_KEYS=1: through the key table as before).  Held to the end of the look-up: seqset_ensure_keys on another thread
  // takes the entries' tables away.
  std::unique_lock<std::mutex> items_lock(const_cast<bg_seqset*>(ss)->keys_mutex);
  static const bool keys_always = [] { const char* e = getenv("BG_READMAP_KEYS"); return e && e[0] == '1'; }();
  const bool through_items = !keys_always && !d_marks && !ss->keys.p && ss->erec.p && ss->oreads.p && ss->edir.p && ss->W && reads->W <= ss->W;
  if (!through_items) {
    items_lock.unlock();
    BG_TRY(seqset_ensure_keys(ctx, const_cast<bg_seqset*>(ss)));
  }
  const bool have_keys = ss->keys.p && ss->W;
  if (ss->n && !have_keys && !through_items && !ss->accum[0].p) { set_error("bg_readmap_build: the seqset is not finalized"); return BG_EINVAL; }
  if (have_keys && reads->W > ss->W && ss->n) { set_error("bg_readmap_build: reads longer than the seqset's entries"); return BG_EINVAL; }
  if (is_paired && (reads->n & 1)) { set_error("bg_readmap_build: paired input needs an even number of reads"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n = reads->n, n_o = 2 * n, n_ent = ss->n;
  const int W = reads->W, EW = ss->W ? ss->W : W;
  // Read ids (rows) are 32-bit, as the reference's are (readmap.h: uint32_t read ids; a 30x human has 1.24 G).  Entry ids
  // are 64-bit throughout (the rows pack 37 bits, make_readmap.h:56-90); only the directory over RESIDENT keys is 32-bit,
  // and resident keys of 2^32 entries would not fit one device anyway -- an opened seqset (FM search) has no such limit.
  if (n_o >= (1ull << 32)) { set_error("bg_readmap_build: more than 2^31 reads"); return BG_EINVAL; }
  if (have_keys && n_ent >= (1ull << 32)) { set_error("bg_readmap_build: more than 2^32 resident entry keys"); return BG_EINVAL; }
  if (n_ent >= kNoLoopEntry) { set_error("bg_readmap_build: more than 2^37 - 1 entries"); return BG_EINVAL; }
  std::unique_ptr<bg_readmap> rm(new bg_readmap);
  rm->ctx = ctx;
  rm->n_entries = n_ent;
  rm->seqset_uuid = ss->uuid;
  // seqset::max_read_len() of the sample's own seqset (biograph_create.cpp:825) = its longest read; a migrated readmap
  // keeps that value on the merged seqset (make_readmap.cpp:459-461)
  rm->max_read_len = d_marks ? reads->max_len : ss->max_len;
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));

  // ---- 1. entry ids
  DevBuf<uint64_t> eid;
  BG_TRY(eid.alloc(ctx, n_o));
  {
    DevBuf<uint32_t> dir;
    ctx->tic("readmap_entry_ids");
    int bases = 0;
    if (through_items) {
      if (n_o) {
        const ItEntries t{reinterpret_cast<const Item*>(ss->erec.p), ss->sizes.p, ss->oreads.p, ss->edir.p, n_ent, ss->W, ss->edir_bases};
        k_rm_entry_ids_items<<<cdiv(n_o, 128), 128, 0, ctx->stream>>>(reads->words.p, reads->len.p, n_o, W, t, eid.p, err.p);
        BG_LAUNCH_CHECK();
      }
    } else if (have_keys) {
      BG_TRY(build_prefix_dir_keys(ctx, ss->keys.p, n_ent, EW, &dir, &bases));
      if (n_o) {
        k_rm_entry_ids<<<cdiv(n_o, 128), 128, 0, ctx->stream>>>(reads->words.p, reads->len.p, n_o, W, ss->keys.p,
                                                                ss->sizes.p, n_ent, EW, dir.p, bases, d_marks, eid.p, err.p);
        BG_LAUNCH_CHECK();
      }
    } else if (n_o) {  // opened from a file (a replica of a merged seqset): FM search
      k_rm_entry_ids_fm<<<cdiv(n_o, 128), 128, 0, ctx->stream>>>(rm_seqset_view(ss), reads->words.p, reads->len.p, n_o, W,
                                                                 d_marks, eid.p, err.p);
      BG_LAUNCH_CHECK();
    }
    ctx->toc(n_o * (W * 8 + 64));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    if (items_lock.owns_lock()) items_lock.unlock();
  }

  // ---- 2. rows, sorted by (hi, lo)
  const uint64_t n_rec = is_paired ? n / 2 : n;
  const uint64_t n_slots = n_rec * (is_paired ? 4 : 2);
  DevBuf<uint64_t> hi_s, lo_s;  // sorted rows
  uint64_t n_rows = 0;
  {
    DevBuf<uint64_t> hi, lo;
    BG_TRY(hi.alloc(ctx, n_slots));
    BG_TRY(lo.alloc(ctx, n_slots));
    if (n_rec) {
      k_rm_rows<<<cdiv(n_rec, 256), 256, 0, ctx->stream>>>(reads->words.p, reads->len.p, n_rec, W, is_paired ? 1 : 0,
                                                          eid.p, hi.p, lo.p);
      BG_LAUNCH_CHECK();
    }
    int ent_bits = 0;
    while (ent_bits < 42 && (n_ent >> ent_bits)) ++ent_bits;
    BG_TRY(rm_sort_rows(ctx, hi, lo, n_slots, &hi_s, &lo_s, &n_rows, 22 + ent_bits));
  }
  eid.reset();
  rm->n_reads = n_rows;
  if (n_rows >= (1ull << 32)) { set_error("bg_readmap_build: too many read ids"); return BG_EINVAL; }

  // ---- 3. per-row fields + sparse_multi
  BG_TRY(rm->read_len.alloc(ctx, n_rows));
  BG_TRY(rm->is_forward.alloc(ctx, n_rows));
  BG_TRY(rm->mate_loop_ptr.alloc(ctx, n_rows));
  BG_TRY(rm->entry_id.alloc(ctx, n_rows));
  BG_TRY(rm->s2m_bits.alloc(ctx, (n_ent + 63) / 64));
  BG_TRY(rm->d2m_bits.alloc(ctx, (n_rows + 63) / 64));
  {
    DevBuf<uint8_t> head, marks;
    BG_TRY(head.alloc(ctx, n_rows));
    BG_TRY(marks.alloc(ctx, n_ent));
    if (n_ent) BG_HIP(hipMemsetAsync(marks.p, 0, n_ent, ctx->stream));
    if (n_rows) {
      BG_HIP(hipMemsetAsync(rm->mate_loop_ptr.p, 0, n_rows * 8, ctx->stream));
      k_rm_row_fields<<<cdiv(n_rows, 256), 256, 0, ctx->stream>>>(hi_s.p, n_rows, rm->read_len.p, rm->is_forward.p,
                                                                 head.p, marks.p, rm->entry_id.p);
      BG_LAUNCH_CHECK();
      k_bytes_to_bits<<<cdiv((n_rows + 63) / 64, 4), 256, 0, ctx->stream>>>(
          head.p, n_rows, reinterpret_cast<unsigned long long*>(rm->d2m_bits.p));
      BG_LAUNCH_CHECK();
    }
    if (n_ent) {
      k_bytes_to_bits<<<cdiv((n_ent + 63) / 64, 4), 256, 0, ctx->stream>>>(
          marks.p, n_ent, reinterpret_cast<unsigned long long*>(rm->s2m_bits.p));
      BG_LAUNCH_CHECK();
    }
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  BG_TRY(bitcount_finalize(ctx, rm->s2m_bits, n_ent, &rm->s2m_sub, &rm->s2m_acc));
  BG_TRY(bitcount_finalize(ctx, rm->d2m_bits, n_rows, &rm->d2m_sub, &rm->d2m_acc));
  BG_TRY(readmap_min_len(ctx, rm.get()));

  // ---- 4. mate loops: three claim hops
  if (n_rows) {
    DevBuf<uint8_t> f;
    DevBuf<uint32_t> starts;
    uint64_t ns = 0;
    BG_TRY(f.alloc(ctx, n_rows));
    k_rm_flag_type<<<cdiv(n_rows, 256), 256, 0, ctx->stream>>>(hi_s.p, n_rows, 0, f.p);
    BG_LAUNCH_CHECK();
    BG_TRY(compact_flagged<uint32_t>(ctx, f.p, n_rows, &starts, &ns));
    DevBuf<uint64_t> ga, gb;
    DevBuf<uint32_t> oa, ob, rc, mate, mrc;
    BG_TRY(ga.alloc(ctx, ns));
    BG_TRY(gb.alloc(ctx, ns));
    BG_TRY(oa.alloc(ctx, ns));
    BG_TRY(ob.alloc(ctx, ns));
    BG_TRY(rc.alloc(ctx, ns));
    BG_TRY(mate.alloc(ctx, ns));
    BG_TRY(mrc.alloc(ctx, ns));
    if (ns) {
      ctx->tic("readmap_mate_loops");
      uint64_t* gs = nullptr;
      uint32_t* os = nullptr;
      k_rm_group_rc<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(hi_s.p, lo_s.p, n_rows, starts.p, ns, ga.p, oa.p);
      BG_LAUNCH_CHECK();
      BG_TRY(sort_pairs(ctx, ga, gb, oa, ob, ns, 64, &gs, &os));
      k_rm_claim<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(gs, os, ns, n_rows, rc.p);
      BG_LAUNCH_CHECK();
      k_rm_after_rc<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(hi_s.p, lo_s.p, n_rows, starts.p, rc.p, ns,
                                                             rm->mate_loop_ptr.p, ga.p, oa.p, err.p);
      BG_LAUNCH_CHECK();
      BG_TRY(sort_pairs(ctx, ga, gb, oa, ob, ns, 64, &gs, &os));
      k_rm_claim<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(gs, os, ns, n_rows, mate.p);
      BG_LAUNCH_CHECK();
      k_rm_after_mate<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(hi_s.p, lo_s.p, n_rows, rc.p, mate.p, ns,
                                                               rm->mate_loop_ptr.p, ga.p, oa.p, err.p);
      BG_LAUNCH_CHECK();
      BG_TRY(sort_pairs(ctx, ga, gb, oa, ob, ns, 64, &gs, &os));
      k_rm_claim<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(gs, os, ns, n_rows, mrc.p);
      BG_LAUNCH_CHECK();
      k_rm_after_mate_rc<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(hi_s.p, lo_s.p, n_rows, starts.p, rc.p, mate.p, mrc.p,
                                                                  ns, rm->mate_loop_ptr.p, err.p);
      BG_LAUNCH_CHECK();
      ctx->toc(ns * 64);
    }
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err & 1) { set_error("bg_readmap_build: a read was not found in the seqset"); return BG_EINTERNAL; }
  if (h_err & 2) { set_error("bg_readmap_build: mate loop claim failed (inconsistent rows)"); return BG_EINTERNAL; }
  if (h_err & 4) { set_error("bg_readmap_build: a read has no marked entry (the marks are not this sample's)"); return BG_EINVAL; }
  *out = rm.release();
  return BG_OK;
}

// ---- a readmap from its per-read columns (the pieces of a sharded build gathered on one device; hand-made input) ----
namespace bg {
__global__ void k_rm_from_arrays(const uint64_t* __restrict__ entry, const uint32_t* __restrict__ len,
                                 const uint32_t* __restrict__ ptr, const uint8_t* __restrict__ fwd, uint64_t n, uint64_t n_ent,
                                 uint64_t* __restrict__ o_entry, uint64_t* __restrict__ o_len, uint64_t* __restrict__ o_ptr,
                                 uint64_t* __restrict__ o_fwd, uint8_t* __restrict__ head,
                                 unsigned long long* __restrict__ s2m_bits, uint32_t* __restrict__ err) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t e = entry[i];
  const bool first = i == 0 || entry[i - 1] != e;
  if (e >= n_ent || (i && entry[i - 1] > e) || ptr[i] >= n) { atomicOr(err, 1u); return; }
  o_entry[i] = e;
  o_len[i] = len[i];
  o_ptr[i] = ptr[i];
  o_fwd[i] = fwd[i] ? 1 : 0;
  head[i] = first ? 1 : 0;
  if (first) atomicOr(&s2m_bits[e >> 6], 1ull << (e & 63));  // one per entry group: a set bit per entry that carries reads
}
}  // namespace bg

extern "C" int bg_readmap_from_arrays(bg_ctx* ctx, uint64_t n_entries, uint32_t max_read_len, const char* seqset_uuid,
                                      uint64_t n_reads, const uint64_t* d_entry_id, const uint32_t* d_read_len,
                                      const uint32_t* d_mate_loop_ptr, const uint8_t* d_is_forward, bg_readmap** out) {
  if (!ctx || !out || (n_reads && (!d_entry_id || !d_read_len || !d_mate_loop_ptr || !d_is_forward))) {
    set_error("bg_readmap_from_arrays: null argument");
    return BG_EINVAL;
  }
  if (n_reads >= (1ull << 32)) { set_error("bg_readmap_from_arrays: more than 2^32 read ids"); return BG_EINVAL; }
  if (n_entries >= kNoLoopEntry) { set_error("bg_readmap_from_arrays: more than 2^37 - 1 entries"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_readmap> rm(new bg_readmap);
  rm->ctx = ctx;
  rm->n_entries = n_entries;
  rm->n_reads = n_reads;
  rm->max_read_len = max_read_len;
  rm->seqset_uuid = seqset_uuid ? seqset_uuid : "";
  const uint64_t n = n_reads;
  BG_TRY(rm->read_len.alloc(ctx, n));
  BG_TRY(rm->is_forward.alloc(ctx, n));
  BG_TRY(rm->mate_loop_ptr.alloc(ctx, n));
  BG_TRY(rm->entry_id.alloc(ctx, n));
  BG_TRY(rm->s2m_bits.alloc(ctx, (n_entries + 63) / 64));
  BG_TRY(rm->d2m_bits.alloc(ctx, (n + 63) / 64));
  BG_HIP(hipMemsetAsync(rm->s2m_bits.p, 0, ((n_entries + 63) / 64 ? (n_entries + 63) / 64 : 1) * 8, ctx->stream));
  DevBuf<uint8_t> head;
  DevBuf<uint32_t> err;
  BG_TRY(head.alloc(ctx, n));
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  if (n) {
    k_rm_from_arrays<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_entry_id, d_read_len, d_mate_loop_ptr, d_is_forward, n, n_entries,
                                                           rm->entry_id.p, rm->read_len.p, rm->mate_loop_ptr.p, rm->is_forward.p,
                                                           head.p, reinterpret_cast<unsigned long long*>(rm->s2m_bits.p), err.p);
    BG_LAUNCH_CHECK();
    k_bytes_to_bits<<<cdiv((n + 63) / 64, 4), 256, 0, ctx->stream>>>(head.p, n, reinterpret_cast<unsigned long long*>(rm->d2m_bits.p));
    BG_LAUNCH_CHECK();
  }
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_readmap_from_arrays: entry ids must ascend and stay below n_entries, loop pointers below n_reads"); return BG_EINVAL; }
  BG_TRY(bitcount_finalize(ctx, rm->s2m_bits, n_entries, &rm->s2m_sub, &rm->s2m_acc));
  BG_TRY(bitcount_finalize(ctx, rm->d2m_bits, n, &rm->d2m_sub, &rm->d2m_acc));
  BG_TRY(readmap_min_len(ctx, rm.get()));
  *out = rm.release();
  return BG_OK;
}

extern "C" uint64_t bg_readmap_size(const bg_readmap* rm) { return rm ? rm->n_reads : 0; }
extern "C" uint32_t bg_readmap_max_read_len(const bg_readmap* rm) { return rm ? rm->max_read_len : 0; }
extern "C" int bg_readmap_has_mate_loop(const bg_readmap* rm) { return rm && !rm->pair_mode ? 1 : 0; }
extern "C" void bg_readmap_free(bg_readmap* rm) {
  bg::readmap_tables_forget(rm);
  delete rm;
}

// ---- bg::readmap_tables (bg_internal.h): the narrow columns the discovery engine reads, made on the device
namespace bg {
__global__ void k_rt_narrow(const uint64_t* __restrict__ entry_id, const uint64_t* __restrict__ mate_loop, const uint64_t* __restrict__ is_forward,
                            const uint64_t* __restrict__ read_len, uint64_t n_reads, uint64_t n_entries, unsigned long long* __restrict__ has_reads,
                            uint8_t* __restrict__ head, uint32_t* __restrict__ mate32, uint8_t* __restrict__ fwd8, uint8_t* __restrict__ len8,
                            uint32_t* __restrict__ err) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_reads) return;
  const uint64_t e = entry_id[i];
  const bool h = i == 0 || entry_id[i - 1] != e;
  if (i && entry_id[i - 1] > e) atomicOr(&err[0], 1u);  // (read ids are grouped by ascending entry)
  mate32[i] = uint32_t(mate_loop[i]);
  fwd8[i] = uint8_t(is_forward[i] & 1);
  const uint64_t len = read_len[i];
  len8[i] = uint8_t(len);
  atomicMin(&err[1], uint32_t(len > 0xffffffffull ? 0xffffffffull : len));
  if (len > 255) atomicOr(&err[0], 4u);
  if (e >= n_entries) { atomicOr(&err[0], 2u); head[i] = 0; return; }
  head[i] = h ? 1 : 0;
  if (h) atomicOr(&has_reads[e >> 6], 1ull << (e & 63));
}
__global__ void k_rt_popc(const unsigned long long* __restrict__ words, uint64_t n, uint32_t* __restrict__ out) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = uint32_t(__popcll(words[i]));
}

ReadmapTables::~ReadmapTables() {
  if (d_word_rank) (void)hipFree(d_word_rank);
  if (d_group_start) (void)hipFree(d_group_start);
  if (d_mate_loop) (void)hipFree(d_mate_loop);
  if (d_is_forward) (void)hipFree(d_is_forward);
}

static std::mutex g_rt_mutex;
static std::map<const bg_readmap*, std::weak_ptr<const ReadmapTables>> g_rt_cache;

void readmap_tables_forget(const bg_readmap* rm) {
  std::lock_guard<std::mutex> lk(g_rt_mutex);
  g_rt_cache.erase(rm);
}

static int readmap_tables_make(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, ReadmapTables* t) {
  const uint64_t n_ent = ss->n, n_reads = rm->n_reads;
  const uint64_t n_words = (n_ent + 63) / 64 + 1;
  if (n_reads >= (uint64_t(1) << 32)) { set_error("readmap tables: 2^32 read ids or more (read ids are 32-bit: readmap.h)"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  DevBuf<unsigned long long> d_has;
  DevBuf<uint8_t> d_head, d_len8;
  DevBuf<uint32_t> d_popc, d_err, d_heads, d_total;
  BG_TRY(d_has.alloc(ctx, n_words));
  BG_TRY(d_head.alloc(ctx, n_reads));
  BG_TRY(d_len8.alloc(ctx, n_reads));
  BG_TRY(d_popc.alloc(ctx, n_words + 1));
  BG_TRY(d_err.alloc(ctx, 2));
  BG_TRY(d_total.alloc(ctx, 1));
  BG_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_word_rank), (n_words + 1) * 4));
  BG_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_mate_loop), (n_reads ? n_reads : 1) * 4));
  BG_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_is_forward), n_reads ? n_reads : 1));
  BG_HIP(hipMemsetAsync(d_has.p, 0, n_words * 8, ctx->stream));
  const uint32_t err0[2] = {0u, 0xffffffffu};
  BG_HIP(copy_h2d(ctx, d_err.p, err0, 8));
  if (n_reads) {
    k_rt_narrow<<<cdiv(n_reads, 256), 256, 0, ctx->stream>>>(rm->entry_id.p, rm->mate_loop_ptr.p, rm->is_forward.p, rm->read_len.p, n_reads, n_ent,
                                                             d_has.p, d_head.p, t->d_mate_loop, t->d_is_forward, d_len8.p, d_err.p);
    BG_LAUNCH_CHECK();
  }
  k_rt_popc<<<cdiv(n_words, 256), 256, 0, ctx->stream>>>(d_has.p, n_words, d_popc.p);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32(ctx, d_popc.p, t->d_word_rank, n_words, d_total.p));
  BG_HIP(hipMemcpyAsync(t->d_word_rank + n_words, d_total.p, 4, hipMemcpyDeviceToDevice, ctx->stream));
  uint64_t n_groups = 0;
  if (n_reads) BG_TRY(compact_flagged<uint32_t>(ctx, d_head.p, n_reads, &d_heads, &n_groups));
  uint32_t h_err[2] = {0, 0}, h_total = 0;
  BG_HIP(copy_d2h(ctx, h_err, d_err.p, 8));
  BG_HIP(copy_d2h(ctx, &h_total, d_total.p, 4));
  if (h_err[0] & 2) { set_error("readmap tables: the readmap names an entry the seqset does not have"); return BG_EINVAL; }
  if (h_err[0] & 4) { set_error("readmap tables: a read of more than 255 bases"); return BG_EINVAL; }
  if ((h_err[0] & 1) || n_groups != h_total) { set_error("the readmap's read ids are not grouped by entry"); return BG_EINVAL; }
  t->min_read_len = n_reads ? h_err[1] : 0;
  BG_HIP(hipMalloc(reinterpret_cast<void**>(&t->d_group_start), (n_groups + 1) * 4));
  if (n_groups) BG_HIP(hipMemcpyAsync(t->d_group_start, d_heads.p, n_groups * 4, hipMemcpyDeviceToDevice, ctx->stream));
  const uint32_t last = uint32_t(n_reads);
  BG_HIP(copy_h2d(ctx, t->d_group_start + n_groups, &last, 4));
  d_heads.reset();
  d_head.reset();
  d_popc.reset();
  // the narrow tables to the host, through the context's pinned double buffer
  auto down = [&](auto& v, const void* src, size_t count) -> int {
    v.resize(count);
    return count ? staged_download(ctx, v.data(), src, count * sizeof(v[0]), 0) : BG_OK;
  };
  BG_TRY(down(t->has_reads, d_has.p, n_words));
  BG_TRY(down(t->word_rank, t->d_word_rank, n_words + 1));
  BG_TRY(down(t->group_start, t->d_group_start, n_groups + 1));
  BG_TRY(down(t->mate_loop, t->d_mate_loop, n_reads));
  BG_TRY(down(t->is_forward, t->d_is_forward, n_reads));
  BG_TRY(down(t->read_len, d_len8.p, n_reads));
  return BG_OK;
}

int readmap_tables(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, std::shared_ptr<const ReadmapTables>* out) {
  // (one maker at a time: a second caller for the same readmap -- the coverage session opened beside the tracer's -- waits and
  // takes the first one's tables)
  std::lock_guard<std::mutex> lk(g_rt_mutex);
  auto it = g_rt_cache.find(rm);
  if (it != g_rt_cache.end()) {
    if (auto have = it->second.lock()) { *out = have; return BG_OK; }
    g_rt_cache.erase(it);
  }
  std::shared_ptr<ReadmapTables> t(new ReadmapTables);
  BG_TRY(readmap_tables_make(ctx, ss, rm, t.get()));
  g_rt_cache[rm] = t;
  *out = t;
  return BG_OK;
}
}  // namespace bg

static unsigned rm_bits_for_value(uint64_t v) { unsigned b = 0; while (v) { ++b; v >>= 1; } return b; }

static int rm_part_bytes(bg_ctx* ctx, const bg_readmap* rm, const std::string& part, std::string* out) {
  auto dl = [&](const uint64_t* d, uint64_t words) -> int {
    out->resize(words * 8);
    if (!words) return BG_OK;
    if (words * 8 >= (8u << 20)) return staged_download(ctx, &(*out)[0], d, words * 8, 0);
    BG_HIP(hipMemcpyAsync(&(*out)[0], d, words * 8, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    return BG_OK;
  };
  const uint64_t n = rm->n_reads, ne = rm->n_entries;
  if (part == "read_ids/source_to_mid/bits") return dl(rm->s2m_bits.p, (ne + 63) / 64);
  if (part == "read_ids/source_to_mid/subaccum") return dl(rm->s2m_sub.p, (ne + 511) / 512);
  if (part == "read_ids/source_to_mid/accum") return dl(rm->s2m_acc.p, (ne + 512) / 512);
  if (part == "read_ids/dest_to_mid/bits") return dl(rm->d2m_bits.p, (n + 63) / 64);
  if (part == "read_ids/dest_to_mid/subaccum") return dl(rm->d2m_sub.p, (n + 511) / 512);
  if (part == "read_ids/dest_to_mid/accum") return dl(rm->d2m_acc.p, (n + 512) / 512);
  const uint64_t* src = nullptr;
  int bits = 0;
  if (part == "read_lengths/elements") { src = rm->read_len.p; bits = int(rm_bits_for_value(rm->max_read_len)); }
  else if (part == "mate_loop_ptr/elements") { src = rm->mate_loop_ptr.p; bits = int(rm_bits_for_value(n)); }
  else if (part == "is_forward/packed_data") { src = rm->is_forward.p; bits = 1; }
  else { set_error("bg_readmap_part: unknown part '%s'", part.c_str()); return BG_EINVAL; }
  uint64_t nwords = (n * uint64_t(bits) + 63) / 64;
  out->resize(nwords * 8);
  if (!nwords) return BG_OK;
  DevBuf<uint64_t> packed;
  BG_TRY(packed.alloc(ctx, nwords));
  k_pack_bits_u64<<<cdiv(nwords, 256), 256, 0, ctx->stream>>>(src, n, bits, nwords, packed.p);
  BG_LAUNCH_CHECK();
  return dl(packed.p, nwords);
}

extern "C" int bg_readmap_part(bg_ctx* ctx, const bg_readmap* rm, const char* part, void* h_out, uint64_t* size) {
  if (!ctx || !rm || !part || !size) { set_error("bg_readmap_part: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::string data;
  BG_TRY(rm_part_bytes(ctx, rm, part, &data));
  if (h_out) {
    if (*size < data.size()) { *size = data.size(); set_error("bg_readmap_part: buffer too small"); return BG_EINVAL; }
    memcpy(h_out, data.data(), data.size());
  }
  *size = data.size();
  return BG_OK;
}

static std::string rm_version_json(const char* type, int maj, int min, int pat) {
  char buf[256];
  snprintf(buf, sizeof buf,
           "{\"part_type\":\"%s\",\"version\":{\"build\":\"\",\"major\":%d,\"minor\":%d,\"patch\":%d,\"pre\":\"\"}}", type,
           maj, min, pat);
  return buf;
}

// Spiral container of the readmap (member order of make_readmap::create_common / create_from_reads,
// as found in father_lambda.bg/coverage/*.readmap).  h_sha1_hex (41 bytes) receives the SHA-1 of the
// written file: the reference names the file after it (biograph_create.cpp:827-828).
static int rm_no_loops(const char* what) {
  set_error("%s: readmap has no mate loop table; use \"biograph upgrade\" to construct it", what);  // readmap.cpp:249-253
  return BG_EINVAL;
}

extern "C" int bg_readmap_write(bg_ctx* ctx, const bg_readmap* rm, const char* path, int argc,
                                const char* const* argv, char* h_sha1_hex) {
  if (!ctx || !rm || !path) { set_error("bg_readmap_write: null argument"); return BG_EINVAL; }
  if (rm->pair_mode) return rm_no_loops("bg_readmap_write");
  BG_HIP(hipSetDevice(ctx->device));
  std::vector<ZipPart> parts;
  time_t now = time(nullptr);
  char tbuf[64];
  ctime_r(&now, tbuf);
  std::string ttext(tbuf);
  if (!ttext.empty() && ttext.back() == '\n') ttext.pop_back();
  std::string cmd = "[";
  for (int i = 0; i < argc; ++i) {
    if (i) cmd += ",";
    cmd += "\"";
    for (const char* c = argv[i] ? argv[i] : ""; *c; ++c) {
      if (*c == '"' || *c == '\\') cmd.push_back('\\');
      cmd.push_back(*c);
    }
    cmd += "\"";
  }
  cmd += "]";
  std::string uuid = make_uuid();
  char fi[4096];
  snprintf(fi, sizeof fi,
           "{\"build_host\":\"\",\"build_is_clean\":true,\"build_revision\":\"biograph_amd\",\"build_timestamp\":0,"
           "\"build_timestamp_text\":\"\",\"build_user\":\"\",\"command_line\":%s,\"create_timestamp\":%lld,"
           "\"create_timestamp_text\":\"%s\",\"uuid\":\"%s\"}",
           cmd.c_str(), (long long)now, ttext.c_str(), uuid.c_str());
  parts.push_back(ZipPart{"file_info.json", fi, false});
  parts.push_back(ZipPart{"part_info.json", rm_version_json("readmap", 1, 2, 0), false});
  parts.push_back(ZipPart{"readmap.json", "{\"seqset_uuid\":\"" + rm->seqset_uuid + "\"}", false});
  parts.push_back(ZipPart{"read_ids/part_info.json", rm_version_json("sparse_multi", 1, 0, 0), false});
  std::string data;
  char js[128];
  for (int sel = 0; sel < 2; ++sel) {
    std::string dir = sel ? "read_ids/dest_to_mid" : "read_ids/source_to_mid";
    parts.push_back(ZipPart{dir + "/part_info.json", rm_version_json("bitcount", 1, 0, 0), false});
    snprintf(js, sizeof js, "{\"nbits\":%llu}", (unsigned long long)(sel ? rm->n_reads : rm->n_entries));
    parts.push_back(ZipPart{dir + "/bitcount.json", js, false});
    for (const char* w : {"bits", "subaccum", "accum"}) {
      BG_TRY(rm_part_bytes(ctx, rm, dir + "/" + w, &data));
      parts.push_back(ZipPart{dir + "/" + w, data, true});
    }
  }
  for (int which = 0; which < 2; ++which) {
    std::string dir = which ? "mate_loop_ptr" : "read_lengths";
    uint64_t maxv = which ? rm->n_reads : rm->max_read_len;
    parts.push_back(ZipPart{dir + "/part_info.json", rm_version_json("packed_varbit_vector", 1, 0, 0), false});
    snprintf(js, sizeof js, "{\"bits_per_value\":%u,\"element_count\":%llu,\"max_value\":%llu}", rm_bits_for_value(maxv),
             (unsigned long long)rm->n_reads, (unsigned long long)maxv);
    parts.push_back(ZipPart{dir + "/packed_varbit_vector.json", js, false});
    BG_TRY(rm_part_bytes(ctx, rm, dir + "/elements", &data));
    parts.push_back(ZipPart{dir + "/elements", data, true});
  }
  parts.push_back(ZipPart{"is_forward/part_info.json", rm_version_json("packed_vector", 1, 0, 0), false});
  BG_TRY(rm_part_bytes(ctx, rm, "is_forward/packed_data", &data));
  parts.push_back(ZipPart{"is_forward/packed_data", data, true});
  snprintf(js, sizeof js, "{\"value_count\":%llu,\"value_width_bits\":1}", (unsigned long long)rm->n_reads);
  parts.push_back(ZipPart{"is_forward/packed_vector.json", js, false});
  BG_TRY(spiral_zip_write(path, parts));
  if (h_sha1_hex) BG_TRY(sha1_file_hex(path, h_sha1_hex));
  return BG_OK;
}

// ---- open a readmap container (readmap::readmap / open_spiral_file, bio_base/readmap.cpp) ----------
static bool rm_json_u64(const std::string& js, const char* key, uint64_t* out) {
  std::string pat = std::string("\"") + key + "\":";
  size_t p = js.find(pat);
  if (p == std::string::npos) return false;
  *out = strtoull(js.c_str() + p + pat.size(), nullptr, 10);
  return true;
}

// ---- bg_readmap_open's columns on the device: the container's packed vectors go up as they are (a seventh of the bytes of the
// unpacked 64-bit columns) and are unpacked there; the entry of read id i = the (rank of i among the group heads)-th marked entry
namespace bg {
__global__ void k_rmo_unpack(const uint64_t* __restrict__ w, uint64_t nw, uint64_t n, unsigned bits, uint64_t* __restrict__ out,
                             unsigned long long* __restrict__ vmax) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t bit = i * bits, q = bit >> 6, r = bit & 63;
  uint64_t x = q < nw ? w[q] >> r : 0;
  if (r + bits > 64 && q + 1 < nw) x |= w[q + 1] << (64 - r);
  x &= bits == 64 ? ~0ull : ((1ull << bits) - 1);
  out[i] = x;
  if (vmax) {  // (a wave's maximum, one atomic)
    unsigned long long m = x;
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned long long o = __shfl_xor(m, d, 64);
      m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(vmax, m);
  }
}
__global__ void k_rmo_popc(const uint64_t* __restrict__ w, uint64_t nwords, uint64_t nbits, uint32_t* __restrict__ out) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  uint64_t x = w[i];
  if (i == nbits / 64 && (nbits & 63)) x &= (1ull << (nbits & 63)) - 1;
  if (i > nbits / 64 || (i == nbits / 64 && !(nbits & 63))) x = 0;
  out[i] = uint32_t(__popcll(x));
}
// mid_to_entry[k] = the k-th marked entry
__global__ void k_rmo_marked(const uint64_t* __restrict__ w, uint64_t nwords, uint64_t nbits, const uint64_t* __restrict__ before,
                             uint64_t* __restrict__ mid_to_entry) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  uint64_t x = w[i];
  if (i == nbits / 64 && (nbits & 63)) x &= (1ull << (nbits & 63)) - 1;
  if (i > nbits / 64 || (i == nbits / 64 && !(nbits & 63))) x = 0;
  uint64_t at = before[i];
  while (x) {
    mid_to_entry[at++] = i * 64 + uint64_t(__ffsll((unsigned long long)x) - 1);
    x &= x - 1;
  }
}
// entry of read id i (sparse_multi::reverse_lookup) and the open path's consistency checks
__global__ void k_rmo_entries(const uint64_t* __restrict__ heads, const uint64_t* __restrict__ heads_before, uint64_t n,
                              const uint64_t* __restrict__ mid_to_entry, uint64_t n_marked, const uint64_t* __restrict__ ptr, int pair_mode,
                              uint64_t* __restrict__ entry, uint32_t* __restrict__ err) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t word = heads[i >> 6];
  const uint64_t mid = heads_before[i >> 6] + uint64_t(__popcll(word & ((2ull << (i & 63)) - 1)));  // heads at or before i
  const uint64_t p = ptr[i];
  if (mid == 0 || mid > n_marked || (p >= n && !(pair_mode && p == 0xffffffffull))) { atomicOr(err, 1u); entry[i] = 0; return; }
  entry[i] = mid_to_entry[mid - 1];
}
}  // namespace bg

extern "C" int bg_readmap_open(bg_ctx* ctx, const char* path, bg_readmap** out) {
  if (!ctx || !path || !out) { set_error("bg_readmap_open: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::map<std::string, std::string> parts;
  BG_TRY(spiral_zip_read(path, &parts));
  auto need = [&](const char* name) -> const std::string* {
    auto it = parts.find(name);
    if (it == parts.end()) { set_error("%s: member %s missing (readmaps older than format 1.2.0 carry no mate loop table; `biograph upgrade` them)", path, name); return nullptr; }
    return &it->second;
  };
  // format 1.1.0 (v3.x): read_lengths = raw uint8, mate_loop_ptr = 32-bit packed_vector; format 1.2.0: both
  // packed_varbit_vectors (readmap.cpp open path, int_map_interface.cpp detect_subpart_or_uint8_membuf)
  const bool legacy = parts.count("read_lengths") != 0;
  // older still: mates as a pair table instead of the loop table (readmap.cpp:48-51).  has_mate / get_mate answer from
  // it (readmap.cpp:207-239); everything that follows loops (get_rev_comp, get_mate_rc, migrate, write) refuses
  // such a readmap like the reference does (readmap.cpp:249-253).
  const bool pair_mode = legacy && !parts.count("mate_loop_ptr/packed_data") && parts.count("mate_pair_ptr/packed_data");
  if (legacy && !pair_mode && !parts.count("mate_loop_ptr/packed_data")) {
    set_error("%s: readmap has no pairing data", path);
    return BG_EIO;
  }
  const std::string *s2m_js = need("read_ids/source_to_mid/bitcount.json"), *s2m = need("read_ids/source_to_mid/bits"),
                    *d2m_js = need("read_ids/dest_to_mid/bitcount.json"), *d2m = need("read_ids/dest_to_mid/bits"),
                    *rl = need(legacy ? "read_lengths" : "read_lengths/elements"),
                    *ml = need(pair_mode ? "mate_pair_ptr/packed_data" : legacy ? "mate_loop_ptr/packed_data" : "mate_loop_ptr/elements"),
                    *isf = need("is_forward/packed_data"), *rmj = need("readmap.json");
  if (!s2m_js || !s2m || !d2m_js || !d2m || !rl || !ml || !isf || !rmj) return BG_EIO;
  uint64_t n_ent = 0, n = 0, rl_bits = 8, ml_bits = 32, max_len = 0;
  bool ok = rm_json_u64(*s2m_js, "nbits", &n_ent) && rm_json_u64(*d2m_js, "nbits", &n);
  if (ok && !legacy) {
    const std::string *rl_js = need("read_lengths/packed_varbit_vector.json"),
                      *ml_js = need("mate_loop_ptr/packed_varbit_vector.json");
    if (!rl_js || !ml_js) return BG_EIO;
    ok = rm_json_u64(*rl_js, "bits_per_value", &rl_bits) && rm_json_u64(*ml_js, "bits_per_value", &ml_bits) &&
         rm_json_u64(*rl_js, "max_value", &max_len);
  }
  if (!ok) {
    set_error("%s: malformed readmap metadata", path);
    return BG_EIO;
  }
  if (n >= (1ull << 32)) { set_error("bg_readmap_open: more than 2^32 read ids"); return BG_EINVAL; }
  std::unique_ptr<bg_readmap> rm(new bg_readmap);
  rm->ctx = ctx;
  rm->n_reads = n;
  rm->n_entries = n_ent;
  rm->max_read_len = uint32_t(max_len);
  rm->pair_mode = pair_mode;
  size_t p = rmj->find("\"seqset_uuid\":\"");
  if (p != std::string::npos) rm->seqset_uuid = rmj->substr(p + 15, rmj->find('"', p + 15) - (p + 15));
  if (rl->size() < (n * rl_bits + 7) / 8 || ml->size() < (n * ml_bits + 7) / 8) { set_error("%s: truncated vectors", path); return BG_EIO; }
  if (s2m->size() < (n_ent + 63) / 64 * 8 || d2m->size() < (n + 63) / 64 * 8) { set_error("%s: truncated bit vectors", path); return BG_EIO; }
  {
    // the packed members to the device as they are (padded to whole words; 130 MB for 26 M read ids: plain copies -- a context
    // that opens a readmap and nothing else need not lock staging buffers for it)
    auto up_blob = [&](const std::string& blob, DevBuf<uint64_t>* d, uint64_t* nw) -> int {
      *nw = (blob.size() + 7) / 8 + 1;
      BG_TRY(d->alloc(ctx, *nw));
      BG_HIP(hipMemsetAsync(d->p, 0, *nw * 8, ctx->stream));
      if (!blob.empty()) BG_HIP(hipMemcpyAsync(d->p, blob.data(), blob.size(), hipMemcpyHostToDevice, ctx->stream));
      return BG_OK;
    };
    DevBuf<uint64_t> b_len, b_ptr, b_fwd;
    uint64_t nw_len = 0, nw_ptr = 0, nw_fwd = 0, nw_s2m = (n_ent + 63) / 64, nw_d2m = (n + 63) / 64;
    BG_TRY(up_blob(*rl, &b_len, &nw_len));
    BG_TRY(up_blob(*ml, &b_ptr, &nw_ptr));
    BG_TRY(up_blob(*isf, &b_fwd, &nw_fwd));
    BG_TRY(rm->s2m_bits.alloc(ctx, nw_s2m));
    BG_TRY(rm->d2m_bits.alloc(ctx, nw_d2m));
    if (nw_s2m) BG_HIP(hipMemcpyAsync(rm->s2m_bits.p, s2m->data(), nw_s2m * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nw_d2m) BG_HIP(hipMemcpyAsync(rm->d2m_bits.p, d2m->data(), nw_d2m * 8, hipMemcpyHostToDevice, ctx->stream));
    BG_TRY(rm->read_len.alloc(ctx, n));
    BG_TRY(rm->mate_loop_ptr.alloc(ctx, n));
    BG_TRY(rm->is_forward.alloc(ctx, n));
    BG_TRY(rm->entry_id.alloc(ctx, n));
    DevBuf<unsigned long long> vmax;
    DevBuf<uint32_t> err, pc_s, pc_d;
    DevBuf<uint64_t> before_s, before_d, mid_to_entry, tot;
    BG_TRY(vmax.alloc(ctx, 1));
    BG_TRY(err.alloc(ctx, 1));
    BG_TRY(tot.alloc(ctx, 2));
    BG_HIP(hipMemsetAsync(vmax.p, 0, 8, ctx->stream));
    BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
    if (n) {
      k_rmo_unpack<<<cdiv(n, 256), 256, 0, ctx->stream>>>(b_len.p, nw_len, n, unsigned(rl_bits), rm->read_len.p, vmax.p);
      BG_LAUNCH_CHECK();
      k_rmo_unpack<<<cdiv(n, 256), 256, 0, ctx->stream>>>(b_ptr.p, nw_ptr, n, unsigned(ml_bits), rm->mate_loop_ptr.p, nullptr);
      BG_LAUNCH_CHECK();
      k_rmo_unpack<<<cdiv(n, 256), 256, 0, ctx->stream>>>(b_fwd.p, nw_fwd, n, 1u, rm->is_forward.p, nullptr);
      BG_LAUNCH_CHECK();
    }
    // marked entries in order, group heads before every word
    BG_TRY(pc_s.alloc(ctx, nw_s2m + 1));
    BG_TRY(pc_d.alloc(ctx, nw_d2m + 1));
    BG_TRY(before_s.alloc(ctx, nw_s2m + 1));
    BG_TRY(before_d.alloc(ctx, nw_d2m + 1));
    uint64_t h_tot[2] = {0, 0};
    if (nw_s2m) {
      k_rmo_popc<<<cdiv(nw_s2m, 256), 256, 0, ctx->stream>>>(rm->s2m_bits.p, nw_s2m, n_ent, pc_s.p);
      BG_LAUNCH_CHECK();
      BG_TRY(exclusive_scan_u32_to_u64(ctx, pc_s.p, before_s.p, nw_s2m, tot.p));
    }
    if (nw_d2m) {
      k_rmo_popc<<<cdiv(nw_d2m, 256), 256, 0, ctx->stream>>>(rm->d2m_bits.p, nw_d2m, n, pc_d.p);
      BG_LAUNCH_CHECK();
      BG_TRY(exclusive_scan_u32_to_u64(ctx, pc_d.p, before_d.p, nw_d2m, tot.p + 1));
    }
    if (nw_s2m) BG_HIP(copy_d2h(ctx, h_tot, tot.p, 8));
    const uint64_t n_marked = h_tot[0];
    BG_TRY(mid_to_entry.alloc(ctx, n_marked));
    if (nw_s2m) {
      k_rmo_marked<<<cdiv(nw_s2m, 256), 256, 0, ctx->stream>>>(rm->s2m_bits.p, nw_s2m, n_ent, before_s.p, mid_to_entry.p);
      BG_LAUNCH_CHECK();
    }
    if (n) {
      k_rmo_entries<<<cdiv(n, 256), 256, 0, ctx->stream>>>(rm->d2m_bits.p, before_d.p, n, mid_to_entry.p, n_marked, rm->mate_loop_ptr.p,
                                                          pair_mode ? 1 : 0, rm->entry_id.p, err.p);
      BG_LAUNCH_CHECK();
    }
    uint32_t h_err = 0;
    unsigned long long h_max = 0;
    BG_HIP(copy_d2h(ctx, &h_err, err.p, 4));
    BG_HIP(copy_d2h(ctx, &h_max, vmax.p, 8));
    if (h_err) { set_error("%s: inconsistent sparse_multi / mate loop table", path); return BG_EIO; }
    if (legacy) max_len = std::max<uint64_t>(max_len, h_max);
    rm->max_read_len = uint32_t(max_len);
  }
  BG_TRY(bitcount_finalize(ctx, rm->s2m_bits, n_ent, &rm->s2m_sub, &rm->s2m_acc));
  BG_TRY(bitcount_finalize(ctx, rm->d2m_bits, n, &rm->d2m_sub, &rm->d2m_acc));
  BG_TRY(readmap_min_len(ctx, rm.get()));
  *out = rm.release();
  return BG_OK;
}

__global__ void k_rm_narrow(const uint64_t* __restrict__ a, uint64_t n, uint32_t* __restrict__ o) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) o[i] = uint32_t(a[i]);
}

// Per read id: seqset entry (sparse_multi::reverse_lookup), read length, mate loop pointer, orientation
// (readmap.h:120-138 accessors as whole arrays).  Any output may be null.
extern "C" int bg_readmap_arrays(bg_ctx* ctx, const bg_readmap* rm, uint64_t* h_entry_id, uint32_t* h_read_len,
                                 uint32_t* h_mate_loop_ptr, uint8_t* h_is_forward) {
  if (!ctx || !rm) { set_error("bg_readmap_arrays: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n = rm->n_reads;
  if (!n) return BG_OK;
  DevBuf<uint32_t> tmp;
  BG_TRY(tmp.alloc(ctx, n));
  if (h_entry_id) BG_HIP(hipMemcpyAsync(h_entry_id, rm->entry_id.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  const uint64_t* srcs[3] = {rm->read_len.p, rm->mate_loop_ptr.p, rm->is_forward.p};
  std::vector<uint32_t> fw(h_is_forward ? n : 0);
  void* dsts[3] = {h_read_len, h_mate_loop_ptr, h_is_forward ? fw.data() : nullptr};
  for (int k = 0; k < 3; ++k) {
    if (!dsts[k]) continue;
    k_rm_narrow<<<cdiv(n, 256), 256, 0, ctx->stream>>>(srcs[k], n, tmp.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipMemcpyAsync(dsts[k], tmp.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_is_forward)
    for (uint64_t i = 0; i < n; ++i) h_is_forward[i] = uint8_t(fw[i]);
  return BG_OK;
}

// =====================================================================================================================
// Sharded readmap of ONE sample (SURVEY §8e + f.1; BASELINE.json configs[3]): the seqset's entries are range-partitioned
// over the ranks (shard q holds the global entries [off_q, off_q+1) with their keys), every rank holds some of the
// sample's pairs.  make_readmap::create_from_reads (make_readmap.cpp:120-392) with every structure owned by one rank:
//   1. a read's entry = seqset::find_existing_unique on the shard its key falls into (rows routed with
//      BG_PART_KEY_BOUNDS; bg_seqset_find_rows answers with GLOBAL entry ids);
//   2. the rows (hi, lo) of make_readmap.h:56-90 go to the owner of their entry, which sorts its rows: the ranks' sorted
//      rows, in rank order, are the global row order of make_readmap.h:187-213, so read id = rows on the ranks before +
//      local position;
//   3. the three claim hops of :283-392 travel as claims (group, loop start's read id, claiming row's read id) to the
//      owner of the group's entry, which hands the k-th claimer of a group (in read-id order of the loop starts: the
//      sequential order the reference claims in) the group's k-th row, answers with that row's read id and issues the
//      next hop's claims.
// The library still never communicates: biograph_amd/sharded.py (sharded_readmap) moves the rows.
struct bg_readmap_shard {
  bg_ctx* ctx = nullptr;
  uint64_t n = 0;
  bg::DevBuf<uint64_t> hi, lo;  // this rank's rows, sorted
  bg::DevBuf<uint64_t> ptr;     // mate loop pointer of every row: a GLOBAL read id (~0 = not set yet)
};

namespace bg {

constexpr int kClaimWords = 4;  // group key, read id of the loop's start, read id of the claiming row, destination << 10

__device__ __forceinline__ uint64_t entry_owner(const uint64_t* __restrict__ off, uint32_t nparts, uint64_t e) {
  uint32_t lo = 0, hi = nparts;  // last q with off[q] <= e (an empty shard shares its offset with the next one)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// entry of every received read row (W key words + length in the low byte of the last word): first entry of the shard
// that has the row as a prefix, as a global id; ~0 for an empty row; err |= 1 when there is none
__global__ __launch_bounds__(128) void k_rm_find_rows(const uint64_t* __restrict__ rows, uint64_t n, int W,
                                                     const uint64_t* __restrict__ ekeys, const uint8_t* __restrict__ esizes,
                                                     uint64_t n_ent, int EW, const uint32_t* __restrict__ dir, int bases,
                                                     uint64_t entry_base, uint64_t* __restrict__ out, uint32_t* __restrict__ err) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* r = rows + i * uint64_t(W + 1);
  const int L = int(r[W] & 0xff);
  if (!L) { out[i] = ~0ull; return; }
  uint64_t key[kMaxWords];
  for (int w = 0; w < EW; ++w) key[w] = w < W ? r[w] : 0;
  const uint32_t p = uint32_t(key[0] >> (64 - 2 * bases));
  uint64_t lo = dir[p], hi = dir[p + 1];
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (key_compare(ekeys + mid * EW, esizes[mid], key, L, EW) < 0) lo = mid + 1; else hi = mid;
  }
  if (lo >= n_ent || key_lcp(ekeys + lo * EW, esizes[lo], key, L, EW) < L) {
    atomicOr(err, 1u);  // "Read ... was not found in seqset" (make_readmap.cpp:147-151)
    out[i] = ~0ull;
    return;
  }
  out[i] = entry_base + lo;
}

__global__ void k_rms_pack_rows(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, const uint32_t* __restrict__ pos,
                                uint64_t m, const uint64_t* __restrict__ off, uint32_t nparts, uint64_t* __restrict__ rows) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const uint64_t h = hi[pos[t]];
  uint64_t* r = rows + t * 3;
  r[0] = h;
  r[1] = lo[pos[t]];
  r[2] = entry_owner(off, nparts, hi_entry(h)) << 10;
}

__global__ void k_rms_split_rows(const uint64_t* __restrict__ rows, uint64_t n, uint64_t* __restrict__ hi, uint64_t* __restrict__ lo) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  hi[i] = rows[i * 3];
  lo[i] = rows[i * 3 + 1];
}

__global__ void k_rms_first_claims(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, const uint32_t* __restrict__ starts,
                                   uint64_t ns, uint64_t gid_base, const uint64_t* __restrict__ off, uint32_t nparts,
                                   uint64_t* __restrict__ claims) {
  const uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  const uint64_t i = starts[s];
  uint64_t* c = claims + s * kClaimWords;
  c[0] = rm_hi(lo[i], 1, hi_len(hi[i]), 0);  // hop 1: the RC rows of (loop_entry, read_len)  (:283-302)
  c[1] = gid_base + i;
  c[2] = gid_base + i;
  c[3] = entry_owner(off, nparts, lo[i]) << 10;
}

__global__ void k_rms_claim_keys(const uint64_t* __restrict__ claims, uint64_t n, int word, uint64_t* __restrict__ keys,
                                 uint32_t* __restrict__ idx) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = claims[i * kClaimWords + word];
  if (idx) idx[i] = uint32_t(i);
}
__global__ void k_rms_gather_claim_word(const uint64_t* __restrict__ claims, const uint32_t* __restrict__ idx, uint64_t n, int word,
                                        uint64_t* __restrict__ out) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = claims[uint64_t(idx[i]) * kClaimWords + word];
}

// whole claims (kClaimWords words each) picked by index: one lane per word, a claim's words land side by side
__global__ void k_rms_gather_claims(const uint64_t* __restrict__ claims, const uint32_t* __restrict__ idx, uint64_t n,
                                    uint64_t* __restrict__ out) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t < n * kClaimWords) out[t] = claims[uint64_t(idx[t / kClaimWords]) * kClaimWords + t % kClaimWords];
}

// Claimers sorted by (group, read id of their loop's start): the k-th of a group takes the group's k-th row.  Answers
// with the row's read id, finishes what ends here and writes the next hop's claim (word 3 = ~0: none).
__global__ void k_rms_apply_claims(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ lo, uint64_t n_rows,
                                   uint64_t gid_base, int hop, const uint64_t* __restrict__ claims, const uint64_t* __restrict__ gs,
                                   const uint32_t* __restrict__ perm, uint64_t n, const uint64_t* __restrict__ off, uint32_t nparts,
                                   uint64_t* __restrict__ ptr, uint64_t* __restrict__ reply, uint64_t* __restrict__ next,
                                   uint8_t* __restrict__ has_next, uint32_t* __restrict__ err) {
  const uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint64_t g = gs[p];
  const uint32_t c = perm[p];
  const uint64_t first = lower_bound_u64(gs, n, g);
  const uint64_t row = lower_bound_u64(hi, n_rows, g) + (p - first);
  has_next[c] = 0;
  if (row >= n_rows || (hi[row] >> 10) != (g >> 10)) {  // entry, type and length of the claimed row (:352-389)
    atomicOr(err, 2u);
    reply[c] = ~0ull;
    return;
  }
  reply[c] = gid_base + row;
  const uint64_t start = claims[uint64_t(c) * kClaimWords + 1];
  uint64_t* nx = next + uint64_t(c) * kClaimWords;
  if (hop == 1) {
    if (lo[row] == kNoLoopEntry) { ptr[row] = start; return; }  // no mate: RC points back to the start
    nx[0] = rm_hi(lo[row], 2, hi_mate_len(hi[row]), 0);
  } else if (hop == 2) {
    nx[0] = rm_hi(lo[row], 3, hi_len(hi[row]), 0);
  } else {
    ptr[row] = start;  // loop back to the beginning (:389)
    return;
  }
  nx[1] = start;
  nx[2] = gid_base + row;
  nx[3] = entry_owner(off, nparts, lo[row]) << 10;
  has_next[c] = 1;
}

__global__ void k_rms_take_replies(const uint64_t* __restrict__ sent, const uint64_t* __restrict__ reply, uint64_t n,
                                   uint64_t gid_base, uint64_t n_rows, uint64_t* __restrict__ ptr, uint32_t* __restrict__ err) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t from = sent[i * kClaimWords + 2];
  if (from < gid_base || from - gid_base >= n_rows || reply[i] == ~0ull) { atomicOr(err, 2u); return; }
  ptr[from - gid_base] = reply[i];
}

__global__ void k_rms_columns(const uint64_t* __restrict__ hi, const uint64_t* __restrict__ ptr, uint64_t n,
                              uint64_t* __restrict__ entry, uint32_t* __restrict__ len, uint32_t* __restrict__ out_ptr,
                              uint8_t* __restrict__ fwd, uint32_t* __restrict__ err) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = hi[i];
  const int t = hi_type(h);
  entry[i] = hi_entry(h);
  len[i] = hi_len(h);
  fwd[i] = (t == 0 || t == 2) ? 1 : 0;
  if (ptr[i] >= (1ull << 32)) atomicOr(err, 2u);  // a row no loop went through
  out_ptr[i] = uint32_t(ptr[i]);
}

static int rms_offsets(bg_ctx* ctx, uint32_t nparts, const uint64_t* h_entry_offsets, DevBuf<uint64_t>* d) {
  BG_TRY(d->alloc(ctx, nparts + 1));
  BG_HIP(hipMemcpyAsync(d->p, h_entry_offsets, (uint64_t(nparts) + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  return BG_OK;
}

static int rms_check(bg_ctx* ctx, DevBuf<uint32_t>& err, const char* who) {
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err & 1) { set_error("%s: a read was not found in the seqset", who); return BG_EINTERNAL; }
  if (h_err & 2) { set_error("%s: mate loop claim failed (inconsistent rows)", who); return BG_EINTERNAL; }
  return BG_OK;
}

}  // namespace bg

extern "C" int bg_seqset_find_rows(bg_ctx* ctx, const bg_seqset* shard, const uint64_t* d_rows, uint64_t n, int W,
                                   uint64_t entry_base, uint64_t* d_entry) {
  if (!ctx || !shard || (n && (!d_rows || !d_entry))) { set_error("bg_seqset_find_rows: null argument"); return BG_EINVAL; }
  BG_TRY(seqset_ensure_keys(ctx, const_cast<bg_seqset*>(shard)));
  if (n && shard->n && !(shard->keys.p && shard->W)) { set_error("bg_seqset_find_rows: the shard has no resident keys"); return BG_EINVAL; }
  if (W < 1 || W > kMaxWords || (shard->n && W > shard->W)) { set_error("bg_seqset_find_rows: rows wider than the shard's keys"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  DevBuf<uint32_t> err, dir;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  int bases = 0;
  const int EW = shard->W ? shard->W : W;
  ctx->tic("readmap_entry_ids");
  if (shard->n) {
    BG_TRY(build_prefix_dir_keys(ctx, shard->keys.p, shard->n, EW, &dir, &bases));
  } else {  // an empty shard answers "not found" for anything but an empty row
    bases = 1;
    BG_TRY(dir.alloc(ctx, 5));
    BG_HIP(hipMemsetAsync(dir.p, 0, 5 * 4, ctx->stream));
  }
  k_rm_find_rows<<<cdiv(n, 128), 128, 0, ctx->stream>>>(d_rows, n, W, shard->keys.p, shard->sizes.p, shard->n, EW, dir.p, bases,
                                                       entry_base, d_entry, err.p);
  BG_LAUNCH_CHECK();
  ctx->toc(n * (uint64_t(W) * 8 + 64));
  return rms_check(ctx, err, "bg_seqset_find_rows");
}

extern "C" int bg_readmap_rows(bg_ctx* ctx, const bg_reads* reads, int is_paired, const uint64_t* d_entry, uint32_t nparts,
                               const uint64_t* h_entry_offsets, uint64_t* d_rows, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !reads || !h_entry_offsets || !n_out || (reads->n && !d_entry)) { set_error("bg_readmap_rows: null argument"); return BG_EINVAL; }
  if (is_paired && (reads->n & 1)) { set_error("bg_readmap_rows: paired input needs an even number of reads"); return BG_EINVAL; }
  if (nparts < 1 || nparts > 126) { set_error("bg_readmap_rows: %u shards outside 1..126", nparts); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n = reads->n, n_rec = is_paired ? n / 2 : n, n_slots = n_rec * (is_paired ? 4 : 2);
  *n_out = 0;
  if (!n_slots) return BG_OK;
  if (n_slots >= (1ull << 32)) { set_error("bg_readmap_rows: more than 2^32 rows on one device"); return BG_EINVAL; }
  DevBuf<uint64_t> hi, lo, off;
  DevBuf<uint8_t> f;
  DevBuf<uint32_t> pos;
  BG_TRY(hi.alloc(ctx, n_slots));
  BG_TRY(lo.alloc(ctx, n_slots));
  BG_TRY(f.alloc(ctx, n_slots));
  BG_TRY(rms_offsets(ctx, nparts, h_entry_offsets, &off));
  k_rm_rows<<<cdiv(n_rec, 256), 256, 0, ctx->stream>>>(reads->words.p, reads->len.p, n_rec, reads->W, is_paired ? 1 : 0, d_entry, hi.p, lo.p);
  BG_LAUNCH_CHECK();
  k_rm_flag_valid<<<cdiv(n_slots, 256), 256, 0, ctx->stream>>>(hi.p, n_slots, f.p);
  BG_LAUNCH_CHECK();
  uint64_t m = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, f.p, n_slots, &pos, &m));
  if (m > cap || (m && !d_rows)) { set_error("bg_readmap_rows: room for %llu rows, %llu needed", (unsigned long long)cap, (unsigned long long)m); return BG_EINVAL; }
  if (m) {
    k_rms_pack_rows<<<cdiv(m, 256), 256, 0, ctx->stream>>>(hi.p, lo.p, pos.p, m, off.p, nparts, d_rows);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = m;
  return BG_OK;
}

extern "C" int bg_readmap_shard_build(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, bg_readmap_shard** out) {
  if (!ctx || !out || (n && !d_rows)) { set_error("bg_readmap_shard_build: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_readmap_shard> sh(new bg_readmap_shard);
  sh->ctx = ctx;
  DevBuf<uint64_t> hi, lo;
  BG_TRY(hi.alloc(ctx, n));
  BG_TRY(lo.alloc(ctx, n));
  if (n) {
    k_rms_split_rows<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows, n, hi.p, lo.p);
    BG_LAUNCH_CHECK();
  }
  uint64_t n_rows = 0;
  BG_TRY(rm_sort_rows(ctx, hi, lo, n, &sh->hi, &sh->lo, &n_rows));
  if (n_rows != n) { set_error("bg_readmap_shard_build: %llu of %llu rows are not rows", (unsigned long long)(n - n_rows), (unsigned long long)n); return BG_EINVAL; }
  sh->n = n;
  BG_TRY(sh->ptr.alloc(ctx, n));
  if (n) BG_HIP(hipMemsetAsync(sh->ptr.p, 0xff, n * 8, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *out = sh.release();
  return BG_OK;
}

extern "C" uint64_t bg_readmap_shard_size(const bg_readmap_shard* s) { return s ? s->n : 0; }
extern "C" void bg_readmap_shard_free(bg_readmap_shard* s) { delete s; }

extern "C" int bg_readmap_shard_first_claims(bg_ctx* ctx, const bg_readmap_shard* s, uint64_t read_id_base, uint32_t nparts,
                                             const uint64_t* h_entry_offsets, uint64_t* d_claims, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !s || !h_entry_offsets || !n_out) { set_error("bg_readmap_shard_first_claims: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  *n_out = 0;
  if (!s->n) return BG_OK;
  DevBuf<uint8_t> f;
  DevBuf<uint32_t> starts;
  DevBuf<uint64_t> off;
  BG_TRY(f.alloc(ctx, s->n));
  BG_TRY(rms_offsets(ctx, nparts, h_entry_offsets, &off));
  k_rm_flag_type<<<cdiv(s->n, 256), 256, 0, ctx->stream>>>(s->hi.p, s->n, 0, f.p);
  BG_LAUNCH_CHECK();
  uint64_t ns = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, f.p, s->n, &starts, &ns));
  if (ns > cap || (ns && !d_claims)) { set_error("bg_readmap_shard_first_claims: room for %llu claims, %llu needed", (unsigned long long)cap, (unsigned long long)ns); return BG_EINVAL; }
  if (ns) {
    k_rms_first_claims<<<cdiv(ns, 256), 256, 0, ctx->stream>>>(s->hi.p, s->lo.p, starts.p, ns, read_id_base, off.p, nparts, d_claims);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = ns;
  return BG_OK;
}

extern "C" int bg_readmap_shard_apply_claims(bg_ctx* ctx, bg_readmap_shard* s, uint64_t read_id_base, int hop, const uint64_t* d_claims,
                                             uint64_t n, uint32_t nparts, const uint64_t* h_entry_offsets, uint64_t* d_reply,
                                             uint64_t* d_next, uint64_t* n_next) {
  if (!ctx || !s || !h_entry_offsets || !n_next || (n && (!d_claims || !d_reply || !d_next))) {
    set_error("bg_readmap_shard_apply_claims: null argument");
    return BG_EINVAL;
  }
  if (hop < 1 || hop > 3) { set_error("bg_readmap_shard_apply_claims: hop %d outside 1..3", hop); return BG_EINVAL; }
  if (n >= (1ull << 32)) { set_error("bg_readmap_shard_apply_claims: more than 2^32 claims"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  *n_next = 0;
  if (!n) return BG_OK;
  DevBuf<uint64_t> ka, kb, off, next;
  DevBuf<uint32_t> va, vb, err, pick;
  DevBuf<uint8_t> has_next;
  BG_TRY(ka.alloc(ctx, n));
  BG_TRY(kb.alloc(ctx, n));
  BG_TRY(va.alloc(ctx, n));
  BG_TRY(vb.alloc(ctx, n));
  BG_TRY(next.alloc(ctx, n * kClaimWords));
  BG_TRY(has_next.alloc(ctx, n));
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  BG_TRY(rms_offsets(ctx, nparts, h_entry_offsets, &off));
  ctx->tic("readmap_mate_loops");
  // by the loop start's read id, then stably by group
  k_rms_claim_keys<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_claims, n, 1, ka.p, va.p);
  BG_LAUNCH_CHECK();
  uint64_t* ks = nullptr;
  uint32_t* vs = nullptr;
  BG_TRY(sort_pairs(ctx, ka, kb, va, vb, n, 33, &ks, &vs));
  uint64_t* k2 = ks == ka.p ? kb.p : ka.p;
  uint32_t* v2 = vs == va.p ? vb.p : va.p;
  k_rms_gather_claim_word<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_claims, vs, n, 0, k2);
  BG_LAUNCH_CHECK();
  bool in_b = false;
  BG_TRY(radix_sort_u64(ctx, k2, ks, vs, v2, n, 0, 64, &in_b));
  const uint64_t* gs = in_b ? ks : k2;
  const uint32_t* perm = in_b ? v2 : vs;
  k_rms_apply_claims<<<cdiv(n, 256), 256, 0, ctx->stream>>>(s->hi.p, s->lo.p, s->n, read_id_base, hop, d_claims, gs, perm, n, off.p,
                                                           nparts, s->ptr.p, d_reply, next.p, has_next.p, err.p);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 96);
  BG_TRY(rms_check(ctx, err, "bg_readmap_shard_apply_claims"));
  uint64_t m = 0;
  BG_TRY(compact_flagged<uint32_t>(ctx, has_next.p, n, &pick, &m));
  if (m) {
    // claims are kClaimWords-word rows: the picked ones, whole, straight to where they go
    k_rms_gather_claims<<<cdiv(m * kClaimWords, 256), 256, 0, ctx->stream>>>(next.p, pick.p, m, d_next);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_next = m;
  return BG_OK;
}

extern "C" int bg_readmap_shard_take_replies(bg_ctx* ctx, bg_readmap_shard* s, uint64_t read_id_base, const uint64_t* d_claims_sent,
                                             const uint64_t* d_reply, uint64_t n) {
  if (!ctx || !s || (n && (!d_claims_sent || !d_reply))) { set_error("bg_readmap_shard_take_replies: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  k_rms_take_replies<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_claims_sent, d_reply, n, read_id_base, s->n, s->ptr.p, err.p);
  BG_LAUNCH_CHECK();
  return rms_check(ctx, err, "bg_readmap_shard_take_replies");
}

extern "C" int bg_readmap_shard_columns(bg_ctx* ctx, const bg_readmap_shard* s, uint64_t* d_entry_id, uint32_t* d_read_len,
                                        uint32_t* d_mate_loop_ptr, uint8_t* d_is_forward) {
  if (!ctx || !s || (s->n && (!d_entry_id || !d_read_len || !d_mate_loop_ptr || !d_is_forward))) {
    set_error("bg_readmap_shard_columns: null argument");
    return BG_EINVAL;
  }
  BG_HIP(hipSetDevice(ctx->device));
  if (!s->n) return BG_OK;
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  k_rms_columns<<<cdiv(s->n, 256), 256, 0, ctx->stream>>>(s->hi.p, s->ptr.p, s->n, d_entry_id, d_read_len, d_mate_loop_ptr, d_is_forward, err.p);
  BG_LAUNCH_CHECK();
  return rms_check(ctx, err, "bg_readmap_shard_columns");
}
