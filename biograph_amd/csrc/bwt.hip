// bwt.hip — `biograph reference`, MAKE_BWT step (modules/bio_mapred/make_bwt.cpp:33-141; file format
// bio_base/bwt_file.h:9-45, bwt_file.cpp:16-44) on the GPU.
//
// The reference sorts every suffix of every extent of the flat reference (plus the empty suffix at each extent's end)
// with std::sort and a comparator that walks the two suffixes (shorter-is-less on a prefix, ties by (extent, offset)).
// Here: every extent gets a terminator that is smaller than all bases and ordered by extent number, which makes all
// suffixes distinct and gives exactly that order, and the suffix array comes from prefix doubling -- rank pairs
// (rank[p], rank[p + h]) sorted with the library's radix sort, h doubling until all ranks are distinct
// (log2 of the longest repeat rounds; the keys only hold as many bits as the ranks need).
// From the suffix array: bit vectors "the base before this suffix is b" (b = A, C, G, T) and "century" marks (offset
// divisible by cent_mod) with the bitcount layout the seqset uses (bits / subaccum / accum back to back,
// io/bitcount.cpp:16-31), the table of the marked suffixes' flat positions, C(a) and the msgpack header.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "bg_internal.h"

namespace bg {

int bitcount_finalize_plain(bg_ctx* ctx, const DevBuf<uint64_t>& bits, uint64_t nbits, DevBuf<uint64_t>* subaccum,
                            DevBuf<uint64_t>* accum);  // readmap.hip

// position p of the concatenation (extent e occupies [start[e], start[e] + size[e]], the last one = its terminator)
__device__ __forceinline__ uint32_t bwt_extent_of(const uint64_t* __restrict__ start, uint32_t n_ext, uint64_t p) {
  uint32_t lo = 0, hi = n_ext;  // last e with start[e] <= p
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (start[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void k_bwt_init(const uint8_t* __restrict__ codes, const uint64_t* __restrict__ start,
                           const uint64_t* __restrict__ flat, const uint64_t* __restrict__ size, uint32_t n_ext, uint64_t n,
                           uint32_t* __restrict__ rank) {
  const uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t e = bwt_extent_of(start, n_ext, p);
  const uint64_t j = p - start[e];
  rank[p] = j == size[e] ? e : n_ext + uint32_t(codes[flat[e] + j] & 3);
}

__global__ void k_bwt_keys(const uint32_t* __restrict__ rank, uint64_t n, uint64_t h, int rbits, uint64_t* __restrict__ key,
                           uint32_t* __restrict__ val) {
  const uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= n) return;
  // (a suffix that ends within its first h symbols has met its unique terminator: its rank is final, the second half moot)
  const uint64_t second = p + h < n ? uint64_t(rank[p + h]) : 0;
  key[p] = (uint64_t(rank[p]) << rbits) | second;
  val[p] = uint32_t(p);
}

__global__ void k_bwt_heads(const uint64_t* __restrict__ key, uint64_t n, uint32_t* __restrict__ head) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

// rank of the suffix at sorted position i = (number of group heads up to and including i) - 1
__global__ void k_bwt_rerank(const uint32_t* __restrict__ excl, const uint32_t* __restrict__ head, const uint32_t* __restrict__ val,
                             uint64_t n, uint32_t* __restrict__ rank) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rank[val[i]] = excl[i] + head[i] - 1;
}

__global__ void k_bwt_sa_from_rank(const uint32_t* __restrict__ rank, uint64_t n, uint32_t* __restrict__ sa) {
  const uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p < n) sa[rank[p]] = uint32_t(p);
}

// per sorted suffix: the previous base (0..3, 4 = none), the century flag and flat position, the suffix's first base
__global__ void k_bwt_rows(const uint32_t* __restrict__ sa, const uint8_t* __restrict__ codes,
                           const uint64_t* __restrict__ start, const uint64_t* __restrict__ flat,
                           const uint64_t* __restrict__ size, uint32_t n_ext, uint64_t n, uint64_t cent_mod,
                           uint8_t* __restrict__ prev, uint8_t* __restrict__ cent, uint8_t* __restrict__ first) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t p = sa[i];
  const uint32_t e = bwt_extent_of(start, n_ext, p);
  const uint64_t j = p - start[e];
  prev[i] = j ? uint8_t(codes[flat[e] + j - 1] & 3) : uint8_t(4);
  const bool has = j != size[e];
  cent[i] = (has && j % cent_mod == 0) ? 1 : 0;
  first[i] = has ? uint8_t(codes[flat[e] + j] & 3) : uint8_t(4);
}

__global__ __launch_bounds__(256) void k_bwt_bits(const uint8_t* __restrict__ src, uint64_t n, int want,
                                                  unsigned long long* __restrict__ bits) {
  const uint64_t w = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (w >= (n + 63) / 64) return;
  const uint64_t t = w * 64 + (threadIdx.x & 63);
  const uint64_t word = __ballot(t < n && int(src[t]) == want);
  if ((threadIdx.x & 63) == 0) bits[w] = word;
}

__global__ void k_bwt_century_table(const uint32_t* __restrict__ pos, uint64_t m, const uint32_t* __restrict__ sa,
                                    const uint64_t* __restrict__ start, const uint64_t* __restrict__ flat, uint32_t n_ext,
                                    uint32_t* __restrict__ table) {
  const uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const uint64_t p = sa[pos[k]];
  const uint32_t e = bwt_extent_of(start, n_ext, p);
  table[k] = uint32_t(flat[e] + (p - start[e]));
}

// first sorted position whose suffix starts with base b (n when there is none): the suffixes are sorted, so `first` is
// non-decreasing over the non-empty ones after the n_ext empty suffixes at the front
__global__ void k_bwt_first_of(const uint8_t* __restrict__ first, uint64_t n, unsigned long long* __restrict__ out) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = first[i];
  if (b < 4 && (i == 0 || first[i - 1] != b)) atomicMin(&out[b], (unsigned long long)i);
}

}  // namespace bg

#include "compact.h"

using namespace bg;

// d_codes: the flat reference, one base code (0..3) per byte; extents as in reference.ref's index (flat offset, size).
// Writes the .bwt file of bwt_file (magic, offset of the header, five bitcounts, the century table, the msgpack header).
extern "C" int bg_bwt_build(bg_ctx* ctx, const uint8_t* d_codes, uint32_t n_extents, const uint64_t* h_ext_flat,
                            const uint64_t* h_ext_size, uint64_t cent_mod, const char* path, bg_bwt_stats* stats) {
  if (!ctx || !path || (n_extents && (!h_ext_flat || !h_ext_size || !d_codes))) { set_error("bg_bwt_build: null argument"); return BG_EINVAL; }
  if (!cent_mod) { set_error("bg_bwt_build: cent_mod must be positive"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  uint64_t n = 0, cent_count = 0;
  std::vector<uint64_t> h_start(n_extents);
  for (uint32_t e = 0; e < n_extents; ++e) {
    h_start[e] = n;
    n += h_ext_size[e] + 1;
    cent_count += (h_ext_size[e] + cent_mod - 1) / cent_mod;
    if (h_ext_flat[e] + h_ext_size[e] >= (1ull << 32)) { set_error("bg_bwt_build: flat positions must fit 32 bits (century table)"); return BG_EINVAL; }
  }
  if (n >= (1ull << 32) - 1) { set_error("bg_bwt_build: %llu suffixes; the format's ranks are 32-bit", (unsigned long long)n); return BG_EINVAL; }
  DevBuf<uint64_t> d_start, d_flat, d_size;
  BG_TRY(d_start.alloc(ctx, n_extents));
  BG_TRY(d_flat.alloc(ctx, n_extents));
  BG_TRY(d_size.alloc(ctx, n_extents));
  if (n_extents) {
    BG_HIP(hipMemcpyAsync(d_start.p, h_start.data(), n_extents * 8, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(d_flat.p, h_ext_flat, n_extents * 8, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(d_size.p, h_ext_size, n_extents * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  // ---- suffix array by prefix doubling
  DevBuf<uint32_t> rank, sa;
  BG_TRY(rank.alloc(ctx, n));
  uint32_t rounds = 0;
  if (n) {
    DevBuf<uint64_t> ka, kb;
    DevBuf<uint32_t> va, vb, head, excl, total;
    BG_TRY(ka.alloc(ctx, n));
    BG_TRY(kb.alloc(ctx, n));
    BG_TRY(va.alloc(ctx, n));
    BG_TRY(vb.alloc(ctx, n));
    BG_TRY(head.alloc(ctx, n));
    BG_TRY(excl.alloc(ctx, n));
    BG_TRY(total.alloc(ctx, 1));
    ctx->tic("bwt_suffix_sort");
    k_bwt_init<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_codes, d_start.p, d_flat.p, d_size.p, n_extents, n, rank.p);
    BG_LAUNCH_CHECK();
    int rbits = 1;
    while ((1ull << rbits) < n + uint64_t(n_extents) + 5) ++rbits;  // ranks < n, initial ranks < n_ext + 4
    for (uint64_t h = 1;; h <<= 1) {
      ++rounds;
      k_bwt_keys<<<cdiv(n, 256), 256, 0, ctx->stream>>>(rank.p, n, h, rbits, ka.p, va.p);
      BG_LAUNCH_CHECK();
      bool in_b = false;
      BG_TRY(radix_sort_u64(ctx, ka.p, kb.p, va.p, vb.p, n, 0, 2 * rbits, &in_b));
      const uint64_t* ks = in_b ? kb.p : ka.p;
      const uint32_t* vs = in_b ? vb.p : va.p;
      k_bwt_heads<<<cdiv(n, 256), 256, 0, ctx->stream>>>(ks, n, head.p);
      BG_LAUNCH_CHECK();
      BG_TRY(exclusive_scan_u32(ctx, head.p, excl.p, n, total.p));
      k_bwt_rerank<<<cdiv(n, 256), 256, 0, ctx->stream>>>(excl.p, head.p, vs, n, rank.p);
      BG_LAUNCH_CHECK();
      uint32_t groups = 0;
      BG_HIP(hipMemcpyAsync(&groups, total.p, 4, hipMemcpyDeviceToHost, ctx->stream));
      BG_HIP(hipStreamSynchronize(ctx->stream));
      if (uint64_t(groups) == n) break;
      if (h >= n) { set_error("bg_bwt_build: suffix ranks did not separate (internal error)"); return BG_EINTERNAL; }
    }
    ctx->toc(uint64_t(rounds) * n * 12 * 2 * 8);
  }
  BG_TRY(sa.alloc(ctx, n));
  if (n) {
    k_bwt_sa_from_rank<<<cdiv(n, 256), 256, 0, ctx->stream>>>(rank.p, n, sa.p);
    BG_LAUNCH_CHECK();
  }
  rank.reset();
  // ---- rows -> bit vectors, century table, C(a)
  const uint64_t nwords = (n + 63) / 64, ngroups = (n + 511) / 512, naccum = (n + 512) / 512;
  const uint64_t bits_size = (nwords + ngroups + naccum) * 8;  // bitcount::compute_size
  DevBuf<uint8_t> prev, cent, first;
  BG_TRY(prev.alloc(ctx, n));
  BG_TRY(cent.alloc(ctx, n));
  BG_TRY(first.alloc(ctx, n));
  DevBuf<unsigned long long> first_of;
  BG_TRY(first_of.alloc(ctx, 4));
  unsigned long long h_first_of[4] = {n, n, n, n};
  BG_HIP(hipMemcpyAsync(first_of.p, h_first_of, 32, hipMemcpyHostToDevice, ctx->stream));
  if (n) {
    ctx->tic("bwt_rows");
    k_bwt_rows<<<cdiv(n, 256), 256, 0, ctx->stream>>>(sa.p, d_codes, d_start.p, d_flat.p, d_size.p, n_extents, n, cent_mod, prev.p,
                                                       cent.p, first.p);
    BG_LAUNCH_CHECK();
    k_bwt_first_of<<<cdiv(n, 256), 256, 0, ctx->stream>>>(first.p, n, first_of.p);
    BG_LAUNCH_CHECK();
    ctx->toc(n * 16);
  }
  BG_HIP(hipMemcpyAsync(h_first_of, first_of.p, 32, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  std::string file(16 + 5 * bits_size + 4 * cent_count, '\0');
  const uint64_t magic = 0x57426C6172697053ull;  // "SpiralBW"
  const uint64_t file_size = file.size();
  memcpy(&file[0], &magic, 8);
  memcpy(&file[8], &file_size, 8);
  for (int v = 0; v < 5; ++v) {
    DevBuf<uint64_t> bits, sub, acc;
    BG_TRY(bits.alloc(ctx, nwords ? nwords : 1));
    if (nwords) {
      k_bwt_bits<<<cdiv(nwords, 4), 256, 0, ctx->stream>>>(v < 4 ? prev.p : cent.p, n, v < 4 ? v : 1,
                                                         reinterpret_cast<unsigned long long*>(bits.p));
      BG_LAUNCH_CHECK();
    }
    BG_TRY(bitcount_finalize_plain(ctx, bits, n, &sub, &acc));
    char* dst = &file[16 + size_t(v) * bits_size];
    if (nwords) BG_HIP(bg::copy_d2h(ctx, dst, bits.p, nwords * 8));
    if (ngroups) BG_HIP(bg::copy_d2h(ctx, dst + nwords * 8, sub.p, ngroups * 8));
    if (naccum) BG_HIP(bg::copy_d2h(ctx, dst + (nwords + ngroups) * 8, acc.p, naccum * 8));
  }
  uint64_t n_cent = 0;
  {
    DevBuf<uint32_t> pos, table;
    BG_TRY(compact_flagged<uint32_t>(ctx, cent.p, n, &pos, &n_cent));
    if (n_cent != cent_count) { set_error("bg_bwt_build: %llu century marks, expected %llu", (unsigned long long)n_cent, (unsigned long long)cent_count); return BG_EINTERNAL; }
    BG_TRY(table.alloc(ctx, n_cent ? n_cent : 1));
    if (n_cent) {
      k_bwt_century_table<<<cdiv(n_cent, 256), 256, 0, ctx->stream>>>(pos.p, n_cent, sa.p, d_start.p, d_flat.p, n_extents, table.p);
      BG_LAUNCH_CHECK();
      BG_HIP(bg::copy_d2h(ctx, &file[16 + 5 * bits_size], table.p, n_cent * 4));
    }
  }
  // ---- msgpack header {1: version {1: major, 2: minor, 3: patch, 4: "", 5: ""}, 2: ca_table} (bwt_file.h:9-18).  The
  // reference never sets `version` (the golden file holds whatever the stack held); this writes zeros.  ca_table = the
  // first position of every base that occurs, in order, then the number of entries and of century marks
  // (make_bwt.cpp:97-115: a base that does not occur gets no entry -- the reference then fails its own CHECK).
  std::string mp;
  auto put_uint = [&mp](uint64_t v) {
    if (v < 128) mp.push_back(char(v));
    else if (v < 256) { mp.push_back(char(0xcc)); mp.push_back(char(v)); }
    else if (v < 65536) { mp.push_back(char(0xcd)); mp.push_back(char(v >> 8)); mp.push_back(char(v)); }
    else if (v < (1ull << 32)) { mp.push_back(char(0xce)); for (int i = 3; i >= 0; --i) mp.push_back(char(v >> (8 * i))); }
    else { mp.push_back(char(0xcf)); for (int i = 7; i >= 0; --i) mp.push_back(char(v >> (8 * i))); }
  };
  std::vector<uint64_t> ca;
  for (int b = 0; b < 4; ++b)
    if (h_first_of[b] < n) ca.push_back(h_first_of[b]);
  if (ca.size() != 4) { set_error("bg_bwt_build: the reference lacks one of the four bases"); return BG_EINVAL; }
  ca.push_back(n);
  ca.push_back(n_cent);
  mp.push_back(char(0x82));
  put_uint(1);
  mp.push_back(char(0x85));
  for (int k = 1; k <= 3; ++k) { put_uint(uint64_t(k)); put_uint(0); }
  put_uint(4); mp.push_back(char(0xa0));
  put_uint(5); mp.push_back(char(0xa0));
  put_uint(2);
  mp.push_back(char(0x90 | ca.size()));
  for (uint64_t v : ca) put_uint(v);
  file += mp;
  FILE* f = fopen(path, "wb");
  if (!f) { set_error("%s: %s", path, strerror(errno)); return BG_EIO; }
  const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
  if (fclose(f) != 0 || !ok) { set_error("%s: write error", path); return BG_EIO; }
  if (stats) { stats->n_entries = n; stats->n_century = n_cent; stats->doubling_rounds = rounds; }
  return BG_OK;
}
