// shard_cover.hip — the exact-cover emission of the sharded build without a replicated cover index
// (DESIGN.md §7).
//
// In the single-GPU build every oriented read is a candidate cover for every suffix, so the index spans all
// reads (build_seqset.hip, suffix_records_impl).  Sharded, an index over ALL corrected reads on every rank
// is what stops the build from scaling (620 M reads of a 30x human sample: > 100 GB per rank, replicated).
// Here the oriented reads are range-partitioned by their leading bases — the partition of the suffix
// records themselves (part_repo::partition_for_sequence, part_repo.cpp:32-45) — and a suffix is tested
// where it would be sorted anyway:
//
//   owner of a prefix range   holds the oriented reads that start in the range (its slice of the index; they
//                             are also its j = 0 records)
//   sender                    ships suffix j of an oriented read s as a CANDIDATE row to the owner of its
//                             leading bases: any read that has s[j:] as a prefix starts with the same bases
//   owner                     answers "covered" per candidate (cover_probe against its slice)
//   sender                    first covered j of each read -> tells the owner which candidates to keep
//                             (j below the first cover: exactly the records the single-GPU rule emits)
//
// Candidates go out in rounds of growing j ranges (most reads are covered within a few bases at 30x), the
// short tails nobody looks up (< K bases) in a last plain round.  Per-rank memory is the rank's own reads,
// its slice of the index and one round of candidates — nothing grows with the total number of reads.
#include "bg_device_funcs.h"
#include "bg_internal.h"
#include "compact.h"

namespace bg {

constexpr uint64_t kRowLenMask = 0xff;
constexpr uint64_t kRowPlain = 1ull << 16;  // a row nobody tests (shorter than K): always a record
constexpr int kRowTagShift = 24;

__global__ void k_oriented_rows(const uint64_t* __restrict__ words, const uint16_t* __restrict__ len, uint64_t n, int W_in,
                                int W_out, uint64_t* __restrict__ rows) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= 2 * n) return;
  const uint64_t i = o >> 1;
  const int L = len[i];
  uint64_t tmp[kMaxWords];
  oriented_words(words + i * W_in, L, W_in, (o & 1) != 0, tmp);
  uint64_t* r = rows + o * uint64_t(W_out + 1);
  for (int w = 0; w < W_out; ++w) r[w] = L ? tmp[w] : 0;
  r[W_out] = uint64_t(L);
}

__global__ void k_rows_prefix_hist(const uint64_t* __restrict__ rows, uint64_t n, int row_words, int bases, uint64_t every,
                                   unsigned long long* __restrict__ hist) {
  const uint64_t i = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * every;
  if (i >= n) return;
  const uint64_t* r = rows + i * uint64_t(row_words);
  if ((r[row_words - 1] & kRowLenMask) == 0) return;
  atomicAdd(&hist[r[0] >> (64 - 2 * bases)], 1ull);
}

// rows -> the arrays the index is built over (+ the sort keys: leading 16 bases | oriented id)
__global__ void k_rows_to_oriented(const uint64_t* __restrict__ rows, uint64_t n, int W, uint64_t* __restrict__ owords,
                                   uint8_t* __restrict__ olen, uint64_t* __restrict__ k0) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o >= n) return;
  const uint64_t* r = rows + o * uint64_t(W + 1);
  for (int w = 0; w < W; ++w) owords[o * W + w] = r[w];
  olen[o] = uint8_t(r[W] & kRowLenMask);
  if (k0) k0[o] = (r[0] & ~((1ull << 32) - 1)) | o;
}

__global__ void k_cover_test_rows(CoverIndex ci, const uint64_t* __restrict__ owords, const uint64_t* __restrict__ rows,
                                  uint64_t n, int W, uint8_t* __restrict__ covered) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* r = rows + i * uint64_t(W + 1);
  const uint64_t meta = r[W];
  const int L = int(meta & kRowLenMask);
  bool c = false;
  if (!(meta & kRowPlain) && L >= ci.K) c = cover_probe(ci, owords, r, L, 0, W);
  covered[i] = c ? 1 : 0;
}

__global__ void k_query_init(const uint8_t* __restrict__ olen, uint64_t n, uint8_t* __restrict__ flags) {
  const uint64_t o = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (o < n) flags[o] = olen[o] >= 2 ? 1 : 0;  // a read of one base has no suffix j >= 1
}

// candidate rows of one round: pending read `slot`, suffixes j_lo + jj, jj < span (rows past the last suffix
// that is looked up — fewer than K bases left — carry length 0 and are dropped by the partition)
__global__ void k_query_candidates(const uint64_t* __restrict__ owords, const uint8_t* __restrict__ olen,
                                   const uint32_t* __restrict__ pending, uint64_t n_pending, int W, int K, int j_lo, int span,
                                   uint64_t* __restrict__ rows) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_pending * uint64_t(span)) return;
  const uint64_t slot = t / uint64_t(span);
  const int j = j_lo + int(t - slot * uint64_t(span));
  const uint32_t o = pending[slot];
  const int L = olen[o];
  uint64_t* r = rows + t * uint64_t(W + 1);
  const int rem = L - j;
  if (rem < K || rem <= 0) {
    for (int w = 0; w <= W; ++w) r[w] = 0;
    return;
  }
  const uint64_t* s = owords + uint64_t(o) * W;
  for (int w = 0; w < W; ++w) r[w] = window32(s, W, j + 32 * w);
  r[W] = uint64_t(rem) | (t << kRowTagShift);
}

__global__ void k_query_first_cover(const uint64_t* __restrict__ rows_sent, const uint8_t* __restrict__ covered, uint64_t n,
                                    int W, const uint32_t* __restrict__ pending, int j_lo, int span,
                                    uint32_t* __restrict__ first_cov32) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n || !covered[i]) return;
  const uint64_t t = rows_sent[i * uint64_t(W + 1) + W] >> kRowTagShift;
  const uint64_t slot = t / uint64_t(span);
  const uint32_t j = uint32_t(j_lo) + uint32_t(t - slot * uint64_t(span));
  atomicMin(&first_cov32[slot], j);
}

__global__ void k_query_keep(const uint64_t* __restrict__ rows_sent, uint64_t n, int W, int j_lo, int span,
                             const uint32_t* __restrict__ first_cov32, uint8_t* __restrict__ keep) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t t = rows_sent[i * uint64_t(W + 1) + W] >> kRowTagShift;
  const uint64_t slot = t / uint64_t(span);
  const uint32_t j = uint32_t(j_lo) + uint32_t(t - slot * uint64_t(span));
  keep[i] = j < first_cov32[slot] ? 1 : 0;
}

__global__ void k_query_still_pending(const uint32_t* __restrict__ first_cov32, uint64_t n_pending, uint8_t* __restrict__ flags) {
  const uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s < n_pending) flags[s] = first_cov32[s] == 0xffffffffu ? 1 : 0;
}

__global__ void k_gather_pending(const uint32_t* __restrict__ pending, const uint32_t* __restrict__ keep_slots, uint64_t m,
                                 uint32_t* __restrict__ out) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < m) out[i] = pending[keep_slots[i]];
}

// the suffixes nobody looks up (fewer than K bases) of the reads no suffix of which was covered
__global__ void k_query_plain(const uint64_t* __restrict__ owords, const uint8_t* __restrict__ olen,
                              const uint32_t* __restrict__ pending, uint64_t n_pending, int W, int K,
                              uint64_t* __restrict__ rows) {
  const int span = K - 1;
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_pending * uint64_t(span)) return;
  const uint64_t slot = t / uint64_t(span);
  const uint32_t o = pending[slot];
  const int L = olen[o];
  const int first = L - K + 1 > 1 ? L - K + 1 : 1;  // first suffix with fewer than K bases
  const int j = first + int(t - slot * uint64_t(span));
  uint64_t* r = rows + t * uint64_t(W + 1);
  if (j >= L) {
    for (int w = 0; w <= W; ++w) r[w] = 0;
    return;
  }
  const uint64_t* s = owords + uint64_t(o) * W;
  for (int w = 0; w < W; ++w) r[w] = window32(s, W, j + 32 * w);
  r[W] = uint64_t(L - j) | kRowPlain;
}

}  // namespace bg

using namespace bg;

extern "C" int bg_oriented_rows(bg_ctx* ctx, const bg_reads* reads, int W_out, uint64_t* d_rows) {
  if (!ctx || !reads || (reads->n && !d_rows)) { set_error("bg_oriented_rows: null argument"); return BG_EINVAL; }
  if (W_out < 1 || W_out > kMaxWords || uint32_t(W_out) * 32 < reads->max_len) {
    set_error("bg_oriented_rows: %d words do not hold reads of up to %u bases", W_out, reads->max_len);
    return BG_EINVAL;
  }
  BG_HIP(hipSetDevice(ctx->device));
  if (reads->n) {
    k_oriented_rows<<<cdiv(2 * reads->n, 256), 256, 0, ctx->stream>>>(reads->words.p, reads->len.p, reads->n, reads->W, W_out, d_rows);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_rows_prefix_hist(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int row_words, int bases, uint64_t every,
                                   uint64_t* d_hist) {
  if (!ctx || !d_hist || (n && !d_rows)) { set_error("bg_rows_prefix_hist: null argument"); return BG_EINVAL; }
  if (bases < 1 || bases > 12 || every < 1 || row_words < 2) { set_error("bg_rows_prefix_hist: bad geometry"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  BG_HIP(hipMemsetAsync(d_hist, 0, (size_t(1) << (2 * bases)) * 8, ctx->stream));
  if (n) {
    k_rows_prefix_hist<<<cdiv(cdiv(n, every), 256), 256, 0, ctx->stream>>>(d_rows, n, row_words, bases, every,
                                                                         reinterpret_cast<unsigned long long*>(d_hist));
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_cover_build(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int W, int cover_k, bg_cover** out) {
  if (!ctx || !out || (n && !d_rows)) { set_error("bg_cover_build: null argument"); return BG_EINVAL; }
  if (W < 1 || W > kMaxWords || cover_k < 1 || cover_k > 31) { set_error("bg_cover_build: bad geometry"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_cover> c(new bg_cover);
  c->ctx = ctx;
  c->n = n;
  c->W = W;
  c->K = cover_k;
  BG_TRY(c->owords.alloc(ctx, n * W));
  BG_TRY(c->olen.alloc(ctx, n));
  DevBuf<uint64_t> k0;
  BG_TRY(k0.alloc(ctx, n));
  ctx->tic("suffix_oriented");
  if (n) {
    k_rows_to_oriented<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows, n, W, c->owords.p, c->olen.p, k0.p);
    BG_LAUNCH_CHECK();
  }
  ctx->toc(n * (uint64_t(W) * 16 + 17));
  BG_TRY(cover_index_build(ctx, c->owords.p, c->olen.p, n, W, cover_k, &k0, &c->cs));
  *out = c.release();
  return BG_OK;
}

extern "C" void bg_cover_free(bg_cover* c) { delete c; }

extern "C" int bg_cover_test(bg_ctx* ctx, const bg_cover* c, const uint64_t* d_rows, uint64_t n, uint8_t* d_covered) {
  if (!ctx || !c || (n && (!d_rows || !d_covered))) { set_error("bg_cover_test: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (n) {
    ctx->tic("suffix_cover_count");
    k_cover_test_rows<<<cdiv(n, 256), 256, 0, ctx->stream>>>(c->cs.ci, c->owords.p, d_rows, n, c->W, d_covered);
    BG_LAUNCH_CHECK();
    ctx->toc(n * (uint64_t(c->W) * 16 + 9));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_cover_query_begin(bg_ctx* ctx, const uint64_t* d_rows, uint64_t n, int W, int cover_k,
                                    bg_cover_query** out) {
  if (!ctx || !out || (n && !d_rows)) { set_error("bg_cover_query_begin: null argument"); return BG_EINVAL; }
  if (W < 1 || W > kMaxWords || cover_k < 1 || cover_k > 31) { set_error("bg_cover_query_begin: bad geometry"); return BG_EINVAL; }
  if (n >= (1ull << 32)) { set_error("bg_cover_query_begin: too many oriented reads for one rank"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_cover_query> q(new bg_cover_query);
  q->ctx = ctx;
  q->n = n;
  q->W = W;
  q->K = cover_k;
  BG_TRY(q->owords.alloc(ctx, n * W));
  BG_TRY(q->olen.alloc(ctx, n));
  DevBuf<uint8_t> flags;
  BG_TRY(flags.alloc(ctx, n));
  if (n) {
    k_rows_to_oriented<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows, n, W, q->owords.p, q->olen.p, nullptr);
    BG_LAUNCH_CHECK();
    k_query_init<<<cdiv(n, 256), 256, 0, ctx->stream>>>(q->olen.p, n, flags.p);
    BG_LAUNCH_CHECK();
  }
  BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, n, &q->pending, &q->n_pending));
  *out = q.release();
  return BG_OK;
}

extern "C" void bg_cover_query_free(bg_cover_query* q) { delete q; }
extern "C" uint64_t bg_cover_query_pending(const bg_cover_query* q) { return q ? q->n_pending : 0; }

extern "C" int bg_cover_query_candidates(bg_ctx* ctx, bg_cover_query* q, int j_lo, int j_hi, uint64_t* d_rows, uint64_t cap,
                                         uint64_t* n_out) {
  if (!ctx || !q || !n_out) { set_error("bg_cover_query_candidates: null argument"); return BG_EINVAL; }
  if (j_lo < 1 || j_hi <= j_lo || j_hi > 256) { set_error("bg_cover_query_candidates: bad suffix range [%d, %d)", j_lo, j_hi); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const int span = j_hi - j_lo;
  const uint64_t total = q->n_pending * uint64_t(span);
  if (total > cap || (total && !d_rows)) { set_error("bg_cover_query_candidates: room for %llu rows, %llu needed",
                                                     (unsigned long long)cap, (unsigned long long)total); return BG_EINVAL; }
  if (total >= (1ull << (64 - kRowTagShift))) { set_error("bg_cover_query_candidates: round too large"); return BG_EINVAL; }
  q->j_lo = j_lo;
  q->span = span;
  if (total) {
    ctx->tic("suffix_emit");
    k_query_candidates<<<cdiv(total, 256), 256, 0, ctx->stream>>>(q->owords.p, q->olen.p, q->pending.p, q->n_pending, q->W, q->K,
                                                                 j_lo, span, d_rows);
    BG_LAUNCH_CHECK();
    ctx->toc(total * uint64_t(q->W + 1) * 8);
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = total;
  return BG_OK;
}

// d_rows_sent: the candidate rows of the round in the order their answers d_covered came back in (the order
// they were sent in); d_keep_out[i] = 1: the owner keeps row i as a record.  Reads with a covered suffix
// leave the pending list.
extern "C" int bg_cover_query_answers(bg_ctx* ctx, bg_cover_query* q, const uint64_t* d_rows_sent, const uint8_t* d_covered,
                                      uint64_t n, uint8_t* d_keep_out) {
  if (!ctx || !q || (n && (!d_rows_sent || !d_covered || !d_keep_out))) { set_error("bg_cover_query_answers: null argument"); return BG_EINVAL; }
  if (!q->span) { set_error("bg_cover_query_answers: no round in flight"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  DevBuf<uint32_t> fc32, keep_slots, np;
  DevBuf<uint8_t> flags;
  BG_TRY(fc32.alloc(ctx, q->n_pending));
  BG_TRY(flags.alloc(ctx, q->n_pending));
  if (q->n_pending) BG_HIP(hipMemsetAsync(fc32.p, 0xff, q->n_pending * 4, ctx->stream));
  if (n) {
    k_query_first_cover<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows_sent, d_covered, n, q->W, q->pending.p, q->j_lo, q->span, fc32.p);
    BG_LAUNCH_CHECK();
    k_query_keep<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_rows_sent, n, q->W, q->j_lo, q->span, fc32.p, d_keep_out);
    BG_LAUNCH_CHECK();
  }
  uint64_t m = 0;
  if (q->n_pending) {
    k_query_still_pending<<<cdiv(q->n_pending, 256), 256, 0, ctx->stream>>>(fc32.p, q->n_pending, flags.p);
    BG_LAUNCH_CHECK();
  }
  BG_TRY(compact_flagged<uint32_t>(ctx, flags.p, q->n_pending, &keep_slots, &m));
  BG_TRY(np.alloc(ctx, m));
  if (m) {
    k_gather_pending<<<cdiv(m, 256), 256, 0, ctx->stream>>>(q->pending.p, keep_slots.p, m, np.p);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  q->pending = std::move(np);
  q->n_pending = m;
  q->span = 0;
  return BG_OK;
}

extern "C" int bg_cover_query_plain(bg_ctx* ctx, bg_cover_query* q, uint64_t* d_rows, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !q || !n_out) { set_error("bg_cover_query_plain: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t total = q->K > 1 ? q->n_pending * uint64_t(q->K - 1) : 0;
  if (total > cap || (total && !d_rows)) { set_error("bg_cover_query_plain: room for %llu rows, %llu needed",
                                                     (unsigned long long)cap, (unsigned long long)total); return BG_EINVAL; }
  if (total) {
    k_query_plain<<<cdiv(total, 256), 256, 0, ctx->stream>>>(q->owords.p, q->olen.p, q->pending.p, q->n_pending, q->W, q->K, d_rows);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = total;
  return BG_OK;
}
