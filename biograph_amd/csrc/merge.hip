// merge.hip — `biograph merge` data path: seqset_flat / make_mergemap / seqset_merger
// (bio_base/make_mergemap.cpp:150-220, seqset_merger.cpp:111-185, driver biograph_merge.cpp:189-290) and
// make_readmap::create_from_fast_migrate (bio_mapred/make_readmap.cpp:455-519).
//
// The reference flattens every input to sorted sequences, k-way merges them (an entry that equals or is
// a prefix of the next one joins its chain; a chain = one merged entry) and rebuilds the arrays by a
// sequential sweep.  Net semantics: merged entries = prefix-free maxima of the union; mergemap of input
// i marks, for every entry e of i, the first merged entry that has e as a prefix.  Here the inputs'
// resident entry keys are concatenated and go through the same sort / prefix-dedup / array kernels as a
// fresh build (the union of pop_front-closed sets is closed), and the maps are one lower_bound per
// input entry.
#include "bg_device_funcs.h"
#include "bg_internal.h"

namespace bg {


__global__ void k_merge_copy_keys(const uint64_t* __restrict__ src, const uint8_t* __restrict__ sizes, uint64_t n,
                                  int Ws, int Wd, uint64_t* __restrict__ dst, uint8_t* __restrict__ dlen) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int w = 0; w < Wd; ++w) dst[i * Wd + w] = w < Ws ? src[i * Ws + w] : 0;
  dlen[i] = sizes[i];
}

// Entry sequences by FM navigation for an input opened from a file (seqset.cpp:609-721): pop_front =
// select on prev_b.  One thread per entry; select = binary search over the rank function.
__global__ void k_merge_keys_fm(SeqsetView v, int W, uint64_t* __restrict__ dst, uint8_t* __restrict__ dlen) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= v.n) return;
  uint64_t cur = i;
  int L = v.sizes[i];
  uint64_t word = 0;
  for (int w = 0; w < W; ++w) dst[i * W + w] = 0;
  for (int j = 0; j < L; ++j) {
    int b = 0;
    while (b < 3 && cur >= v.fixed[b + 1]) ++b;
    word |= uint64_t(b) << (62 - 2 * (j & 31));
    if ((j & 31) == 31 || j + 1 == L) { dst[i * W + (j >> 5)] = word; word = 0; }
    uint64_t rank = cur - v.fixed[b];
    uint64_t lo = 0, hi = v.n;  // smallest t with count(b, t+1) > rank
    while (lo < hi) {
      uint64_t mid = (lo + hi) >> 1;
      if (v.count(b, mid + 1) > rank) hi = mid; else lo = mid + 1;
    }
    cur = lo;
  }
  dlen[i] = uint8_t(L);
}

// merged index of every input entry = first merged entry having it as a prefix
__global__ __launch_bounds__(128) void k_merge_map(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ klen,
                                                   uint64_t n, int W, const uint64_t* __restrict__ ekeys,
                                                   const uint8_t* __restrict__ esizes, uint64_t n_ent,
                                                   const uint32_t* __restrict__ dir, int bases,
                                                   uint64_t* __restrict__ map, uint32_t* __restrict__ err) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* key = keys + i * W;
  int L = klen[i];
  uint32_t p = uint32_t(key[0] >> (64 - 2 * bases));
  uint64_t lo = dir[p], hi = dir[p + 1];
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (key_compare(ekeys + mid * W, esizes[mid], key, L, W) < 0) lo = mid + 1; else hi = mid;
  }
  if (lo >= n_ent || key_lcp(ekeys + lo * W, esizes[lo], key, L, W) < L) { atomicOr(err, 1u); lo = 0; }
  map[i] = lo;
}

__global__ void k_migrate_entries(const uint64_t* __restrict__ old_entry, uint64_t n, const uint64_t* __restrict__ map,
                                  uint64_t n_old_entries, uint64_t* __restrict__ new_entry,
                                  uint8_t* __restrict__ marks, uint32_t* __restrict__ err) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t e = old_entry[i];
  if (e >= n_old_entries) { atomicOr(err, 1u); return; }
  uint64_t t = map[e];
  new_entry[i] = t;
  marks[t] = 1;
}

__global__ __launch_bounds__(256) void k_merge_bytes_to_bits(const uint8_t* __restrict__ marks, uint64_t n,
                                                             unsigned long long* __restrict__ bits) {
  uint64_t w = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (w >= (n + 63) / 64) return;
  uint64_t t = w * 64 + (threadIdx.x & 63);
  uint64_t word = __ballot(t < n && marks[t] != 0);
  if ((threadIdx.x & 63) == 0) bits[w] = word;
}

// rows (W key words + the length) of the entries [lo, lo + cnt) of one input: what bg_merge_sharded sends through the
// records exchange.  One thread per entry; resident keys are copied, an input opened from a file is walked (k_merge_keys_fm).
__global__ void k_stripe_rows_keys(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ sizes, uint64_t lo, uint64_t cnt,
                                   int Ws, int W, uint64_t* __restrict__ rows) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  uint64_t* row = rows + i * uint64_t(W + 1);
  for (int w = 0; w < W; ++w) row[w] = w < Ws ? keys[(lo + i) * Ws + w] : 0;
  row[W] = sizes[lo + i];
}
__global__ void k_stripe_rows_fm(SeqsetView v, uint64_t lo, uint64_t cnt, int W, uint64_t* __restrict__ rows) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  uint64_t* row = rows + i * uint64_t(W + 1);
  uint64_t cur = lo + i;
  const int L = v.sizes[cur];
  uint64_t word = 0;
  for (int w = 0; w < W; ++w) row[w] = 0;
  for (int j = 0; j < L; ++j) {
    int b = 0;
    while (b < 3 && cur >= v.fixed[b + 1]) ++b;
    word |= uint64_t(b) << (62 - 2 * (j & 31));
    if ((j & 31) == 31 || j + 1 == L) { row[j >> 5] = word; word = 0; }
    const uint64_t rank = cur - v.fixed[b];
    uint64_t a = 0, z = v.n;  // smallest t with count(b, t + 1) > rank
    while (a < z) {
      const uint64_t mid = (a + z) >> 1;
      if (v.count(b, mid + 1) > rank) z = mid; else a = mid + 1;
    }
    cur = a;
  }
  row[W] = uint64_t(L);
}

static SeqsetView merge_view(const bg_seqset* ss) {
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

int bitcount_finalize_plain(bg_ctx* ctx, const DevBuf<uint64_t>& bits, uint64_t nbits, DevBuf<uint64_t>* subaccum,
                            DevBuf<uint64_t>* accum);  // readmap.hip

int seqset_stripe_rows(bg_ctx* ctx, const bg_seqset* s, uint64_t lo, uint64_t cnt, int W, uint64_t* d_rows) {
  if (!cnt) return BG_OK;
  if (lo + cnt > s->n) { set_error("seqset_stripe_rows: stripe past the end"); return BG_EINVAL; }
  if (!s->accum[0].p) { set_error("bg_merge_sharded: an input is not finalized"); return BG_EINVAL; }
  BG_TRY(seqset_ensure_keys(ctx, const_cast<bg_seqset*>(s)));
  if (s->keys.p && s->W) k_stripe_rows_keys<<<cdiv(cnt, 256), 256, 0, ctx->stream>>>(s->keys.p, s->sizes.p, lo, cnt, s->W, W, d_rows);
  else k_stripe_rows_fm<<<cdiv(cnt, 64), 64, 0, ctx->stream>>>(merge_view(s), lo, cnt, W, d_rows);
  BG_LAUNCH_CHECK();
  return BG_OK;
}

}  // namespace bg

using namespace bg;

extern "C" int bg_seqset_merge(bg_ctx* ctx, uint32_t n_in, const bg_seqset* const* in, bg_seqset** out,
                               uint64_t* const* d_maps) {
  if (!ctx || !in || !out || !n_in) { set_error("bg_seqset_merge: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  uint64_t total = 0;
  uint32_t max_len = 0;
  for (uint32_t i = 0; i < n_in; ++i) {
    if (!in[i]) { set_error("bg_seqset_merge: null input"); return BG_EINVAL; }
    if (in[i]->n && !in[i]->accum[0].p) { set_error("bg_seqset_merge: input %u not finalized", i); return BG_EINVAL; }
    total += in[i]->n;
    max_len = std::max(max_len, in[i]->max_len);  // seqset_merger.cpp:57-60
  }
  const int W = std::max(1u, (max_len + 31) / 32);
  std::unique_ptr<bg_seqset> ss(new bg_seqset);
  ss->ctx = ctx;
  ss->W = W;
  ss->uuid = make_uuid();
  DevBuf<uint64_t> rec, keep;  // all inputs' keys, W words each; `keep` = a copy for the maps
  DevBuf<uint8_t> rlen, keep_len;
  BG_TRY(rec.alloc(ctx, total * W));
  BG_TRY(rlen.alloc(ctx, total));
  ctx->tic("merge_gather_keys");
  uint64_t pos = 0;
  for (uint32_t i = 0; i < n_in; ++i) {
    const bg_seqset* s = in[i];
    if (!s->n) continue;
    BG_TRY(seqset_ensure_keys(ctx, const_cast<bg_seqset*>(s)));
    if (s->keys.p && s->W) {
      k_merge_copy_keys<<<cdiv(s->n, 256), 256, 0, ctx->stream>>>(s->keys.p, s->sizes.p, s->n, s->W, W, rec.p + pos * W,
                                                                 rlen.p + pos);
    } else {
      k_merge_keys_fm<<<cdiv(s->n, 64), 64, 0, ctx->stream>>>(merge_view(s), W, rec.p + pos * W, rlen.p + pos);
    }
    BG_LAUNCH_CHECK();
    pos += s->n;
  }
  ctx->toc(total * W * 8);
  if (d_maps) {
    BG_TRY(keep.alloc(ctx, total * W));
    BG_TRY(keep_len.alloc(ctx, total));
    if (total) {
      BG_HIP(hipMemcpyAsync(keep.p, rec.p, total * W * 8, hipMemcpyDeviceToDevice, ctx->stream));
      BG_HIP(hipMemcpyAsync(keep_len.p, rlen.p, total, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  uint64_t n_tied = 0;
  BG_TRY(entries_from_records_impl(ctx, rec, rlen, total, W, ss.get(), &n_tied));
  BG_TRY(prev_bits_local_impl(ctx, ss.get()));
  ss->max_len = ss->n ? max_len : 0;
  BG_TRY(seqset_finalize_device(ctx, ss.get()));
  if (d_maps && total) {
    DevBuf<uint32_t> dir, err;
    BG_TRY(err.alloc(ctx, 1));
    BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
    ctx->tic("merge_maps");
    int bases = 0;
    BG_TRY(build_prefix_dir_keys(ctx, ss->keys.p, ss->n, W, &dir, &bases));
    pos = 0;
    for (uint32_t i = 0; i < n_in; ++i) {
      uint64_t n = in[i]->n;
      if (n && d_maps[i]) {
        k_merge_map<<<cdiv(n, 128), 128, 0, ctx->stream>>>(keep.p + pos * W, keep_len.p + pos, n, W, ss->keys.p,
                                                           ss->sizes.p, ss->n, dir.p, bases, d_maps[i], err.p);
        BG_LAUNCH_CHECK();
      }
      pos += n;
    }
    ctx->toc(total * (W * 8 + 64));
    uint32_t h_err = 0;
    BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    if (h_err) { set_error("bg_seqset_merge: an input entry has no extension in the merged seqset"); return BG_EINTERNAL; }
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  *out = ss.release();
  return BG_OK;
}

extern "C" int bg_readmap_migrate(bg_ctx* ctx, const bg_readmap* old_rm, const uint64_t* d_map, const bg_seqset* merged,
                                  bg_readmap** out) {
  if (!merged) { set_error("bg_readmap_migrate: null argument"); return BG_EINVAL; }
  return bg_readmap_migrate_to(ctx, old_rm, d_map, merged->n, merged->uuid.c_str(), out);
}

extern "C" int bg_readmap_migrate_to(bg_ctx* ctx, const bg_readmap* old_rm, const uint64_t* d_map, uint64_t merged_entries,
                                     const char* merged_uuid, bg_readmap** out) {
  if (!ctx || !old_rm || !merged_uuid || !out || (old_rm->n_entries && !d_map)) { set_error("bg_readmap_migrate: null argument"); return BG_EINVAL; }
  if (old_rm->pair_mode) { set_error("bg_readmap_migrate: readmap has no mate loop table; use \"biograph upgrade\" to construct it"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n = old_rm->n_reads, n_ent = merged_entries;
  std::unique_ptr<bg_readmap> rm(new bg_readmap);
  rm->ctx = ctx;
  rm->n_reads = n;
  rm->n_entries = n_ent;
  rm->max_read_len = old_rm->max_read_len;  // create_common(..., old_readmap.max_read_len()) (:459-461)
  rm->min_read_len = old_rm->min_read_len;
  rm->seqset_uuid = merged_uuid;
  BG_TRY(rm->read_len.alloc(ctx, n));
  BG_TRY(rm->mate_loop_ptr.alloc(ctx, n));
  BG_TRY(rm->is_forward.alloc(ctx, n));
  BG_TRY(rm->entry_id.alloc(ctx, n));
  BG_TRY(rm->s2m_bits.alloc(ctx, (n_ent + 63) / 64));
  BG_TRY(rm->d2m_bits.alloc(ctx, (n + 63) / 64));
  DevBuf<uint8_t> marks;
  DevBuf<uint32_t> err;
  BG_TRY(marks.alloc(ctx, n_ent));
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  if (n_ent) BG_HIP(hipMemsetAsync(marks.p, 0, n_ent, ctx->stream));
  if (n) {
    BG_HIP(hipMemcpyAsync(rm->read_len.p, old_rm->read_len.p, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(rm->mate_loop_ptr.p, old_rm->mate_loop_ptr.p, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(rm->is_forward.p, old_rm->is_forward.p, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    // the translation is strictly increasing, so the groups of read ids (dest_to_mid) do not change
    BG_HIP(hipMemcpyAsync(rm->d2m_bits.p, old_rm->d2m_bits.p, (n + 63) / 64 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    k_migrate_entries<<<cdiv(n, 256), 256, 0, ctx->stream>>>(old_rm->entry_id.p, n, d_map, old_rm->n_entries,
                                                             rm->entry_id.p, marks.p, err.p);
    BG_LAUNCH_CHECK();
  }
  if (n_ent) {
    k_merge_bytes_to_bits<<<cdiv((n_ent + 63) / 64, 4), 256, 0, ctx->stream>>>(
        marks.p, n_ent, reinterpret_cast<unsigned long long*>(rm->s2m_bits.p));
    BG_LAUNCH_CHECK();
  }
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_readmap_migrate: the map does not cover the readmap's seqset"); return BG_EINVAL; }
  BG_TRY(bitcount_finalize_plain(ctx, rm->s2m_bits, n_ent, &rm->s2m_sub, &rm->s2m_acc));
  BG_TRY(bitcount_finalize_plain(ctx, rm->d2m_bits, n, &rm->d2m_sub, &rm->d2m_acc));
  *out = rm.release();
  return BG_OK;
}
