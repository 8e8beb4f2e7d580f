// walk.hip — batched overlap-walk primitives over a seqset resident in HBM (row G1 of SURVEY §8a).
// One lane per query: push_front (bio_base/seqset.cpp:319-338), push_front_drop (:350-429) and
// find (:146-155).  Pure integer, latency-bound random access: 2 rank queries (bits + subaccum +
// accum words) and <= 2 entry-size bytes per step, plus shared[] / min-index probes when dropping.
#include "bg_device_funcs.h"
#include "bg_internal.h"

namespace bg {

static SeqsetView make_view(const bg_seqset* ss) {
  SeqsetView v;
  v.n = ss->n;
  v.sizes = ss->sizes.p;
  v.shared = ss->shared.p;
  for (int b = 0; b < 4; ++b) {
    v.bits[b] = ss->bits[b].p;
    v.subaccum[b] = ss->subaccum[b].p;
    v.accum[b] = ss->accum[b].p;
  }
  for (int b = 0; b < 5; ++b) v.fixed[b] = ss->fixed[b];
  v.min1 = ss->min1.p;
  v.min2 = ss->min2.p;
  return v;
}

__global__ __launch_bounds__(256) void k_push_front(SeqsetView v, uint64_t n, const uint64_t* __restrict__ begin,
                                                    const uint64_t* __restrict__ end,
                                                    const uint32_t* __restrict__ size,
                                                    const uint8_t* __restrict__ base, uint64_t* __restrict__ ob,
                                                    uint64_t* __restrict__ oe, uint32_t* __restrict__ os) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t b0 = begin[i], e0 = end[i];
  uint64_t rb = 0, re = 0;
  uint32_t rs = 0;
  if (b0 < e0 && e0 <= v.n) v.push_front(b0, e0, size[i], base[i] & 3, &rb, &re, &rs);
  ob[i] = rb; oe[i] = re; os[i] = rs;
}

__global__ __launch_bounds__(256) void k_push_front_drop(SeqsetView v, uint64_t n,
                                                         const uint64_t* __restrict__ begin,
                                                         const uint64_t* __restrict__ end,
                                                         const uint32_t* __restrict__ size,
                                                         const uint8_t* __restrict__ base,
                                                         const uint32_t* __restrict__ min_ctx,
                                                         uint64_t* __restrict__ ob, uint64_t* __restrict__ oe,
                                                         uint32_t* __restrict__ os) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t b0 = begin[i], e0 = end[i];
  uint64_t rb = 0, re = 0;
  uint32_t rs = 0;
  if (b0 < e0 && e0 <= v.n) v.push_front_drop(b0, e0, size[i], base[i] & 3, min_ctx[i], &rb, &re, &rs);
  ob[i] = rb; oe[i] = re; os[i] = rs;
}

__global__ __launch_bounds__(256) void k_find(SeqsetView v, uint64_t n, const uint8_t* __restrict__ seq,
                                              const uint64_t* __restrict__ offs, uint64_t* __restrict__ ob,
                                              uint64_t* __restrict__ oe) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t s = offs[i], e = offs[i + 1];
  uint64_t rb = 0, re = v.n;
  uint32_t rs = 0;
  for (uint64_t p = e; p > s; --p) {  // seqset.cpp:149-153: push_front from the last base
    if (rb >= re) break;
    uint32_t c = seq[p - 1];
    uint32_t code = (c >> 1) & 3;
    code ^= code >> 1;
    v.push_front(rb, re, rs, int(code), &rb, &re, &rs);
  }
  ob[i] = rb; oe[i] = re;
}

}  // namespace bg

using namespace bg;

static int check_ss(bg_ctx* ctx, const bg_seqset* ss, const char* who) {
  if (!ctx || !ss) { set_error("%s: null argument", who); return BG_EINVAL; }
  if (ss->n && !ss->accum[0].p) { set_error("%s: seqset not finalized", who); return BG_EINVAL; }
  return BG_OK;
}

extern "C" int bg_push_front_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n, const uint64_t* d_begin,
                                   const uint64_t* d_end, const uint32_t* d_size, const uint8_t* d_base,
                                   uint64_t* d_obegin, uint64_t* d_oend, uint32_t* d_osize) {
  BG_TRY(check_ss(ctx, ss, "bg_push_front_batch"));
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  ctx->tic("walk_push_front");
  k_push_front<<<cdiv(n, 256), 256, 0, ctx->stream>>>(make_view(ss), n, d_begin, d_end, d_size, d_base, d_obegin,
                                                      d_oend, d_osize);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 52);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_push_front_drop_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n, const uint64_t* d_begin,
                                        const uint64_t* d_end, const uint32_t* d_size, const uint8_t* d_base,
                                        const uint32_t* d_min_ctx, uint64_t* d_obegin, uint64_t* d_oend,
                                        uint32_t* d_osize) {
  BG_TRY(check_ss(ctx, ss, "bg_push_front_drop_batch"));
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  ctx->tic("walk_push_front_drop");
  k_push_front_drop<<<cdiv(n, 256), 256, 0, ctx->stream>>>(make_view(ss), n, d_begin, d_end, d_size, d_base,
                                                           d_min_ctx, d_obegin, d_oend, d_osize);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 52);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_find_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n, const uint8_t* d_seq,
                             const uint64_t* d_offs, uint64_t* d_obegin, uint64_t* d_oend) {
  BG_TRY(check_ss(ctx, ss, "bg_find_batch"));
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  ctx->tic("walk_find");
  k_find<<<cdiv(n, 256), 256, 0, ctx->stream>>>(make_view(ss), n, d_seq, d_offs, d_obegin, d_oend);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 52);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}
