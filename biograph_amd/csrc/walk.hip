// walk.hip — batched overlap-walk primitives over a seqset resident in HBM (row G1 of SURVEY §8a).
// One lane per query: push_front (bio_base/seqset.cpp:319-338), push_front_drop (:350-429) and
// find (:146-155).  Pure integer, latency-bound random access: 2 rank queries (bits + subaccum +
// accum words) and <= 2 entry-size bytes per step, plus shared[] / min-index probes when dropping.
#include <string.h>

#include <vector>

#include "bg_device_funcs.h"
#include "bg_internal.h"

namespace bg {

// bg_walk_counters: two device counters every walk kernel of this process adds to while they are switched on
static unsigned long long* g_walk_counters = nullptr;

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
  v.walk_counters = g_walk_counters;
  static const bool no_blocks = [] { const char* e = getenv("BG_WALK_RANK_BLOCKS"); return e && e[0] == '0'; }();
  for (int b = 0; b < 4; ++b) v.rank64[b] = no_blocks ? nullptr : ss->rank64[b].p;
  static const bool no_rank4 = [] { const char* e = getenv("BG_WALK_RANK4"); return e && e[0] == '0'; }();
  v.rank4 = (no_blocks || no_rank4) ? nullptr : ss->rank4.p;
  v.rank4_super = ss->rank4_super.p;
  // (read at every call: the tests compare the two ways inside one process)
  const char* e_wblk = getenv("BG_WALK_BLOCKS");
  if (!no_blocks && !(e_wblk && e_wblk[0] == '0')) {
    seqset_ensure_walk_blocks(ss);
    if (ss->wblk_state.load(std::memory_order_acquire) == 1) { v.wblk = ss->wblk.p; v.wblk_super = ss->wblk_super.p; }
  }
  static const bool no_select = [] { const char* e = getenv("BG_WALK_SELECT"); return e && e[0] == '0'; }();
  for (int b = 0; b < 4; ++b) v.sel256[b] = (no_blocks || no_select) ? nullptr : ss->sel256[b].p;
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

// seqset_range::pop_front (bio_base/seqset.cpp:609-628): the first base b is the `fixed` bucket the range begins in;
// the entry that push_front(b) maps onto `begin` is the (begin - fixed[b])-th set bit of prev_b (select = binary search
// over the counts); then the range widens over the neighbours that share the shorter context.
__device__ __forceinline__ void pop_front_range(const SeqsetView& v, uint64_t b0, uint64_t e0, uint32_t s0, uint64_t* ob, uint64_t* oe,
                                                uint32_t* os) {
  if (!(b0 < e0 && e0 <= v.n) || s0 == 0) { *ob = 0; *oe = 0; *os = 0; return; }  // (the reference throws)
  int b = 0;
  while (b < 3 && b0 >= v.fixed[b + 1]) ++b;
  const uint64_t rank = b0 - v.fixed[b];
  uint64_t lo = 0, hi = v.n;  // smallest t with count(b, t + 1) > rank = the entry that carries the rank-th set bit
  if (SeqsetView::uniform_nonnull(v.sel256[0]) && SeqsetView::uniform_nonnull(v.rank64[0])) {  // (four together)
    // between two samples: the last rank block that starts at or before the bit, then the bit inside it
    const uint64_t j = rank >> 8;
    uint64_t g_lo = v.sel256[b][j] / 384, g_hi = v.sel256[b][j + 1] / 384;
    const uint64_t last_block = v.n / 384;
    if (g_hi > last_block) g_hi = last_block;
    while (g_lo < g_hi) {  // largest g with before(g) <= rank
      const uint64_t mid = (g_lo + g_hi + 1) >> 1;
      if (v.rank64[b][mid * 8] <= rank) g_lo = mid; else g_hi = mid - 1;
    }
    const uint64_t* blk = v.rank64[b] + g_lo * 8;
    unsigned need = unsigned(rank - blk[0]);
    int w = 0;
    unsigned before_w = 0;
    for (; w < 5; ++w) {
      const unsigned upto = unsigned((blk[1] >> (9 * w)) & 0x1ff);  // set bits of the block before word w + 1
      if (upto > need) break;
      before_w = upto;
    }
    need -= before_w;
    uint64_t x = blk[2 + w];
    unsigned pos = 0;
    for (unsigned width = 32; width; width >>= 1) {  // the need-th set bit of x
      const uint64_t low = x & ((1ull << width) - 1);
      const unsigned c = unsigned(__popcll(low));
      if (need >= c) { need -= c; x >>= width; pos += width; } else x = low;
    }
    lo = g_lo * 384 + uint64_t(w) * 64 + pos;
    hi = lo;
  }
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (v.count(b, mid + 1) > rank) hi = mid; else lo = mid + 1;
  }
  const uint32_t ctx_len = s0 - 1;
  uint64_t nb = lo, ne = lo + 1;
  nb = v.next_backward_lt(nb, ctx_len);  // (the entries around `lo` that share the shorter context, eight a load)
  ne = v.next_forward_lt(ne, ctx_len);
  *ob = nb;
  *oe = ne;
  *os = ctx_len;
}

__global__ __launch_bounds__(256) void k_pop_front(SeqsetView v, uint64_t n, const uint64_t* __restrict__ begin,
                                                   const uint64_t* __restrict__ end, const uint32_t* __restrict__ size,
                                                   uint64_t* __restrict__ ob, uint64_t* __restrict__ oe,
                                                   uint32_t* __restrict__ os) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pop_front_range(v, begin[i], end[i], size[i], &ob[i], &oe[i], &os[i]);
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

extern "C" int bg_pop_front_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n, const uint64_t* d_begin,
                                  const uint64_t* d_end, const uint32_t* d_size, uint64_t* d_obegin, uint64_t* d_oend,
                                  uint32_t* d_osize) {
  BG_TRY(check_ss(ctx, ss, "bg_pop_front_batch"));
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  ctx->tic("walk_pop_front");
  k_pop_front<<<cdiv(n, 256), 256, 0, ctx->stream>>>(make_view(ss), n, d_begin, d_end, d_size, d_obegin, d_oend, d_osize);
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

// =================================================================================================
// Frontier kernels (rows G2, G3, G7): one wave per frontier / assembly, lane = (pushed base, range).

namespace bg {

// BG_WALK_PACK=0: branch trace, propagate and push-to-pair trace with a wave an assembly whatever its starting set, as before
// round 6 (the tests compare the two ways)
static bool walk_pack() {
  const char* e = getenv("BG_WALK_PACK");
  return !(e && e[0] == '0');
}

constexpr int kFrontierMax = 16;  // ranges per frontier ("#rs typically 1-10")

struct Tri {
  uint64_t b, e;
  uint32_t s;
};

__device__ __forceinline__ bool tri_precedes(const Tri& x, const Tri& y) {  // seqset.h:211-218
  if (x.b != y.b) return x.b < y.b;
  if (x.e != y.e) return x.e > y.e;
  return x.s < y.s;
}

// Lanes 0..15 hold candidate ranges (cv = valid).  Removes exact duplicates (keeping the lowest lane,
// OR-ing `flag`), optionally removes ranges that contain another one (seqset_set_dedup_prefixes,
// variants/assemble.cpp:1473-1498) and returns, for surviving lanes, their rank in set order.
__device__ __forceinline__ void frontier_dedup(Tri& c, bool& cv, bool& flag, bool prefixes, int lane, int* rank,
                                               int* count) {
  {  // the common frontier holds a single range: nothing to compare
    bool in = cv && lane < kFrontierMax;
    uint64_t vm = __ballot(in);
    if (__popcll(vm) <= 1) {
      cv = in;
      *rank = 0;
      *count = __popcll(vm);
      return;
    }
  }
  bool dup = false, f = flag;
  for (int j = 0; j < kFrontierMax; ++j) {
    uint64_t jb = __shfl(c.b, j, 64), je = __shfl(c.e, j, 64);
    uint32_t js = __shfl(c.s, j, 64);
    bool jv = __shfl(int(cv), j, 64) != 0, jf = __shfl(int(flag), j, 64) != 0;
    bool same = cv && jv && jb == c.b && je == c.e && js == c.s;
    if (same && j < lane) dup = true;
    if (same && jf) f = true;
  }
  cv = cv && !dup && lane < kFrontierMax;
  flag = f;
  if (prefixes) {
    bool discard = false;
    for (int j = 0; j < kFrontierMax; ++j) {
      uint64_t jb = __shfl(c.b, j, 64), je = __shfl(c.e, j, 64);
      uint32_t js = __shfl(c.s, j, 64);
      bool jv = __shfl(int(cv), j, 64) != 0;
      if (!cv || !jv || j == lane) continue;
      bool contained = jb >= c.b && je <= c.e;
      bool strictly = jb > c.b || je < c.e;
      if (contained && (strictly || js > c.s)) discard = true;
    }
    // every lane is judged against the pre-discard set: containment is transitive, so this equals the
    // fixpoint of the reference's repeat-until-stable loop
    cv = cv && !discard;
  }
  int r = 0, n = 0;
  for (int j = 0; j < kFrontierMax; ++j) {
    Tri t{__shfl(c.b, j, 64), __shfl(c.e, j, 64), __shfl(c.s, j, 64)};
    bool jv = __shfl(int(cv), j, 64) != 0;
    if (jv) {
      ++n;
      if (j != lane && tri_precedes(t, c)) ++r;
    }
  }
  *rank = r;
  *count = n;
}

// Moves the surviving ranges to lane == rank, so that the frontier sits in set order in lanes 0..count-1.
__device__ __forceinline__ void frontier_reseat(Tri& c, bool& cv, int rank, int lane) {
  uint64_t vm = __ballot(cv);
  if (__popcll(vm) <= 1) {
    int src = vm ? __ffsll((long long)vm) - 1 : 0;
    Tri t{__shfl(c.b, src, 64), __shfl(c.e, src, 64), __shfl(c.s, src, 64)};
    c = t;
    cv = vm != 0 && lane == 0;
    return;
  }
  Tri nxt{0, 0, 0};
  bool nxv = false;
  for (int j = 0; j < kFrontierMax; ++j) {
    Tri t{__shfl(c.b, j, 64), __shfl(c.e, j, 64), __shfl(c.s, j, 64)};
    bool jv = __shfl(int(cv), j, 64) != 0;
    int jr = __shfl(rank, j, 64);
    if (jv && jr == lane) { nxt = t; nxv = true; }
  }
  c = nxt;
  cv = nxv;
}

// push_to_pair_discover::trace_one_base (graph_discover/push_to_pair.cpp:159-240) minus the per-trace
// state (m_seen_ranges, readmap) which stays with the host driver.
__global__ __launch_bounds__(256) void k_trace_one_base(SeqsetView v, uint64_t nf, const uint32_t* __restrict__ foff,
                                                        const uint64_t* __restrict__ fbegin,
                                                        const uint64_t* __restrict__ fend,
                                                        const uint32_t* __restrict__ fsize,
                                                        const uint32_t* __restrict__ min_ctx,
                                                        uint8_t* __restrict__ best_base, uint8_t* __restrict__ n_cand,
                                                        uint64_t* __restrict__ cb, uint64_t* __restrict__ ce,
                                                        uint32_t* __restrict__ cs, uint8_t* __restrict__ cdrop,
                                                        uint8_t* __restrict__ n_next, uint64_t* __restrict__ nb,
                                                        uint64_t* __restrict__ ne, uint32_t* __restrict__ ns,
                                                        uint32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  uint64_t f = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (f >= nf) return;  // wave-uniform
  uint32_t lo = foff[f], hi = foff[f + 1];
  if (hi - lo > uint32_t(kFrontierMax)) {
    if (lane == 0) { atomicOr(err, 1u); best_base[f] = 255; n_cand[f] = 0; n_next[f] = 0; }
    return;
  }
  const int r = lane & 15, b = lane >> 4;
  bool rv = uint32_t(r) < hi - lo;
  uint64_t rb = rv ? fbegin[lo + r] : 0, re = rv ? fend[lo + r] : 0;
  uint32_t rs = rv ? fsize[lo + r] : 0;
  rv = rv && rb < re && re <= v.n;
  uint64_t ob = 0, oe = 0;
  uint32_t os = 0;
  if (rv) v.push_front_drop(rb, re, rs, b, min_ctx[f], &ob, &oe, &os);
  bool valid = oe > ob;
  bool drop = valid && os != rs + 1;
  uint32_t m = valid ? os : 0;
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) {
    uint32_t t = __shfl_xor(m, d, 64);
    m = t > m ? t : m;
  }
  uint32_t best_len = 0;
  int best = -1;
  for (int bb = 0; bb < 4; ++bb) {  // strict > : first base wins ties (:196)
    uint32_t l = __shfl(m, bb * 16, 64);
    if (l > best_len) { best_len = l; best = bb; }
  }
  if (best < 0) {
    if (lane == 0) { best_base[f] = 255; n_cand[f] = 0; n_next[f] = 0; }
    return;
  }
  int src = best * 16 + (lane & 15);
  Tri c{__shfl(ob, src, 64), __shfl(oe, src, 64), __shfl(os, src, 64)};
  bool cv = (__shfl(int(valid), src, 64) != 0) && lane < kFrontierMax;
  bool cd = __shfl(int(drop), src, 64) != 0;
  int rank, count;
  frontier_dedup(c, cv, cd, false, lane, &rank, &count);
  if (cv) {
    uint64_t o = f * kFrontierMax + rank;
    cb[o] = c.b; ce[o] = c.e; cs[o] = c.s; cdrop[o] = cd ? 1 : 0;
  }
  if (lane == 0) { best_base[f] = uint8_t(best); n_cand[f] = uint8_t(count); }
  frontier_dedup(c, cv, cd, true, lane, &rank, &count);
  if (cv) {
    uint64_t o = f * kFrontierMax + rank;
    nb[o] = c.b; ne[o] = c.e; ns[o] = c.s;
  }
  if (lane == 0) n_next[f] = uint8_t(count);
}

// branch_discover::on_trace (graph_discover/branch.cpp:20-80): walk the assembly's bases from frontier
// `cur`; report the first alternative base with any valid push_front_drop(rc_b, min_overlap) at each
// offset; advance cur along the reference base.
__global__ __launch_bounds__(256) void k_branch_trace(SeqsetView v, uint64_t na, const uint64_t* __restrict__ soff,
                                                      const uint8_t* __restrict__ seq,
                                                      const uint32_t* __restrict__ foff,
                                                      const uint64_t* __restrict__ fbegin,
                                                      const uint64_t* __restrict__ fend,
                                                      const uint32_t* __restrict__ fsize, uint32_t min_overlap,
                                                      uint32_t ev_cap, uint32_t* __restrict__ ev_off,
                                                      uint8_t* __restrict__ ev_base, uint32_t* __restrict__ n_ev,
                                                      uint8_t* __restrict__ n_out, uint64_t* __restrict__ outb,
                                                      uint64_t* __restrict__ oute, uint32_t* __restrict__ outs,
                                                      uint32_t* __restrict__ err, int pack) {
  const int lane = threadIdx.x & 63;
  // A wave takes 16 consecutive assemblies.  A frontier never grows along a trace (the next one is the reference base's
  // pushes of this one's ranges, less duplicates and containers), and nearly every trace starts from ONE range: then four
  // lanes -- one a base -- are an assembly and the wave walks all 16 side by side.  The kernel is a chain of dependent
  // loads (three round trips a step), so what counts is how many chains the chip holds: one assembly a wave were 6 000 of
  // 64 443 at a time, ten rounds of 1 256 steps; sixteen a wave are all of them at once (57.6 -> 10.0 ms on the bench's chr20).
  // A group with a wider frontier in it takes the assemblies one after the other, a wave each, as before.
  const uint64_t a0 = (uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16;
  if (a0 >= na) return;
  {
    const int sub = lane >> 2, p = lane & 3;  // p = pushed (rc) base
    const uint64_t a = a0 + uint64_t(sub);
    const bool act = a < na;
    const uint32_t lo = act ? foff[a] : 0, hi = act ? foff[a + 1] : 0;
    if (pack && __all(hi - lo <= 1u)) {  // (pack = 0, BG_WALK_PACK=0: every group the old way -- comparisons)
      Tri cur{0, 0, 0};
      bool cv = hi > lo;
      if (cv) { cur.b = fbegin[lo]; cur.e = fend[lo]; cur.s = fsize[lo]; }
      cv = cv && cur.b < cur.e && cur.e <= v.n;
      const uint64_t s0 = act ? soff[a] : 0, s1 = act ? soff[a + 1] : 0;
      uint32_t nev = 0;
      for (uint64_t step = 0;; ++step) {
        const bool live = s0 + step < s1;
        if (!__any(live)) break;
        const int rc_ref = live ? 3 - int(seq[s0 + step] & 3) : 0;
        uint64_t ob = 0, oe = 0;
        uint32_t os = 0;
        if (live && cv) v.push_front_drop(cur.b, cur.e, cur.s, p, p == rc_ref ? 0u : min_overlap, &ob, &oe, &os);
        const bool valid = oe > ob;
        const uint32_t nib = uint32_t(__ballot(valid) >> (sub * 4)) & 0xfu;  // which of the four bases extend
        if (live) {
          for (int b = 0; b < 4; ++b) {  // :41-68 alternatives in base order
            const int rc_b = 3 - b;
            if (rc_b == rc_ref) continue;
            if ((nib >> rc_b) & 1u) {
              if (p == 0 && nev < ev_cap) {
                ev_off[a * ev_cap + nev] = uint32_t(step);
                ev_base[a * ev_cap + nev] = uint8_t(b);
              }
              ++nev;
              break;
            }
          }
        }
        const int src = sub * 4 + rc_ref;
        const uint64_t nb = __shfl(ob, src, 64), ne = __shfl(oe, src, 64);
        const uint32_t ns = __shfl(os, src, 64);
        if (live) { cur.b = nb; cur.e = ne; cur.s = ns; cv = ne > nb; }
      }
      if (act && p == 0) {
        if (cv) { outb[a * kFrontierMax] = cur.b; oute[a * kFrontierMax] = cur.e; outs[a * kFrontierMax] = cur.s; }
        n_ev[a] = nev;
        n_out[a] = cv ? 1 : 0;
      }
      return;
    }
  }
  for (uint64_t a = a0; a < a0 + 16 && a < na; ++a) {
  uint32_t lo = foff[a], hi = foff[a + 1];
  if (hi - lo > uint32_t(kFrontierMax)) {
    if (lane == 0) { atomicOr(err, 1u); n_ev[a] = 0; n_out[a] = 0; }
    continue;
  }
  Tri cur{0, 0, 0};
  bool cv = uint32_t(lane) < hi - lo;
  if (cv) { cur.b = fbegin[lo + lane]; cur.e = fend[lo + lane]; cur.s = fsize[lo + lane]; }
  cv = cv && cur.b < cur.e && cur.e <= v.n;
  const int r = lane & 15, p = lane >> 4;  // p = pushed (rc) base
  uint32_t nev = 0;
  for (uint64_t off = soff[a]; off < soff[a + 1]; ++off) {
    int rc_ref = 3 - int(seq[off] & 3);
    uint64_t rb = __shfl(cur.b, r, 64), re = __shfl(cur.e, r, 64);
    uint32_t rs = __shfl(cur.s, r, 64);
    bool rv = __shfl(int(cv), r, 64) != 0;
    uint64_t ob = 0, oe = 0;
    uint32_t os = 0;
    if (rv) v.push_front_drop(rb, re, rs, p, p == rc_ref ? 0u : min_overlap, &ob, &oe, &os);
    bool valid = oe > ob;
    uint64_t ball = __ballot(valid);
    for (int b = 0; b < 4; ++b) {  // :41-68 alternatives in base order
      int rc_b = 3 - b;
      if (rc_b == rc_ref) continue;
      if ((ball >> (rc_b * 16)) & 0xffffull) {
        if (lane == 0 && nev < ev_cap) {
          ev_off[a * ev_cap + nev] = uint32_t(off - soff[a]);
          ev_base[a * ev_cap + nev] = uint8_t(b);
        }
        ++nev;
        break;
      }
    }
    int src = rc_ref * 16 + (lane & 15);
    Tri c{__shfl(ob, src, 64), __shfl(oe, src, 64), __shfl(os, src, 64)};
    bool nv = (__shfl(int(valid), src, 64) != 0) && lane < kFrontierMax;
    bool dummy = false;
    int rank, count;
    frontier_dedup(c, nv, dummy, true, lane, &rank, &count);
    frontier_reseat(c, nv, rank, lane);  // the frontier stays in set order
    cur = c;
    cv = nv;
  }
  if (cv) { outb[a * kFrontierMax + lane] = cur.b; oute[a * kFrontierMax + lane] = cur.e; outs[a * kFrontierMax + lane] = cur.s; }
  uint64_t live = __ballot(cv);
  if (lane == 0) { n_ev[a] = nev; n_out[a] = uint8_t(__popcll(live)); }
  }
}

// ref_map::build (variants/ref_map.cpp:72-185): one lane per chain.  A chain pushes the complement of
// `len` reference bases (walking +1 or, for the reverse strand, -1 and complemented) starting from the
// empty context; after `pre` warm-up bases every position whose range is exactly one full entry
// bumps that entry.  Chains are finer than the reference's chunks (same 256-base warm-up): the walk
// state after >= max_read_len pushes does not depend on where the chain began.
// (Measured and put away, round 6, on top of the walk blocks: the next base's rank words fetched with the size a push checks --
// a push that drops nothing in one round trip instead of two -- 15.7 -> 16.1 ms; the chain as a state machine whose every pass
// is one round of loads whatever its lanes are at, so that no lane waits for another's drop loop: 16.2-17.1 ms against 14.9.  The
// kernel is not short of round trips: 148 M lines of walk blocks and 120 M atomics on 64-bit cells are what it moves -- without
// the atomics (a build that skips them) 11.1 ms of the 15-16.)
__global__ __launch_bounds__(64) void k_ref_map_chains(SeqsetView v, uint64_t nchains,
                                                      const uint64_t* __restrict__ cstart,
                                                      const uint32_t* __restrict__ clen,
                                                      const uint32_t* __restrict__ cpre,
                                                      const int8_t* __restrict__ cdir,
                                                      const uint8_t* __restrict__ ref, unsigned long long* __restrict__ cells) {
  uint64_t c = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= nchains) return;
  uint64_t pos = cstart[c];
  int dir = cdir[c];
  uint32_t len = clen[c], pre = cpre[c];
  uint64_t b = 0, e = v.n;
  uint32_t s = 0;
  for (uint32_t i = 0; i < len; ++i) {
    int base = ref[pos] & 3;
    if (dir < 0) base = 3 - base;   // reverse-complement slice
    pos += dir;
    v.push_front_drop(b, e, s, 3 - base, 0, &b, &e, &s);
    if (e <= b) { b = 0; e = v.n; s = 0; continue; }  // cannot happen on a closed seqset (CHECK(r.valid()))
    if (i < pre) continue;
    if (b + 1 != e || s != v.size_at(b)) continue;
    // ONE atomic per marked position (a device-scope atomic is a 64-byte request at the memory side whatever it carries;
    // a count and a flag word were two): the forward slices count in the low half of the cell, the reverse-complement ones in
    // the high half, and k_ref_map_pack turns the halves into count and flags
    atomicAdd(&cells[b], dir < 0 ? (1ull << 32) : 1ull);
  }
}

// tracer::advance_read_ahead_to (variants/tracer.cpp:346-400) for one extent of a scaffold: the range after pushing
// the complement of every base in turn, from the empty context, reduced to what the tracer keeps -- the entry id when
// the range is exactly one full entry (seqset_range::is_seqset_entry, seqset.h:298-300), ~0 otherwise.  Chain c covers
// positions [c * chain, ...) and warms up on the `warm` bases before them: a chain's context is always a prefix of the
// sequential walk's and grows by one base per step until they are equal, so after max_read_len steps they are.
// dup_first: the first base of the extent is pushed twice (tracer::skip_ahead_to pushes the base it lands on and
// advance_read_ahead_to pushes it again, tracer.cpp:327-345).
__global__ __launch_bounds__(64) void k_ref_walk_entries(SeqsetView v, const uint8_t* __restrict__ codes, uint64_t n,
                                                        uint32_t chain, uint32_t warm, int dup_first,
                                                        uint64_t* __restrict__ out) {
  const uint64_t c = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t p0 = c * chain;
  if (p0 >= n) return;
  const uint64_t p1 = p0 + chain < n ? p0 + chain : n;
  uint64_t pos = p0 >= warm ? p0 - warm : 0;
  uint64_t b = 0, e = v.n;
  uint32_t s = 0;
  if (pos == 0 && dup_first) {
    v.push_front_drop(b, e, s, 3 - (codes[0] & 3), 0, &b, &e, &s);
    if (e <= b) { b = 0; e = v.n; s = 0; }
  }
  for (; pos < p1; ++pos) {
    v.push_front_drop(b, e, s, 3 - (codes[pos] & 3), 0, &b, &e, &s);
    if (e <= b) { b = 0; e = v.n; s = 0; if (pos >= p0) out[pos] = ~0ull; continue; }
    if (pos >= p0) out[pos] = (b + 1 == e && s == v.size_at(b)) ? b : ~0ull;
  }
}

// path_group::add_base (variants/path_group.cpp:113-127) for whole sequences: every head of a path group is a chain of
// push_front_drop(complement, 0) along its group's bases, and heads only ever merge (equal ranges stay equal), so the
// chains are independent: one lane per chain, the range after every base is written out and the host
// (csrc/coverage.cpp) replays the merging and the distant-object bookkeeping.  Ranges are 3 words (begin, end, size).
__global__ __launch_bounds__(64) void k_walk_chains(SeqsetView v, uint32_t n, const bg_walk_chain* __restrict__ chains,
                                                   const uint8_t* __restrict__ codes, uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bg_walk_chain c = chains[i];
  uint64_t b = c.begin, e = c.end;
  uint32_t s = c.size;
  const uint8_t* cd = codes + c.code_offset;
  uint64_t* o = out + c.out_offset;
  for (uint32_t t = 0; t < c.len; ++t) {
    if (e > b) v.push_front_drop(b, e, s, 3 - (cd[t] & 3), 0, &b, &e, &s);
    o[0] = b;
    o[1] = e;
    o[2] = s;
    o += c.out_stride;
  }
}

// The base-by-base loop of tracer::expand_next_path (variants/tracer.cpp:545-620) for as long as it needs nothing but
// the seqset: the four push_front_drop(complement, min_overlap + pushed_since_read) of :553-554 with the loop check of
// :568-577, and, while exactly one of them is valid and the path has not reached a read, the step onto it.  One
// lane per path; the host (csrc/tracer.cpp) replays the costs of the steps and does everything that touches the
// readmap and the reference positions.
// Four lanes per path, one per base: the four pushes of a step are one memory round trip deep instead of four (the kernel
// is a chain of dependent look-ups: a wave of 16 paths steps as fast as its slowest path's slowest push).
__global__ __launch_bounds__(64) void k_tracer_extend(SeqsetView v, const uint64_t* __restrict__ has_reads, uint64_t n,
                                                     const bg_trace_query* __restrict__ q, uint32_t step_cap,
                                                     bg_trace_reply* __restrict__ r, uint32_t* __restrict__ steps) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t i = t >> 2;
  if (i >= n) return;
  const int lane = threadIdx.x & 63, base = lane & 3, quad0 = lane & ~3;
  const bg_trace_query qq = q[i];
  uint64_t b = qq.begin, e = qq.end;
  uint32_t s = qq.size, min_ctx = qq.min_ctx;
  bool pushed_any = (qq.flags & 1u) != 0;
  const uint32_t cap = qq.max_steps < step_cap ? qq.max_steps : step_cap;
  auto is_read = [&](uint64_t rb, uint64_t re, uint32_t rs) {
    return rb + 1 == re && rs == v.size_at(rb) && ((has_reads[rb >> 6] >> (rb & 63)) & 1);
  };
  uint32_t n_steps = 0, reason, valid;
  uint64_t pb, pe;
  uint32_t ps;
  for (;;) {
    v.push_front_drop(b, e, s, 3 - base, min_ctx, &pb, &pe, &ps);
    bool ok = pe > pb;
    if (ok && pb + 1 == pe && ps == v.size_at(pb) && pb == qq.loop_check) { pb = 0; pe = 0; ok = false; }  // a loop
    valid = uint32_t((__ballot(ok) >> quad0) & 15ull);
    if (__popc(valid) != 1) { reason = BG_TRACE_AMBIGUOUS; break; }
    if (pushed_any && is_read(b, e, s)) { reason = BG_TRACE_AT_READ; break; }
    if (n_steps >= cap) { reason = BG_TRACE_STEP_CAP; break; }
    const int next = __ffs(int(valid)) - 1;
    b = __shfl(pb, quad0 + next, 64);
    e = __shfl(pe, quad0 + next, 64);
    s = __shfl(ps, quad0 + next, 64);
    if (base == 0) steps[i * step_cap + n_steps] = uint32_t(next) | (s << 8);
    ++n_steps;
    ++min_ctx;  // pushed_since_read grows with every base that does not land on a read
    pushed_any = true;
    if (is_read(b, e, s)) {
      reason = BG_TRACE_STEPPED_ON_READ;
      // (the pushes of the step that was just taken: what the one-lane version leaves in the reply at this exit)
      break;
    }
  }
  bg_trace_reply* out = r + i;
  out->pb[base] = pb;
  out->pe[base] = pe;
  out->ps[base] = ps;
  if (base == 0) {
    out->begin = b; out->end = e; out->size = s; out->n_steps = n_steps; out->reason = reason; out->valid_mask = valid;
  }
}

// ---- the never-branching traces, whole (bg_internal.h TraceSeed / TraceSimpleOut).  Four lanes per trace as in
// k_tracer_extend (one per pushed base); lane 0 of the quad keeps the search's books.
struct SimpleOpts {
  int64_t base_cost, decrease_overlap_cost, increase_max_between_pair_cost, max_cost, traverse_ref_cost, pairs_used_cost,
      rejoin_local_cost, size_change_cost;
  uint64_t max_ambiguous_bases, max_search_steps, max_bases_between_pairs, max_pair_distance, max_pairs_per_read;
  uint32_t min_overlap;
  int trace_ambiguous_ref, forward_pairs_face_inward;
};
struct EntryPos { uint64_t entry; int32_t off; int32_t pad; };   // std::pair<uint64_t, int32_t>
struct MatePos { uint32_t read; int32_t off; };                  // std::pair<uint32_t, int32_t>

__device__ __forceinline__ bool sim_has_rc_mate_in_range(const TraceSimpleTables& tb, const TraceSeed& sd, uint32_t read_id, int64_t start,
                                                         int64_t limit) {  // tracer.cpp:466-475 through the window
  const int64_t lo = start > int64_t(sd.win_lo) ? start : int64_t(sd.win_lo);
  const int64_t hi = limit < int64_t(sd.win_hi) + 1 ? limit : int64_t(sd.win_hi) + 1;
  if (lo >= hi) return false;
  const MatePos* v = static_cast<const MatePos*>(tb.mate_positions);
  const int32_t lo32 = int32_t(lo);
  uint64_t a = 0, z = tb.n_mate_positions;
  while (a < z) {  // lower_bound of (read_id, lo)
    const uint64_t mid = (a + z) >> 1;
    const MatePos m = v[mid];
    if (m.read < read_id || (m.read == read_id && m.off < lo32)) a = mid + 1; else z = mid;
  }
  if (a >= tb.n_mate_positions) return false;
  const MatePos m = v[a];
  return m.read == read_id && int64_t(m.off) < hi;
}

__global__ __launch_bounds__(64) void k_trace_simple(SeqsetView v, const uint64_t* __restrict__ has_reads, TraceSimpleTables tb, SimpleOpts o,
                                                    const TraceSeed* __restrict__ seeds, uint64_t n, TraceSimpleOut* __restrict__ outs) {
  const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t i = t >> 2;
  if (i >= n) return;
  const int lane = threadIdx.x & 63, base = lane & 3, quad0 = lane & ~3;
  const bool book = base == 0;
  const TraceSeed sd = seeds[i];
  TraceSimpleOut* out = outs + i;
  auto is_read = [&](uint64_t rb, uint64_t re, uint32_t rs) {
    return rb + 1 == re && rs == v.size_at(rb) && ((has_reads[rb >> 6] >> (rb & 63)) & 1);
  };
  uint32_t counters = 0, status = kSimpleHost;
  // ---- tracer::trace (:477-519)
  const unsigned mc = tb.ref_map[sd.entry] & 63u;
  bool run = true;
  if (mc > 1) {
    counters |= kSimAmbiguousRefReads;
    if (!o.trace_ambiguous_ref) { status = kSimpleNothing; run = false; }
  } else {
    counters |= kSimRefReads;
  }
  unsigned min_overlap = sd.size & 0xffu;  // uint8_t(left_anchor.size())
  unsigned ambiguous_bases = sd.ref_ambiguous;
  int64_t cost = 0;
  // push_next_path (:879-905): cost 0; trace_next_paths (:943-1021) pops it
  if (run && cost > o.max_cost) { counters |= kSimMaxBranchCost; status = kSimpleNothing; run = false; }
  if (run && uint64_t(ambiguous_bases) > o.max_ambiguous_bases) { counters |= kSimTooManyAmbiguousBases; status = kSimpleNothing; run = false; }
  if (run) {
    counters |= kSimStepCount;
    if (1 > o.max_search_steps) { counters |= kSimTooManySteps; status = kSimpleNothing; run = false; }
  }
  if (!run) {  // (uniform over the quad: every lane of it read the same seed)
    if (book) { out->status = status; out->counters = counters; out->n_steps = 0; out->n_ids = 0; out->found_pairs = 0; }
    return;
  }
  // ---- expand_next_path (:528-673) for path 0: the walk, with add_base's books (:847-877) kept per step
  uint64_t b = sd.entry, e = sd.entry + 1;
  uint32_t s = sd.size, min_ctx = o.min_overlap;
  unsigned pushed_since_read = 0, path_bases = 0, pushed_since_pair = 0, max_between_pairs = 0, n_steps = 0;
  bool too_far = false, on_read = false, dead = false;
  uint64_t pb, pe;
  uint32_t ps, valid = 0;
  for (;;) {
    v.push_front_drop(b, e, s, 3 - base, min_ctx, &pb, &pe, &ps);
    const bool ok = pe > pb;  // (loop check: next_path::loop_check_seqset_id is unset before the first read)
    valid = uint32_t((__ballot(ok) >> quad0) & 15ull);
    if (__popc(valid) != 1) { dead = valid == 0; break; }       // AMBIGUOUS: no push (a dead end of path 0) or a branch
    if (n_steps >= uint32_t(kSimpleSteps)) break;              // longer than the fast path keeps: the host's
    const int next = __ffs(int(valid)) - 1;
    b = __shfl(pb, quad0 + next, 64);
    e = __shfl(pe, quad0 + next, 64);
    s = __shfl(ps, quad0 + next, 64);
    if (book) out->bases[n_steps] = uint8_t(next);
    ++n_steps;
    ++min_ctx;
    // add_base
    ++pushed_since_read;
    ++path_bases;
    if (ambiguous_bases) ++ambiguous_bases;
    cost += o.base_cost;
    const unsigned overlap = s - pushed_since_read;
    if (overlap < min_overlap) {
      const unsigned dec = min_overlap - overlap;
      cost += o.decrease_overlap_cost * int64_t(dec);
      min_overlap = overlap & 0xffu;
      max_between_pairs += dec * 2;
    }
    ++pushed_since_pair;
    if (pushed_since_pair > max_between_pairs) {
      ++max_between_pairs;
      cost += o.increase_max_between_pair_cost;
    }
    if (is_read(b, e, s)) { on_read = true; break; }            // STEPPED_ON_READ: add_pairs and the rejoin test below
    if (uint64_t(pushed_since_pair) > o.max_bases_between_pairs) { too_far = true; break; }
  }
  if (!book) return;
  out->n_steps = n_steps;
  out->n_ids = 0;
  out->found_pairs = 0;
  if (too_far) { out->status = kSimpleNothing; out->counters = counters | kSimTooFarWithoutPair; return; }
  if (!on_read) {
    // no push at all from path 0: ++ambiguous_steps, nothing to branch into, no dead-end rejoin for a path without a
    // predecessor (:726-735): the trace ends without an assembly.  Anything else is the host's.
    out->status = (dead && n_steps < uint32_t(kSimpleSteps)) ? kSimpleNothing : kSimpleHost;
    out->counters = counters | (dead ? 0u : (n_steps >= uint32_t(kSimpleSteps) ? (2u << 20) : (1u << 20)));  // why the host: a branch / too long
    return;
  }
  // ---- add_pairs (:737-845) at the read the walk landed on
  const uint64_t sid = b;
  uint32_t first = 0, second = 0;
  {
    const uint64_t word = has_reads[sid >> 6], bit = 1ull << (sid & 63);
    if (word & bit) {
      const uint32_t g = tb.word_rank[sid >> 6] + uint32_t(__popcll(word & (bit - 1)));
      first = tb.group_start[g];
      second = tb.group_start[g + 1];
    }
  }
  if (first != second) {
    pushed_since_read = 0;
    const unsigned ref_count = tb.ref_map[sid] & 63u;
    if (ref_count > 1) {
      ambiguous_bases = ambiguous_bases > s ? ambiguous_bases : s;
      if (!o.trace_ambiguous_ref) { cost += o.max_cost; counters |= kSimPruneAmbiguousRef; }
    } else if (ref_count == 1) {
      cost += o.traverse_ref_cost;
    }
  }
  bool matched_pair = false;
  unsigned added_pairs = 0, added_reads = 0, found_pairs = 0;
  for (uint32_t rid = first; rid != second; ++rid) {
    const uint32_t m1 = tb.mate_loop[rid];
    if (tb.mate_loop[m1] == rid) continue;  // !has_mate
    if (added_reads < o.max_pairs_per_read) {
      if (added_reads < uint32_t(kSimpleIds)) out->ids[added_reads] = rid;
      ++added_reads;
    }
    if (bool(tb.is_forward[rid]) == bool(o.forward_pairs_face_inward)) {
      if (matched_pair) continue;
      int64_t ambiguous_in_ref = 0;
      if (ambiguous_bases > path_bases) ambiguous_in_ref = int64_t(ambiguous_bases) - int64_t(path_bases);
      if (sim_has_rc_mate_in_range(tb, sd, rid, int64_t(sd.cur_offset) + int64_t(path_bases) - int64_t(o.max_pair_distance),
                                   int64_t(sd.cur_offset) - (ambiguous_in_ref > 0 ? ambiguous_in_ref : 0))) {
        matched_pair = true;
        continue;
      }
      // path_has_read_in_range (:1129-1160) over path 0 alone: its pairs were all seen at this very base, at distance 0
      // ("cur_distance > 0" never holds), and the path has no predecessor -- false
    } else {
      if (added_pairs < o.max_pairs_per_read) { ++added_pairs; ++found_pairs; }
      if (matched_pair) continue;
      if (sim_has_rc_mate_in_range(tb, sd, rid, int64_t(sd.cur_offset), int64_t(sd.cur_offset) + int64_t(path_bases) + int64_t(o.max_pair_distance)))
        matched_pair = true;
    }
  }
  if (matched_pair) {
    pushed_since_pair = 0;
    ambiguous_bases = 0;
    cost += o.pairs_used_cost;
    counters |= kSimMatchedPairs;
  }
  out->found_pairs = found_pairs;
  out->n_ids = added_reads;
  if (added_reads > uint32_t(kSimpleIds)) { out->status = kSimpleHost; out->counters = counters | (3u << 20); return; }
  if (uint64_t(pushed_since_pair) > o.max_bases_between_pairs) { out->status = kSimpleNothing; out->counters = counters | kSimTooFarWithoutPair; return; }
  // ---- add_rejoins (:675-724): the nearest place after cur_offset where this read lies on the scaffold
  int64_t best_distance = -1;
  int32_t best_offset = -1;
  {
    const EntryPos* ep = static_cast<const EntryPos*>(tb.entry_positions);
    uint64_t a = 0, z = tb.n_entry_positions;
    while (a < z) {  // upper_bound of (sid, cur_offset)
      const uint64_t mid = (a + z) >> 1;
      const EntryPos m = ep[mid];
      if (m.entry < sid || (m.entry == sid && m.off <= sd.cur_offset)) a = mid + 1; else z = mid;
    }
    for (uint64_t k = a; k < tb.n_entry_positions; ++k) {
      const EntryPos m = ep[k];
      if (m.entry != sid || m.off > sd.win_hi) break;
      int64_t distance = int64_t(sd.cur_offset) + int64_t(path_bases) - int64_t(m.off);
      if (distance < 0) distance = -distance;
      if (best_distance < 0 || distance < best_distance) { best_distance = distance; best_offset = m.off; }
    }
  }
  if (best_distance < 0 || (ambiguous_bases && best_distance > int64_t(ambiguous_bases))) {
    out->status = kSimpleHost;  // the walk goes on from this read: the host's
    out->counters = counters | ((best_distance < 0 ? 4u : 5u) << 20);  // why: the read lies nowhere ahead / ambiguous rejoin
    return;
  }
  const int64_t rejoin_cost = o.rejoin_local_cost + best_distance * o.size_change_cost - o.traverse_ref_cost;
  if (cost + rejoin_cost > o.max_cost) {  // push_rejoin (:907-925) refuses it: the trace ends without an assembly
    out->status = kSimpleNothing;
    out->counters = counters | kSimMaxBranchCost;
    return;
  }
  out->status = kSimpleAssembly;
  out->counters = counters | kSimOutput;
  out->right_offset = best_offset;
  out->right_anchor_len = s;
  out->min_overlap = min_overlap;
}

int tracer_simple_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const TraceSimpleTables& tb, const bg_trace_options& op,
                        const TraceSeed* d_seeds, uint64_t n, TraceSimpleOut* d_out) {
  if (!n) return BG_OK;
  BG_HIP(hipSetDevice(ctx->device));
  SimpleOpts o;
  o.base_cost = op.base_cost;
  o.decrease_overlap_cost = op.decrease_overlap_cost;
  o.increase_max_between_pair_cost = op.increase_max_between_pair_cost;
  o.max_cost = op.max_cost;
  o.traverse_ref_cost = op.traverse_ref_cost;
  o.pairs_used_cost = op.pairs_used_cost;
  o.rejoin_local_cost = op.rejoin_local_cost;
  o.size_change_cost = op.size_change_cost;
  o.max_ambiguous_bases = op.max_ambiguous_bases;
  o.max_search_steps = op.max_search_steps;
  o.max_bases_between_pairs = op.max_bases_between_pairs;
  o.max_pair_distance = op.max_pair_distance;
  o.max_pairs_per_read = op.max_pairs_per_read;
  o.min_overlap = uint32_t(op.min_overlap);
  o.trace_ambiguous_ref = op.trace_ambiguous_ref;
  o.forward_pairs_face_inward = op.forward_pairs_face_inward;
  k_trace_simple<<<cdiv(n * 4, 64), 64, 0, ctx->stream>>>(make_view(ss), rm->s2m_bits.p, tb, o, d_seeds, n, d_out);
  BG_LAUNCH_CHECK();
  return BG_OK;
}

__global__ void k_ref_map_pack(const unsigned long long* __restrict__ cells, uint64_t n, uint8_t* __restrict__ out) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long v = cells[i];
  const uint64_t fw = v & 0xffffffffull, rc = v >> 32;  // (a half overflows at 2^32 positions of one strand on one entry)
  const uint64_t c = fw + rc;
  // rc slice sets k_fwd_flag, forward slice k_rev_flag (ref_map.cpp:61-65)
  out[i] = uint8_t((c > 63 ? 63 : c) | (rc ? 0x80u : 0u) | (fw ? 0x40u : 0u));
}

}  // namespace bg

extern "C" int bg_trace_one_base_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n_frontiers, const uint32_t* d_foff,
                                       const uint64_t* d_begin, const uint64_t* d_end, const uint32_t* d_size,
                                       const uint32_t* d_min_ctx, uint8_t* d_best_base, uint8_t* d_n_cand,
                                       uint64_t* d_cbegin, uint64_t* d_cend, uint32_t* d_csize, uint8_t* d_cdrop,
                                       uint8_t* d_n_next, uint64_t* d_nbegin, uint64_t* d_nend, uint32_t* d_nsize) {
  BG_TRY(check_ss(ctx, ss, "bg_trace_one_base_batch"));
  BG_HIP(hipSetDevice(ctx->device));
  if (!n_frontiers) return BG_OK;
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  ctx->tic("walk_trace_one_base");
  k_trace_one_base<<<cdiv(n_frontiers, 4), 256, 0, ctx->stream>>>(make_view(ss), n_frontiers, d_foff, d_begin, d_end,
                                                                  d_size, d_min_ctx, d_best_base, d_n_cand, d_cbegin,
                                                                  d_cend, d_csize, d_cdrop, d_n_next, d_nbegin, d_nend,
                                                                  d_nsize, err.p);
  BG_LAUNCH_CHECK();
  ctx->toc(n_frontiers * 4 * 52);
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_trace_one_base_batch: a frontier holds more than %d ranges", kFrontierMax); return BG_EINVAL; }
  return BG_OK;
}

extern "C" int bg_branch_trace_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n_asm, const uint64_t* d_seq_off,
                                     const uint8_t* d_seq_codes, const uint32_t* d_foff, const uint64_t* d_begin,
                                     const uint64_t* d_end, const uint32_t* d_size, uint32_t min_overlap,
                                     uint32_t ev_cap, uint32_t* d_ev_off, uint8_t* d_ev_base, uint32_t* d_n_ev,
                                     uint8_t* d_n_out, uint64_t* d_obegin, uint64_t* d_oend, uint32_t* d_osize) {
  BG_TRY(check_ss(ctx, ss, "bg_branch_trace_batch"));
  BG_HIP(hipSetDevice(ctx->device));
  if (!n_asm) return BG_OK;
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  ctx->tic("walk_branch_trace");
  k_branch_trace<<<cdiv(n_asm, 64), 256, 0, ctx->stream>>>(make_view(ss), n_asm, d_seq_off, d_seq_codes, d_foff, d_begin,
                                                          d_end, d_size, min_overlap, ev_cap, d_ev_off, d_ev_base,
                                                          d_n_ev, d_n_out, d_obegin, d_oend, d_osize, err.p, walk_pack() ? 1 : 0);
  BG_LAUNCH_CHECK();
  ctx->toc(0);
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_branch_trace_batch: a frontier holds more than %d ranges", kFrontierMax); return BG_EINVAL; }
  return BG_OK;
}

extern "C" int bg_ref_map_build(bg_ctx* ctx, const bg_seqset* ss, const uint8_t* d_ref_codes, uint64_t n_extents,
                                const uint64_t* h_ext_start, const uint64_t* h_ext_size, uint32_t chain_bases,
                                uint8_t* d_out) {
  BG_TRY(check_ss(ctx, ss, "bg_ref_map_build"));
  if ((n_extents && (!h_ext_start || !h_ext_size || !d_ref_codes)) || (ss->n && !d_out)) {
    set_error("bg_ref_map_build: null argument");
    return BG_EINVAL;
  }
  BG_HIP(hipSetDevice(ctx->device));
  // One lane per chain and a dozen dependent loads per base: what bounds the kernel is how many chains are in flight, not
  // bytes.  4096-base chains over a chr20 (both strands) were 31 000 lanes on a chip that holds 500 000; shorter chains pay
  // their 256-base warm-up more often and still finish sooner -- 25.9 ms at 4096, 20.9 at 2048, 18.4 at 1024, 21.8 at 512,
  // 24.4 at 384 (profiles/r06_walk_ab.txt).  0 = about 130 000 chains, never shorter than 1024 bases; BG_REFMAP_CHAIN
  // overrides whatever the caller passes (measurements).
  if (const char* e = getenv("BG_REFMAP_CHAIN")) { if (atoi(e) >= 256) chain_bases = uint32_t(atoi(e)); }
  if (chain_bases < 256) {
    uint64_t bases = 0;
    for (uint64_t x = 0; x < n_extents; ++x) bases += h_ext_size[x];
    const uint64_t want = std::max<uint64_t>(1024, (2 * bases) / 131072);
    chain_bases = uint32_t(std::min<uint64_t>(4096, (want + 255) / 256 * 256));
  }
  // The reference warms a chunk up over 256 bases.  A chain's context is a prefix of the sequential walk's and grows by a base
  // a step until the two are equal, and no context is longer than the longest entry: max_len + 1 steps are enough (what the
  // tracer's read-ahead chains use, bg_ref_walk_entries) -- 151 instead of 256 for 150-base reads, an eighth of a 1024-base
  // chain's steps less.  Same cells (tests/test_gpu_discover.py: the oracle walks the reference's 256).
  const uint64_t warm = std::min<uint64_t>(256, uint64_t(ss->max_len) + 1);
  std::vector<uint64_t> cstart;
  std::vector<uint32_t> clen, cpre;
  std::vector<int8_t> cdir;
  uint64_t total = 0;
  for (uint64_t x = 0; x < n_extents; ++x) {
    uint64_t st = h_ext_start[x], len = h_ext_size[x];
    for (uint64_t cs = 0; cs < len; cs += chain_bases) {
      uint64_t ce = cs + chain_bases < len ? cs + chain_bases : len;
      uint64_t pre = cs >= warm ? cs - warm : 0;               // ref_map.cpp:95-100
      uint64_t pre_rc = ce + warm < len ? ce + warm : len;     // :102-106
      cstart.push_back(st + pre); clen.push_back(uint32_t(ce - pre)); cpre.push_back(uint32_t(cs - pre)); cdir.push_back(1);
      cstart.push_back(st + pre_rc - 1); clen.push_back(uint32_t(pre_rc - cs)); cpre.push_back(uint32_t(pre_rc - ce)); cdir.push_back(-1);
      total += (ce - pre) + (pre_rc - cs);
    }
  }
  const uint64_t nch = cstart.size(), n = ss->n;
  DevBuf<uint64_t> dstart;
  DevBuf<uint32_t> dlen, dpre;
  DevBuf<uint64_t> cells;
  DevBuf<int8_t> ddir;
  BG_TRY(dstart.alloc(ctx, nch));
  BG_TRY(dlen.alloc(ctx, nch));
  BG_TRY(dpre.alloc(ctx, nch));
  BG_TRY(ddir.alloc(ctx, nch));
  BG_TRY(cells.alloc(ctx, n));
  if (n) BG_HIP(hipMemsetAsync(cells.p, 0, n * 8, ctx->stream));
  if (nch) {
    BG_HIP(hipMemcpyAsync(dstart.p, cstart.data(), nch * 8, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(dlen.p, clen.data(), nch * 4, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(dpre.p, cpre.data(), nch * 4, hipMemcpyHostToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(ddir.p, cdir.data(), nch, hipMemcpyHostToDevice, ctx->stream));
    ctx->tic("walk_ref_map");
    k_ref_map_chains<<<cdiv(nch, 64), 64, 0, ctx->stream>>>(make_view(ss), nch, dstart.p, dlen.p, dpre.p, ddir.p,
                                                            d_ref_codes, reinterpret_cast<unsigned long long*>(cells.p));
    BG_LAUNCH_CHECK();
    ctx->toc(total * 52);
  }
  if (n) {
    k_ref_map_pack<<<cdiv(n, 256), 256, 0, ctx->stream>>>(reinterpret_cast<const unsigned long long*>(cells.p), n, d_out);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// See k_tracer_extend.  rm supplies "this entry has reads" (readmap::get_bit, bio_base/readmap.cpp:197-200).
extern "C" int bg_tracer_extend_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n,
                                      const bg_trace_query* d_queries, uint32_t step_cap, bg_trace_reply* d_replies,
                                      uint32_t* d_steps) {
  BG_TRY(check_ss(ctx, ss, "bg_tracer_extend_batch"));
  if (!rm || (n && (!d_queries || !d_replies || !d_steps)) || !step_cap) { set_error("bg_tracer_extend_batch: null argument"); return BG_EINVAL; }
  if (rm->n_entries != ss->n) { set_error("bg_tracer_extend_batch: the readmap belongs to another seqset"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  ctx->tic("walk_tracer_extend");
  k_tracer_extend<<<cdiv(n * 4, 64), 64, 0, ctx->stream>>>(make_view(ss), rm->s2m_bits.p, n, d_queries, step_cap, d_replies, d_steps);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 4 * 52);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

namespace bg {
// the launch alone, for the tracer's lock-step loop (csrc/tracer.cpp): the caller queues its download behind it and waits once
int tracer_extend_launch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n, const bg_trace_query* d_queries,
                         uint32_t step_cap, bg_trace_reply* d_replies, uint32_t* d_steps) {
  if (!n) return BG_OK;
  k_tracer_extend<<<cdiv(n * 4, 64), 64, 0, ctx->stream>>>(make_view(ss), rm->s2m_bits.p, n, d_queries, step_cap, d_replies, d_steps);
  BG_LAUNCH_CHECK();
  return BG_OK;
}
}  // namespace bg

// Chains of push_front_drop(complement, 0): chain i starts at (begin, end, size), follows len bases (2-bit codes at
// h_codes + code_offset) and writes the range after base t as three u64 at h_out + out_offset + t * out_stride.  All
// buffers are the HOST's: the callers are host state machines that need the answer before their next step; one call =
// one upload, one launch, one download.  An invalid range (end <= begin: no entry starts with the base; the reference
// would not return) stays invalid for the rest of its chain.
extern "C" int bg_walk_chains_batch(bg_ctx* ctx, const bg_seqset* ss, uint32_t n_chains, const bg_walk_chain* h_chains,
                                    const uint8_t* h_codes, uint64_t n_codes, uint64_t* h_out, uint64_t out_words) {
  BG_TRY(check_ss(ctx, ss, "bg_walk_chains_batch"));
  if (!n_chains) return BG_OK;
  if (!h_chains || (n_codes && !h_codes) || (out_words && !h_out)) { set_error("bg_walk_chains_batch: null argument"); return BG_EINVAL; }
  for (uint32_t i = 0; i < n_chains; ++i) {
    const bg_walk_chain& c = h_chains[i];
    if (c.code_offset + c.len > n_codes || (c.len && (c.out_stride < 3 || c.out_offset + uint64_t(c.len - 1) * c.out_stride + 3 > out_words))) {
      set_error("bg_walk_chains_batch: chain %u reaches outside the buffers", i);
      return BG_EINVAL;
    }
  }
  BG_HIP(hipSetDevice(ctx->device));
  const size_t chain_words = (size_t(n_chains) * sizeof(bg_walk_chain) + 7) / 8, code_words = (size_t(n_codes) + 7) / 8;
  DevBuf<uint64_t> buf;
  BG_TRY(buf.alloc(ctx, chain_words + code_words + size_t(out_words)));
  BG_HIP(hipMemcpyAsync(buf.p, h_chains, size_t(n_chains) * sizeof(bg_walk_chain), hipMemcpyHostToDevice, ctx->stream));
  if (n_codes) BG_HIP(hipMemcpyAsync(buf.p + chain_words, h_codes, n_codes, hipMemcpyHostToDevice, ctx->stream));
  if (out_words) BG_HIP(hipMemsetAsync(buf.p + chain_words + code_words, 0, size_t(out_words) * 8, ctx->stream));  // gaps between strides
  ctx->tic("walk_chains");
  k_walk_chains<<<cdiv(n_chains, 64), 64, 0, ctx->stream>>>(make_view(ss), n_chains, reinterpret_cast<const bg_walk_chain*>(buf.p),
                                                            reinterpret_cast<const uint8_t*>(buf.p + chain_words),
                                                            buf.p + chain_words + code_words);
  BG_LAUNCH_CHECK();
  ctx->toc(uint64_t(out_words) / 3 * 52);
  if (out_words) BG_HIP(hipMemcpyAsync(h_out, buf.p + chain_words + code_words, out_words * 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// Chains of seqset_range::push_front (seqset.cpp:319-338) from the empty context: chain i pushes its bases in order
// and stops at the first one after which no entry is left -- what pop_tracer::add_anchor_drop does with a left-anchored
// drop (pop_tracer.cpp:258-266: the longest suffix of the assembly the seqset knows).  One lane per chain.
__global__ __launch_bounds__(64) void k_push_front_chains(SeqsetView v, uint32_t n, const uint64_t* __restrict__ offs,
                                                         const uint8_t* __restrict__ codes, uint64_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t b = 0, e = v.n;
  uint32_t s = 0;
  for (uint64_t t = offs[i]; t < offs[i + 1]; ++t) {
    uint64_t nb, ne;
    uint32_t ns;
    v.push_front(b, e, s, codes[t] & 3, &nb, &ne, &ns);
    if (nb >= ne) break;
    b = nb; e = ne; s = ns;
  }
  out[3 * uint64_t(i)] = b;
  out[3 * uint64_t(i) + 1] = e;
  out[3 * uint64_t(i) + 2] = s;
}

extern "C" int bg_push_front_chains(bg_ctx* ctx, const bg_seqset* ss, uint32_t n, const uint64_t* h_offsets, const uint8_t* h_codes,
                                    uint64_t* h_out) {
  BG_TRY(check_ss(ctx, ss, "bg_push_front_chains"));
  if (!n) return BG_OK;
  if (!h_offsets || !h_out || (h_offsets[n] && !h_codes)) { set_error("bg_push_front_chains: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n_codes = h_offsets[n];
  const size_t off_words = size_t(n) + 1, code_words = (n_codes + 7) / 8 + 1;
  DevBuf<uint64_t> buf;
  BG_TRY(buf.alloc(ctx, off_words + code_words + size_t(n) * 3));
  BG_HIP(hipMemcpyAsync(buf.p, h_offsets, off_words * 8, hipMemcpyHostToDevice, ctx->stream));
  if (n_codes) BG_HIP(hipMemcpyAsync(buf.p + off_words, h_codes, n_codes, hipMemcpyHostToDevice, ctx->stream));
  ctx->tic("walk_push_front_chains");
  k_push_front_chains<<<cdiv(n, 64), 64, 0, ctx->stream>>>(make_view(ss), n, buf.p, reinterpret_cast<const uint8_t*>(buf.p + off_words),
                                                          buf.p + off_words + code_words);
  BG_LAUNCH_CHECK();
  ctx->toc(n_codes * 52);
  BG_HIP(hipMemcpyAsync(h_out, buf.p + off_words + code_words, size_t(n) * 24, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// The heads of ONE path group along one sequence: h_start / h_out hold (begin, end, size) as three u64 per range,
// h_out[(t * n_chains + i) * 3 ..] = chain i after base t.
extern "C" int bg_walk_chains(bg_ctx* ctx, const bg_seqset* ss, uint32_t n_chains, const uint64_t* h_start,
                              const uint8_t* h_codes, uint32_t len, uint64_t* h_out) {
  if (!n_chains || !len) return BG_OK;
  if (!h_start || !h_codes || !h_out) { set_error("bg_walk_chains: null argument"); return BG_EINVAL; }
  std::vector<bg_walk_chain> chains(n_chains);
  for (uint32_t i = 0; i < n_chains; ++i) {
    chains[i].begin = h_start[3 * i];
    chains[i].end = h_start[3 * i + 1];
    chains[i].size = uint32_t(h_start[3 * i + 2]);
    chains[i].len = len;
    chains[i].code_offset = 0;
    chains[i].out_offset = uint64_t(i) * 3;
    chains[i].out_stride = uint64_t(n_chains) * 3;
  }
  return bg_walk_chains_batch(ctx, ss, n_chains, chains.data(), h_codes, len, h_out, uint64_t(len) * n_chains * 3);
}


// d_codes = the n bases (codes 0..3) of ONE extent in walking order; d_entry[i] = the full entry the read-ahead range
// is after base i, or ~0.  See k_ref_walk_entries.
extern "C" int bg_ref_walk_entries(bg_ctx* ctx, const bg_seqset* ss, const uint8_t* d_codes, uint64_t n, int dup_first,
                                   uint64_t* d_entry) {
  BG_TRY(check_ss(ctx, ss, "bg_ref_walk_entries"));
  if (n && (!d_codes || !d_entry)) { set_error("bg_ref_walk_entries: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  const uint32_t chain = 256, warm = ss->max_len + 1;  // a lane walks chain + warm dependent steps (1.7 ms per Mb piece; 1024: 4.8 ms)
  ctx->tic("walk_ref_entries");
  k_ref_walk_entries<<<cdiv(cdiv(n, chain), 64), 64, 0, ctx->stream>>>(make_view(ss), d_codes, n, chain, warm, dup_first ? 1 : 0,
                                                                   d_entry);
  BG_LAUNCH_CHECK();
  ctx->toc(n * 52);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// ------------------------------------------------------------------------------------------------
// Row G4: readmap queries and seqset_path::propagate_from_end.

struct ReadmapView {
  uint64_t n;
  const uint64_t* entry_id;
  const uint64_t* read_len;
  const uint64_t* ptr;
  const uint64_t* is_fwd;
  int min_read_len;
};

static ReadmapView make_rm_view(const bg_readmap* rm) {
  ReadmapView r;
  r.n = rm->n_reads;
  r.entry_id = rm->entry_id.p;
  r.read_len = rm->read_len.p;
  r.ptr = rm->mate_loop_ptr.p;
  r.is_fwd = rm->is_forward.p;
  r.min_read_len = int(rm->min_read_len);
  return r;
}

__device__ __forceinline__ bool rm_has_mate(const ReadmapView& rm, uint64_t id) {  // readmap.cpp:207-218
  return rm.ptr[rm.ptr[id]] != id;
}
__device__ __forceinline__ uint64_t rm_follow(const ReadmapView& rm, uint64_t id, int n) {
  for (int c = 0; c < n; ++c) id = rm.ptr[id];
  return id;
}
// read::get_seqset_entry (readmap.h:249-251): ctx_entry(id).truncate(read length) (seqset.cpp:630-661)
__device__ __forceinline__ Tri rm_read_entry(const SeqsetView& v, const ReadmapView& rm, uint64_t id) {
  uint64_t e = rm.entry_id[id];
  uint32_t len = uint32_t(rm.read_len[id]);
  Tri t{e, e + 1, v.size_at(e)};
  if (t.s > len) {
    t.b = v.next_backward_lt(e, len);
    t.e = e + 1 < v.n ? v.next_forward_lt(e + 1, len) : v.n;
    t.s = len;
  }
  return t;
}

// readmap::get_prefix_reads (readmap.cpp:434-447) + read_iterator (:538-649): reads whose sequence is a
// prefix of the range's, first scanning read ids upward from the range's first entry, then downward;
// the admissible read length shrinks to the running minimum of `shared` over the entries passed.
// f(read_id) returns false to stop.
template <class F>
__device__ __forceinline__ void for_prefix_reads(const SeqsetView& v, const ReadmapView& rm, const Tri& r,
                                                 int read_len_limit, F f) {
  const int min_size = read_len_limit > rm.min_read_len ? read_len_limit : rm.min_read_len;
  if (int(r.s) < min_size || rm.n == 0 || r.b >= r.e) return;
  uint64_t lo = 0, hi = rm.n;  // sparse_multi::lookup_lower_bound (io/sparse_multi.cpp:32-38)
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (uint64_t(rm.entry_id[mid]) < r.b) lo = mid + 1; else hi = mid;
  }
  const uint64_t id0 = lo;
  uint64_t sid = r.b;
  int max_len = int(r.s);
  for (uint64_t id = id0; id < rm.n; ++id) {
    const uint64_t e = rm.entry_id[id];
    bool stop = false;
    while (sid != e) {
      ++sid;
      int sh = v.shared_at(sid);
      if (sh < min_size) { stop = true; break; }
      if (sh < max_len) max_len = sh;
    }
    if (stop) break;
    if (int(rm.read_len[id]) <= max_len && !f(id)) return;
  }
  sid = r.b;
  max_len = int(r.s);
  for (uint64_t id = id0; id-- > 0;) {
    const uint64_t e = rm.entry_id[id];
    bool stop = false;
    while (sid != e) {
      int sh = v.shared_at(sid);
      --sid;
      if (sh < min_size) { stop = true; break; }
      if (sh < max_len) max_len = sh;
    }
    if (stop) break;
    if (int(rm.read_len[id]) <= max_len && !f(id)) return;
  }
}

__global__ void k_prefix_reads(SeqsetView v, ReadmapView rm, uint64_t n, const uint64_t* __restrict__ qb,
                               const uint64_t* __restrict__ qe, const uint32_t* __restrict__ qs, int limit,
                               uint32_t cap, uint32_t* __restrict__ ids, uint32_t* __restrict__ count) {
  uint64_t q = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= n) return;
  uint32_t c = 0;
  Tri r{qb[q], qe[q], qs[q]};
  if (r.e <= v.n)
    for_prefix_reads(v, rm, r, limit, [&](uint64_t id) {
      if (c < cap) ids[q * cap + c] = uint32_t(id);
      ++c;
      return true;
    });
  count[q] = c;
}

__global__ void k_read_info(SeqsetView v, ReadmapView rm, uint64_t n, const uint32_t* __restrict__ read_ids,
                            uint8_t* __restrict__ has_mate, uint32_t* __restrict__ mate_rc,
                            uint32_t* __restrict__ rev_comp, uint64_t* __restrict__ ob, uint64_t* __restrict__ oe,
                            uint32_t* __restrict__ os) {
  uint64_t q = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= n) return;
  uint64_t id = read_ids[q];
  if (id >= rm.n) { has_mate[q] = 0; mate_rc[q] = 0xffffffffu; rev_comp[q] = 0xffffffffu; ob[q] = oe[q] = 0; os[q] = 0; return; }
  bool fwd = rm.is_fwd[id] != 0;
  bool hm = rm_has_mate(rm, id);
  has_mate[q] = hm ? 1 : 0;
  mate_rc[q] = hm ? uint32_t(rm_follow(rm, id, fwd ? 3 : 1)) : 0xffffffffu;  // readmap.cpp:270-290
  rev_comp[q] = uint32_t(rm_follow(rm, id, fwd ? 1 : 3));                      // :248-268
  Tri t = rm_read_entry(v, rm, id);
  ob[q] = t.b; oe[q] = t.e; os[q] = t.s;
}

// seqset_path::propagate_from_end (variants/assemble.cpp:1345-1418): one wave per assembly.  The set
// `cur` lives in lanes 0..15; each step pushes seq[offset] (offset descending) on every range, removes
// ranges that contain another one, records the set at the marked offsets and at 0, and — with a
// readmap — collects get_mate_rc().get_seqset_entry() of the mated prefix reads of every range that has
// at most max_pairs_per_read of them (:1394-1414).  Mates are emitted unsorted with repeats (the
// reference keeps a set; the host folds them).
// k_propagate_single: the same for assemblies that start from at most ONE range and collect no mates -- a set of one range
// stays a set of one range, so a LANE is an assembly (one push a step, nothing to compare): 64 chains of dependent loads a
// wave instead of one.  *n_wide = how many assemblies it left alone (a wider starting set): the caller runs the kernel below
// on those.
__global__ __launch_bounds__(256) void k_propagate_single(
    SeqsetView v, uint64_t na, const uint64_t* __restrict__ soff, const uint8_t* __restrict__ seq, const uint32_t* __restrict__ foff,
    const uint64_t* __restrict__ fbegin, const uint64_t* __restrict__ fend, const uint32_t* __restrict__ fsize,
    const uint32_t* __restrict__ mark_off, const uint32_t* __restrict__ marks, uint8_t* __restrict__ mark_n,
    uint64_t* __restrict__ mb, uint64_t* __restrict__ me, uint32_t* __restrict__ ms, uint8_t* __restrict__ n_out,
    uint64_t* __restrict__ outb, uint64_t* __restrict__ oute, uint32_t* __restrict__ outs, uint32_t* __restrict__ n_mates,
    uint8_t* __restrict__ wide, unsigned long long* __restrict__ n_wide) {
  const uint64_t a = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= na) return;
  const uint32_t lo = foff[a], hi = foff[a + 1];
  if (hi - lo > 1u) { wide[a] = 1; atomicAdd(n_wide, 1ull); return; }
  wide[a] = 0;
  Tri cur{0, 0, 0};
  bool cv = hi > lo;
  if (cv) { cur.b = fbegin[lo]; cur.e = fend[lo]; cur.s = fsize[lo]; }
  cv = cv && cur.b < cur.e && cur.e <= v.n;
  const uint64_t s0 = soff[a];
  const uint64_t len = soff[a + 1] - s0;
  const int64_t mlo = mark_off ? int64_t(mark_off[a]) : 0;
  int64_t mi = mark_off ? int64_t(mark_off[a + 1]) - 1 : -1;
  for (uint64_t off = len; off-- > 0;) {
    const int base = seq[s0 + off] & 3;
    Tri c{0, 0, 0};
    if (cv) v.push_front_drop(cur.b, cur.e, cur.s, base, 0u, &c.b, &c.e, &c.s);
    cv = cv && c.e > c.b;
    cur = c;
    if (mi >= mlo && marks[mi] == uint32_t(off)) {
      if (cv) { mb[uint64_t(mi) * kFrontierMax] = cur.b; me[uint64_t(mi) * kFrontierMax] = cur.e; ms[uint64_t(mi) * kFrontierMax] = cur.s; }
      mark_n[mi] = cv ? 1 : 0;
      --mi;
    }
  }
  if (cv) { outb[a * kFrontierMax] = cur.b; oute[a * kFrontierMax] = cur.e; outs[a * kFrontierMax] = cur.s; }
  n_out[a] = cv ? 1 : 0;
  n_mates[a] = 0;
}

// (only: non-null = the assemblies k_propagate_single left)
__global__ __launch_bounds__(256) void k_propagate_from_end(
    SeqsetView v, ReadmapView rm, int use_rm, const uint8_t* __restrict__ only, uint64_t na, const uint64_t* __restrict__ soff,
    const uint8_t* __restrict__ seq, const uint32_t* __restrict__ foff, const uint64_t* __restrict__ fbegin,
    const uint64_t* __restrict__ fend, const uint32_t* __restrict__ fsize, uint32_t max_pairs,
    const uint32_t* __restrict__ mark_off, const uint32_t* __restrict__ marks, uint8_t* __restrict__ mark_n,
    uint64_t* __restrict__ mb, uint64_t* __restrict__ me, uint32_t* __restrict__ ms, uint8_t* __restrict__ n_out,
    uint64_t* __restrict__ outb, uint64_t* __restrict__ oute, uint32_t* __restrict__ outs, uint32_t mate_cap,
    uint32_t* __restrict__ n_mates, uint64_t* __restrict__ mateb, uint64_t* __restrict__ matee,
    uint32_t* __restrict__ mates, uint32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  uint64_t a = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (a >= na) return;
  if (only && !only[a]) return;
  uint32_t lo = foff[a], hi = foff[a + 1];
  if (hi - lo > uint32_t(kFrontierMax)) {
    if (lane == 0) { atomicOr(err, 1u); n_out[a] = 0; n_mates[a] = 0; }
    return;
  }
  Tri cur{0, 0, 0};
  bool cv = uint32_t(lane) < hi - lo;
  if (cv) { cur.b = fbegin[lo + lane]; cur.e = fend[lo + lane]; cur.s = fsize[lo + lane]; }
  cv = cv && cur.b < cur.e && cur.e <= v.n;
  bool dummy = false;
  int rank, count;
  frontier_dedup(cur, cv, dummy, true, lane, &rank, &count);  // :1360-1361
  frontier_reseat(cur, cv, rank, lane);
  const uint64_t s0 = soff[a];
  const uint64_t len = soff[a + 1] - s0;
  const int64_t mlo = mark_off ? int64_t(mark_off[a]) : 0;
  int64_t mi = mark_off ? int64_t(mark_off[a + 1]) - 1 : -1;
  uint32_t nm = 0;
  for (uint64_t off = len; off-- > 0;) {
    const int base = seq[s0 + off] & 3;
    Tri c{0, 0, 0};
    if (cv) v.push_front_drop(cur.b, cur.e, cur.s, base, 0u, &c.b, &c.e, &c.s);
    bool nv = cv && c.e > c.b;
    dummy = false;
    frontier_dedup(c, nv, dummy, true, lane, &rank, &count);
    frontier_reseat(c, nv, rank, lane);
    cur = c;
    cv = nv;
    if (mi >= mlo && marks[mi] == uint32_t(off)) {
      if (cv) { mb[uint64_t(mi) * kFrontierMax + lane] = cur.b; me[uint64_t(mi) * kFrontierMax + lane] = cur.e; ms[uint64_t(mi) * kFrontierMax + lane] = cur.s; }
      if (lane == 0) mark_n[mi] = uint8_t(count);
      --mi;
    }
    if (use_rm) {
      uint32_t cnt = 0;
      if (cv)
        for_prefix_reads(v, rm, cur, 0, [&](uint64_t id) {
          if (rm_has_mate(rm, id)) { ++cnt; if (cnt > max_pairs) return false; }
          return true;
        });
      uint32_t want = (cv && cnt <= max_pairs) ? cnt : 0u;
      uint32_t incl = want;
      for (int d = 1; d < kFrontierMax; d <<= 1) {
        uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
      }
      uint32_t total = __shfl(incl, kFrontierMax - 1, 64);
      uint32_t slot = nm + incl - want;
      if (want)
        for_prefix_reads(v, rm, cur, 0, [&](uint64_t id) {
          if (rm_has_mate(rm, id)) {
            uint64_t m = rm_follow(rm, id, rm.is_fwd[id] ? 3 : 1);
            Tri t = rm_read_entry(v, rm, m);
            if (slot < mate_cap) {
              uint64_t o = a * uint64_t(mate_cap) + slot;
              mateb[o] = t.b; matee[o] = t.e; mates[o] = t.s;
            }
            ++slot;
          }
          return true;
        });
      nm += total;
    }
  }
  if (cv) { outb[a * kFrontierMax + lane] = cur.b; oute[a * kFrontierMax + lane] = cur.e; outs[a * kFrontierMax + lane] = cur.s; }
  uint64_t live = __ballot(cv);
  if (lane == 0) { n_out[a] = uint8_t(__popcll(live)); n_mates[a] = nm; }
}

static int check_rm(const bg_seqset* ss, const bg_readmap* rm, const char* who) {
  if (!rm) { set_error("%s: null readmap", who); return BG_EINVAL; }
  if (rm->n_entries != ss->n) { set_error("%s: Cannot use a readmap with a seqset it doesn't belong to.", who); return BG_EINVAL; }
  return BG_OK;
}

extern "C" int bg_prefix_reads_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n,
                                     const uint64_t* d_begin, const uint64_t* d_end, const uint32_t* d_size,
                                     int read_len_limit, uint32_t cap, uint32_t* d_ids, uint32_t* d_count) {
  BG_TRY(check_ss(ctx, ss, "bg_prefix_reads_batch"));
  BG_TRY(check_rm(ss, rm, "bg_prefix_reads_batch"));
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  ctx->tic("readmap_prefix_reads");
  k_prefix_reads<<<cdiv(n, 64), 64, 0, ctx->stream>>>(make_view(ss), make_rm_view(rm), n, d_begin, d_end, d_size,
                                                      read_len_limit, cap, d_ids, d_count);
  BG_LAUNCH_CHECK();
  ctx->toc(0);
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// The prefix reads of many ranges with what the pop tracer asks of each read (readmap::get_rev_comp, has_mate, get_mate_rc,
// get_readlength, get_is_forward: pop_tracer.cpp:76-159), packed: a reference stretch of a million positions has a few
// hundred thousand prefix reads in all, not a million times `cap` slots, and each of those attributes is one to five
// dependent loads in tables of a hundred megabytes -- here, not on the host.
__global__ void k_pr_kept(const uint32_t* __restrict__ count, uint64_t n, uint32_t cap, uint32_t* __restrict__ kept) {
  const uint64_t q = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q < n) kept[q] = count[q] < cap ? count[q] : cap;
}
__global__ void k_pr_pack(ReadmapView rm, uint64_t n, uint32_t cap, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ kept,
                          const uint64_t* __restrict__ off, uint32_t* __restrict__ o_id, uint32_t* __restrict__ o_rc,
                          uint32_t* __restrict__ o_mate_rc, uint8_t* __restrict__ o_flags, uint8_t* __restrict__ o_len) {
  const uint64_t q = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const uint32_t k = kept[q];
  for (uint32_t j = 0; j < k; ++j) {
    const uint64_t id = ids[q * cap + j], p = off[q] + j;
    const bool fwd = rm.is_fwd[id] != 0, hm = rm_has_mate(rm, id);
    o_id[p] = uint32_t(id);
    o_rc[p] = uint32_t(rm_follow(rm, id, fwd ? 1 : 3));
    o_mate_rc[p] = hm ? uint32_t(rm_follow(rm, id, fwd ? 3 : 1)) : 0xffffffffu;
    o_flags[p] = uint8_t((hm ? 1 : 0) | (fwd ? 2 : 0));
    const uint64_t len = rm.read_len[id];
    o_len[p] = uint8_t(len < 255 ? len : 255);
  }
}

// pop_tracer::pop_pass (pop_tracer.cpp:364-445) for several passes ahead: every popper pops one base per pass until too
// little context is left, and takes the reverse complements of the first `cap` prefix reads of what remains (add_entry_reads,
// :458-467).  What a pass does to a popper depends on nothing but the popper -- until the host's matching merges it with a
// front; the host replays the passes in order from these records and asks again from where a merge changed a range.
__global__ __launch_bounds__(64) void k_pop_passes(SeqsetView v, ReadmapView rm, uint64_t n, const uint64_t* __restrict__ qb,
                                                  const uint64_t* __restrict__ qe, const uint32_t* __restrict__ qs, uint32_t passes,
                                                  uint32_t min_pop_overlap, int limit, uint32_t cap, bg::PopPassRec* __restrict__ recs,
                                                  uint32_t* __restrict__ ids, unsigned int* __restrict__ ids_used) {
  // (ids_used == null: a record's ids at the fixed place (popper * passes + pass) * cap -- no counter to share; what the
  // zero-copy rounds use, whose buffers are the host's)
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t b = qb[i], e = qe[i];
  uint32_t s = qs[i];
  for (uint32_t j = 0; j < passes; ++j) {
    bg::PopPassRec rec;
    pop_front_range(v, b, e, s, &rec.b, &rec.e, &rec.s);
    rec.n_ids = 0;
    rec.ids_at = 0;
    rec.pad = 0;
    if (rec.s < min_pop_overlap) { recs[i * passes + j] = rec; break; }  // discarded by the host at this pass
    uint32_t tmp[bg::kPopPassMaxIds];
    uint32_t c = 0;
    Tri r{rec.b, rec.e, rec.s};
    if (r.e <= v.n)
      for_prefix_reads(v, rm, r, limit, [&](uint64_t id) {
        tmp[c++] = uint32_t(rm_follow(rm, id, rm.is_fwd[id] ? 1 : 3));  // readmap::get_rev_comp
        return c < cap;
      });
    if (c) {
      const unsigned at = SeqsetView::uniform_nonnull(ids_used) ? atomicAdd(ids_used, c) : unsigned((i * passes + j) * cap);
      for (uint32_t k = 0; k < c; ++k) ids[at + k] = tmp[k];
      rec.ids_at = at;
      rec.n_ids = c;
    }
    recs[i * passes + j] = rec;
    b = rec.b; e = rec.e; s = rec.s;
  }
}

namespace bg {
size_t pop_passes_in_bytes(uint64_t n) { return 8 + n * 16 + ((n * 4 + 7) & ~size_t(7)); }
size_t pop_passes_out_bytes(uint64_t n, uint32_t passes, uint32_t cap) { return 8 + n * passes * sizeof(PopPassRec) + n * passes * cap * 4; }

// One upload (the ranges, from pinned memory), one launch, one download (two when the ids are many) into pinned memory: a pass
// of the pop tracer is a link in a chain of a few hundred launches per work item, and each link used to be six blocking calls.
int pop_passes(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n, const uint64_t* h_begin, const uint64_t* h_end,
               const uint32_t* h_size, uint32_t passes, uint32_t min_pop_overlap, int read_len_limit, uint32_t cap, void* pin_in,
               void* pin_out, std::vector<PopPassRec>* recs, std::vector<uint32_t>* ids) {
  BG_TRY(check_ss(ctx, ss, "pop_passes"));
  BG_TRY(check_rm(ss, rm, "pop_passes"));
  if (rm->pair_mode) { set_error("pop_passes: readmap has no mate loop table; use \"biograph upgrade\" to construct it"); return BG_EINVAL; }
  if (cap < 1 || cap > kPopPassMaxIds || passes < 1) { set_error("pop_passes: bad cap or pass count"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  recs->clear();
  ids->clear();
  if (!n) return BG_OK;
  if (n * passes * cap >= (1ull << 30)) { set_error("pop_passes: batch too large"); return BG_EINVAL; }
  const size_t in_bytes = pop_passes_in_bytes(n), rec_bytes = n * passes * sizeof(PopPassRec), id_bytes = n * passes * cap * 4;
  // A small round (nearly all of them): the kernel reads the ranges from and writes its records and ids to the page-locked host
  // buffers themselves -- a launch and a wait instead of a copy either way around them (BG_TRACER_ZEROCOPY=0: never).
  static const bool zc_off = [] { const char* e = getenv("BG_TRACER_ZEROCOPY"); return e && e[0] == '0'; }();
  if (!zc_off && id_bytes <= (size_t(1) << 20)) {
    uint8_t* hin = static_cast<uint8_t*>(pin_in);
    memcpy(hin, h_begin, n * 8);
    memcpy(hin + n * 8, h_end, n * 8);
    memcpy(hin + n * 16, h_size, n * 4);
    uint8_t* hout = static_cast<uint8_t*>(pin_out);
    PopPassRec* h_recs = reinterpret_cast<PopPassRec*>(hout + 8);
    uint32_t* h_ids = reinterpret_cast<uint32_t*>(hout + 8 + rec_bytes);
    k_pop_passes<<<cdiv(n, 64), 64, 0, ctx->stream>>>(make_view(ss), make_rm_view(rm), n, reinterpret_cast<const uint64_t*>(hin),
                                                       reinterpret_cast<const uint64_t*>(hin) + n, reinterpret_cast<const uint32_t*>(hin + n * 16),
                                                       passes, min_pop_overlap, read_len_limit, cap, h_recs, h_ids, nullptr);
    BG_LAUNCH_CHECK();
    BG_HIP(hipStreamSynchronize(ctx->stream));
    // the records as the caller reads them: a popper's passes up to the one that discards it, their ids side by side
    recs->assign(h_recs, h_recs + n * passes);
    for (uint64_t i = 0; i < n; ++i)
      for (uint32_t j = 0; j < passes; ++j) {
        PopPassRec& r = (*recs)[i * passes + j];
        if (r.s < min_pop_overlap) break;  // (the kernel wrote nothing behind it)
        const uint32_t at = uint32_t(ids->size());
        ids->insert(ids->end(), h_ids + r.ids_at, h_ids + r.ids_at + r.n_ids);
        r.ids_at = at;
      }
    return BG_OK;
  }
  // device: [b n x 8][e n x 8][s n x 4, padded][counter 8][recs][ids]; the upload carries everything before the counter's end
  DevBuf<uint8_t> d;
  BG_TRY(d.alloc(ctx, in_bytes + rec_bytes + id_bytes));
  uint8_t* hin = static_cast<uint8_t*>(pin_in);
  memcpy(hin, h_begin, n * 8);
  memcpy(hin + n * 8, h_end, n * 8);
  memcpy(hin + n * 16, h_size, n * 4);
  memset(hin + in_bytes - 8, 0, 8);
  BG_HIP(hipMemcpyAsync(d.p, hin, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  uint64_t* db = reinterpret_cast<uint64_t*>(d.p);
  uint64_t* de = db + n;
  uint32_t* ds = reinterpret_cast<uint32_t*>(de + n);
  unsigned int* d_used = reinterpret_cast<unsigned int*>(d.p + in_bytes - 8);
  PopPassRec* d_recs = reinterpret_cast<PopPassRec*>(d.p + in_bytes);
  uint32_t* d_ids = reinterpret_cast<uint32_t*>(d.p + in_bytes + rec_bytes);
  k_pop_passes<<<cdiv(n, 64), 64, 0, ctx->stream>>>(make_view(ss), make_rm_view(rm), n, db, de, ds, passes, min_pop_overlap, read_len_limit,
                                                     cap, d_recs, d_ids, d_used);
  BG_LAUNCH_CHECK();
  uint8_t* hout = static_cast<uint8_t*>(pin_out);
  const bool one_copy = id_bytes <= (size_t(256) << 10);
  BG_HIP(hipMemcpyAsync(hout, d.p + in_bytes - 8, 8 + rec_bytes + (one_copy ? id_bytes : 0), hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  uint32_t used = 0;
  memcpy(&used, hout, 4);
  if (!one_copy && used) {
    BG_HIP(hipMemcpyAsync(hout + 8 + rec_bytes, d_ids, size_t(used) * 4, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  const PopPassRec* hr = reinterpret_cast<const PopPassRec*>(hout + 8);
  recs->assign(hr, hr + n * passes);
  const uint32_t* hi = reinterpret_cast<const uint32_t*>(hout + 8 + rec_bytes);
  ids->assign(hi, hi + used);
  return BG_OK;
}
}  // namespace bg

namespace bg {
int prefix_reads_packed(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n, const uint64_t* d_begin,
                        const uint64_t* d_end, const uint32_t* d_size, int read_len_limit, uint32_t cap, PackedPrefixReads* out) {
  BG_TRY(check_ss(ctx, ss, "prefix_reads_packed"));
  BG_TRY(check_rm(ss, rm, "prefix_reads_packed"));
  if (rm->pair_mode) { set_error("prefix_reads_packed: readmap has no mate loop table; use \"biograph upgrade\" to construct it"); return BG_EINVAL; }
  if (rm->max_read_len > 255) { set_error("prefix_reads_packed: reads longer than 255 bases"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  out->count.assign(n, 0);
  out->first.assign(n + 1, 0);
  out->id.clear(); out->rc.clear(); out->mate_rc.clear(); out->flags.clear(); out->len.clear();
  if (!n) return BG_OK;
  DevBuf<uint32_t> ids, count, kept;
  DevBuf<uint64_t> off, tot;
  BG_TRY(ids.alloc(ctx, n * cap)); BG_TRY(count.alloc(ctx, n)); BG_TRY(kept.alloc(ctx, n)); BG_TRY(off.alloc(ctx, n)); BG_TRY(tot.alloc(ctx, 1));
  const SeqsetView v = make_view(ss);
  const ReadmapView rv = make_rm_view(rm);
  k_prefix_reads<<<cdiv(n, 64), 64, 0, ctx->stream>>>(v, rv, n, d_begin, d_end, d_size, read_len_limit, cap, ids.p, count.p);
  BG_LAUNCH_CHECK();
  k_pr_kept<<<cdiv(n, 256), 256, 0, ctx->stream>>>(count.p, n, cap, kept.p);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32_to_u64(ctx, kept.p, off.p, n, tot.p));
  uint64_t total = 0;
  BG_HIP(copy_d2h(ctx, &total, tot.p, 8));
  if (total >= (1ull << 32)) { set_error("prefix_reads_packed: more than 2^32 reads in one batch"); return BG_EINVAL; }
  DevBuf<uint32_t> pk;   // id, rc, mate_rc
  DevBuf<uint8_t> pb;    // flags, len
  BG_TRY(pk.alloc(ctx, total * 3 + 1)); BG_TRY(pb.alloc(ctx, total * 2 + 1));
  k_pr_pack<<<cdiv(n, 64), 64, 0, ctx->stream>>>(rv, n, cap, ids.p, kept.p, off.p, pk.p, pk.p + total, pk.p + 2 * total, pb.p, pb.p + total);
  BG_LAUNCH_CHECK();
  std::vector<uint64_t> h_off(n);
  BG_HIP(copy_d2h(ctx, out->count.data(), count.p, n * 4));
  BG_HIP(copy_d2h(ctx, h_off.data(), off.p, n * 8));
  for (uint64_t i = 0; i < n; ++i) out->first[i] = uint32_t(h_off[i]);
  out->first[n] = uint32_t(total);
  out->id.resize(total); out->rc.resize(total); out->mate_rc.resize(total); out->flags.resize(total); out->len.resize(total);
  if (total) {
    BG_HIP(copy_d2h(ctx, out->id.data(), pk.p, total * 4));
    BG_HIP(copy_d2h(ctx, out->rc.data(), pk.p + total, total * 4));
    BG_HIP(copy_d2h(ctx, out->mate_rc.data(), pk.p + 2 * total, total * 4));
    BG_HIP(copy_d2h(ctx, out->flags.data(), pb.p, total));
    BG_HIP(copy_d2h(ctx, out->len.data(), pb.p + total, total));
  }
  return BG_OK;
}
}  // namespace bg

extern "C" int bg_read_info_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n,
                                  const uint32_t* d_read_ids, uint8_t* d_has_mate, uint32_t* d_mate_rc,
                                  uint32_t* d_rev_comp, uint64_t* d_begin, uint64_t* d_end, uint32_t* d_size) {
  BG_TRY(check_ss(ctx, ss, "bg_read_info_batch"));
  BG_TRY(check_rm(ss, rm, "bg_read_info_batch"));
  if (rm->pair_mode) { set_error("bg_read_info_batch: readmap has no mate loop table; use \"biograph upgrade\" to construct it"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!n) return BG_OK;
  k_read_info<<<cdiv(n, 128), 128, 0, ctx->stream>>>(make_view(ss), make_rm_view(rm), n, d_read_ids, d_has_mate,
                                                     d_mate_rc, d_rev_comp, d_begin, d_end, d_size);
  BG_LAUNCH_CHECK();
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_propagate_from_end_batch(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, uint64_t n_asm,
                                           const uint64_t* d_seq_off, const uint8_t* d_seq_codes,
                                           const uint32_t* d_foff, const uint64_t* d_begin, const uint64_t* d_end,
                                           const uint32_t* d_size, uint32_t max_pairs_per_read,
                                           const uint32_t* d_mark_off, const uint32_t* d_marks, uint8_t* d_mark_n,
                                           uint64_t* d_mbegin, uint64_t* d_mend, uint32_t* d_msize, uint8_t* d_n_out,
                                           uint64_t* d_obegin, uint64_t* d_oend, uint32_t* d_osize, uint32_t mate_cap,
                                           uint32_t* d_n_mates, uint64_t* d_mate_begin, uint64_t* d_mate_end,
                                           uint32_t* d_mate_size) {
  BG_TRY(check_ss(ctx, ss, "bg_propagate_from_end_batch"));
  if (rm) BG_TRY(check_rm(ss, rm, "bg_propagate_from_end_batch"));
  if (rm && rm->pair_mode) { set_error("bg_propagate_from_end_batch: readmap has no mate loop table; use \"biograph upgrade\" to construct it"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!n_asm) return BG_OK;
  DevBuf<uint32_t> err;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  ReadmapView rv{};
  if (rm) rv = make_rm_view(rm);
  ctx->tic("walk_propagate_from_end");
  DevBuf<uint8_t> wide;
  DevBuf<unsigned long long> n_wide;
  bool rest = true;
  if (!rm && walk_pack()) {  // the assemblies that start from one range, a lane each; the others (if any) a wave each as before
    BG_TRY(wide.alloc(ctx, n_asm));
    BG_TRY(n_wide.alloc(ctx, 1));
    BG_HIP(hipMemsetAsync(n_wide.p, 0, 8, ctx->stream));
    k_propagate_single<<<cdiv(n_asm, 256), 256, 0, ctx->stream>>>(make_view(ss), n_asm, d_seq_off, d_seq_codes, d_foff, d_begin, d_end, d_size, d_mark_off,
                                                                  d_marks, d_mark_n, d_mbegin, d_mend, d_msize, d_n_out, d_obegin, d_oend, d_osize,
                                                                  d_n_mates, wide.p, n_wide.p);
    BG_LAUNCH_CHECK();
    unsigned long long h_wide = 0;
    BG_HIP(copy_d2h(ctx, &h_wide, n_wide.p, 8));
    rest = h_wide != 0;
  }
  if (rest)
  k_propagate_from_end<<<cdiv(n_asm, 4), 256, 0, ctx->stream>>>(
      make_view(ss), rv, rm ? 1 : 0, wide.p, n_asm, d_seq_off, d_seq_codes, d_foff, d_begin, d_end, d_size, max_pairs_per_read,
      d_mark_off, d_marks, d_mark_n, d_mbegin, d_mend, d_msize, d_n_out, d_obegin, d_oend, d_osize, mate_cap, d_n_mates,
      d_mate_begin, d_mate_end, d_mate_size, err.p);
  BG_LAUNCH_CHECK();
  ctx->toc(0);
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_propagate_from_end_batch: a frontier holds more than %d ranges", kFrontierMax); return BG_EINVAL; }
  return BG_OK;
}

// ------------------------------------------------------------------------------------------------
// Row G2 driver loop: the extension trace of push_to_pair_discover::on_trace
// (graph_discover/push_to_pair.cpp:271-343) — repeat trace_one_base from the assembly's rc start ranges
// until nothing extends (reason 0), a dropped range of the chosen base repeats (m_seen_ranges, :221-225;
// reason 1), max_steps is reached (2), or one of the bounded per-assembly buffers is full (3 = seen list,
// 4 = range buffer; the host resumes from the last recorded frontier).  One wave per assembly; a step is
// 64 concurrent push_front_drop calls (4 bases x 16 ranges).  The frontier after every step is recorded
// for the host's mate / anchor bookkeeping (:289-342), which cuts the path but never changes it.
// SEEN_LDS: the per-trace list of dropped ranges (m_seen_ranges) lives in the workgroup's LDS (seen_cap x 20 bytes a wave)
// instead of global memory: a step with a dropped candidate was three device-scope loads a lane, a store and a
// __threadfence -- round trips to L2 in a kernel that is nothing but dependent round trips.
// k_push_to_pair_single: the traces that start from at most ONE range (the set never grows: they stay one range), sixteen a
// wave -- four lanes, one a base, are a trace.  The seen list of a trace lies in global memory and is written without waiting;
// what keeps it out of the loop is a 2048-bit filter per trace in LDS: only a dropped range whose bit is already set (a repeat,
// or one time in ten a collision once the list is long) is compared with the list itself, behind a fence.  Same answers as the
// kernel below, which takes the traces this one leaves (*n_wide of them, wide[a] = 1).
constexpr uint32_t kSeenFilterBits = 2048;
__global__ __launch_bounds__(256) void k_push_to_pair_single(
    SeqsetView v, uint64_t na, const uint32_t* __restrict__ foff, const uint64_t* __restrict__ fbegin,
    const uint64_t* __restrict__ fend, const uint32_t* __restrict__ fsize, uint32_t min_overlap, uint32_t max_steps,
    uint32_t range_cap, uint32_t seen_cap, uint64_t* __restrict__ seen_b, uint64_t* __restrict__ seen_e,
    uint32_t* __restrict__ seen_s, uint32_t* __restrict__ n_steps, uint8_t* __restrict__ reason,
    uint8_t* __restrict__ bases, uint8_t* __restrict__ counts, uint32_t* __restrict__ n_ranges,
    uint64_t* __restrict__ rb_out, uint64_t* __restrict__ re_out, uint32_t* __restrict__ rs_out,
    uint8_t* __restrict__ wide, unsigned long long* __restrict__ n_wide) {
  __shared__ uint32_t filt[4][16][kSeenFilterBits / 32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane >> 2, p = lane & 3;
  const uint64_t a = (uint64_t(blockIdx.x) * 4 + uint64_t(wv)) * 16 + uint64_t(sub);
  const bool act = a < na;
  const uint32_t lo = act ? foff[a] : 0, hi = act ? foff[a + 1] : 0;
  const bool mine = act && hi - lo <= 1u;
  if (act && p == 0) {
    wide[a] = mine ? 0 : 1;
    if (!mine) atomicAdd(n_wide, 1ull);
  }
  uint32_t* f = filt[wv][sub];
  for (uint32_t i = uint32_t(p); i < kSeenFilterBits / 32; i += 4) f[i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  Tri cur{0, 0, 0};
  bool cv = mine && hi > lo;
  if (cv) { cur.b = fbegin[lo]; cur.e = fend[lo]; cur.s = fsize[lo]; }
  cv = cv && cur.b < cur.e && cur.e <= v.n;
  uint64_t* sb = seen_b + a * uint64_t(seen_cap);
  uint64_t* se = seen_e + a * uint64_t(seen_cap);
  uint32_t* ssz = seen_s + a * uint64_t(seen_cap);
  uint32_t steps = 0, nseen = 0, nr = 0;
  int why = 0;
  bool done = !mine;
  for (;;) {
    if (!done) {
      if (!cv) { why = 0; done = true; }
      else if (steps >= max_steps) { why = 2; done = true; }
    }
    if (__all(done)) break;
    uint64_t ob = 0, oe = 0;
    uint32_t os = 0;
    if (!done) v.push_front_drop(cur.b, cur.e, cur.s, p, min_overlap, &ob, &oe, &os);
    const bool valid = oe > ob;
    const uint32_t m = valid ? os : 0;
    uint32_t best_len = 0;
    int best = -1;
    for (int bb = 0; bb < 4; ++bb) {  // strict > : first base wins ties (:196)
      const uint32_t l = __shfl(m, sub * 4 + bb, 64);
      if (l > best_len) { best_len = l; best = bb; }
    }
    const int src = sub * 4 + (best < 0 ? 0 : best);
    const Tri c{__shfl(ob, src, 64), __shfl(oe, src, 64), __shfl(os, src, 64)};
    if (!done && best < 0) { why = 0; done = true; }
    const bool cd = !done && c.s != cur.s + 1;  // the chosen base's range gave context up
    // m_seen_ranges: every dropped candidate must be new
    uint32_t h = uint32_t(c.b) * 0x9E3779B1u ^ uint32_t(c.b >> 32) * 0x85EBCA77u ^ uint32_t(c.e - c.b) * 0xC2B2AE3Du ^ c.s * 0x27D4EB2Fu;
    h = (h ^ (h >> 15)) & (kSeenFilterBits - 1);
    const bool maybe = cd && ((f[h >> 5] >> (h & 31)) & 1u);
    if (__any(maybe)) {
      __threadfence();  // the list's earlier stores have landed before anybody compares with them
      bool hit = false;
      if (maybe)
        for (uint32_t i = uint32_t(p); i < nseen; i += 4)
          hit |= __hip_atomic_load(&sb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.b &&
                 __hip_atomic_load(&se[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.e &&
                 __hip_atomic_load(&ssz[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.s;
      const uint32_t hits = uint32_t(__ballot(hit) >> (sub * 4)) & 0xfu;
      if (maybe && hits) { why = 1; done = true; }
    }
    if (cd && !done) {
      if (nseen >= seen_cap) { why = 3; done = true; }
      else {
        if (p == 0) {
          __hip_atomic_store(&sb[nseen], c.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&se[nseen], c.e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&ssz[nseen], c.s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          f[h >> 5] |= 1u << (h & 31);
        }
        ++nseen;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the filter: written by lane 0 of a trace, read by all four)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!done) {
      if (nr + 1 > range_cap) { why = 4; done = true; }
      else {
        if (p == 0) {
          const uint64_t o = a * uint64_t(range_cap) + nr;
          rb_out[o] = c.b; re_out[o] = c.e; rs_out[o] = c.s;
          bases[a * uint64_t(max_steps) + steps] = uint8_t(3 - best);  // appended base = complement (:236)
          counts[a * uint64_t(max_steps) + steps] = 1;
        }
        ++nr;
        ++steps;
        cur = c;
        cv = true;
      }
    }
  }
  if (mine && p == 0) { n_steps[a] = steps; reason[a] = uint8_t(why); n_ranges[a] = nr; }
}

template <bool SEEN_LDS>
__global__ __launch_bounds__(256) void k_push_to_pair_trace(
    SeqsetView v, const uint8_t* __restrict__ only, uint64_t na, const uint32_t* __restrict__ foff, const uint64_t* __restrict__ fbegin,
    const uint64_t* __restrict__ fend, const uint32_t* __restrict__ fsize, uint32_t min_overlap, uint32_t max_steps,
    uint32_t range_cap, uint32_t seen_cap, uint64_t* __restrict__ seen_b, uint64_t* __restrict__ seen_e,
    uint32_t* __restrict__ seen_s, uint32_t* __restrict__ n_steps, uint8_t* __restrict__ reason,
    uint8_t* __restrict__ bases, uint8_t* __restrict__ counts, uint32_t* __restrict__ n_ranges,
    uint64_t* __restrict__ rb_out, uint64_t* __restrict__ re_out, uint32_t* __restrict__ rs_out,
    uint32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  uint64_t a = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (a >= na) return;
  if (only && !only[a]) return;
  uint32_t lo = foff[a], hi = foff[a + 1];
  if (hi - lo > uint32_t(kFrontierMax)) {
    if (lane == 0) { atomicOr(err, 1u); n_steps[a] = 0; reason[a] = 0; n_ranges[a] = 0; }
    return;
  }
  Tri cur{0, 0, 0};
  bool cv = uint32_t(lane) < hi - lo;
  if (cv) { cur.b = fbegin[lo + lane]; cur.e = fend[lo + lane]; cur.s = fsize[lo + lane]; }
  cv = cv && cur.b < cur.e && cur.e <= v.n;
  const int r = lane & 15, b = lane >> 4;
  uint32_t steps = 0, nseen = 0, nr = 0;
  int why = 0;
  extern __shared__ uint64_t seen_lds[];  // SEEN_LDS: per wave seen_cap begins, seen_cap ends, seen_cap sizes
  uint64_t* sb = seen_b + a * uint64_t(seen_cap);
  uint64_t* se = seen_e + a * uint64_t(seen_cap);
  uint32_t* ssz = seen_s + a * uint64_t(seen_cap);
  if (SEEN_LDS) {
    const uint32_t per_wave = seen_cap * 2 + (seen_cap + 1) / 2;  // in 8-byte words
    sb = seen_lds + uint32_t(threadIdx.x >> 6) * per_wave;
    se = sb + seen_cap;
    ssz = reinterpret_cast<uint32_t*>(se + seen_cap);
  }
  for (;;) {
    if (!__ballot(cv)) { why = 0; break; }
    if (steps >= max_steps) { why = 2; break; }
    uint64_t qb = __shfl(cur.b, r, 64), qe = __shfl(cur.e, r, 64);
    uint32_t qs = __shfl(cur.s, r, 64);
    bool qv = __shfl(int(cv), r, 64) != 0;
    uint64_t ob = 0, oe = 0;
    uint32_t os = 0;
    if (qv) v.push_front_drop(qb, qe, qs, b, min_overlap, &ob, &oe, &os);
    bool valid = oe > ob;
    bool drop = valid && os != qs + 1;
    uint32_t m = valid ? os : 0;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
      uint32_t t = __shfl_xor(m, d, 64);
      m = t > m ? t : m;
    }
    uint32_t best_len = 0;
    int best = -1;
    for (int bb = 0; bb < 4; ++bb) {  // strict > : first base wins ties (:196)
      uint32_t l = __shfl(m, bb * 16, 64);
      if (l > best_len) { best_len = l; best = bb; }
    }
    if (best < 0) { why = 0; break; }
    int src = best * 16 + (lane & 15);
    Tri c{__shfl(ob, src, 64), __shfl(oe, src, 64), __shfl(os, src, 64)};
    bool nv = (__shfl(int(valid), src, 64) != 0) && lane < kFrontierMax;
    bool cd = __shfl(int(drop), src, 64) != 0;
    int rank, count;
    frontier_dedup(c, nv, cd, false, lane, &rank, &count);
    // m_seen_ranges: every dropped candidate must be new
    uint64_t dmask = __ballot(nv && cd);
    bool dup = false, full = false;
    while (dmask) {
      int j = __ffsll((long long)dmask) - 1;
      dmask &= dmask - 1;
      Tri t{__shfl(c.b, j, 64), __shfl(c.e, j, 64), __shfl(c.s, j, 64)};
      bool hit = false;
      if (SEEN_LDS) {
        for (uint32_t i = lane; i < nseen; i += 64) hit |= sb[i] == t.b && se[i] == t.e && ssz[i] == t.s;
        if (__ballot(hit)) { dup = true; break; }
        if (nseen >= seen_cap) { full = true; break; }
        if (lane == 0) { sb[nseen] = t.b; se[nseen] = t.e; ssz[nseen] = t.s; }
        ++nseen;
        // (one wave writes and reads its own list: LDS operations of a wave complete in order; the compiler must not move them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        continue;
      }
      for (uint32_t i = lane; i < nseen; i += 64)  // agent-scope loads: lane 0's stores below must be seen
        hit |= __hip_atomic_load(&sb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == t.b &&
               __hip_atomic_load(&se[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == t.e &&
               __hip_atomic_load(&ssz[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == t.s;
      if (__ballot(hit)) { dup = true; break; }
      if (nseen >= seen_cap) { full = true; break; }
      if (lane == 0) {
        __hip_atomic_store(&sb[nseen], t.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&se[nseen], t.e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ssz[nseen], t.s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ++nseen;
      __threadfence();
    }
    if (dup) { why = 1; break; }
    if (full) { why = 3; break; }
    frontier_dedup(c, nv, cd, true, lane, &rank, &count);
    if (nr + uint32_t(count) > range_cap) { why = 4; break; }
    if (nv) {
      uint64_t o = a * uint64_t(range_cap) + nr + rank;
      rb_out[o] = c.b; re_out[o] = c.e; rs_out[o] = c.s;
    }
    if (lane == 0) {
      bases[a * uint64_t(max_steps) + steps] = uint8_t(3 - best);  // appended base = complement (:236)
      counts[a * uint64_t(max_steps) + steps] = uint8_t(count);
    }
    nr += uint32_t(count);
    ++steps;
    frontier_reseat(c, nv, rank, lane);  // the frontier stays in set order
    cur = c;
    cv = nv;
  }
  if (lane == 0) { n_steps[a] = steps; reason[a] = uint8_t(why); n_ranges[a] = nr; }
}

extern "C" int bg_push_to_pair_trace_batch(bg_ctx* ctx, const bg_seqset* ss, uint64_t n_asm, const uint32_t* d_foff,
                                           const uint64_t* d_begin, const uint64_t* d_end, const uint32_t* d_size,
                                           uint32_t min_overlap, uint32_t max_steps, uint32_t range_cap,
                                           uint32_t seen_cap, uint32_t* d_n_steps, uint8_t* d_reason,
                                           uint8_t* d_bases, uint8_t* d_counts, uint32_t* d_n_ranges,
                                           uint64_t* d_rbegin, uint64_t* d_rend, uint32_t* d_rsize) {
  BG_TRY(check_ss(ctx, ss, "bg_push_to_pair_trace_batch"));
  if (!max_steps || !range_cap || !seen_cap) { set_error("bg_push_to_pair_trace_batch: zero capacity"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  if (!n_asm) return BG_OK;
  DevBuf<uint32_t> err, seen_s;
  DevBuf<uint64_t> seen_b, seen_e;
  BG_TRY(err.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(err.p, 0, 4, ctx->stream));
  // 1. the traces that start from one range, sixteen a wave (k_push_to_pair_single); 2. the others a wave each, their seen
  // lists in LDS while four waves' lists leave room for eight workgroups a CU (20 KB each: seen_cap <= 256).
  // BG_WALK_PACK=0: every trace through 2; BG_SEEN_GLOBAL=1: 2 with its lists in global memory.
  const bool single = walk_pack();
  DevBuf<uint8_t> wide;
  DevBuf<unsigned long long> n_wide;
  unsigned long long h_wide = n_asm;
  ctx->tic("walk_push_to_pair_trace");
  if (single) {
    BG_TRY(wide.alloc(ctx, n_asm));
    BG_TRY(n_wide.alloc(ctx, 1));
    BG_TRY(seen_b.alloc(ctx, n_asm * seen_cap));
    BG_TRY(seen_e.alloc(ctx, n_asm * seen_cap));
    BG_TRY(seen_s.alloc(ctx, n_asm * seen_cap));
    BG_HIP(hipMemsetAsync(n_wide.p, 0, 8, ctx->stream));
    k_push_to_pair_single<<<cdiv(n_asm, 64), 256, 0, ctx->stream>>>(
        make_view(ss), n_asm, d_foff, d_begin, d_end, d_size, min_overlap, max_steps, range_cap, seen_cap, seen_b.p, seen_e.p, seen_s.p,
        d_n_steps, d_reason, d_bases, d_counts, d_n_ranges, d_rbegin, d_rend, d_rsize, wide.p, n_wide.p);
    BG_LAUNCH_CHECK();
    BG_HIP(copy_d2h(ctx, &h_wide, n_wide.p, 8));
  }
  const size_t lds = size_t(4) * (size_t(seen_cap) * 2 + (size_t(seen_cap) + 1) / 2) * 8;
  const char* sl = getenv("BG_SEEN_GLOBAL");
  if (!h_wide) {
  } else if (lds <= 20480 && !(sl && sl[0] == '1')) {
    k_push_to_pair_trace<true><<<cdiv(n_asm, 4), 256, lds, ctx->stream>>>(
        make_view(ss), single ? wide.p : nullptr, n_asm, d_foff, d_begin, d_end, d_size, min_overlap, max_steps, range_cap, seen_cap, nullptr, nullptr,
        nullptr, d_n_steps, d_reason, d_bases, d_counts, d_n_ranges, d_rbegin, d_rend, d_rsize, err.p);
  } else {
    if (!seen_b.p) {
      BG_TRY(seen_b.alloc(ctx, n_asm * seen_cap));
      BG_TRY(seen_e.alloc(ctx, n_asm * seen_cap));
      BG_TRY(seen_s.alloc(ctx, n_asm * seen_cap));
    }
    k_push_to_pair_trace<false><<<cdiv(n_asm, 4), 256, 0, ctx->stream>>>(
        make_view(ss), single ? wide.p : nullptr, n_asm, d_foff, d_begin, d_end, d_size, min_overlap, max_steps, range_cap, seen_cap, seen_b.p,
        seen_e.p, seen_s.p, d_n_steps, d_reason, d_bases, d_counts, d_n_ranges, d_rbegin, d_rend, d_rsize, err.p);
  }
  BG_LAUNCH_CHECK();
  ctx->toc(0);
  uint32_t h_err = 0;
  BG_HIP(hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err) { set_error("bg_push_to_pair_trace_batch: a frontier holds more than %d ranges", kFrontierMax); return BG_EINVAL; }
  return BG_OK;
}

// ---- the tables of tracer::advance_read_ahead_to (tracer.cpp:365-400) for one walked stretch, on the device -----------------
// d_ent[i] = the entry the read-ahead range is after base i (bg_ref_walk_entries) or ~0.  Wanted: (offset, entry) of the
// positions whose entry carries reads, the same sorted by (entry, offset), and (reverse complement of the mate, offset) of
// every mated read on such an entry, sorted -- m_entry_positions / m_rc_mate_read_positions of a whole chunk.  On the host
// that is a chain of dependent look-ups in arrays of 10^8 bytes per read (entry -> read ids -> mate -> its reverse
// complement): 0.07 s per work item with the session's thread pool, 0.6 s without it when 16 sessions run side by side.
namespace bg {

__global__ void k_tt_count(const uint64_t* __restrict__ ent, uint64_t n, const uint64_t* __restrict__ has_reads,
                           const uint8_t* __restrict__ ref_map, int skip_ambiguous, ReadmapView rm, uint32_t* __restrict__ cnt_pe,
                           uint32_t* __restrict__ cnt_m, uint32_t* __restrict__ first_read, uint32_t* __restrict__ n_reads) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t e = ent[i];
  uint32_t pe = 0, m = 0, lb = 0, cnt = 0;
  if (e != ~0ull && ((has_reads[e >> 6] >> (e & 63)) & 1)) {
    pe = 1;
    if (!(skip_ambiguous && (ref_map[e] & 63u) > 1)) {  // pairing_skipped (tracer.cpp:386-388)
      uint64_t lo = 0, hi = rm.n;
      while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (rm.entry_id[mid] < e) lo = mid + 1; else hi = mid;
      }
      lb = uint32_t(lo);
      for (uint64_t id = lo; id < rm.n && rm.entry_id[id] == e; ++id) {
        ++cnt;
        if (rm_has_mate(rm, id)) ++m;
      }
    }
  }
  cnt_pe[i] = pe;
  cnt_m[i] = m;
  first_read[i] = lb;
  n_reads[i] = cnt;
}

__global__ void k_tt_emit(const uint64_t* __restrict__ ent, uint64_t n, const uint32_t* __restrict__ cnt_pe, const uint64_t* __restrict__ off_pe,
                          const uint64_t* __restrict__ off_m, const uint32_t* __restrict__ first_read, const uint32_t* __restrict__ n_reads,
                          ReadmapView rm, uint32_t* __restrict__ pe_idx, uint64_t* __restrict__ pe_ent, uint64_t* __restrict__ pe_key,
                          uint64_t* __restrict__ m_key) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n || !cnt_pe[i]) return;
  const uint64_t e = ent[i], p = off_pe[i];
  pe_idx[p] = uint32_t(i);
  pe_ent[p] = e;
  pe_key[p] = (e << 24) | i;
  uint64_t q = off_m[i];
  for (uint32_t k = 0; k < n_reads[i]; ++k) {
    const uint64_t id = uint64_t(first_read[i]) + k;
    if (!rm_has_mate(rm, id)) continue;
    const uint64_t mate = rm.ptr[rm.ptr[id]];                         // readmap::get_mate (readmap.cpp:220-239)
    const uint64_t rc = rm_follow(rm, mate, rm.is_fwd[mate] ? 1 : 3);  // get_rev_comp (:248-268)
    m_key[q++] = (rc << 24) | i;
  }
}

// the tables as the host keeps them: (offset, entry) by position, (entry, offset) and (rc mate, offset) sorted -- two words per
// pair of the first two (the int32 member with its padding in one word), one word per pair of the third
__global__ void k_tt_finish(uint64_t n_pe, uint64_t n_m, const uint32_t* __restrict__ pe_idx, const uint64_t* __restrict__ pe_ent,
                            const uint64_t* __restrict__ key, const uint64_t* __restrict__ mkey, int64_t first_off,
                            uint64_t* __restrict__ o_pe, uint64_t* __restrict__ o_ep, uint64_t* __restrict__ o_mp,
                            const uint8_t* __restrict__ sizes, const uint8_t* __restrict__ ref_map, uint16_t* __restrict__ o_info) {
  const uint64_t p = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p < n_pe) {
    o_pe[2 * p] = uint64_t(uint32_t(int32_t(first_off + int64_t(pe_idx[p]))));
    o_pe[2 * p + 1] = pe_ent[p];
    // what the sweep over the positions asks of every entry (tracer.cpp: the seed's size, "the reference holds it more than once"):
    // two random reads in 140 MB host tables a position there, a coalesced column here
    if (o_info) o_info[p] = uint16_t(sizes[pe_ent[p]]) | uint16_t((ref_map[pe_ent[p]] & 63u) > 1 ? 256 : 0);
    o_ep[2 * p] = key[p] >> 24;
    o_ep[2 * p + 1] = uint64_t(uint32_t(int32_t(first_off + int64_t(key[p] & 0xffffff))));
  }
  if (p < n_m) o_mp[p] = (mkey[p] >> 24 & 0xffffffffull) | (uint64_t(uint32_t(int32_t(first_off + int64_t(mkey[p] & 0xffffff)))) << 32);
}

int tracer_read_tables(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const uint8_t* d_ref_map, bool skip_ambiguous,
                       const uint64_t* d_ent, uint64_t n, int64_t first_off, std::vector<std::pair<int32_t, uint64_t>>* pos_entries,
                       std::vector<std::pair<uint64_t, int32_t>>* entry_pos, std::vector<std::pair<uint32_t, int32_t>>* mate_pos,
                       DevBuf<uint64_t>* keep_entry_pos, DevBuf<uint64_t>* keep_mate_pos, std::vector<uint16_t>* pos_info) {
  if (!n) return BG_OK;
  if (n >= (1ull << 24) || ss->n >= (1ull << 39) || rm->n_reads >= (1ull << 32)) { set_error("tracer_read_tables: stretch too long"); return BG_EINVAL; }
  DevBuf<uint32_t> cnt_pe, cnt_m, first_read, n_reads;
  DevBuf<uint64_t> off_pe, off_m, tot;
  BG_TRY(cnt_pe.alloc(ctx, n)); BG_TRY(cnt_m.alloc(ctx, n)); BG_TRY(first_read.alloc(ctx, n)); BG_TRY(n_reads.alloc(ctx, n));
  BG_TRY(off_pe.alloc(ctx, n)); BG_TRY(off_m.alloc(ctx, n)); BG_TRY(tot.alloc(ctx, 2));
  const ReadmapView rv = make_rm_view(rm);
  ctx->tic("tracer_read_tables");
  k_tt_count<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_ent, n, rm->s2m_bits.p, d_ref_map, skip_ambiguous ? 1 : 0, rv, cnt_pe.p, cnt_m.p,
                                                    first_read.p, n_reads.p);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32_to_u64(ctx, cnt_pe.p, off_pe.p, n, tot.p));
  BG_TRY(exclusive_scan_u32_to_u64(ctx, cnt_m.p, off_m.p, n, tot.p + 1));
  uint64_t h_tot[2] = {0, 0};
  BG_HIP(copy_d2h(ctx, h_tot, tot.p, 16));
  const uint64_t n_pe = h_tot[0], n_m = h_tot[1];
  DevBuf<uint32_t> pe_idx;
  DevBuf<uint64_t> pe_ent, ka, kb, ma, mb;
  BG_TRY(pe_idx.alloc(ctx, n_pe)); BG_TRY(pe_ent.alloc(ctx, n_pe)); BG_TRY(ka.alloc(ctx, n_pe)); BG_TRY(kb.alloc(ctx, n_pe));
  BG_TRY(ma.alloc(ctx, n_m)); BG_TRY(mb.alloc(ctx, n_m));
  k_tt_emit<<<cdiv(n, 256), 256, 0, ctx->stream>>>(d_ent, n, cnt_pe.p, off_pe.p, off_m.p, first_read.p, n_reads.p, rv, pe_idx.p, pe_ent.p,
                                                   ka.p, ma.p);
  BG_LAUNCH_CHECK();
  bool pe_in_b = false, m_in_b = false;
  if (n_pe) BG_TRY(radix_sort_u64(ctx, ka.p, kb.p, nullptr, nullptr, n_pe, 0, 64, &pe_in_b));
  if (n_m) BG_TRY(radix_sort_u64(ctx, ma.p, mb.p, nullptr, nullptr, n_m, 0, 56, &m_in_b));
  ctx->toc(n * 40 + (n_pe + n_m) * 64);
  // the three host tables in their final layout, straight into the vectors' storage: std::pair<int32_t, uint64_t>,
  // std::pair<uint64_t, int32_t> (16 bytes each, the second member at 8) and std::pair<uint32_t, int32_t> (8 bytes)
  static_assert(sizeof(std::pair<int32_t, uint64_t>) == 16 && sizeof(std::pair<uint64_t, int32_t>) == 16 && sizeof(std::pair<uint32_t, int32_t>) == 8,
                "pair layouts");
  DevBuf<uint64_t> o_pe, o_ep, o_mp;
  DevBuf<uint16_t> o_info;
  BG_TRY(o_pe.alloc(ctx, n_pe * 2)); BG_TRY(o_ep.alloc(ctx, n_pe * 2)); BG_TRY(o_mp.alloc(ctx, n_m));
  if (pos_info) BG_TRY(o_info.alloc(ctx, n_pe));
  const uint64_t n_fin = n_pe > n_m ? n_pe : n_m;
  if (n_fin) {
    k_tt_finish<<<cdiv(n_fin, 256), 256, 0, ctx->stream>>>(n_pe, n_m, pe_idx.p, pe_ent.p, pe_in_b ? kb.p : ka.p, m_in_b ? mb.p : ma.p, first_off,
                                                         o_pe.p, o_ep.p, o_mp.p, ss->sizes.p, d_ref_map, pos_info ? o_info.p : nullptr);
    BG_LAUNCH_CHECK();
  }
  const size_t pe0 = pos_entries->size(), ep0 = entry_pos->size(), mp0 = mate_pos->size();
  pos_entries->resize(pe0 + n_pe);
  entry_pos->resize(ep0 + n_pe);
  mate_pos->resize(mp0 + n_m);
  if (n_pe) {
    BG_HIP(hipMemcpyAsync(static_cast<void*>(pos_entries->data() + pe0), o_pe.p, n_pe * 16, hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipMemcpyAsync(static_cast<void*>(entry_pos->data() + ep0), o_ep.p, n_pe * 16, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (n_m) BG_HIP(hipMemcpyAsync(static_cast<void*>(mate_pos->data() + mp0), o_mp.p, n_m * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (pos_info) {
    pos_info->resize(pe0 + n_pe);
    if (n_pe) BG_HIP(hipMemcpyAsync(pos_info->data() + pe0, o_info.p, n_pe * 2, hipMemcpyDeviceToHost, ctx->stream));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (keep_entry_pos) *keep_entry_pos = std::move(o_ep);
  if (keep_mate_pos) *keep_mate_pos = std::move(o_mp);
  return BG_OK;
}

}  // namespace bg


// Measurement: while on, every push_front_drop of every walk kernel of this process adds to two device counters (calls,
// iterations of the drop loop: N_pfd and N_drop of SURVEY.md section 8(d)'s walk bytes).  on = 1 zeroes and switches them on,
// on = 0 reads them into h_out2 (may be NULL) and switches them off.  Atomics on two addresses: never inside a timed region.
extern "C" int bg_walk_counters(bg_ctx* ctx, int on, uint64_t* h_out2) {
  if (!ctx) return BG_EINVAL;
  BG_HIP(hipSetDevice(ctx->device));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  static unsigned long long* buf = nullptr;
  if (!buf) BG_HIP(hipMalloc(reinterpret_cast<void**>(&buf), 16));
  if (on) {
    BG_HIP(hipMemset(buf, 0, 16));
    bg::g_walk_counters = buf;
    return BG_OK;
  }
  unsigned long long h[2] = {0, 0};
  BG_HIP(hipMemcpy(h, buf, 16, hipMemcpyDeviceToHost));
  bg::g_walk_counters = nullptr;
  if (h_out2) { h_out2[0] = h[0]; h_out2[1] = h[1]; }
  return BG_OK;
}
