// sharded.hip — `biograph create` for ONE sample sharded over N ranks, one process per GPU: bg_create_sharded / bg_sharded_write.
//
// The reference's driver is SEQSETMain::run (modules/biograph/biograph_create.cpp:431-811): kmerize -> correct -> part_repo
// partitions on disk (part_repo.cpp:32-45: a record lives in the partition of its leading bases) -> per-partition sort / dedup /
// expand -> builder -> make_readmap, with `parallel_for` workers sharing the files.  Here every structure has ONE owner rank and
// lives in that rank's HBM; what crosses ranks goes through the communicator (csrc/comm.hip: RCCL over xGMI):
//
//   1  super-k-mer records -> owner of their minimiser's bucket        all_to_all        (256 hash partitions, kmer_counter.h:167-178)
//   1b presence bits of every record's windows back to its reader       reply
//   2  routed correction: the look-ups a read's own windows do not answer travel as probe strings   all_to_all + reply per step
//   3  oriented corrected reads -> owner of their leading 8 bases: the owner's slice of the cover index and its j = 0 records
//   4  cover rounds: the owners' first-K-mer filters on every rank (all_gather), candidates -> owner + answers per scan round,
//      then the records in windows, each travelling while the next is generated
//   5  first / last entry of every shard                                 host all_gather   (part_repo.cpp:294-303, builder.cpp:225)
//   6  two-word tails -> shard of lower_bound(tail), full tails for the few the two words do not settle   (builder.cpp:85-108)
//   7  512-aligned slices: every rank writes its bytes of every container member in place (spiral_file_mmap.cpp:350-457)
//   8  the sample's readmap, sharded like the seqset (make_readmap.cpp:120-392), gathered on the writing rank
//
// The phases are the C ABI's own entry points (shard.hip, shard_cover.hip, build_kmer.hip, readmap.hip); this file is the
// orchestration biograph_amd/sharded.py::sharded_create holds for simulated ranks, for one real rank per process.
#include "comm.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

struct bg_correction;
struct bg_readmap_shard;

struct bg_sharded {
  bg_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  bg_seqset* slice = nullptr;
  uint64_t first = 0, n_total = 0;
  uint32_t max_len = 0;
  bool is_last = false;
  uint64_t base_pop[4] = {0, 0, 0, 0}, fixed[5] = {0, 0, 0, 0, 0};
  // this rank's stretch of the readmap's columns (global entry ids, global read ids)
  bool has_readmap = false;
  uint64_t rm_first = 0, rm_n = 0, rm_total = 0, rm_entries = 0;
  uint32_t rm_max_read_len = 0;
  bg::DevBuf<uint64_t> rm_entry;
  bg::DevBuf<uint32_t> rm_len, rm_loop;
  bg::DevBuf<uint8_t> rm_fwd;
  ~bg_sharded() { if (slice) bg_seqset_free(slice); }
};

namespace bg {
namespace {

constexpr int kPrefixBases = 8;  // splitter resolution: 4^8 bins of leading bases

struct Rows {  // n rows of w u64 words on the device
  DevBuf<uint64_t> b;
  uint64_t n = 0;
  int w = 0;
  int alloc(bg_ctx* ctx, uint64_t rows, int width) { n = rows; w = width; return b.alloc(ctx, rows * uint64_t(width)); }
  Rows view() const {  // the same memory, not owned
    Rows v;
    v.b.p = b.p; v.b.n = b.n; v.b.ctx = nullptr; v.n = n; v.w = w;
    return v;
  }
  void reset() { b.reset(); n = 0; }
};

struct Phases {  // BG_SHARD_TIMING=1: wall time per phase on rank 0's stderr (device synchronised)
  bg_ctx* ctx;
  bool on;
  std::string name;
  double t = 0;
  std::string line;
  Phases(bg_ctx* c, int rank) : ctx(c) { const char* e = getenv("BG_SHARD_TIMING"); on = e && *e == '1' && rank == 0; }
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void mark(const char* n) {
    if (!on) return;
    (void)hipStreamSynchronize(ctx->stream);
    const double w = now();
    if (!name.empty()) { char buf[96]; snprintf(buf, sizeof buf, "%s%s %.1f", line.empty() ? "" : ", ", name.c_str(), (w - t) * 1e3); line += buf; }
    name = n;
    t = w;
  }
  void done() { if (on) { mark(""); fprintf(stderr, "bg_create_sharded phases (ms): %s\n", line.c_str()); } }
};

int gather_u64(bg_comm* c, const std::vector<uint64_t>& mine, std::vector<uint64_t>* all) {
  all->assign(mine.size() * size_t(comm_world(c)), 0);
  return comm_all_gather_host(c, mine.data(), mine.size() * 8, all->data());
}
int all_sum(bg_comm* c, uint64_t v, uint64_t* out) {
  std::vector<uint64_t> all;
  BG_TRY(gather_u64(c, {v}, &all));
  uint64_t s = 0;
  for (uint64_t x : all) s += x;
  *out = s;
  return BG_OK;
}

// rows travel to the ranks `counts` names; the whole counts matrix comes from ONE host all-gather (cm[s * P + d]).
// keep_send: the caller goes on using `send` (a rank alone then gets a view of it instead of the buffer itself).
int a2a_v(bg_comm* c, bg_ctx* ctx, Rows& send, const std::vector<uint64_t>& counts, Rows* recv, std::vector<uint64_t>* cm, bool keep_send = false) {
  const int P = comm_world(c), me = comm_rank(c);
  std::vector<uint64_t> m;
  BG_TRY(gather_u64(c, counts, &m));
  if (comm_is_shortcut(c)) {
    if (keep_send) *recv = send.view(); else *recv = std::move(send);
    recv->n = counts[0];
  } else {
    std::vector<uint64_t> rr(P);
    uint64_t tot = 0;
    for (int s = 0; s < P; ++s) { rr[s] = m[size_t(s) * P + me]; tot += rr[s]; }
    BG_TRY(recv->alloc(ctx, tot, send.w));
    BG_TRY(comm_exchange(c, send.b.p, counts.data(), rr.data(), size_t(send.w) * 8, recv->b.p));
    if (!keep_send) send.reset();  // everything is ordered on the context's stream: the memory is not reused before the exchange has run
  }
  if (cm) *cm = std::move(m);
  return BG_OK;
}

// the way back of an exchange whose counts matrix was cm: one row of row_bytes per row received then, in received order ->
// one row per row sent then, in sent order
int reply(bg_comm* c, const void* d_send, size_t row_bytes, const std::vector<uint64_t>& cm, void* d_recv) {
  const int P = comm_world(c), me = comm_rank(c);
  std::vector<uint64_t> sr(P), rr(P);
  for (int p = 0; p < P; ++p) { sr[p] = cm[size_t(p) * P + me]; rr[p] = cm[size_t(me) * P + p]; }
  return comm_exchange(c, d_send, sr.data(), rr.data(), row_bytes, d_recv);
}

int partition(bg_ctx* ctx, const uint64_t* rows, uint64_t n, int w, int mode, int P, const uint64_t* params, Rows* out, std::vector<uint64_t>* counts) {
  counts->assign(P, 0);
  BG_TRY(out->alloc(ctx, n, w));
  BG_TRY(bg_partition_rows(ctx, rows, n, w, mode, uint32_t(P), params, out->b.p, counts->data()));
  uint64_t tot = 0;
  for (uint64_t x : *counts) tot += x;
  out->n = tot;
  return BG_OK;
}

// Super-k-mer records of a reads object grouped by owner rank (sharded.py::_skm_extract): buffers sized for first_room of
// bg_skm_capacity's 3x-the-expected-density worst case, the whole of it only when an input beats that.
int skm_extract(bg_ctx* ctx, const bg_reads* reads, int k, int P, double first_room, Rows* recs, DevBuf<uint64_t>* org, std::vector<uint64_t>* counts,
                uint64_t* n_windows) {
  uint64_t n = 0, cap = 0, nrec = 0;
  BG_TRY(bg_kmer_instances(ctx, reads, k, &n));
  BG_TRY(bg_skm_capacity(ctx, reads, k, n, &cap));
  *n_windows = n;
  counts->assign(P, 0);
  if (!cap) {
    BG_TRY(recs->alloc(ctx, 0, 2));
    BG_TRY(org->alloc(ctx, 0));
    return BG_OK;
  }
  const uint64_t rooms[2] = {std::min<uint64_t>(uint64_t(double(cap) * first_room) + 1024, cap), cap};
  for (int t = 0; t < 2; ++t) {
    const uint64_t room = rooms[t];
    BG_TRY(recs->alloc(ctx, room, 2));
    BG_TRY(org->alloc(ctx, room));
    int rc = bg_skm_extract_by_owner(ctx, reads, k, n, uint32_t(P), recs->b.p, org->p, room, counts->data(), &nrec);
    if (rc == BG_ENOMEM && room < cap) { recs->reset(); org->reset(); continue; }
    BG_TRY(rc);
    break;
  }
  recs->b.shrink(nrec * 2);  // the unused tail goes back to the arena in place
  recs->n = nrec;
  org->shrink(nrec);
  return BG_OK;
}

// Balanced contiguous bin ranges from the global leading-bases histogram (sharded.py::choose_splitters): splitter j = first
// bin at which the running count reaches j / P of the total (ascending, may repeat).
std::vector<uint64_t> choose_splitters(const std::vector<uint64_t>& hist, int P) {
  std::vector<double> cum(hist.size());
  double run = 0;
  for (size_t i = 0; i < hist.size(); ++i) { run += double(hist[i]); cum[i] = run; }
  const double total = hist.empty() ? 0.0 : cum.back();
  std::vector<uint64_t> sp;
  for (int j = 1; j < P; ++j) {
    uint64_t b = hist.size();
    if (total > 0) b = uint64_t(std::lower_bound(cum.begin(), cum.end(), total * double(j) / double(P)) - cum.begin()) + 1;
    if (!sp.empty()) b = std::max(b, sp.back());
    sp.push_back(std::min<uint64_t>(b, hist.size()));
  }
  return sp;
}

__global__ void k_tag_rows(uint64_t* __restrict__ rows, uint64_t n, int rw) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) rows[i * rw + rw - 1] |= i << 8;  // the row's way home
}
__global__ void k_scatter_by_tag(const uint64_t* __restrict__ sent, uint64_t n, int rw, const uint64_t* __restrict__ ans, uint64_t* __restrict__ out) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[sent[i * rw + rw - 1] >> 8] = ans[i];
}
__global__ void k_sum_bytes(const uint8_t* __restrict__ v, uint64_t n, unsigned long long* __restrict__ out) {
  uint64_t s = 0;
  for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) s += v[i];
  for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, (unsigned long long)s);
}

int sum_bytes(bg_ctx* ctx, const uint8_t* d, uint64_t n, uint64_t* out) {
  *out = 0;
  if (!n) return BG_OK;
  DevBuf<unsigned long long> acc;
  BG_TRY(acc.alloc(ctx, 1));
  BG_HIP(hipMemsetAsync(acc.p, 0, 8, ctx->stream));
  k_sum_bytes<<<std::min<unsigned>(cdiv(n, 256), 2048u), 256, 0, ctx->stream>>>(d, n, acc.p);
  BG_LAUNCH_CHECK();
  unsigned long long h = 0;
  BG_HIP(copy_d2h(ctx, &h, acc.p, 8));
  *out = h;
  return BG_OK;
}

struct Handles {  // what must not outlive an error return
  bg_reads* reads = nullptr;
  bg_reads* corrected = nullptr;
  bg_reads* rm_reads = nullptr;
  bg_kmerset* owned = nullptr;
  bg_correction* corr = nullptr;
  bg_cover* cover = nullptr;
  bg_cover_query* query = nullptr;
  bg_records* records = nullptr;
  bg_seqset* shard = nullptr;
  bg_readmap_shard* rsh = nullptr;
  ~Handles() {
    if (rsh) bg_readmap_shard_free(rsh);
    if (shard) bg_seqset_free(shard);
    if (records) bg_records_free(records);
    if (query) bg_cover_query_free(query);
    if (cover) bg_cover_free(cover);
    if (corr) bg_corr_free(corr);
    if (owned) bg_kmerset_free(owned);
    if (rm_reads) bg_reads_free(rm_reads);
    if (corrected) bg_reads_free(corrected);
    if (reads) bg_reads_free(reads);
  }
};

std::vector<int> parse_ints(const char* env, const char* dflt) {
  const char* s = getenv(env);
  if (!s || !*s) s = dflt;
  std::vector<int> v;
  while (*s) {
    v.push_back(int(strtol(s, const_cast<char**>(&s), 10)));
    while (*s == ',' || *s == ' ') ++s;
  }
  return v;
}

// ---- 2: read correction with the kept k-mers left on their owners (sharded.py::_correct_routed)
int correct_routed(bg_comm* c, bg_ctx* ctx, Handles& H, int k, const bg_create_params* prm, Rows& skm_recs, DevBuf<uint64_t>& skm_org,
                   DevBuf<uint32_t>& skm_bits, bg_correct_stats* cst, uint32_t* steps_out, uint64_t* probes_out, Phases& ph) {
  const int P = comm_world(c);
  BG_TRY(bg_corr_begin(ctx, H.reads, k, &prm->correct, skm_recs.b.p, skm_org.p, skm_bits.p, skm_recs.n, &H.corr));
  uint32_t steps = 0;
  for (;;) {
    ph.mark("correct_step");
    uint64_t npend = 0, tot = 0;
    bg_reads* probes = nullptr;
    BG_TRY(bg_corr_step(ctx, H.corr, &npend, &probes));
    ++steps;
    int rc = all_sum(c, npend, &tot);
    if (rc != BG_OK || !tot) { if (probes) bg_reads_free(probes); BG_TRY(rc); break; }
    ph.mark("correct_probe_exchange");
    Rows recs, rows;
    DevBuf<uint64_t> porg;
    std::vector<uint64_t> counts(P, 0), cm;
    if (probes) {
      uint64_t nw = 0;
      rc = skm_extract(ctx, probes, k, P, 0.5, &recs, &porg, &counts, &nw);
      bg_reads_free(probes);
      BG_TRY(rc);
    } else {
      BG_TRY(recs.alloc(ctx, 0, 2));
      BG_TRY(porg.alloc(ctx, 0));
    }
    const uint64_t n_sent = recs.n;
    BG_TRY(a2a_v(c, ctx, recs, counts, &rows, &cm));
    DevBuf<uint32_t> bits, ans;
    BG_TRY(bits.alloc(ctx, rows.n));
    // every window of a probe is looked up: a probe answers by K-MER, and a search reaches a probe's later windows by other
    // routes too (tests/fuzz_parity.py seed 300028)
    BG_TRY(bg_skm_lookup(ctx, H.owned, rows.b.p, rows.n, bits.p, 0));
    BG_TRY(ans.alloc(ctx, n_sent));
    BG_TRY(reply(c, bits.p, 4, cm, ans.p));
    BG_TRY(bg_corr_answers(ctx, H.corr, porg.p, ans.p, n_sent));
  }
  BG_TRY(bg_corr_finish(ctx, H.corr, &H.corrected, cst));
  *probes_out = bg_corr_probes(H.corr);
  bg_corr_free(H.corr);
  H.corr = nullptr;
  *steps_out = steps;
  return BG_OK;
}

// ---- 3 + 4: suffix records without a replicated cover index (sharded.py::_records_sharded_cover + _filtered_cover_rounds)
int records_sharded_cover(bg_comm* c, bg_ctx* ctx, Handles& H, int W, int k, std::vector<Rows>* parts, uint64_t* n_records, Phases& ph,
                          Rows* first_covered) {
  const int P = comm_world(c);
  const int rw = W + 1;
  ph.mark("cover_oriented_rows");
  const uint64_t n = bg_reads_count(H.corrected);
  Rows rows;
  BG_TRY(rows.alloc(ctx, 2 * n, rw));
  BG_TRY(bg_oriented_rows(ctx, H.corrected, W, rows.b.p));
  bg_reads_free(H.corrected);
  H.corrected = nullptr;
  const size_t bins = size_t(1) << (2 * kPrefixBases);
  DevBuf<uint64_t> hist;
  BG_TRY(hist.alloc(ctx, bins));
  uint64_t sample = 8;
  if (const char* e = getenv("BG_PREFIX_SAMPLE")) sample = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  BG_TRY(bg_rows_prefix_hist(ctx, rows.b.p, 2 * n, rw, kPrefixBases, sample, hist.p));
  BG_TRY(bg_cover_query_begin(ctx, rows.b.p, 2 * n, W, k, &H.query));
  BG_TRY(comm_all_reduce_sum_u64(c, hist.p, bins));
  std::vector<uint64_t> gh(bins);
  BG_HIP(copy_d2h(ctx, gh.data(), hist.p, bins * 8));
  hist.reset();
  std::vector<uint64_t> pp = {uint64_t(kPrefixBases)};
  for (uint64_t s : choose_splitters(gh, P)) pp.push_back(s);
  ph.mark("cover_index_exchange");
  {
    Rows send, got;
    std::vector<uint64_t> counts;
    BG_TRY(partition(ctx, rows.b.p, rows.n, rw, BG_PART_PREFIX, P, pp.data(), &send, &counts));
    rows.reset();
    BG_TRY(a2a_v(c, ctx, send, counts, &got, nullptr));
    ph.mark("cover_index_build");
    BG_TRY(bg_cover_build(ctx, got.b.p, got.n, W, k, &H.cover));
    parts->push_back(std::move(got));  // the oriented reads a rank indexes are its j = 0 records
  }
  // the owners' first-K-mer filters on every rank: 32 bits per indexed read in 64-bit words, as long as the gathered filters stay
  // a small share of the device (they are the one structure that grows with the number of ranks: P x 2^log2w x 8 bytes)
  ph.mark("cover_filters");
  std::vector<uint64_t> owned;
  BG_TRY(gather_u64(c, {(*parts)[0].n}, &owned));
  uint64_t most = 1;
  for (uint64_t x : owned) most = std::max(most, x);
  int log2w = std::max(10, int(std::ceil(std::log2(double(most) / 2.0))));
  {
    size_t free_b = 0, total_b = 0;
    BG_HIP(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = total_b / 8;
    if (const char* e = getenv("BG_COVER_FILTER_BUDGET")) budget = strtoull(e, nullptr, 10);
    while (log2w > 10 && (uint64_t(P) << log2w) * 8 > budget) --log2w;  // fewer bits per read: more candidates to verify, the same records
  }
  DevBuf<uint64_t> filt;
  {
    DevBuf<uint64_t> bits;
    BG_TRY(bits.alloc(ctx, size_t(1) << log2w));
    BG_TRY(bg_cover_filter_build(ctx, H.cover, log2w, bits.p));
    BG_TRY(filt.alloc(ctx, size_t(P) << log2w));
    std::vector<uint64_t> by(P, (uint64_t(1) << log2w) * 8);
    BG_TRY(comm_all_gather_dev(c, bits.p, by.data(), filt.p));
  }
  const std::vector<int> hits_plan = parse_ints("BG_COVER_SCAN_HITS", "1,2,4,16,64");
  for (size_t rnd = 0;; ++rnd) {
    uint64_t pend = bg_cover_query_pending(H.query), tot = 0;
    BG_TRY(all_sum(c, pend, &tot));
    if (!tot) break;
    ph.mark("cover_scan");
    const uint32_t hits = uint32_t(hits_plan[std::min(rnd, hits_plan.size() - 1)]);
    const uint64_t cap = pend * hits;
    Rows cand, got;
    BG_TRY(cand.alloc(ctx, std::max<uint64_t>(cap, 1), rw));
    std::vector<uint64_t> counts(P, 0), cm;
    uint64_t n_c = 0;
    BG_TRY(bg_cover_query_scan(ctx, H.query, filt.p, log2w, hits, uint32_t(P), pp.data(), cand.b.p, cap, counts.data(), &n_c));
    cand.n = n_c;
    BG_TRY(a2a_v(c, ctx, cand, counts, &got, &cm, true));
    DevBuf<uint8_t> cv, ans;
    BG_TRY(cv.alloc(ctx, got.n));
    BG_TRY(bg_cover_test(ctx, H.cover, got.b.p, got.n, cv.p));
    got.reset();
    BG_TRY(ans.alloc(ctx, n_c));
    BG_TRY(reply(c, cv.p, 1, cm, ans.p));
    BG_TRY(bg_cover_query_scan_answers(ctx, H.query, cand.b.p, ans.p, n_c));
  }
  filt.reset();
  bg_cover_free(H.cover);
  H.cover = nullptr;
  // the records, in suffix windows: one window travels (the communicator's side stream) while the next is generated
  std::vector<int> wins = parse_ints("BG_COVER_RECORD_WINDOWS", "1,3,5,9,17,33,256");
  if (wins.back() < 256) wins.push_back(256);
  Rows fly_send, fly_recv;
  bool flying = false;
  auto land = [&]() -> int {
    if (!flying) return BG_OK;
    flying = false;
    BG_TRY(comm_exchange_wait(c));
    fly_send.reset();
    parts->push_back(std::move(fly_recv));
    return BG_OK;
  };
  // No host collective stands between the launch of a window's exchange and the generation of the next window (a host
  // collective waits for the exchange in flight first, comm.hip): how many reads are still live at the NEXT window, on
  // every rank, travels with the counts gather of this one.
  uint64_t live = 0, tot = 0;
  BG_TRY(bg_cover_query_records_live(ctx, H.query, wins[0], &live));
  BG_TRY(all_sum(c, live, &tot));
  for (size_t wi = 0; wi + 1 < wins.size(); ++wi) {
    const int j_lo = wins[wi], j_hi = wins[wi + 1];
    ph.mark("cover_records");
    if (!tot) break;
    const uint64_t cap = live * uint64_t(j_hi - j_lo);
    Rows send;
    BG_TRY(send.alloc(ctx, std::max<uint64_t>(cap, 1), rw));
    std::vector<uint64_t> counts(P, 0), m;
    uint64_t n_r = 0;
    BG_TRY(bg_cover_query_records_split(ctx, H.query, j_lo, j_hi, uint32_t(P), pp.data(), send.b.p, cap, counts.data(), &n_r));
    send.n = n_r;
    send.b.shrink(std::max<uint64_t>(n_r, 1) * rw);
    BG_TRY(land());
    uint64_t live_next = 0;
    if (wi + 2 < wins.size()) BG_TRY(bg_cover_query_records_live(ctx, H.query, j_hi, &live_next));
    {
      std::vector<uint64_t> mine(counts);
      mine.push_back(live_next);
      BG_TRY(gather_u64(c, mine, &m));  // (P + 1 values per rank: its counts, then its live reads at the next window)
    }
    live = live_next;
    tot = 0;
    for (int s = 0; s < P; ++s) tot += m[size_t(s) * (P + 1) + P];
    if (comm_is_shortcut(c)) {
      parts->push_back(std::move(send));
    } else {
      const int me = comm_rank(c);
      std::vector<uint64_t> rr(P);
      uint64_t in = 0;
      for (int s = 0; s < P; ++s) { rr[s] = m[size_t(s) * (P + 1) + me]; in += rr[s]; }
      BG_TRY(fly_recv.alloc(ctx, in, rw));
      fly_send = std::move(send);
      BG_TRY(comm_exchange(c, fly_send.b.p, counts.data(), rr.data(), size_t(rw) * 8, fly_recv.b.p, true));
      flying = true;
    }
  }
  BG_TRY(land());
  if (first_covered) {  // the reads whose whole-read record is their last: look-up rows for the tails phase
    uint64_t m = 0, got_m = 0;
    BG_TRY(bg_cover_query_first_covered_rows(ctx, H.query, nullptr, 0, &m));
    BG_TRY(first_covered->alloc(ctx, std::max<uint64_t>(m, 1), rw));
    first_covered->n = m;
    BG_TRY(bg_cover_query_first_covered_rows(ctx, H.query, first_covered->b.p, m, &got_m));
  }
  bg_cover_query_free(H.query);
  H.query = nullptr;
  uint64_t mine = 0;
  for (auto it = parts->begin(); it != parts->end();) {
    if (!it->n) { it = parts->erase(it); continue; }
    mine += it->n;
    ++it;
  }
  return all_sum(c, mine, n_records);
}

// ---- 8: the sample's readmap with one owner per structure (sharded.py::sharded_readmap; make_readmap.cpp:120-392)
int sharded_readmap(bg_comm* c, bg_ctx* ctx, Handles& H, const std::vector<uint64_t>& part_n, DevBuf<uint64_t>& words, DevBuf<uint16_t>& lens,
                    uint64_t n_reads, uint32_t max_len, bool paired, int W, bg_sharded* out, Phases& ph) {
  const int P = comm_world(c), me = comm_rank(c);
  const int rw = W + 1;
  std::vector<uint64_t> off(P + 1, 0);
  for (int r = 0; r < P; ++r) off[r + 1] = off[r] + part_n[r];
  ph.mark("readmap_lookup");
  // 1. entry ids: oriented rows (tagged with their index) to the shard their key falls into
  std::vector<uint64_t> fi(rw), la(rw), kb;
  BG_TRY(bg_seqset_boundary_rows(ctx, H.shard, fi.data(), la.data()));
  BG_TRY(gather_u64(c, la, &kb));
  BG_TRY(bg_reads_import(ctx, n_reads, W, max_len, words.p, lens.p, &H.rm_reads));
  words.reset();
  lens.reset();
  if (paired && (n_reads & 1)) { set_error("bg_create_sharded: a rank holds an odd number of reads of a paired sample"); return BG_EINVAL; }
  const uint64_t n_or = 2 * n_reads;
  Rows sent, got;
  std::vector<uint64_t> counts, cm;
  {
    Rows rows;
    BG_TRY(rows.alloc(ctx, n_or, rw));
    BG_TRY(bg_oriented_rows(ctx, H.rm_reads, W, rows.b.p));
    if (n_or) k_tag_rows<<<cdiv(n_or, 256), 256, 0, ctx->stream>>>(rows.b.p, n_or, rw);
    BG_LAUNCH_CHECK();
    // a dropped read (length 0) needs no look-up but keeps its place: it travels to the first shard and comes back as ~0
    BG_TRY(partition(ctx, rows.b.p, n_or, rw, BG_PART_KEY_BOUNDS, P, kb.data(), &sent, &counts));
  }
  BG_TRY(a2a_v(c, ctx, sent, counts, &got, &cm, true));
  DevBuf<uint64_t> eid;
  {
    DevBuf<uint64_t> ent, ans;
    BG_TRY(ent.alloc(ctx, got.n));
    BG_TRY(bg_seqset_find_rows(ctx, H.shard, got.b.p, got.n, W, off[me], ent.p));
    got.reset();
    BG_TRY(ans.alloc(ctx, sent.n));
    BG_TRY(reply(c, ent.p, 8, cm, ans.p));
    ph.mark("readmap_rows");
    BG_TRY(eid.alloc(ctx, std::max<uint64_t>(n_or, 1)));
    BG_HIP(hipMemsetAsync(eid.p, 0xff, std::max<uint64_t>(n_or, 1) * 8, ctx->stream));
    if (sent.n) k_scatter_by_tag<<<cdiv(sent.n, 256), 256, 0, ctx->stream>>>(sent.b.p, sent.n, rw, ans.p, eid.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  sent.reset();
  // 2. rows to the owner of their entry; sorted there
  {
    const uint64_t n_slots = paired ? (n_reads / 2) * 4 : n_reads * 2;
    Rows rows, send, rgot;
    BG_TRY(rows.alloc(ctx, std::max<uint64_t>(n_slots, 1), 3));
    uint64_t m = 0;
    BG_TRY(bg_readmap_rows(ctx, H.rm_reads, paired ? 1 : 0, eid.p, uint32_t(P), off.data(), rows.b.p, n_slots, &m));
    bg_reads_free(H.rm_reads);
    H.rm_reads = nullptr;
    eid.reset();
    BG_TRY(partition(ctx, rows.b.p, m, 3, BG_PART_ROW_DEST, P, nullptr, &send, &counts));
    rows.reset();
    BG_TRY(a2a_v(c, ctx, send, counts, &rgot, nullptr));
    BG_TRY(bg_readmap_shard_build(ctx, rgot.b.p, rgot.n, &H.rsh));
  }
  std::vector<uint64_t> sizes;
  BG_TRY(gather_u64(c, {bg_readmap_shard_size(H.rsh)}, &sizes));
  std::vector<uint64_t> base(P + 1, 0);
  for (int r = 0; r < P; ++r) base[r + 1] = base[r] + sizes[r];
  const uint64_t n_total = base[P];
  if (n_total >= (uint64_t(1) << 32)) { set_error("bg_create_sharded: more than 2^32 read ids (readmap.h holds them in 32 bits)"); return BG_EINVAL; }
  ph.mark("readmap_mate_loops");
  // 3. the claim hops (make_readmap.cpp:283-392)
  Rows claims;
  {
    const uint64_t n = sizes[me];
    BG_TRY(claims.alloc(ctx, std::max<uint64_t>(n, 1), 4));
    uint64_t m = 0;
    BG_TRY(bg_readmap_shard_first_claims(ctx, H.rsh, base[me], uint32_t(P), off.data(), claims.b.p, n, &m));
    claims.n = m;
  }
  for (int hop = 1; hop <= 3; ++hop) {
    uint64_t tot = 0;
    BG_TRY(all_sum(c, claims.n, &tot));
    if (!tot) break;
    Rows csend, cgot, nx;
    std::vector<uint64_t> ccm;
    BG_TRY(partition(ctx, claims.b.p, claims.n, 4, BG_PART_ROW_DEST, P, nullptr, &csend, &counts));
    claims.reset();
    BG_TRY(a2a_v(c, ctx, csend, counts, &cgot, &ccm, true));
    DevBuf<uint64_t> rep, home;
    const uint64_t n = cgot.n;
    BG_TRY(rep.alloc(ctx, n));
    BG_TRY(nx.alloc(ctx, std::max<uint64_t>(n, 1), 4));
    uint64_t m = 0;
    BG_TRY(bg_readmap_shard_apply_claims(ctx, H.rsh, base[me], hop, cgot.b.p, n, uint32_t(P), off.data(), rep.p, nx.b.p, &m));
    nx.n = m;
    cgot.reset();
    BG_TRY(home.alloc(ctx, csend.n));
    BG_TRY(reply(c, rep.p, 8, ccm, home.p));
    BG_TRY(bg_readmap_shard_take_replies(ctx, H.rsh, base[me], csend.b.p, home.p, csend.n));
    claims = std::move(nx);
  }
  ph.mark("readmap_columns");
  std::vector<uint64_t> ml;
  BG_TRY(gather_u64(c, {uint64_t(max_len)}, &ml));
  uint32_t mlen = 0;
  for (uint64_t x : ml) mlen = std::max<uint32_t>(mlen, uint32_t(x));
  const uint64_t n = sizes[me];
  BG_TRY(out->rm_entry.alloc(ctx, n));
  BG_TRY(out->rm_len.alloc(ctx, n));
  BG_TRY(out->rm_loop.alloc(ctx, n));
  BG_TRY(out->rm_fwd.alloc(ctx, n));
  BG_TRY(bg_readmap_shard_columns(ctx, H.rsh, out->rm_entry.p, out->rm_len.p, out->rm_loop.p, out->rm_fwd.p));
  bg_readmap_shard_free(H.rsh);
  H.rsh = nullptr;
  out->has_readmap = true;
  out->rm_first = base[me];
  out->rm_n = n;
  out->rm_total = n_total;
  out->rm_entries = off[P];
  out->rm_max_read_len = mlen;
  return BG_OK;
}

// ---- 7: 512-aligned slices (sharded.py::_aligned_slices): container members are addressed in 64-bit words and 512-bit groups;
// every rank passes its last < 512 entries on (sizes, shared, one prev-mask byte each) so that rank r holds [B_r, B_r+1),
// B_r = its global start rounded down to 512.
int aligned_slice(bg_comm* c, bg_ctx* ctx, Handles& H, const std::vector<uint64_t>& part_n, uint32_t max_len, bg_sharded* out) {
  const int P = comm_world(c), me = comm_rank(c);
  std::vector<uint64_t> off(P + 1, 0);
  for (int r = 0; r < P; ++r) off[r + 1] = off[r] + part_n[r];
  const uint64_t n_total = off[P];
  std::vector<uint64_t> B(P + 1);
  for (int r = 0; r < P; ++r) B[r] = (off[r] / 512) * 512;
  B[P] = n_total;
  const uint64_t n = part_n[me];
  DevBuf<uint8_t> cols;  // three planes of n: sizes, shared, mask
  BG_TRY(cols.alloc(ctx, 3 * std::max<uint64_t>(n, 1)));
  BG_TRY(bg_seqset_slice_masks(ctx, H.shard, cols.p, cols.p + n, cols.p + 2 * n));
  constexpr uint64_t kTail = 511;
  const uint64_t nt = std::min(n, kTail);
  std::vector<uint8_t> mine(8 + 3 * kTail, 0), all;
  memcpy(mine.data(), &nt, 8);
  for (int pl = 0; pl < 3; ++pl)
    if (nt) BG_HIP(copy_d2h(ctx, mine.data() + 8 + pl * kTail, cols.p + pl * n + (n - nt), nt));
  all.resize(mine.size() * P);
  BG_TRY(comm_all_gather_host(c, mine.data(), mine.size(), all.data()));
  // entries [B_me, off_me) sit in the tails of the ranks before me
  std::vector<uint8_t> pre[3];
  for (int q = 0; q < me; ++q) {
    const uint8_t* slot = all.data() + size_t(q) * mine.size();
    uint64_t tq = 0;
    memcpy(&tq, slot, 8);
    const uint64_t start = off[q + 1] - tq;
    const uint64_t lo = B[me] > start ? B[me] - start : 0;
    for (int pl = 0; pl < 3; ++pl)
      for (uint64_t i = lo; i < tq; ++i) pre[pl].push_back(slot[8 + pl * kTail + i]);
  }
  const uint64_t m = B[me + 1] - B[me];
  const int64_t n_own = me + 1 < P ? int64_t(n) - int64_t(off[me + 1] - B[me + 1]) : int64_t(n);
  // n_own < 0: this rank's entries and some of the inherited ones belong to the next rank
  const uint64_t from_pre = std::min<uint64_t>(pre[0].size(), m);
  const uint64_t from_own = m - from_pre;
  if ((n_own >= 0 && (pre[0].size() + uint64_t(n_own) != m)) || from_own > n) {
    set_error("bg_create_sharded: slice arithmetic (rank %d: %llu inherited + %lld own != %llu)", me, (unsigned long long)pre[0].size(), (long long)n_own,
              (unsigned long long)m);
    return BG_EINTERNAL;
  }
  DevBuf<uint8_t> allc;
  BG_TRY(allc.alloc(ctx, 3 * std::max<uint64_t>(m, 1)));
  for (int pl = 0; pl < 3; ++pl) {
    if (from_pre) BG_HIP(copy_h2d(ctx, allc.p + pl * m, pre[pl].data(), from_pre));
    if (from_own) BG_HIP(hipMemcpyAsync(allc.p + pl * m + from_pre, cols.p + pl * n, from_own, hipMemcpyDeviceToDevice, ctx->stream));
  }
  BG_TRY(bg_seqset_slice_build(ctx, m, allc.p, allc.p + m, allc.p + 2 * m, n_total ? max_len : 0, &out->slice));
  uint64_t pc[4];
  BG_TRY(bg_seqset_slice_popcounts(out->slice, pc));
  std::vector<uint64_t> allp;
  BG_TRY(gather_u64(c, {pc[0], pc[1], pc[2], pc[3]}, &allp));
  out->fixed[0] = 0;
  for (int b = 0; b < 4; ++b) {
    uint64_t tot = 0, before = 0;
    for (int q = 0; q < P; ++q) { tot += allp[size_t(q) * 4 + b]; if (q < me) before += allp[size_t(q) * 4 + b]; }
    out->fixed[b + 1] = out->fixed[b] + tot;
    out->base_pop[b] = before;
  }
  if (out->fixed[4] != n_total) {  // seqset.cpp:123-126
    set_error("Invalid seqset in finalize: %llu != %llu", (unsigned long long)out->fixed[4], (unsigned long long)n_total);
    return BG_EINTERNAL;
  }
  out->first = B[me];
  out->n_total = n_total;
  out->max_len = n_total ? max_len : 0;
  out->is_last = me == P - 1;
  return BG_OK;
}

const uint64_t* next_nonempty(const std::vector<uint64_t>& rows, int rw, int P, int r, int step) {
  for (int q = r + step; q >= 0 && q < P; q += step)
    if (rows[size_t(q) * rw + rw - 1] != ~uint64_t(0)) return rows.data() + size_t(q) * rw;
  return nullptr;
}


// The back half of a sharded build, from the rows a rank received to its shard with prev bits (sharded.py::_records_to_output):
// entries of the shard, the cross-shard prefix drop and shared[0], every entry's tail to the shard that holds it.  What
// bg_create_sharded and bg_merge_sharded share.
int shard_from_parts(bg_comm* c, bg_ctx* ctx, Handles& H, std::vector<Rows>& parts, int W, int rank, uint64_t* tails_in_full, Phases& ph,
                     Rows* first_covered /* a create: its rows carry the emission's flags; the reads whose whole-read record is their last */) {
  const bool rows_carry_flags = first_covered != nullptr;
  const int P = comm_world(c);
  // ---- entries of this shard
  ph.mark("entries_from_records");
  {
    std::vector<const uint64_t*> ptrs;
    std::vector<uint64_t> cnts;
    for (auto& p : parts) { ptrs.push_back(p.b.p); cnts.push_back(p.n); }
    if (ptrs.empty()) { ptrs.push_back(nullptr); cnts.push_back(0); }
    BG_TRY(bg_records_from_row_parts(ctx, uint32_t(parts.size()), ptrs.data(), cnts.data(), W, &H.records));
    parts.clear();  // the rows are in the library now
    bg_records_set_meta(H.records, rows_carry_flags ? 1 : 0);
    bg_build_stats bst;
    BG_TRY(bg_entries_from_records(ctx, H.records, &H.shard, &bst));
    bg_records_free(H.records);
    H.records = nullptr;
  }

  // ---- 5: shard boundaries (cross-shard prefix drop, shared[0] across the cut)
  ph.mark("boundaries");
  const int rw = W + 1;
  std::vector<uint64_t> fi(rw), la(rw), firsts, lasts;
  BG_TRY(bg_seqset_boundary_rows(ctx, H.shard, fi.data(), la.data()));
  BG_TRY(gather_u64(c, fi, &firsts));
  if (const uint64_t* nx = next_nonempty(firsts, rw, P, rank, +1)) BG_TRY(bg_seqset_fix_boundary(ctx, H.shard, nullptr, nx));
  BG_TRY(bg_seqset_boundary_rows(ctx, H.shard, fi.data(), la.data()));
  BG_TRY(gather_u64(c, la, &lasts));
  if (const uint64_t* pv = next_nonempty(lasts, rw, P, rank, -1)) BG_TRY(bg_seqset_fix_boundary(ctx, H.shard, pv, nullptr));

  // ---- 6: prev bits: an entry's tail goes to the shard holding lower_bound(tail); short form first.  A shard of a create has
  // most of its bits already (set where its records met): first the look-ups that decide which further entries must send
  // their tail (base + a record dropped as a proper prefix: if that is an entry where it sorts, its tail is that record),
  // then only the flagged entries' tails travel.
  const bool local_prev = bg_seqset_local_prev(H.shard) != 0;
  if (local_prev) {
    ph.mark("tails_lookups");
    const uint64_t* cr = nullptr;
    uint64_t n_c = 0;
    BG_TRY(bg_seqset_cond_rows(H.shard, &cr, &n_c));
    const uint64_t n_f = first_covered ? first_covered->n : 0;
    Rows all, sent, got;
    BG_TRY(all.alloc(ctx, std::max<uint64_t>(n_c + n_f, 1), rw));
    all.n = n_c + n_f;
    if (n_c) BG_HIP(hipMemcpyAsync(all.b.p, cr, n_c * rw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (n_f) BG_HIP(hipMemcpyAsync(all.b.p + n_c * rw, first_covered->b.p, n_f * rw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (first_covered) first_covered->reset();
    std::vector<uint64_t> counts;
    BG_TRY(partition(ctx, all.b.p, all.n, rw, BG_PART_KEY_BOUNDS, P, lasts.data(), &sent, &counts));
    all.reset();
    BG_TRY(a2a_v(c, ctx, sent, counts, &got, nullptr));
    BG_TRY(bg_seqset_mark_need(ctx, H.shard, got.b.p, got.n));
  }
  ph.mark("tails");
  {
    uint64_t n = bg_seqset_size(H.shard);
    Rows rows, shortr, got;
    std::vector<uint64_t> counts, tcm;
    if (local_prev) {
      BG_TRY(bg_seqset_tail_rows2_needed(ctx, H.shard, uint32_t(P), lasts.data(), nullptr, 0, &n));
      BG_TRY(rows.alloc(ctx, std::max<uint64_t>(n, 1), 3));
      rows.n = n;
      uint64_t m2 = 0;
      BG_TRY(bg_seqset_tail_rows2_needed(ctx, H.shard, uint32_t(P), lasts.data(), rows.b.p, n, &m2));
    } else {
      BG_TRY(rows.alloc(ctx, n, 3));
      BG_TRY(bg_seqset_tail_rows2(ctx, H.shard, uint32_t(P), lasts.data(), rows.b.p));
    }
    BG_TRY(partition(ctx, rows.b.p, n, 3, BG_PART_ROW_DEST, P, nullptr, &shortr, &counts));
    rows.reset();
    BG_TRY(a2a_v(c, ctx, shortr, counts, &got, &tcm, true));
    ph.mark("apply_tails");
    DevBuf<uint8_t> need, needs;
    BG_TRY(need.alloc(ctx, got.n));
    BG_TRY(bg_seqset_apply_tails2(ctx, H.shard, got.b.p, got.n, need.p));
    got.reset();
    BG_TRY(needs.alloc(ctx, shortr.n));
    BG_TRY(reply(c, need.p, 1, tcm, needs.p));
    need.reset();
    ph.mark("tails_full");
    uint64_t cap = 0, m = 0, tot = 0;
    BG_TRY(sum_bytes(ctx, needs.p, shortr.n, &cap));
    Rows t, full, fgot;
    BG_TRY(t.alloc(ctx, std::max<uint64_t>(cap, 1), rw));
    BG_TRY(bg_seqset_tail_rows_full(ctx, H.shard, shortr.b.p, needs.p, shortr.n, t.b.p, cap, &m));
    shortr.reset();
    needs.reset();
    BG_TRY(partition(ctx, t.b.p, m, rw, BG_PART_ROW_DEST, P, nullptr, &full, &counts));
    t.reset();
    BG_TRY(all_sum(c, m, &tot));
    *tails_in_full = tot;
    if (tot) {
      BG_TRY(a2a_v(c, ctx, full, counts, &fgot, nullptr));
      BG_TRY(bg_seqset_apply_tails(ctx, H.shard, fgot.b.p, fgot.n));
    }
  }
  return BG_OK;
}

}  // namespace
}  // namespace bg

using namespace bg;

static int create_sharded_impl(bg_ctx* ctx, bg_comm* comm, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                               const bg_create_params* params, int readmap, bg_sharded** out, bg_sharded_stats* stats);

// Errors: what a rank can know about its own input is checked before the first collective and AGREED on (one host
// all-gather of the ranks' verdicts), so a bad input takes every rank out together with the offending rank's message
// instead of leaving the others in a collective.  A rank that fails later (out of memory, an internal error) lands what it
// has in flight before its buffers go back to the arena and tells the ranks of a shared-memory communicator (comm_fail);
// over RCCL its peers stay in their next send / receive -- `bgbinary create --gpus N` ends them, a library caller must.
extern "C" int bg_create_sharded(bg_ctx* ctx, bg_comm* comm, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads, const bg_create_params* params,
                                 int readmap, bg_sharded** out, bg_sharded_stats* stats) {
  if (!ctx || !comm || !out || (n_reads && (!d_seq || !d_offs)) || readmap < 0 || readmap > 2) {
    set_error("bg_create_sharded: bad arguments");
    return BG_EINVAL;
  }
  {
    uint64_t verdict = BG_OK;
    if (readmap == 2 && (n_reads & 1)) verdict = uint64_t(-BG_EINVAL);
    std::vector<uint64_t> all;
    BG_TRY(gather_u64(comm, {verdict}, &all));
    for (size_t r = 0; r < all.size(); ++r)
      if (all[r]) {
        set_error("bg_create_sharded: rank %zu holds an odd number of reads of a paired sample", r);
        return BG_EINVAL;
      }
  }
  const int rc = create_sharded_impl(ctx, comm, d_seq, d_offs, n_reads, params, readmap, out, stats);
  if (rc != BG_OK) comm_fail(comm);
  return rc;
}

static int create_sharded_impl(bg_ctx* ctx, bg_comm* comm, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                               const bg_create_params* params, int readmap, bg_sharded** out, bg_sharded_stats* stats) {
  bg_create_params dflt;
  bg_create_params_default(&dflt);
  const bg_create_params* prm = params ? params : &dflt;
  const int k = prm->kmer_size;
  if (!bg_skm_supported(k)) {
    set_error("bg_create_sharded: the routed build takes k-mer sizes with a 12-base minimiser (14..31), not %d (sharded.py keeps the replicated mode)", k);
    return BG_EINVAL;
  }
  BG_HIP(hipSetDevice(ctx->device));
  const int P = comm_world(comm);
  Handles H;
  Phases ph(ctx, comm_rank(comm));
  std::unique_ptr<bg_sharded> res(new bg_sharded);
  res->ctx = ctx;
  res->rank = comm_rank(comm);
  res->world = P;
  bg_sharded_stats st;
  memset(&st, 0, sizeof st);

  ph.mark("pack");
  BG_TRY(bg_pack_reads(ctx, d_seq, d_offs, n_reads, &H.reads));
  const uint64_t my_bases = bg_reads_total_bases(H.reads);

  // ---- 1: windows -> owner ranks as super-k-mer records
  ph.mark("kmer_extract_partition");
  Rows skm_recs;
  DevBuf<uint64_t> skm_org;
  DevBuf<uint32_t> skm_bits;
  uint64_t n_windows = 0, my_distinct = 0, my_kept = 0;
  bool gathered_set = false;
  {
    std::vector<uint64_t> counts, kcm;
    BG_TRY(skm_extract(ctx, H.reads, k, P, 0.385, &skm_recs, &skm_org, &counts, &n_windows));
    ph.mark("kmer_all_to_all");
    Rows keys;
    BG_TRY(a2a_v(comm, ctx, skm_recs, counts, &keys, &kcm, true));
    ph.mark("kmer_count");
    DevBuf<uint32_t> bits;
    BG_TRY(bits.alloc(ctx, keys.n));
    BG_TRY(bg_skm_count(ctx, keys.b.p, keys.n, k, prm->min_kmer_count, &H.owned, bits.p));
    my_distinct = bg_kmerset_distinct(H.owned);
    my_kept = bg_kmerset_size(H.owned);
    // Where the kept k-mers live for the correction.  They are the one structure of the build that is small beside the reads
    // (8 bytes a k-mer: 0.5 GB for a 30x chr20, 25 GB for a 30x human): when all of them fit in a sixteenth of the device,
    // every rank takes a copy (one all-gather) and corrects its reads with no further exchange, as the single-GPU build does
    // -- 288 GB of HBM is what makes that the cheap way.  Otherwise (BG_SHARD_KMERSET=routed forces it) they stay with
    // their owners and the correction asks them (correct_routed).
    // The verdict is a collective one: every rank sees every rank's device size and switch and derives the same answer (the
    // smallest device bounds the copy; a rank that forces "routed" wins over one that forces "gathered") -- ranks deciding each on
    // its own hipMemGetInfo / getenv would enter different collectives.
    size_t free_b = 0, total_b = 0;
    BG_HIP(hipMemGetInfo(&free_b, &total_b));
    const char* where = getenv("BG_SHARD_KMERSET");
    const uint64_t my_force = where && !strcmp(where, "routed") ? 1 : where && !strcmp(where, "gathered") ? 2 : 0;
    std::vector<uint64_t> verdict_in;
    BG_TRY(gather_u64(comm, {my_kept, uint64_t(total_b), my_force}, &verdict_in));
    std::vector<uint64_t> kept_all(P);
    uint64_t total_kept = 0, min_total_b = ~0ull;
    bool forced_routed = false, forced_gathered = false;
    for (int r = 0; r < P; ++r) {
      kept_all[r] = verdict_in[size_t(r) * 3];
      total_kept += kept_all[r];
      min_total_b = std::min(min_total_b, verdict_in[size_t(r) * 3 + 1]);
      forced_routed |= verdict_in[size_t(r) * 3 + 2] == 1;
      forced_gathered |= verdict_in[size_t(r) * 3 + 2] == 2;
    }
    gathered_set = !forced_routed && (forced_gathered || total_kept * 8 * 4 <= min_total_b / 16);  // (keys + hash table + flags)
    if (!gathered_set) {
      // the owner keeps its k-mers (hash set) and settles the records its counting tables gave up on
      BG_TRY(bg_kmerset_build_hash(ctx, H.owned));
      BG_TRY(bg_skm_lookup(ctx, H.owned, keys.b.p, keys.n, bits.p, 1));
    }
    keys.reset();
    BG_TRY(skm_bits.alloc(ctx, skm_recs.n));
    BG_TRY(reply(comm, bits.p, 4, kcm, skm_bits.p));  // 1b: the presence bits of every record go back to the rank it came from
    if (gathered_set) {
      ph.mark("kmerset_all_gather");
      DevBuf<uint64_t> mine, all;
      BG_TRY(mine.alloc(ctx, my_kept));
      BG_TRY(all.alloc(ctx, total_kept));
      if (my_kept) BG_TRY(bg_kmerset_export(ctx, H.owned, mine.p));
      bg_kmerset_free(H.owned);
      H.owned = nullptr;
      std::vector<uint64_t> bytes(kept_all);
      for (uint64_t& v : bytes) v *= 8;
      BG_TRY(bg_comm_all_gather(comm, mine.p, bytes.data(), all.p));
      mine.reset();
      // (records whose bucket overflowed the owner's counting table came back "unknown": attach_presence looks them up here)
      BG_TRY(bg_kmerset_from_kmers(ctx, all.p, total_kept, k, nullptr, &H.owned));  // (no starts-a-read flags: the build never reads them)
      all.reset();
      BG_TRY(bg_kmerset_attach_presence(ctx, H.owned, H.reads, skm_recs.b.p, skm_org.p, skm_bits.p, skm_recs.n));
    }
  }

  // ---- 2: correction -- local over the gathered set, or routed
  bg_correct_stats cst;
  memset(&cst, 0, sizeof cst);
  if (gathered_set) {
    ph.mark("correct");
    BG_TRY(bg_correct_reads(ctx, H.reads, H.owned, &prm->correct, &H.corrected, &cst));
  } else {
    BG_TRY(correct_routed(comm, ctx, H, k, prm, skm_recs, skm_org, skm_bits, &cst, &st.correct_steps, &st.correct_probes, ph));
  }
  bg_reads_free(H.reads);
  H.reads = nullptr;
  bg_kmerset_free(H.owned);
  H.owned = nullptr;
  skm_recs.reset();
  skm_org.reset();
  skm_bits.reset();

  std::vector<uint64_t> g;
  BG_TRY(gather_u64(comm, {n_reads, my_bases, n_windows, my_distinct, my_kept, cst.corrected_reads, cst.corrected_bases, uint64_t(bg_reads_max_len(H.corrected)),
                           st.correct_probes},
                    &g));
  uint32_t max_len = 0;
  st.correct_probes = 0;
  for (int r = 0; r < P; ++r) {
    const uint64_t* v = g.data() + size_t(r) * 9;
    st.build.n_reads += v[0]; st.build.n_bases += v[1]; st.build.n_kmers += v[2]; st.build.n_distinct_kmers += v[3];
    st.build.n_kept_kmers += v[4]; st.build.n_corrected_reads += v[5]; st.build.n_corrected_bases += v[6];
    max_len = std::max<uint32_t>(max_len, uint32_t(v[7]));
    st.correct_probes += v[8];
  }
  const int W = std::max(1, int((max_len + 31) / 32));
  st.build.max_len = max_len;

  // this rank's corrected reads, packed: what the readmap is built from once the entries exist
  DevBuf<uint64_t> kept_words;
  DevBuf<uint16_t> kept_lens;
  const uint64_t n_corr = bg_reads_count(H.corrected);
  if (readmap) {
    BG_TRY(kept_words.alloc(ctx, n_corr * uint64_t(W)));
    BG_TRY(kept_lens.alloc(ctx, n_corr));
    BG_TRY(bg_reads_export(ctx, H.corrected, W, kept_words.p, kept_lens.p));
  }

  // ---- 3 + 4: records
  std::vector<Rows> parts;
  Rows first_covered;
  BG_TRY(records_sharded_cover(comm, ctx, H, W, k, &parts, &st.build.n_records, ph, &first_covered));

  // ---- entries of this shard, boundaries, prev bits
  BG_TRY(shard_from_parts(comm, ctx, H, parts, W, res->rank, &st.tails_sent_in_full, ph, &first_covered));
  std::vector<uint64_t> part_n;
  BG_TRY(gather_u64(comm, {bg_seqset_size(H.shard)}, &part_n));
  for (uint64_t x : part_n) st.build.n_entries += x;
  st.shard_entries = part_n[res->rank];

  // ---- 8: the sample's readmap while the keys are resident
  if (readmap) BG_TRY(sharded_readmap(comm, ctx, H, part_n, kept_words, kept_lens, n_corr, max_len, readmap == 2, W, res.get(), ph));

  // ---- 7: slices
  ph.mark("slices");
  BG_TRY(aligned_slice(comm, ctx, H, part_n, max_len, res.get()));
  bg_seqset_free(H.shard);
  H.shard = nullptr;
  BG_HIP(hipStreamSynchronize(ctx->stream));
  ph.done();
  st.first_entry = res->first;
  st.slice_entries = bg_seqset_size(res->slice);
  st.n_read_ids = res->rm_total;
  st.first_read_id = res->rm_first;
  st.slice_read_ids = res->rm_n;
  if (stats) *stats = st;
  *out = res.release();
  return BG_OK;
}

__global__ void k_or_last_word(uint64_t* __restrict__ rows, uint64_t n, int rw, uint64_t tag) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) rows[i * rw + rw - 1] |= tag;
}

static int merge_sharded_impl(bg_ctx* ctx, bg_comm* comm, uint32_t n_in, const bg_seqset* const* in, bg_sharded** out,
                              bg_sharded_stats* stats, uint64_t* const* d_maps) {
  const int P = comm_world(comm), me = comm_rank(comm);
  Handles H;
  Phases ph(ctx, me);
  std::unique_ptr<bg_sharded> res(new bg_sharded);
  res->ctx = ctx;
  res->rank = me;
  res->world = P;
  bg_sharded_stats st;
  memset(&st, 0, sizeof st);
  uint32_t max_len = 0;
  uint64_t total = 0;
  std::vector<uint64_t> lo(n_in), cnt(n_in);
  for (uint32_t i = 0; i < n_in; ++i) {
    max_len = std::max(max_len, in[i]->max_len);  // seqset_merger.cpp:57-60: the longest input sets the key width
    const uint64_t n = in[i]->n;
    lo[i] = n / uint64_t(P) * uint64_t(me) + std::min<uint64_t>(n % uint64_t(P), uint64_t(me));  // stripes of n / P, the first n % P one longer
    cnt[i] = n / uint64_t(P) + (uint64_t(me) < n % uint64_t(P) ? 1 : 0);
    total += cnt[i];
  }
  const int W = std::max(1u, (max_len + 31) / 32), rw = W + 1;
  bool want_maps = false;
  for (uint32_t i = 0; d_maps && i < n_in; ++i) want_maps = want_maps || d_maps[i];

  // ---- this rank's stripes as rows; a copy with every row's number in it for the maps
  ph.mark("merge_rows");
  Rows rows, look;
  BG_TRY(rows.alloc(ctx, std::max<uint64_t>(total, 1), rw));
  rows.n = total;
  {
    uint64_t at = 0;
    for (uint32_t i = 0; i < n_in; ++i) {
      BG_TRY(seqset_stripe_rows(ctx, in[i], lo[i], cnt[i], W, rows.b.p + at * rw));
      at += cnt[i];
    }
  }
  if (want_maps) {
    BG_TRY(look.alloc(ctx, std::max<uint64_t>(total, 1), rw));
    look.n = total;
    if (total) {
      BG_HIP(hipMemcpyAsync(look.b.p, rows.b.p, total * rw * 8, hipMemcpyDeviceToDevice, ctx->stream));
      k_tag_rows<<<cdiv(total, 256), 256, 0, ctx->stream>>>(look.b.p, total, rw);
      BG_LAUNCH_CHECK();
    }
  }
  {  // the input's number rides in the length word, whose low byte is read (sharded.py::sharded_merge)
    uint64_t at = 0;
    for (uint32_t i = 0; i < n_in; ++i) {
      if (cnt[i] && i) {
        k_or_last_word<<<cdiv(cnt[i], 256), 256, 0, ctx->stream>>>(rows.b.p + at * rw, cnt[i], rw, uint64_t(i) << 8);
        BG_LAUNCH_CHECK();
      }
      at += cnt[i];
    }
  }

  // ---- common splitters, the records exchange (part_repo.cpp:32-45, part_counts.h:8-45)
  ph.mark("merge_partition");
  std::vector<Rows> parts;
  {
    const size_t bins = size_t(1) << (2 * kPrefixBases);
    DevBuf<uint64_t> hist;
    BG_TRY(hist.alloc(ctx, bins));
    uint64_t sample = 8;
    if (const char* e = getenv("BG_PREFIX_SAMPLE")) sample = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    BG_TRY(bg_rows_prefix_hist(ctx, rows.b.p, total, rw, kPrefixBases, sample, hist.p));
    BG_TRY(comm_all_reduce_sum_u64(comm, hist.p, bins));
    std::vector<uint64_t> gh(bins);
    BG_HIP(copy_d2h(ctx, gh.data(), hist.p, bins * 8));
    std::vector<uint64_t> pp = {uint64_t(kPrefixBases)};
    for (uint64_t sp : choose_splitters(gh, P)) pp.push_back(sp);
    Rows send, got;
    std::vector<uint64_t> counts;
    BG_TRY(partition(ctx, rows.b.p, total, rw, BG_PART_PREFIX, P, pp.data(), &send, &counts));
    rows.reset();
    ph.mark("merge_all_to_all");
    BG_TRY(a2a_v(comm, ctx, send, counts, &got, nullptr));
    parts.push_back(std::move(got));
  }
  uint64_t n_in_total = 0;
  BG_TRY(all_sum(comm, total, &n_in_total));
  st.build.n_records = n_in_total;

  // ---- the shard
  BG_TRY(shard_from_parts(comm, ctx, H, parts, W, me, &st.tails_sent_in_full, ph, nullptr));  // (a merge's rows carry the input's number where a create's carry flags)
  std::vector<uint64_t> part_n;
  BG_TRY(gather_u64(comm, {bg_seqset_size(H.shard)}, &part_n));
  for (uint64_t x : part_n) st.build.n_entries += x;
  st.shard_entries = part_n[me];
  st.build.max_len = max_len;

  // ---- merge maps: every stripe entry to the shard its key falls into; the global id of the entry that holds it comes back
  if (want_maps) {
    ph.mark("merge_maps");
    std::vector<uint64_t> off(P + 1, 0);
    for (int r = 0; r < P; ++r) off[r + 1] = off[r] + part_n[r];
    std::vector<uint64_t> fi(rw), la(rw), kb;
    BG_TRY(bg_seqset_boundary_rows(ctx, H.shard, fi.data(), la.data()));
    BG_TRY(gather_u64(comm, la, &kb));
    Rows sent, got;
    std::vector<uint64_t> counts, cm;
    BG_TRY(partition(ctx, look.b.p, total, rw, BG_PART_KEY_BOUNDS, P, kb.data(), &sent, &counts));
    look.reset();
    BG_TRY(a2a_v(comm, ctx, sent, counts, &got, &cm, true));
    DevBuf<uint64_t> ent, ans, ids;
    BG_TRY(ent.alloc(ctx, got.n));
    BG_TRY(bg_seqset_find_rows(ctx, H.shard, got.b.p, got.n, W, off[me], ent.p));
    got.reset();
    BG_TRY(ans.alloc(ctx, sent.n));
    BG_TRY(reply(comm, ent.p, 8, cm, ans.p));
    BG_TRY(ids.alloc(ctx, std::max<uint64_t>(total, 1)));
    if (sent.n) k_scatter_by_tag<<<cdiv(sent.n, 256), 256, 0, ctx->stream>>>(sent.b.p, sent.n, rw, ans.p, ids.p);
    BG_LAUNCH_CHECK();
    uint64_t at = 0;
    for (uint32_t i = 0; i < n_in; ++i) {
      if (cnt[i] && d_maps[i]) BG_HIP(hipMemcpyAsync(d_maps[i], ids.p + at, cnt[i] * 8, hipMemcpyDeviceToDevice, ctx->stream));
      at += cnt[i];
    }
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }

  ph.mark("slices");
  BG_TRY(aligned_slice(comm, ctx, H, part_n, max_len, res.get()));
  bg_seqset_free(H.shard);
  H.shard = nullptr;
  BG_HIP(hipStreamSynchronize(ctx->stream));
  ph.done();
  st.first_entry = res->first;
  st.slice_entries = bg_seqset_size(res->slice);
  if (stats) *stats = st;
  *out = res.release();
  return BG_OK;
}

extern "C" int bg_merge_sharded(bg_ctx* ctx, bg_comm* comm, uint32_t n_in, const bg_seqset* const* in, bg_sharded** out,
                                bg_sharded_stats* stats, uint64_t* const* d_maps) {
  if (!ctx || !comm || !in || !out || !n_in) { set_error("bg_merge_sharded: null argument"); return BG_EINVAL; }
  for (uint32_t i = 0; i < n_in; ++i)
    if (!in[i] || (in[i]->n && !in[i]->accum[0].p)) { set_error("bg_merge_sharded: input %u is missing or not finalized", i); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  {  // every rank must hold the same inputs: their sizes are agreed before the first data collective
    std::vector<uint64_t> mine, all;
    for (uint32_t i = 0; i < n_in; ++i) mine.push_back(in[i]->n);
    BG_TRY(gather_u64(comm, mine, &all));
    for (size_t x = 0; x < all.size(); ++x)
      if (all[x] != mine[x % n_in]) { set_error("bg_merge_sharded: the ranks hold different inputs (input %zu)", x % n_in); return BG_EINVAL; }
  }
  const int rc = merge_sharded_impl(ctx, comm, n_in, in, out, stats, d_maps);
  if (rc != BG_OK) comm_fail(comm);
  return rc;
}

extern "C" void bg_sharded_free(bg_sharded* s) { delete s; }
extern "C" const bg_seqset* bg_sharded_slice(const bg_sharded* s) { return s ? s->slice : nullptr; }

// All ranks write ONE seqset container (sharded.py::write_slices): rank 0 lays the file out (headers, JSON members, `fixed`,
// room for the rest), every rank then writes its bytes of every member in place; the readmap's columns are gathered on rank 0,
// which writes the readmap container (readmap_path: NULL on every rank = no readmap file).
extern "C" int bg_sharded_write(bg_ctx* ctx, bg_comm* comm, const bg_sharded* s, const char* seqset_path, const char* readmap_path, int argc,
                                const char* const* argv, char* h_uuid37, char* h_readmap_sha41) {
  if (!ctx || !comm || !s || !seqset_path) { set_error("bg_sharded_write: null argument"); return BG_EINVAL; }
  const int P = comm_world(comm), me = comm_rank(comm);
  BG_HIP(hipSetDevice(ctx->device));
  struct Layout { uint64_t offs[15]; char uuid[40]; int32_t rc; char err[256]; } mine, zero;
  memset(&mine, 0, sizeof mine);
  if (me == 0) {
    mine.rc = bg_seqset_sharded_create_file(seqset_path, s->n_total, s->max_len, s->fixed, argc, argv, mine.offs, mine.uuid);
    if (mine.rc != BG_OK) snprintf(mine.err, sizeof mine.err, "%s", bg_last_error());
  }
  std::vector<Layout> all(P);
  BG_TRY(comm_all_gather_host(comm, &mine, sizeof mine, all.data()));
  zero = all[0];
  if (zero.rc != BG_OK) { set_error("bg_sharded_write: rank 0 could not lay out %s: %s", seqset_path, zero.err); return zero.rc; }
  BG_TRY(comm_barrier(comm));
  int rc = bg_seqset_slice_write(ctx, s->slice, seqset_path, zero.offs, s->first, s->is_last ? 1 : 0, s->base_pop);
  // every rank learns whether everybody wrote
  struct Verdict { int32_t rc; char err[252]; } v, bad;
  memset(&v, 0, sizeof v);
  v.rc = rc;
  if (rc != BG_OK) snprintf(v.err, sizeof v.err, "%s", bg_last_error());
  std::vector<Verdict> vs(P);
  BG_TRY(comm_all_gather_host(comm, &v, sizeof v, vs.data()));
  for (int r = 0; r < P; ++r)
    if (vs[r].rc != BG_OK) { bad = vs[r]; set_error("bg_sharded_write: rank %d: %s", r, bad.err); return bad.rc; }
  if (h_uuid37) memcpy(h_uuid37, zero.uuid, 37);
  if (h_readmap_sha41) h_readmap_sha41[0] = 0;
  if (!s->has_readmap || !readmap_path) return BG_OK;
  // the columns travel to rank 0 (17 bytes per read id over the links once)
  std::vector<uint64_t> send_rows(P, 0), recv_rows(P, 0), ns;
  BG_TRY(gather_u64(comm, {s->rm_n}, &ns));
  send_rows[0] = s->rm_n;
  if (me == 0) recv_rows = ns;
  const uint64_t tot = me == 0 ? s->rm_total : 0;
  DevBuf<uint64_t> e;
  DevBuf<uint32_t> rl, ml;
  DevBuf<uint8_t> fw;
  BG_TRY(e.alloc(ctx, tot));
  BG_TRY(rl.alloc(ctx, tot));
  BG_TRY(ml.alloc(ctx, tot));
  BG_TRY(fw.alloc(ctx, tot));
  BG_TRY(comm_exchange(comm, s->rm_entry.p, send_rows.data(), recv_rows.data(), 8, e.p));
  BG_TRY(comm_exchange(comm, s->rm_len.p, send_rows.data(), recv_rows.data(), 4, rl.p));
  BG_TRY(comm_exchange(comm, s->rm_loop.p, send_rows.data(), recv_rows.data(), 4, ml.p));
  BG_TRY(comm_exchange(comm, s->rm_fwd.p, send_rows.data(), recv_rows.data(), 1, fw.p));
  struct Sha { int32_t rc; char sha[44]; char err[208]; } sm, s0;
  memset(&sm, 0, sizeof sm);
  if (me == 0) {
    bg_readmap* rm = nullptr;
    sm.rc = bg_readmap_from_arrays(ctx, s->rm_entries, s->rm_max_read_len, zero.uuid, s->rm_total, e.p, rl.p, ml.p, fw.p, &rm);
    if (sm.rc == BG_OK) sm.rc = bg_readmap_write(ctx, rm, readmap_path, argc, argv, sm.sha);
    if (sm.rc != BG_OK) snprintf(sm.err, sizeof sm.err, "%s", bg_last_error());
    if (rm) bg_readmap_free(rm);
  }
  std::vector<Sha> shas(P);
  BG_TRY(comm_all_gather_host(comm, &sm, sizeof sm, shas.data()));
  s0 = shas[0];
  if (s0.rc != BG_OK) { set_error("bg_sharded_write: readmap: %s", s0.err); return s0.rc; }
  if (h_readmap_sha41) memcpy(h_readmap_sha41, s0.sha, 41);
  return BG_OK;
}
