// coverage.cpp — row f.2, third slice: the coverage and genotype end of the reference's assembly pipeline
// (pipeline.cpp:99-101) and the VCF record line:
//
//   path_group              modules/variants/path_group.cpp:21-277
//   calc_coverage           modules/variants/calc_coverage.cpp:9-533
//   simple_genotype_filter  modules/variants/simple_genotype_filter.cpp:27-151
//   ploidless_vcf_export    modules/variants/ploidless_vcf_export.cpp:12-166 (on_assembly: one record line)
//
// calc_coverage walks "path groups" through the seqset: the reference path group follows the scaffold around every
// variant section (max_read_len bases to either side), a variant's path group splits off at its left end, follows the
// variant's sequence and joins the reference group again at its right end; wherever a path stands on a seqset entry
// that carries reads, the reads are counted into the coverage trackers ("distant objects") still within reach.
// A path is a chain of push_front_drop(complement, 0) and paths only merge, so a whole stretch of bases is ONE device
// call for all paths of a group (bg_walk_chains_batch, csrc/walk.hip): the ranges after every base come back and the
// host replays the merging, the trimming to max_coverage_paths and the tracker bookkeeping in the reference's order.
// Runs of variants further apart than 2 x max_read_len share nothing (the reference path group is re-initialised
// between them): such islands advance in lock step on worker threads and every round is one launch for all of them.
//
// The reference reports from destructors (a tracker when the last path lets go of it, an assembly when its last
// tracker has reported).  What a tracker holds cannot change after the last path dropped it and everything reported is
// a minimum or a sum, so trackers are kept to the end and reported then; the output is put in canon order, the order
// the reference's sorted output queue with its tracked left offsets produces.
#include <limits.h>
#include <linux/futex.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>

#include "assembly.h"
#include "fiber.h"
#include "bg_internal.h"

using namespace bg;

extern "C" void bg_coverage_options_default(bg_coverage_options* o) {
  if (!o) return;
  o->max_coverage_paths = 4;             // assemble.h:715
  o->penalize_directional_coverage = 1;  // assemble.h:719
  o->simple_genotype_filter = 1;         // assemble.h:607
  o->min_depth_portion = .23;            // assemble.h:704
  o->min_avg_pair_depth = 1;             // assemble.h:711
  o->min_pair_depth = 0;                 // assemble.h:710
  o->min_read_depth = 1;                 // assemble.h:707
  o->vcf_sv_size_threshold = 20;         // assemble.h:724
  o->output_ml_features = 0;             // assemble.h:730 (biograph_variants.cpp:686 switches it on)
  o->rvg_exclude = 0;                    // assemble.h:699 (biograph_variants.cpp:194,689: --rvg-exclude, true there)
}

namespace {

void check(bool cond, const char* what) {
  if (!cond) throw CheckFailed{what};
}

struct Range { uint64_t b = 0, e = 0; uint32_t s = 0; };

struct Accum {  // coverage_accum (calc_coverage.cpp:23-88)
  std::vector<int> starts, ends;
  explicit Accum(size_t n = 0) : starts(n, 0), ends(n, 0) {}
  int size() const { return int(starts.size()); }
  bool add(int64_t start, int64_t end) {
    const int64_t n = size();
    if (start < 0) start = 0;
    if (end >= n) end = n - 1;
    if (start > n - 1 || end < 0) return false;
    ++starts[size_t(start)];
    ++ends[size_t(end)];
    return true;
  }
  std::vector<int> coverage() const {
    std::vector<int> out;
    int cur = 0;
    const size_t n = starts.size();
    for (size_t i = 0; i <= n; ++i) {
      if (i < n) cur += starts[i];
      if (i > 0) cur -= ends[i - 1];
      if (i == n) check(cur == 0, "coverage_accum: starts and ends do not balance"); else out.push_back(cur);
    }
    return out;
  }
};

struct Shared {  // shared_assembly (calc_coverage.h:44-58)
  bg_trace_assembly_rec a;
  std::set<uint32_t> pair_read_ids;
  std::vector<std::pair<int, int>> other_min;
  std::vector<std::vector<int>> other_ref_coverage;
};

struct Tracker {  // cov_tracker (calc_coverage.h:60-69)
  int var = -1;  // index into shared, -1: a stretch of reference
  std::vector<int> others;
  Accum fwd, rev, pair;
};

using DobjSet = std::vector<int>;  // tracker ids in insertion order, no duplicates (path_group.h dobj_set)

void dobj_add(DobjSet& s, int d) {  // path_group.cpp:231-241
  auto it = std::find(s.begin(), s.end(), d);
  if (it != s.end()) s.erase(it);
  s.push_back(d);
}
void dobj_add_all(DobjSet& s, DobjSet&& more) {  // :243-268
  if (more.empty()) return;
  if (s.empty()) { s = std::move(more); return; }
  DobjSet kept;
  for (int d : s)
    if (std::find(more.begin(), more.end(), d) == more.end()) kept.push_back(d);
  kept.insert(kept.end(), more.begin(), more.end());
  s = std::move(kept);
}

struct Path {
  std::map<int, DobjSet> objects;
  int unprop = 0;
};

struct CurKey {  // cur_comparer (path_group.h:111-121): longest context first, then by begin
  uint32_t size;
  uint64_t begin;
  bool operator<(const CurKey& r) const { return size != r.size ? size > r.size : begin < r.begin; }
};

class Coverage;

static double cov_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return double(ts.tv_sec) + double(ts.tv_nsec) * 1e-9;
}

// One stretch of bases for the heads of one path group, waiting for the device.
struct WalkRequest {
  std::vector<bg_walk_chain> chains;  // offsets relative to this request's codes / out
  std::vector<uint8_t> codes;
  std::vector<uint64_t> out;
};

// Where an island's state machine hands its stretches of bases in and gets the ranges back (Coverage::run_walks).
struct WalkWaiter {
  virtual ~WalkWaiter() {}
  virtual void walk(const std::vector<WalkRequest*>& reqs) = 0;  // returns when every request's `out` is filled
};

// The islands of a scaffold advance in lock step, a group of workers at a time: every worker thread runs one island's state
// machine and stops at each stretch of bases it needs walked; when all live workers of the group have stopped, the group's
// launcher thread (the only one that talks to HIP for the group, on the group's own context and stream) ships all stretches
// in one bg_walk_chains_batch and wakes them.  A round costs a launch, three copies and a wake-up of every worker whatever
// it carries, and an island takes about nine of them: several groups run their rounds side by side on the device, and the
// workers of a group sleep on the round counter itself (a futex: one wake call, no mutex for the woken to queue behind).
class Batcher {
 public:
  Batcher(bg_ctx* ctx, const bg_seqset* ss, int workers) : ctx_(ctx), ss_(ss), live_(workers) {}
  void submit_and_wait(const std::vector<WalkRequest*>& reqs) {
    uint32_t round;
    {
      std::lock_guard<std::mutex> l(mu_);
      pending_.insert(pending_.end(), reqs.begin(), reqs.end());
      round = round_.load(std::memory_order_relaxed);
      if (++waiting_ == live_) cv_all_.notify_one();
    }
    while (round_.load(std::memory_order_acquire) == round)
      syscall(SYS_futex, reinterpret_cast<uint32_t*>(&round_), FUTEX_WAIT_PRIVATE, round, nullptr, nullptr, 0);
    if (failed_.load(std::memory_order_acquire)) {
      std::lock_guard<std::mutex> l(mu_);
      throw CheckFailed{error_};
    }
  }
  void worker_exit() {
    std::unique_lock<std::mutex> l(mu_);
    --live_;
    cv_all_.notify_one();
  }
  void run() {  // on the group's launcher thread
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      cv_all_.wait(l, [&] { return live_ == 0 || waiting_ == live_; });
      if (live_ == 0) return;
      std::vector<WalkRequest*> batch;
      batch.swap(pending_);
      l.unlock();
      const std::string err = launch(batch);
      l.lock();
      if (!err.empty() && error_.empty()) { error_ = err; failed_.store(true, std::memory_order_release); }
      waiting_ = 0;
      round_.fetch_add(1, std::memory_order_release);
      syscall(SYS_futex, reinterpret_cast<uint32_t*>(&round_), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    }
  }
  uint64_t launches = 0, steps = 0;

 private:
  // Page-locked staging for the round's chains, bases and replies (a copy of pageable memory is staged by the runtime under a lock
  // that all groups' launchers and the tracer sessions share) -- off by default, see launch().
  struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    bool ensure(size_t bytes) {
      if (bytes <= cap) return true;
      if (p) (void)hipHostFree(p);
      p = nullptr;
      cap = 0;
      const size_t want = bytes + bytes / 2 + 4096;
      if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
      cap = want;
      return true;
    }
  };
  PinnedBuf pin_chains_, pin_codes_, pin_out_;
  std::string launch(const std::vector<WalkRequest*>& batch) {
    size_t n_chains = 0, n_codes = 0, out_words = 0;
    for (WalkRequest* r : batch) { n_chains += r->chains.size(); n_codes += r->codes.size(); out_words += r->out.size(); }
    // (pageable staging by default: the page-locked buffers are this run's own and have to grow to a few MB per group first,
    // which costs what they save -- workers 0.32-0.34 s against 0.35-0.37 s, A/B on one box; BG_COVERAGE_PINNED=1: page-locked)
    static const bool pinned = [] { const char* e = getenv("BG_COVERAGE_PINNED"); return e && e[0] == '1'; }();
    std::vector<bg_walk_chain> v_chains;
    std::vector<uint8_t> v_codes;
    std::vector<uint64_t> v_out;
    bg_walk_chain* chains;
    uint8_t* codes;
    uint64_t* out;
    if (pinned) {
      if (!pin_chains_.ensure(n_chains * sizeof(bg_walk_chain) + 8) || !pin_codes_.ensure(n_codes + 8) || !pin_out_.ensure(out_words * 8 + 8))
        return std::string("coverage: pinned staging buffers refused");
      chains = static_cast<bg_walk_chain*>(pin_chains_.p);
      codes = static_cast<uint8_t*>(pin_codes_.p);
      out = static_cast<uint64_t*>(pin_out_.p);
    } else {
      v_chains.resize(n_chains + 1); v_codes.resize(n_codes + 8); v_out.resize(out_words + 1);
      chains = v_chains.data(); codes = v_codes.data(); out = v_out.data();
    }
    std::vector<size_t> out_at;
    size_t nc = 0, nb = 0, nw = 0;
    for (WalkRequest* r : batch) {
      out_at.push_back(nw);
      for (bg_walk_chain c : r->chains) {
        c.code_offset += nb;
        c.out_offset += nw;
        chains[nc++] = c;
        steps += c.len;
      }
      if (!r->codes.empty()) memcpy(codes + nb, r->codes.data(), r->codes.size());
      nb += r->codes.size();
      nw += r->out.size();
    }
    if (bg_walk_chains_batch(ctx_, ss_, uint32_t(n_chains), chains, codes, n_codes, out, out_words) != BG_OK)
      return std::string("bg_walk_chains_batch: ") + bg_last_error();
    ++launches;
    for (size_t i = 0; i < batch.size(); ++i)
      std::copy(out + out_at[i], out + out_at[i] + batch[i]->out.size(), batch[i]->out.begin());
    return std::string();
  }
  bg_ctx* ctx_;
  const bg_seqset* ss_;
  std::mutex mu_;
  std::condition_variable cv_all_;
  std::vector<WalkRequest*> pending_;
  int live_, waiting_ = 0;
  std::atomic<uint32_t> round_{0};
  static_assert(sizeof(std::atomic<uint32_t>) == 4, "the round counter is the futex word");
  std::atomic<bool> failed_{false};
  std::string error_;
};

// What the islands share, read-only: the readmap columns on_seqset_entry reads (readmap::entry_to_index,
// get_readlength, get_is_forward), the scaffold, the options.
struct CovWorld {
  const bg_seqset* ss = nullptr;
  const ScaffoldView* sc = nullptr;
  bg_coverage_options o;
  // the readmap's narrow columns (bg_internal.h ReadmapTables: made on the device, shared with the tracer sessions of the same
  // readmap): read lengths and strands, and readmap::entry_to_index without a binary search over all read ids -- a bit per entry
  // that carries reads, the number of such entries before every 64-entry word, the first read id of the k-th such entry
  std::shared_ptr<const bg::ReadmapTables> rt;
  const uint8_t* read_len = nullptr;
  const uint8_t* is_forward = nullptr;
  const uint64_t* has_reads = nullptr;
  const uint32_t *word_rank = nullptr, *group_start = nullptr;
  uint32_t min_read_len = 0;
  int64_t max_read_len = 0;
  std::pair<uint32_t, uint32_t> entry_to_index(uint64_t e) const {
    const uint64_t word = has_reads[e >> 6], bit = 1ull << (e & 63);
    if (!(word & bit)) return {0u, 0u};
    const uint32_t g = word_rank[e >> 6] + uint32_t(__builtin_popcountll(word & (bit - 1)));
    return {group_start[g], group_start[g + 1]};
  }
  int load(bg_ctx* ctx, const bg_seqset* seqset, const bg_readmap* rm) {
    ss = seqset;
    BG_TRY(bg::readmap_tables(ctx, seqset, rm, &rt));
    read_len = rt->read_len.data();
    is_forward = rt->is_forward.data();
    has_reads = rt->has_reads.data();
    word_rank = rt->word_rank.data();
    group_start = rt->group_start.data();
    min_read_len = rt->min_read_len;
    max_read_len = int64_t(ss->max_len);  // seqset::max_read_len() = the longest entry (seqset.h:71)
    return BG_OK;
  }
};

class PathGroup {
 public:
  PathGroup(Coverage* c, size_t max_size) : c_(c), max_size_(max_size) {}
  void start(const Range& r) { add_path(r, Path{}); }
  void add_distant_object(int dobj, int size) {  // :97-104
    for (auto& kv : cur_) dobj_add(kv.second.second.objects[-(kv.second.second.unprop + size)], dobj);
  }
  // add_base (:113-151) for every base of a sequence in two halves: begin_sequence says what has to be walked
  // (false: nothing), Coverage::run_walks gets it walked together with whatever else its island has waiting, and
  // finish_sequence replays the bases over the ranges that came back
  bool begin_sequence(const char* seq, size_t len);
  void finish_sequence();
  WalkRequest* request() { return &req_; }
  void flush() {
    for (auto& kv : cur_) flush_path(kv.second.first, kv.second.second);
  }
  void join(PathGroup& rhs) {  // :203-220
    for (auto& kv : rhs.cur_) add_path(kv.second.first, std::move(kv.second.second));
    rhs.cur_.clear();
  }
  std::unique_ptr<PathGroup> split() const {  // :222-234
    std::unique_ptr<PathGroup> out(new PathGroup(c_, max_size_));
    out->cur_ = cur_;
    return out;
  }
  template <class F>
  void visit_distant_objects(const Range& r, F&& visit) {  // :171-201
    auto it = cur_.find(CurKey{r.s, r.b});
    check(it != cur_.end(), "path_group: visiting a range that is not a head");
    const Path& p = it->second.second;
    std::set<int> done;
    for (const auto& o : p.objects) {
      const int dist = o.first + p.unprop;
      if (dist > int(r.s)) break;
      for (int d : o.second)
        if (done.insert(d).second) visit(d, dist);
    }
  }

 private:
  static void flush_path(const Range& r, Path& p) {  // :82-95
    if (!p.unprop) return;
    std::map<int, DobjSet> old;
    std::swap(old, p.objects);
    for (auto& o : old)
      if (o.first + p.unprop <= int(r.s)) p.objects.emplace(o.first + p.unprop, std::move(o.second));
    p.unprop = 0;
  }
  void add_path(const Range& r, Path p) {  // :41-52
    const CurKey k{r.s, r.b};
    auto it = cur_.find(k);
    if (it == cur_.end()) {
      cur_.emplace(k, std::make_pair(r, std::move(p)));
      return;
    }
    Path& other = it->second.second;  // merge_paths(r, p, existing) :60-80
    if (p.unprop != other.unprop) {
      if (p.unprop) flush_path(r, p);
      if (other.unprop) flush_path(r, other);
    }
    check(p.unprop == other.unprop, "path_group: merged paths keep different distances");
    for (auto& o : other.objects) dobj_add_all(p.objects[o.first], std::move(o.second));
    it->second.second = std::move(p);
  }

  Coverage* c_;
  size_t max_size_;
  std::map<CurKey, std::pair<Range, Path>> cur_;
  WalkRequest req_;
  std::vector<std::pair<int, Path>> heads_;  // (chain, path) in cur order, while a walk is out
};

class Coverage {
 public:
  Coverage(const CovWorld& w, WalkWaiter* waiter)
      : world_(w), ss_(w.ss), sc_(*w.sc), o_(w.o), read_len_(w.read_len), is_forward_(w.is_forward),
        min_read_len_(w.min_read_len), max_read_len_(w.max_read_len), batcher_(waiter) {
    it_offset_ = sc_.ext.empty() ? sc_.end_pos : sc_.ext[0].offset;
    cur_offset_ = it_offset_;
    init_ref_pg();
  }

  // Everything this island has waiting for the device -- the variant groups whose sequence walk was put off, plus
  // `now` -- goes out as one hand-in to the batcher (one round for the island instead of one per group); then the
  // groups replay their bases.  The order of the replays does not matter: they touch disjoint path groups and add to
  // the trackers' counts.
  void run_walks(PathGroup* now) {
    std::vector<WalkRequest*> reqs;
    for (PathGroup* pg : deferred_) reqs.push_back(pg->request());
    if (now) reqs.push_back(now->request());
    if (reqs.empty()) return;
    batcher_->walk(reqs);
    for (PathGroup* pg : deferred_) {
      pg->finish_sequence();
      pg->flush();
    }
    deferred_.clear();
    if (now) now->finish_sequence();
  }

  void on_seqset_entry(const Range& r, PathGroup* pg) {  // calc_coverage.cpp:455-527
    if (r.s < min_read_len_) return;
    const std::pair<uint32_t, uint32_t> reads = world_.entry_to_index(r.b);
    if (reads.first == reads.second) return;
    const uint32_t first = reads.first, last = reads.second;
    pg->visit_distant_objects(r, [&](int d, int distance) {
      Tracker& cov = trackers_[size_t(d)];
      for (uint32_t rid = first; rid != last; ++rid) {
        const int64_t rl = int64_t(read_len_[rid]);
        if (rl > int64_t(r.s)) continue;
        Accum& acc = is_forward_[rid] ? cov.fwd : cov.rev;
        const int64_t start = int64_t(distance) - rl + 1 + acc.size() - 1, end = int64_t(distance) - 1 + acc.size() - 1;
        if (acc.add(start, end) && cov.var >= 0 && shared_[size_t(cov.var)].pair_read_ids.count(rid))
          check(cov.pair.add(start, end), "calc_coverage: pair coverage outside the read coverage");
      }
    });
  }

  void add(bg_trace_assembly_rec&& a) {  // on_assembly :206-279
    if (a.matches_reference) return;
    check(a.left_offset >= last_left_, "calc_coverage: input not sorted by left offset");
    last_left_ = a.left_offset;
    skip_to(a.left_offset - max_read_len_);
    while (cur_offset_ < a.left_offset) advance_towards(a.left_offset);
    const int sa = int(shared_.size());
    shared_.emplace_back();
    Shared& s = shared_.back();
    s.pair_read_ids.insert(a.left_pair_matches.begin(), a.left_pair_matches.end());
    s.pair_read_ids.insert(a.right_pair_matches.begin(), a.right_pair_matches.end());
    s.a = std::move(a);
    const int t = int(trackers_.size());
    trackers_.emplace_back();
    Tracker& tr = trackers_.back();
    tr.var = sa;
    for (const auto& x : active_) tr.others.push_back(x.second.first);
    const size_t n = s.a.seq.size();
    tr.fwd = Accum(n + 1);
    tr.rev = Accum(n + 1);
    tr.pair = Accum(n + 1);
    std::unique_ptr<PathGroup> pg = ref_pg_->split();
    pg->add_distant_object(t, int(n));
    if (pg->begin_sequence(s.a.seq.data(), n)) deferred_.push_back(pg.get());  // walked with the island's next stretch
    else pg->flush();
    const int64_t right = shared_[size_t(sa)].a.right_offset;
    need_coverage_to_ = std::max(need_coverage_to_, right + max_read_len_);
    active_.emplace(right, std::make_pair(sa, std::move(pg)));
    need_ref_dobj_ = true;
  }

  void finish(std::vector<bg_trace_assembly_rec>* out) {  // ~calc_coverage, then the destructors' reports (:105-204)
    skip_to(0x7fffffff);
    check(active_.empty() && deferred_.empty(), "calc_coverage: assemblies still active at the end");
    for (Tracker& t : trackers_) {
      const std::vector<int> f = t.fwd.coverage(), r = t.rev.coverage();
      std::vector<int> both(f.size());
      for (size_t i = 0; i < f.size(); ++i) both[i] = sum_dir_depth(f[i], r[i]);
      if (t.var >= 0) {
        std::vector<int> pair = t.pair.coverage();
        const std::pair<int, int> m{*std::min_element(both.begin(), both.end()), *std::min_element(pair.begin(), pair.end())};
        for (int other : t.others) shared_[size_t(other)].other_min.push_back(m);
        shared_[size_t(t.var)].a.coverage = std::move(both);
        shared_[size_t(t.var)].a.pair_coverage = std::move(pair);
      } else {
        check(!t.others.empty(), "calc_coverage: reference coverage nobody asked for");
        for (int other : t.others) shared_[size_t(other)].other_ref_coverage.push_back(both);
      }
    }
    for (Shared& s : shared_) {
      bg_trace_assembly_rec& a = s.a;
      a.other_depth = a.other_pair_depth = 0;
      if (!s.other_min.empty()) {
        a.other_depth = s.other_min[0].first;
        a.other_pair_depth = s.other_min[0].second;
        for (const auto& m : s.other_min) { a.other_depth = std::min(a.other_depth, m.first); a.other_pair_depth = std::min(a.other_pair_depth, m.second); }
      }
      check(!s.other_ref_coverage.empty(), "calc_coverage: an assembly without reference coverage");
      int min_ref = 0x7fffffff;
      for (const auto& c : s.other_ref_coverage) min_ref = std::min(min_ref, *std::min_element(c.begin(), c.end()));
      a.ref_depth = min_ref;
      a.has_coverage = true;
      out->push_back(std::move(a));
    }
  }

 private:
  friend class PathGroup;
  int sum_dir_depth(int fwd, int rev) const {  // :9-21
    if (o_.penalize_directional_coverage) {
      const int total = fwd + rev;
      if (abs(fwd - rev) * 4 > total * 3) return std::min(fwd, rev) * 2;
      return total;
    }
    return fwd + rev;
  }
  void init_ref_pg() {  // :362-370
    ref_pg_.reset(new PathGroup(this, o_.max_coverage_paths));
    Range r;
    r.b = 0;
    r.e = ss_->n;
    r.s = 0;
    ref_pg_->start(r);
  }
  void add_ref_dobj(int64_t ref_len) {  // :289-319
    if (!active_.empty()) {
      const int t = int(trackers_.size());
      trackers_.emplace_back();
      Tracker& tr = trackers_.back();
      tr.fwd = Accum(size_t(ref_len) + 1);
      tr.rev = Accum(size_t(ref_len) + 1);
      for (const auto& x : active_) tr.others.push_back(x.second.first);
      ref_pg_->add_distant_object(t, int(ref_len));
    }
    need_ref_dobj_ = false;
  }
  void flush_active_to_here() {  // :321-343
    while (!active_.empty() && active_.begin()->first <= cur_offset_) {
      if (need_ref_dobj_) add_ref_dobj(0);
      if (!deferred_.empty()) run_walks(nullptr);  // (an insertion joins before the reference moves on)
      ref_pg_->join(*active_.begin()->second.second);
      active_.erase(active_.begin());
    }
  }
  // scaffold::iterator (scaffold.h:14-60)
  bool it_at_end() const { return it_ext_ >= sc_.ext.size(); }
  void it_next_extent() {
    ++it_ext_;
    it_pos_ = 0;
    it_offset_ = it_at_end() ? sc_.end_pos : sc_.ext[it_ext_].offset;
  }
  void it_skip_to(int64_t target) {  // scaffold.cpp:205-247
    while (it_offset_ < target) {
      const int64_t adv = std::min(target - it_offset_, sc_.ext[it_ext_].n - it_pos_);
      it_pos_ += adv;
      it_offset_ += adv;
      if (it_pos_ == sc_.ext[it_ext_].n) {
        it_next_extent();
        if (it_at_end()) return;
      }
    }
  }
  void skip_to(int64_t offset) {  // :345-360
    while (cur_offset_ < offset) {
      if (active_.empty() && cur_offset_ >= need_coverage_to_) {
        if (!it_at_end() && it_offset_ < offset) {
          it_skip_to(offset);
          cur_offset_ = it_offset_;
        } else {
          cur_offset_ = offset;
        }
        init_ref_pg();
      } else {
        advance_towards(offset);
      }
    }
  }
  void advance_towards(int64_t target) {  // :372-453
    flush_active_to_here();
    check(target > cur_offset_, "calc_coverage: advancing backwards");
    if (!active_.empty() && target > active_.begin()->first) target = active_.begin()->first;
    check(target > cur_offset_, "calc_coverage: an active assembly ends behind the cursor");
    if (it_at_end()) { cur_offset_ = target; return; }
    if (it_pos_ == 0 && cur_offset_ < it_offset_) {
      if (target >= it_offset_) {
        cur_offset_ = it_offset_;
        init_ref_pg();
      } else {
        cur_offset_ = target;
      }
      return;
    }
    flush_active_to_here();
    const int64_t extent_end = sc_.ext[it_ext_].offset + sc_.ext[it_ext_].n;
    if (extent_end < target) target = extent_end;
    if (need_coverage_to_ > cur_offset_ && need_coverage_to_ < target) target = need_coverage_to_;
    const int64_t ref_len = target - cur_offset_;
    check(ref_len > 0, "calc_coverage: empty reference stretch");
    add_ref_dobj(ref_len);
    if (ref_pg_->begin_sequence(sc_.ext[it_ext_].p + it_pos_, size_t(ref_len))) run_walks(ref_pg_.get());
    else if (!deferred_.empty()) run_walks(nullptr);
    it_pos_ += ref_len;
    it_offset_ += ref_len;
    cur_offset_ += ref_len;
    if (it_pos_ == sc_.ext[it_ext_].n) it_next_extent();
    ref_pg_->flush();
  }

  const CovWorld& world_;
  const bg_seqset* ss_;
  const ScaffoldView& sc_;
  bg_coverage_options o_;
  const uint8_t* read_len_;
  const uint8_t* is_forward_;
  uint32_t min_read_len_ = 0;
  int64_t max_read_len_ = 0;
  WalkWaiter* batcher_;
  size_t it_ext_ = 0;
  int64_t it_pos_ = 0, it_offset_ = 0, cur_offset_ = 0, need_coverage_to_ = 0, last_left_ = INT32_MIN;
  bool need_ref_dobj_ = false;
  std::multimap<int64_t, std::pair<int, std::unique_ptr<PathGroup>>> active_;
  std::vector<PathGroup*> deferred_;  // variant groups (owned by active_) whose sequence walk has not been run yet
  std::unique_ptr<PathGroup> ref_pg_;
  std::vector<Tracker> trackers_;
  std::vector<Shared> shared_;
};

bool PathGroup::begin_sequence(const char* seq, size_t len) {
  if (!len || cur_.empty()) return false;
  // chain i = the i-th head in order; a merged path keeps following either of its (equal) chains
  heads_.clear();
  const size_t n = cur_.size();
  req_.chains.assign(n, bg_walk_chain());
  size_t i = 0;
  for (auto& kv : cur_) {
    bg_walk_chain& c = req_.chains[i];
    c.begin = kv.second.first.b;
    c.end = kv.second.first.e;
    c.size = kv.second.first.s;
    c.len = uint32_t(len);
    c.code_offset = 0;
    c.out_offset = uint64_t(i) * 3;
    c.out_stride = uint64_t(n) * 3;
    heads_.emplace_back(int(i), std::move(kv.second.second));
    ++i;
  }
  cur_.clear();
  req_.codes.resize(len);
  for (size_t k = 0; k < len; ++k) req_.codes[k] = seq[k] == 'A' ? 0 : seq[k] == 'C' ? 1 : seq[k] == 'G' ? 2 : 3;
  req_.out.assign(len * n * 3, 0);
  return true;
}

void PathGroup::finish_sequence() {
  const size_t n = req_.chains.size(), len = req_.codes.size();
  const std::vector<uint64_t>& ranges = req_.out;
  for (size_t t = 0; t < len; ++t) {
    cur_.clear();
    std::map<CurKey, int> next_chain;
    for (auto& h : heads_) {
      h.second.unprop += 1;
      const uint64_t* o = &ranges[(t * n + size_t(h.first)) * 3];
      Range r;
      r.b = o[0];
      r.e = o[1];
      r.s = uint32_t(o[2]);
      check(r.e > r.b, "path_group: no entry starts with a base of the sequence (the reference would not return)");
      next_chain.emplace(CurKey{r.s, r.b}, h.first);
      add_path(r, std::move(h.second));
    }
    if (max_size_ && cur_.size() > max_size_) {  // trim_cur :153-169: the shortest contexts go
      while (cur_.size() > max_size_) cur_.erase(std::prev(cur_.end()));
    }
    for (auto& kv : cur_) {
      Path& p = kv.second.second;
      const Range& r = kv.second.first;
      if (!p.objects.empty() && p.objects.begin()->first + p.unprop > int(r.s)) flush_path(r, p);
      c_->on_seqset_entry(r, this);
    }
    if (t + 1 < len) {
      heads_.clear();
      for (auto& kv : cur_) heads_.emplace_back(next_chain[kv.first], std::move(kv.second.second));
    }
  }
  heads_.clear();
  req_.chains.clear();
  req_.codes.clear();
  req_.out.clear();
}

std::string subscaffold_str(const ScaffoldView& sc, int64_t start, int64_t len) {  // scaffold.cpp:82-131: gaps read as N
  std::string out(size_t(std::max<int64_t>(len, 0)), 'N');
  for (const auto& e : sc.ext) {
    const int64_t lo = std::max(start, e.offset), hi = std::min(start + len, e.offset + e.n);
    for (int64_t p = lo; p < hi; ++p) out[size_t(p - start)] = e.p[p - e.offset];
  }
  return out;
}

int build_scaffold(const char* who, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                   const uint64_t* ext_len, int64_t end_pos, ScaffoldView* sc, bool check_bases = true) {
  int64_t calc_end = 0;
  for (uint64_t i = 0; i < n_extents; ++i) {
    if (ext_offset[i] < calc_end) { set_error("%s: extents must ascend without overlap", who); return BG_EINVAL; }
    if (check_bases && !all_acgt(ext_seq[i], ext_len[i])) { set_error("%s: extents hold ACGT only", who); return BG_EINVAL; }
    sc->ext.push_back(ScaffoldView::Ext{ext_offset[i], ext_seq[i], int64_t(ext_len[i])});
    calc_end = ext_offset[i] + int64_t(ext_len[i]);
  }
  if (end_pos < calc_end) { set_error("%s: end_pos lies before the end of the last extent", who); return BG_EINVAL; }
  sc->end_pos = end_pos;
  return BG_OK;
}

// simple_genotype_filter::on_assembly (simple_genotype_filter.cpp:27-151): false = discarded
bool genotype(const bg_coverage_options& o, bg_trace_assembly_rec& a) {
  if (a.matches_reference) return false;
  if (!o.simple_genotype_filter) return true;
  check(!a.coverage.empty(), "simple_genotype_filter: assembly without coverage");
  const int min_depth = *std::min_element(a.coverage.begin(), a.coverage.end());
  int min_pair = 0;
  double avg_pair = 0;
  if (!a.pair_coverage.empty()) {
    min_pair = *std::min_element(a.pair_coverage.begin(), a.pair_coverage.end());
    int tot = 0;
    for (int c : a.pair_coverage) tot += c;
    avg_pair = tot * 1. / double(a.pair_coverage.size());
  }
  // simple_genotype_filter::discard (:9-25): an assembly of the pop tracer that fails is passed on untouched
  if (avg_pair < o.min_avg_pair_depth || min_pair < o.min_pair_depth || min_depth < o.min_read_depth) return a.from_pop;
  const int total = min_depth + a.other_depth + a.ref_depth;
  if (!total) return a.from_pop;
  const double min_p = min_depth * 1. / total, alt_p = a.other_depth * 1. / total, ref_p = a.ref_depth * 1. / total;
  const double f = o.min_depth_portion;
  if (min_p < f) return a.from_pop;
  if (ref_p > f) {
    a.strand_count = 1;
    a.genotype_quality = std::min(min_p, ref_p) / (min_p + ref_p);
  } else if (alt_p > f) {
    a.strand_count = 1;
    a.genotype_quality = std::min(min_p, alt_p) / (min_p + alt_p);
  } else {
    a.strand_count = 2;
    a.genotype_quality = min_p;
  }
  return true;
}

// rvg_exclude (rvg_exclude.cpp:5-67), the last step of the standard pipeline (pipeline.cpp:102-104): a small variant
// section without pair support at some position is held back; it leaves only if a structural section of the same
// traced assembly (same assembly_id: "in phase") shows up, before or after it.
class RvgExclude : public SortedOutput {
 public:
  RvgExclude(const bg_coverage_options& o, AssemblySink out) : SortedOutput(std::move(out)), o_(o) {}
  void add(bg_trace_assembly_rec&& a) {
    flush_sorted_to(a.left_offset);
    if (a.matches_reference) { sort_and_output(std::move(a)); return; }
    unsigned reflen = unsigned(a.right_offset - a.left_offset), seqlen = unsigned(a.seq.size());
    if (!reflen || !seqlen) { ++reflen; ++seqlen; }  // compare as if VCF padded
    if (seqlen >= o_.vcf_sv_size_threshold || reflen >= o_.vcf_sv_size_threshold) {
      in_phase_.insert(a.assembly_id);
      auto it = backlog_.find(a.assembly_id);
      if (it != backlog_.end()) {
        for (auto& old : it->second) {
          untrack_left_offset(old.left_offset);
          sort_and_output(std::move(old));
        }
        backlog_.erase(it);
      }
      sort_and_output(std::move(a));
      return;
    }
    bool output_this;
    if (a.other_pair_depth || in_phase_.count(a.assembly_id)) {
      output_this = true;
    } else {
      check(!a.pair_coverage.empty(), "rvg_exclude: assembly without pair coverage");
      output_this = true;
      for (int depth : a.pair_coverage)
        if (!depth) { output_this = false; break; }
    }
    if (output_this) { sort_and_output(std::move(a)); return; }
    track_left_offset(a.left_offset);
    backlog_[a.assembly_id].push_back(std::move(a));
  }
  void finish() {  // flush(): what is still held back is gone
    for (auto& e : backlog_)
      for (auto& a : e.second) untrack_left_offset(a.left_offset);
    backlog_.clear();
    flush();
  }

 private:
  bg_coverage_options o_;
  std::map<uint64_t, std::vector<bg_trace_assembly_rec>> backlog_;
  std::set<uint64_t> in_phase_;
};

void rvg_exclude_all(const bg_coverage_options& o, std::vector<bg_trace_assembly_rec>&& in, std::vector<bg_trace_assembly_rec>* out) {
  RvgExclude step(o, [&](bg_trace_assembly_rec&& a) { out->push_back(std::move(a)); });
  for (auto& a : in) step.add(std::move(a));
  step.finish();
}

}  // namespace

// The readmap columns calc_coverage reads, copied to the host once (a reference has thousands of scaffolds, each its
// own bg_coverage_run).
// A worker thread runs several islands at a time, each on a stack of its own (ucontext): an island that needs bases walked
// parks (WalkWaiter::walk) and the thread goes on with its next island; when all of them are parked or finished the thread
// hands their stretches to its group's Batcher as ONE participant.  In flight are threads x fibers islands per group while
// only the threads meet at the round's barrier and are woken after it: with one island per thread (round 3: 128 threads per
// group) a round cost 1.6 ms of which the kernel was half, and more threads only made the wake-ups slower.
struct FiberThread {
  Batcher* batcher = nullptr;
  ucontext_t sched;
  std::vector<WalkRequest*> reqs;  // of the islands parked in this round
  bool failed = false;
  std::string error;
};
struct IslandFiber : WalkWaiter {  // csrc/fiber.h with the island's way of asking for bases
  FiberThread* th = nullptr;
  Fiber fiber;
  void walk(const std::vector<WalkRequest*>& reqs) override {
    th->reqs.insert(th->reqs.end(), reqs.begin(), reqs.end());
    fiber.park();  // back here when the thread has the replies
    if (th->failed) throw CheckFailed{th->error};
  }
};

struct bg_coverage {
  bg_ctx* ctx = nullptr;
  const bg_seqset* ss = nullptr;
  CovWorld tables;
  std::vector<bg_ctx*> group_ctx;  // contexts (streams) of the worker groups beyond the first, made when first needed
  // a scaffold's sections arriving in stretches (bg_coverage_feed): the variant sections of the island still open, what the
  // finished islands gave (in island order), where the input stands
  struct Stream {
    std::vector<bg_trace_assembly_rec> open, covered;
    int64_t need_to = INT64_MIN, last_left = INT64_MIN;
    uint64_t islands = 0, rounds = 0, steps = 0;
    double t_workers = 0;
  } stream;
  ~bg_coverage() { for (bg_ctx* g : group_ctx) bg_ctx_destroy(g); }
};

extern "C" int bg_coverage_open(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, bg_coverage** out) {
  if (!ctx || !ss || !rm || !out) { set_error("bg_coverage_open: null argument"); return BG_EINVAL; }
  if (rm->n_entries != ss->n) { set_error("bg_coverage_open: the readmap belongs to another seqset"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_coverage> c(new bg_coverage);
  c->ctx = ctx;
  c->ss = ss;
  BG_TRY(c->tables.load(ctx, ss, rm));
  *out = c.release();
  return BG_OK;
}

extern "C" void bg_coverage_close(bg_coverage* c) { delete c; }

extern "C" int bg_assemblies_coverage(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const bg_trace_result* in,
                                      uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                                      const uint64_t* ext_len, int64_t end_pos, const bg_coverage_options* options,
                                      int genotype_filter, bg_trace_result** out) {
  bg_coverage* c = nullptr;
  BG_TRY(bg_coverage_open(ctx, ss, rm, &c));
  const int rc = bg_coverage_run(c, in, n_extents, ext_offset, ext_seq, ext_len, end_pos, options, genotype_filter, out);
  bg_coverage_close(c);
  return rc;
}

// The islands of `islands` (variant sections sorted by left offset, each island a run that shares nothing with the next) through
// calc_coverage, every island a state machine of its own on the worker groups; done[i] = what island i gave.
static void run_islands(bg_coverage* c, CovWorld& w, const std::vector<std::vector<const bg_trace_assembly_rec*>>& islands,
                        std::vector<std::vector<bg_trace_assembly_rec>>* done_out) {
  bg_ctx* ctx = c->ctx;
  const bg_seqset* ss = c->ss;
  std::vector<std::vector<bg_trace_assembly_rec>>& done = *done_out;
  done.assign(islands.size(), {});
  // BG_COVERAGE_GROUPS groups (default 4: contexts, streams, launcher threads) of up to BG_COVERAGE_WORKERS worker threads
  // (default 16) with up to BG_COVERAGE_FIBERS islands in flight each (default 32); fiber v of all runs islands v, v + all, ...
  // (chr20-shaped sample, coverage + genotype, one island per thread: 1 x 128 threads with a condition variable 2.06 s; with the
  // futex 1 x 128 1.86 s, 8 x 64 0.93 s, 16 x 64 1.16 s, 4 x 128 0.87 s, 8 x 128 0.85 s -- 2 700 rounds a second whatever the
  // grouping: the wake-ups.)
  size_t max_workers = 16, max_groups = 4, max_fibers = 32;
  if (const char* e = getenv("BG_COVERAGE_WORKERS")) max_workers = size_t(std::max(1, atoi(e)));
  if (const char* e = getenv("BG_COVERAGE_GROUPS")) max_groups = size_t(std::max(1, atoi(e)));
  if (const char* e = getenv("BG_COVERAGE_FIBERS")) max_fibers = size_t(std::max(1, atoi(e)));
  const size_t n_threads = std::min<size_t>(islands.size(), max_workers * max_groups);
  const size_t fibers_per_thread = n_threads ? std::min<size_t>(max_fibers, (islands.size() + n_threads - 1) / n_threads) : 0;
  const size_t all = n_threads * fibers_per_thread;  // islands in flight
  const size_t n_groups = n_threads ? std::min<size_t>(max_groups, (n_threads + max_workers - 1) / max_workers) : 0;
  while (c->group_ctx.size() + 1 < n_groups) {
    bg_ctx* g = nullptr;
    if (bg_ctx_create_own_stream(ctx->device, &g) != BG_OK) throw CheckFailed{std::string("a worker group's context: ") + bg_last_error()};
    c->group_ctx.push_back(g);
  }
  std::vector<std::unique_ptr<Batcher>> batchers;
  std::vector<size_t> first_worker(n_groups + 1, 0);
  for (size_t g = 0; g < n_groups; ++g) {
    first_worker[g + 1] = n_threads * (g + 1) / n_groups;
    batchers.emplace_back(new Batcher(g ? c->group_ctx[g - 1] : ctx, ss, int(first_worker[g + 1] - first_worker[g])));
  }
  std::mutex err_mu;
  std::string first_error;
  auto note_error = [&](const std::string& what) {
    std::lock_guard<std::mutex> l(err_mu);
    if (first_error.empty()) first_error = what;
  };
  std::vector<std::thread> pool;
  for (size_t g = 0; g < n_groups; ++g)
    for (size_t u = first_worker[g]; u < first_worker[g + 1]; ++u)
      pool.emplace_back([&, g, u] {
        Batcher* b = batchers[g].get();
        FiberThread th;
        th.batcher = b;
        std::vector<std::unique_ptr<IslandFiber>> fibers;
        for (size_t k = 0; k < fibers_per_thread; ++k) {
          const size_t v = u * fibers_per_thread + k;  // this fiber's first island
          if (v >= islands.size()) break;
          fibers.emplace_back(new IslandFiber);
          IslandFiber& f = *fibers.back();
          f.th = &th;
          IslandFiber* self = &f;
          auto body = [&, v, self] {
            try {
              for (size_t i = v; i < islands.size(); i += all) {
                Coverage cov(w, self);
                for (const bg_trace_assembly_rec* a : islands[i]) {
                  bg_trace_assembly_rec copy = *a;
                  cov.add(std::move(copy));
                }
                cov.finish(&done[i]);
              }
            } catch (const CheckFailed& e) {
              note_error(e.what);
            } catch (const std::exception& e) {  // nothing may unwind past a fiber's entry frame (fiber.h): bad_alloc, length_error
              note_error(std::string("coverage: ") + e.what());
            } catch (...) {
              note_error("coverage: an island ended on an unknown exception");
            }
          };
          if (!f.fiber.start(&th.sched, size_t(512) << 10, body)) { note_error("coverage: no memory for an island's stack"); fibers.pop_back(); break; }
        }
        // rounds: every island of this thread runs until it parks or ends, then the thread takes part in the group's round
        for (;;) {
          th.reqs.clear();
          bool any = false;
          for (auto& f : fibers)
            if (!f->fiber.done) {
              any = true;
              f->fiber.resume();
            }
          if (!any) break;
          if (th.reqs.empty() || th.failed) continue;  // (the ones that parked before a failure wake up, throw and end)
          try {
            b->submit_and_wait(th.reqs);
          } catch (const CheckFailed& e) {
            th.failed = true;
            th.error = e.what;
            note_error(e.what);
          }
        }
        b->worker_exit();
      });
  for (size_t g = 1; g < n_groups; ++g)
    pool.emplace_back([&, g] {
      (void)hipSetDevice(ctx->device);
      batchers[g]->run();
    });
  if (n_groups) batchers[0]->run();
  for (auto& th : pool) th.join();
  if (!first_error.empty()) throw CheckFailed{first_error};
  for (auto& b : batchers) {
    c->stream.rounds += b->launches;
    c->stream.steps += b->steps;
  }
  c->stream.islands += islands.size();
}

// A scaffold's sections in stretches, in order (bgbinary discovery: as the pipeline behind the tracers hands them out).  Islands
// (see below) that a stretch completes are covered at once; the one still open waits for the next stretch.  last != 0: the
// scaffold is through -- the open island, then sort, genotype filter and rvg_exclude over everything, *out = the result
// (bg_coverage_run = one stretch that is the last).  One scaffold at a time per session; bg_coverage_discard forgets a scaffold
// that is not going to be finished.  Same result whatever the stretches: the islands are the same and stay in input order.
extern "C" int bg_coverage_feed(bg_coverage* c, const bg_trace_result* in, uint64_t n_extents, const int64_t* ext_offset,
                                const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos,
                                const bg_coverage_options* options, int genotype_filter, int last, bg_trace_result** out) {
  if (!c || (last && !out) || (n_extents && (!ext_offset || !ext_seq || !ext_len))) {
    set_error("bg_assemblies_coverage: null argument");
    return BG_EINVAL;
  }
  const double t_enter = cov_now();
  bg_ctx* ctx = c->ctx;
  bg_coverage_options o;
  if (options) o = *options; else bg_coverage_options_default(&o);
  ScaffoldView sc;
  BG_TRY(build_scaffold("bg_assemblies_coverage", n_extents, ext_offset, ext_seq, ext_len, end_pos, &sc));
  BG_HIP(hipSetDevice(ctx->device));
  bg_coverage::Stream& st = c->stream;
  try {
    CovWorld& w = c->tables;  // the scaffold and the options are this call's
    w.sc = &sc;
    w.o = o;
    // Islands: calc_coverage re-initialises its reference path group whenever nothing is active and the next variant
    // starts more than max_read_len behind everything still owed coverage (skip_to, calc_coverage.cpp:345-360), so
    // runs of variants separated like that share nothing and each gets its own state machine.
    // (the stretch's variant sections join the open island's; every island but the last one is closed by its successor)
    std::vector<bg_trace_assembly_rec> now = std::move(st.open);
    st.open.clear();
    std::vector<size_t> first_of;  // where every island starts in `now`
    int64_t need_to = INT64_MIN;
    for (size_t i = 0; i < now.size(); ++i) {  // (the open island: one island by construction)
      if (first_of.empty()) first_of.push_back(0);
      need_to = std::max(need_to, now[i].right_offset + w.max_read_len);
    }
    if (in)
      for (const auto& a : in->a) {
        if (a.matches_reference) continue;
        check(a.left_offset >= st.last_left, "calc_coverage: input not sorted by left offset");
        check(a.right_offset >= a.left_offset, "calc_coverage: right offset before the left one");
        st.last_left = a.left_offset;
        if (first_of.empty() || a.left_offset - w.max_read_len > need_to) first_of.push_back(now.size());
        now.push_back(a);
        need_to = std::max(need_to, a.right_offset + w.max_read_len);
      }
    size_t n_closed = first_of.size();
    if (!last && n_closed) {  // the last island stays open
      --n_closed;
      st.open.assign(std::make_move_iterator(now.begin() + long(first_of[n_closed])), std::make_move_iterator(now.end()));
      now.resize(first_of[n_closed]);
    }
    std::vector<std::vector<const bg_trace_assembly_rec*>> islands(n_closed);
    for (size_t k = 0; k < n_closed; ++k) {
      const size_t b = first_of[k], e = k + 1 < first_of.size() ? std::min(first_of[k + 1], now.size()) : now.size();
      for (size_t i = b; i < e; ++i) islands[k].push_back(&now[i]);
    }
    const double t_islands = cov_now();
    std::vector<std::vector<bg_trace_assembly_rec>> done;
    run_islands(c, w, islands, &done);
    for (auto& d : done)
      for (auto& a : d) st.covered.push_back(std::move(a));
    const double t_workers = cov_now();
    st.t_workers += t_workers - t_islands;
    if (!last) {
      if (getenv("BG_COVERAGE_TIMING"))
        fprintf(stderr, "bg_coverage_feed: %zu islands of this stretch in %.3f s (%zu sections stay open)\n", islands.size(), t_workers - t_enter, st.open.size());
      return BG_OK;
    }
    std::unique_ptr<bg_trace_result> res(new bg_trace_result);
    memset(&res->stats, 0, sizeof(res->stats));
    std::vector<bg_trace_assembly_rec> covered = std::move(st.covered);
    std::stable_sort(covered.begin(), covered.end(), bg::canon_assembly_less);
    const double t_sorted = cov_now();
    std::vector<bg_trace_assembly_rec> passed;
    for (auto& a : covered)
      if (!genotype_filter || genotype(o, a)) passed.push_back(std::move(a));
    const double t_geno = cov_now();
    if (o.rvg_exclude) rvg_exclude_all(o, std::move(passed), &res->a);  // pipeline.cpp:102-104
    else res->a = std::move(passed);
    if (getenv("BG_COVERAGE_TIMING"))
      fprintf(stderr, "bg_coverage_run: %llu islands: scaffold + islands of the last stretch %.3f s, workers %.3f s over all stretches (%llu rounds), "
                      "collect + sort %.3f s, genotype %.3f s, rvg_exclude %.3f s\n", (unsigned long long)st.islands, t_islands - t_enter, st.t_workers,
              (unsigned long long)st.rounds, t_sorted - t_workers, t_geno - t_sorted, cov_now() - t_geno);
    res->stats.device_rounds += st.rounds;
    res->stats.device_queries += st.steps;
    res->stats.traces = st.islands;
    st = bg_coverage::Stream();
    *out = res.release();
  } catch (const CheckFailed& f) {
    st = bg_coverage::Stream();
    set_error("bg_assemblies_coverage: %s", f.what.c_str());
    return BG_EINVAL;
  }
  return BG_OK;
}

extern "C" void bg_coverage_discard(bg_coverage* c) {
  if (c) c->stream = bg_coverage::Stream();
}

extern "C" int bg_coverage_run(bg_coverage* c, const bg_trace_result* in, uint64_t n_extents, const int64_t* ext_offset,
                               const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos,
                               const bg_coverage_options* options, int genotype_filter, bg_trace_result** out) {
  if (!c || !in || !out) { set_error("bg_assemblies_coverage: null argument"); return BG_EINVAL; }
  bg_coverage_discard(c);
  return bg_coverage_feed(c, in, n_extents, ext_offset, ext_seq, ext_len, end_pos, options, genotype_filter, 1, out);
}

extern "C" int bg_trace_result_coverage(const bg_trace_result* r, uint64_t i, bg_assembly_coverage* out) {
  if (!r || !out || i >= r->a.size()) { set_error("bg_trace_result_coverage: bad argument"); return BG_EINVAL; }
  const bg_trace_assembly_rec& a = r->a[i];
  out->has_coverage = a.has_coverage ? 1 : 0;
  out->coverage = a.coverage.data();
  out->pair_coverage = a.pair_coverage.data();
  out->n_coverage = a.coverage.size();
  out->n_pair_coverage = a.pair_coverage.size();
  out->other_depth = a.other_depth;
  out->other_pair_depth = a.other_pair_depth;
  out->ref_depth = a.ref_depth;
  out->strand_count = a.strand_count;
  out->genotype_quality = a.genotype_quality;
  return BG_OK;
}

extern "C" int bg_trace_result_set_coverage(bg_trace_result* r, uint64_t i, const bg_assembly_coverage* in) {
  if (!r || !in || i >= r->a.size() || (in->n_coverage && !in->coverage) || (in->n_pair_coverage && !in->pair_coverage)) {
    set_error("bg_trace_result_set_coverage: bad argument");
    return BG_EINVAL;
  }
  bg_trace_assembly_rec& a = r->a[i];
  a.has_coverage = in->has_coverage != 0;
  a.coverage.assign(in->coverage, in->coverage + in->n_coverage);
  a.pair_coverage.assign(in->pair_coverage, in->pair_coverage + in->n_pair_coverage);
  a.other_depth = in->other_depth;
  a.other_pair_depth = in->other_pair_depth;
  a.ref_depth = in->ref_depth;
  a.strand_count = in->strand_count;
  a.genotype_quality = in->genotype_quality;
  return BG_OK;
}

// rvg_exclude as a step of its own (host only): `in` = what calc_coverage / simple_genotype_filter emit for one scaffold
extern "C" int bg_assemblies_rvg_exclude(const bg_trace_result* in, const bg_coverage_options* options, bg_trace_result** out) {
  if (!in || !out) { set_error("bg_assemblies_rvg_exclude: null argument"); return BG_EINVAL; }
  bg_coverage_options o;
  if (options) o = *options; else bg_coverage_options_default(&o);
  std::unique_ptr<bg_trace_result> res(new bg_trace_result);
  res->stats = in->stats;
  try {
    std::vector<bg_trace_assembly_rec> copy = in->a;
    rvg_exclude_all(o, std::move(copy), &res->a);
  } catch (const CheckFailed& f) {
    set_error("bg_assemblies_rvg_exclude: %s", f.what.c_str());
    return BG_EINVAL;
  }
  *out = res.release();
  return BG_OK;
}

// ploidless_vcf_export::on_assembly (ploidless_vcf_export.cpp:12-166) without assembly ids and ML features.
extern "C" int bg_vcf_record(const bg_trace_result* r, uint64_t i, const char* scaffold_name, uint64_t n_extents,
                             const int64_t* ext_offset, const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos,
                             const bg_coverage_options* options, uint64_t assembly_id, char* buf, uint64_t cap, uint64_t* len) {
  if (!r || i >= r->a.size() || !scaffold_name || !len || (n_extents && (!ext_offset || !ext_seq || !ext_len))) {
    set_error("bg_vcf_record: bad argument");
    return BG_EINVAL;
  }
  bg_coverage_options o;
  if (options) o = *options; else bg_coverage_options_default(&o);
  ScaffoldView sc;
  // one record reads a few bases around its own position: the extents are not scanned (bg_assemblies_coverage did)
  BG_TRY(build_scaffold("bg_vcf_record", n_extents, ext_offset, ext_seq, ext_len, end_pos, &sc, false));
  const bg_trace_assembly_rec& a = r->a[i];
  if (a.matches_reference) { *len = 0; return BG_OK; }
  if (a.strand_count < 0 || a.strand_count > 2) { set_error("bg_vcf_record: more than 2 strands"); return BG_EINVAL; }
  const int min_depth = a.coverage.empty() ? 0 : *std::min_element(a.coverage.begin(), a.coverage.end());
  const int min_pair = a.pair_coverage.empty() ? 0 : *std::min_element(a.pair_coverage.begin(), a.pair_coverage.end());
  std::string var_seq = a.seq, ref_seq = subscaffold_str(sc, a.left_offset, a.right_offset - a.left_offset);
  int64_t left = a.left_offset;
  if (var_seq.empty() || ref_seq.empty()) {
    if (left == 0) {
      const std::string b = subscaffold_str(sc, a.right_offset, 1);
      var_seq += b;
      ref_seq += b;
    } else {
      const std::string b = subscaffold_str(sc, a.left_offset - 1, 1);
      var_seq = b + var_seq;
      ref_seq = b + ref_seq;
      --left;
    }
  }
  const char* gt = a.strand_count == 2 ? "1/1" : "0/1";
  const char* pg = a.strand_count == 2 ? "1|1" : "0|1";
  std::string line = std::string(scaffold_name) + "\t" + std::to_string(left + 1) + "\t.\t" + ref_seq + "\t" + var_seq + "\t100\tPASS\tNS=1";
  if (var_seq.size() >= o.vcf_sv_size_threshold || ref_seq.size() >= o.vcf_sv_size_threshold) {
    const int64_t svlen = int64_t(var_seq.size()) - int64_t(ref_seq.size());
    line += ";END=" + std::to_string(left + int64_t(ref_seq.size())) + ";SVLEN=" + std::to_string(svlen) + ";SVTYPE=" +
            (svlen < 0 ? "DEL" : svlen > 0 ? "INS" : "CPX");
  }
  char sample[256];
  snprintf(sample, sizeof(sample), "%s:%s:%d:%llu:%u:%d:%d,%d:%d:%d,%d", gt, pg, int(a.genotype_quality * 100),
           (unsigned long long)assembly_id, a.min_overlap, a.other_depth + min_depth + a.ref_depth, a.ref_depth, min_depth,
           a.other_pair_depth + min_pair, 0, min_pair);
  line += "\tGT:PG:GQ:PI:OV:DP:AD:PDP:PAD";
  if (o.output_ml_features) {  // ploidless_vcf_export.cpp:128-130,155-165 (LARANCH prints the LEFT anchor, as there)
    if (!a.has_ml) { set_error("bg_vcf_record: missing ML features on an assembly (bg_assemblies_process with output_ml_features)"); return BG_EINVAL; }
    line += ":LASCORE:LAREFSPAN:LARANCH:LALANCH:LAREFGC:LAALTGC:LAALTSEQLEN:NUMASM";
  }
  line += "\t";
  line += sample;
  if (o.output_ml_features) {
    char ml[256];
    snprintf(ml, sizeof(ml), ":%d:%d:%d:%d:%f:%f:%lu:%lu", a.ml_score, a.ml_refspan, a.ml_lanch, a.ml_ranch, double(a.ml_refgc),
             double(a.ml_altgc), (unsigned long)a.ml_alt_len, (unsigned long)a.merged + 1);
    line += ml;
  }
  line += "\n";
  *len = line.size();
  if (buf && cap) {
    const size_t n = std::min<size_t>(line.size(), size_t(cap - 1));
    memcpy(buf, line.data(), n);
    buf[n] = 0;
  }
  return BG_OK;
}
