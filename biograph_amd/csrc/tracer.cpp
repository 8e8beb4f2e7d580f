// tracer.cpp — row f.2, first slice: the reference's production "push" tracer (modules/variants/tracer.cpp) with the
// direction wrapper of reversable_tracer.cpp:76-145 and the pair_counter step (pair_counter.cpp), as a host state
// machine over two device walks.
//
// The reference sweeps a scaffold chunk base by base; at every offset where the read-ahead range is a read it runs one
// cost-ordered search ("trace") through the seqset and emits up to max_rejoins assemblies.  What couples one offset to
// the next is small and computable up front:
//   * the read-ahead itself (tracer.cpp:346-400): a chain of push_front_drop along the reference.  A chain started
//     anywhere catches up with the sequential walk within max_read_len bases, so the chunk is walked in 1024-base
//     chains on the device (bg_ref_walk_entries) and gives the entry id at every offset;
//   * the sliding tables m_entry_positions / m_rc_mate_read_positions (:372-394, :402-445): at the trace of offset c
//     they hold the offsets in [c - 1 - max_pair_distance, c - 1 + read_ahead_distance]; kept here as whole-chunk
//     tables that are asked with the window;
//   * m_cur_left_anchor and m_cur_ref_ambiguous_bases (:262-296, :493-503): one sequential pass over the offsets.
// Given those, the traces of different offsets are independent (the path storage is reset per trace, :478-481), so all
// traces of a chunk advance in lock step: every round each live trace asks for one extension
// (bg_tracer_extend_batch: the push_front_drop fan-out of :553-554 and the run of forced steps after it) and the
// host does the cost bookkeeping, the pairing look-ups and the queues exactly as the reference orders them
// (std::push_heap / pop_heap on the next paths, the multiset of rejoins, dead_end_sorter).
//
// The anchor_dropper between tracer and pair_counter (anchor_drop.cpp) and the head of the serial pipeline behind the
// tracers (sort, ref trim, dedup, align, split) are in align.cpp.  Not here: the pop tracer, the bidirectional tracer
// and what follows the align_splitter (normalize, coverage, genotype, VCF).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <array>
#include <type_traits>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <queue>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "assembly.h"
#include "bg_internal.h"

using namespace bg;

extern "C" void bg_trace_options_default(bg_trace_options* o) {
  if (!o) return;
  o->ambiguous_branch_cost = 1;          // assemble.h:488-
  o->rejoin_local_cost = 1;
  o->base_cost = 1;
  o->increase_max_between_pair_cost = 1;
  o->max_cost = 1000 * 1000;
  o->pairs_used_cost = -100;
  o->traverse_ref_cost = 100 * 1000;
  o->decrease_overlap_cost = 1000;
  o->size_change_cost = 5;
  o->dead_end_cost = 500 * 1000;
  o->anchor_drop_score = -10000;
  o->pair_match_score = 130;
  o->min_overlap_score = 100;
  o->max_pairs_per_read = 20;
  o->max_next_paths = 1024;
  o->max_ambiguous_bases = 300;
  o->min_overlap = 100;
  o->read_ahead_distance = 100000;
  o->scaffold_split_size = 1000 * 1000;
  o->max_branches_between_pairs = 2;
  o->max_rejoins = 5;
  o->max_bases_between_pairs = 300;
  o->max_search_steps = 1000;
  o->max_ambiguous_search_steps = 100;
  o->initial_search_steps = 100;
  o->max_search_steps_per_read = 3;
  o->max_pair_distance = 1000;
  o->forward_pairs_face_inward = 1;
  o->ignore_ambiguous_ref_pairs = 1;
  o->trace_ambiguous_ref = 0;
  o->trace_dead_ends = 1;
  o->only_trace_forward = 0;
  o->trace_reference_assemblies = 0;
  o->anchor_drop_size_multiplier = 1.5;
  o->min_anchor_drop_overlap = 15;       // assemble.h:504
  o->ref_align_factor = 4;               // assemble.h:677
  o->max_ref_align_bases = 30;           // assemble.h:682
  o->anchor_drop = 1;                    // reversable_tracer.cpp:124-125 always has the step
  o->max_anchor_kmer_size = 31;          // anchor_dropper::g_max_kmer_size (anchor_drop.cpp:9)
  o->output_ml_features = 0;             // assemble.h:730
  o->min_pop_overlap = 15;               // assemble.h:523
  o->pop_tracer_offset_slop = 1000;      // assemble.h:527
  o->min_pair_distance = 100;            // assemble.h:588
}

namespace {

constexpr uint64_t kNoEntry = ~0ull;

struct TRange {
  uint64_t b = 0, e = 0;
  uint32_t s = 0;
};

struct Path {  // tracer::path (tracer.h:131-157)
  int prev = -1;
  int64_t cost = 0;
  uint8_t min_overlap = 0;
  TRange range;
  std::string seq;
  bool part_of_assembly = false;
  std::vector<std::pair<uint32_t, int32_t>> seen_pairs;  // (read id, offset)
  std::vector<uint32_t> seen_read_ids;
};

struct NextPath {  // tracer::next_path (tracer.h:159-176)
  int path = -1;
  unsigned path_bases = 0, pushed_since_read = 0, pushed_since_pair = 0, ambiguous_bases = 0, branch_count_since_pair = 0,
           max_between_pairs = 0, pairs_used = 0, num_reads = 0;
  uint64_t loop_check_seqset_id = ~0ull;
  unsigned reads_until_loop_check = 0;
};

struct Rejoin {  // tracer::rejoin (tracer.h:178-185)
  int path = -1;
  int64_t rejoin_cost = 0;
  int32_t right_offset = 0x7fffffff;
  int right_anchor_len = 0;
  bool anchor_drop = false;
};

std::string rev_comp_str(const std::string& s) {
  std::string o(s.rbegin(), s.rend());
  for (char& c : o) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
  return o;
}

}  // namespace

namespace {

// (n bytes that nobody fills before the download does: a vector's zeros were 280 MB of page faults and stores on one thread
// before the copy threads wrote the same pages again)
struct RawBytes {
  std::unique_ptr<uint8_t[]> p;
  void resize(uint64_t n) { p.reset(n ? new uint8_t[n] : nullptr); }
  uint8_t* data() { return p.get(); }
  const uint8_t* data() const { return p.get(); }
};
struct HostTables {
  RawBytes sizes, ref_map;
  std::shared_ptr<const ReadmapTables> rt;  // the readmap's narrow columns, host and device (bg_internal.h): shared with calc_coverage
};

// A pinned host buffer that grows: the per-round queries / replies of a session cross the bus from and to page-locked
// memory (a copy of pageable memory is staged by the runtime under a lock all sessions share).
struct Pinned {
  void* p = nullptr;
  size_t cap = 0;
  Pinned() {}
  Pinned(const Pinned&) = delete;
  Pinned& operator=(const Pinned&) = delete;
  ~Pinned() { if (p) (void)hipHostFree(p); }
  int ensure(size_t bytes) {
    if (bytes <= cap) return BG_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; set_error("tracer: pinned staging buffer of %zu bytes refused", want); return BG_ENOMEM; }
    cap = want;
    return BG_OK;
  }
};

// The device copies of the readmap columns the never-branching-trace path consults (walk.hip k_trace_simple): pointers into the
// shared ReadmapTables, which the session's HostTables keep alive.
struct DeviceTables {
  const uint32_t* word_rank = nullptr;
  const uint32_t* group_start = nullptr;
  const uint32_t* mate_loop = nullptr;
  const uint8_t* is_forward = nullptr;
};

// Everything the traces of one chunk share.
struct World {
  bg_ctx* ctx;
  const bg_seqset* ss;
  const bg_readmap* rm;
  bg_trace_options o;
  uint64_t n_entries = 0;
  // host copies of the seqset / readmap / ref_map columns the tracer consults: read-only, shared with the sessions
  // cloned from this one (bg_tracer_clone: several sessions trace side by side)
  std::shared_ptr<const HostTables> tables;
  std::shared_ptr<const DeviceTables> dev_tables;  // null: every trace runs on the host's state machine
  const uint8_t* sizes = nullptr;       // entry sizes
  const uint8_t* d_ref_map = nullptr;   // the caller's device ref_map (stays valid while the session lives)
  const uint8_t* ref_map = nullptr;     // ref_map bytes (match count = & 63)
  const uint32_t* mate_loop = nullptr;  // per read id
  const uint8_t* is_forward = nullptr;  // per read id
  const uint64_t* has_reads = nullptr;  // readmap::get_bit as a bit per entry
  const uint32_t* word_rank = nullptr;  // entries with reads before each 64-entry word
  const uint32_t* group_start = nullptr;  // first read id of the k-th entry that carries reads (+ one past the end)
  size_t n_reads = 0;
  void bind_tables() {
    sizes = tables->sizes.data();
    ref_map = tables->ref_map.data();
    mate_loop = tables->rt->mate_loop.data();
    is_forward = tables->rt->is_forward.data();
    has_reads = tables->rt->has_reads.data();
    word_rank = tables->rt->word_rank.data();
    group_start = tables->rt->group_start.data();
    n_reads = tables->rt->mate_loop.size();
  }
  int64_t scaffold_end = 0;
  // reads along the (oriented) scaffold: offset (after the base) -> entry, ascending by offset
  std::vector<std::pair<int32_t, uint64_t>> position_entries;
  std::vector<uint16_t> position_info;  // beside position_entries: the entry's size | 256 when the reference holds it more than once
  std::vector<std::pair<uint64_t, int32_t>> entry_positions;        // (entry, offset), sorted
  std::vector<std::pair<uint32_t, int32_t>> rc_mate_read_positions;  // (read, offset), sorted
  Pinned pin_q, pin_r, pin_s;  // the round's queries, replies and step lists
  bg_trace_stats stats;
  double seconds_device = 0, seconds_host = 0, seconds_tables = 0, seconds_start = 0, seconds_teardown = 0;
  double seconds_tab_codes = 0, seconds_tab_walk = 0, seconds_tab_read = 0;  // inside seconds_tables (BG_TRACER_TIMING)
  int threads = 1;

  unsigned match_count(uint64_t e) const { return ref_map[e] & 63u; }
  // readmap.h:114-116 (sparse_multi::lookup): the read ids of entry e.  Rank of e among the entries that carry reads
  // (a popcount prefix per 64 entries), then that group's first read id: three touches instead of a binary search
  // over tens of millions of read ids.  An entry without reads gives an empty range.
  std::pair<uint32_t, uint32_t> entry_to_index(uint64_t e) const {
    const uint64_t word = has_reads[e >> 6], bit = 1ull << (e & 63);
    if (!(word & bit)) return {0u, 0u};
    const uint32_t g = word_rank[e >> 6] + uint32_t(__builtin_popcountll(word & (bit - 1)));
    return {group_start[g], group_start[g + 1]};
  }
  bool get_bit(uint64_t e) const { return (has_reads[e >> 6] >> (e & 63)) & 1; }
  bool is_entry(const TRange& r) const { return r.b + 1 == r.e && r.s == sizes[r.b]; }     // seqset.h:298-300
  bool is_read(const TRange& r) const { return is_entry(r) && get_bit(r.b); }                // tracer.cpp:521-526
  bool has_mate(uint32_t id) const { return mate_loop[mate_loop[id]] != id; }               // readmap.cpp:207-218
  uint32_t get_mate(uint32_t id) const { return mate_loop[mate_loop[id]]; }                  // :220-239
  uint32_t get_rev_comp(uint32_t id) const {                                                  // :248-268
    for (int c = is_forward[id] ? 1 : 3; c; --c) id = mate_loop[id];
    return id;
  }
  bool pairing_skipped(uint64_t e) const { return o.ignore_ambiguous_ref_pairs && match_count(e) > 1; }
};

// the end of tracer::output_assembly (tracer.cpp:1199-1215): matches_reference, anchors no longer than the assembly
void close_assembly(bg_trace_assembly_rec* a) {
  a->matches_reference = (a->left_anchor_len + a->right_anchor_len) >= (a->right_offset - a->left_offset) &&
                         (a->right_offset - a->left_offset) == int64_t(a->seq.size());
  if (a->matches_reference) a->left_anchor_len = a->right_anchor_len = 0;
  const int64_t max_tot_anchor = std::min<int64_t>(int64_t(a->seq.size()), a->right_offset - a->left_offset);
  if (a->right_anchor_len + a->left_anchor_len > max_tot_anchor) {
    a->right_anchor_len = int(std::min<int64_t>(a->right_anchor_len, max_tot_anchor / 2));
    a->left_anchor_len = int(std::min<int64_t>(a->left_anchor_len, max_tot_anchor / 2));
  }
}

// One trace (tracer::trace + trace_next_paths + output_rejoins for one offset) as a resumable machine.
struct Trace {
  World* w;
  bg_trace_stats st = {};  // this trace's share of assemble_stats (traces run on several host threads)
  int32_t cur_offset = 0;
  TRange cur_range;
  std::string left_anchor;
  unsigned ref_ambiguous_bases = 0;
  // window of the sliding tables at this offset
  int32_t win_lo = 0, win_hi = 0;

  std::vector<Path> paths;   // referenced by index only (no reference is held across alloc())
  std::vector<NextPath> next_paths;
  std::vector<Rejoin> rejoins;   // std::multiset order: worst first, best last
  std::vector<int> dead_ends;    // std::set<const path*>: kept in allocation order
  size_t search_steps = 0, ambiguous_steps = 0;
  bool done = false;
  // expansion in flight
  bool expanding = false;
  NextPath np;
  bool pushed_any = false;
  std::vector<bg_trace_assembly_rec> out;

  int alloc() { paths.emplace_back(); return int(paths.size()) - 1; }
  const bg_trace_options& o() const { return w->o; }

  // ---- tables through the window
  bool has_rc_mate_in_range(uint32_t read_id, int64_t start, int64_t limit) const {  // tracer.cpp:466-475
    const int64_t lo = std::max<int64_t>(start, win_lo), hi = std::min<int64_t>(limit, int64_t(win_hi) + 1);
    if (lo >= hi) return false;
    const auto& v = w->rc_mate_read_positions;
    auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(read_id, int32_t(lo)));
    return it != v.end() && it->first == read_id && it->second < hi;
  }

  // ---- tracer.cpp:847-877
  void add_base(NextPath& x, int base) {
    Path& p = paths[x.path];
    p.seq.push_back("ACGT"[base]);
    ++x.pushed_since_read;
    ++x.path_bases;
    if (x.ambiguous_bases) ++x.ambiguous_bases;
    p.cost += o().base_cost;
    const unsigned overlap = p.range.s - x.pushed_since_read;
    if (overlap < p.min_overlap) {
      const unsigned dec = p.min_overlap - overlap;
      p.cost += o().decrease_overlap_cost * int64_t(dec);
      p.min_overlap = uint8_t(overlap);
      x.max_between_pairs += dec * 2;
    }
    ++x.pushed_since_pair;
    if (x.pushed_since_pair > x.max_between_pairs) {
      ++x.max_between_pairs;
      p.cost += o().increase_max_between_pair_cost;
    }
    add_pairs(x);
  }

  bool path_has_read_in_range(int start_path, uint32_t read_id, int64_t start, int64_t limit) const {  // :1129-1160
    // (as written there: every hop re-reads the START path's pairs and length)
    const Path& sp = paths[start_path];
    int64_t cur_distance = 0;
    int cur = start_path;
    while (cur >= 0) {
      const int64_t last_offset = int64_t(sp.seq.size());
      for (auto it = sp.seen_pairs.rbegin(); it != sp.seen_pairs.rend(); ++it) {
        cur_distance += last_offset - it->second;
        if (cur_distance >= limit) return false;
        if (cur_distance < start) continue;
        if (it->first == read_id && cur_distance > 0) return true;
      }
      cur_distance += last_offset;
      if (cur_distance >= limit) return false;
      cur = paths[cur].prev;
    }
    return false;
  }

  void add_pairs(NextPath& x) {  // :737-845
    Path& p = paths[x.path];
    if (!w->is_read(p.range)) return;
    const uint64_t sid = p.range.b;
    ++x.num_reads;
    if (x.reads_until_loop_check) {
      --x.reads_until_loop_check;
    } else {
      x.loop_check_seqset_id = sid;
      x.reads_until_loop_check = x.num_reads / 2;
    }
    const auto reads = w->entry_to_index(sid);
    if (reads.first != reads.second) {
      x.pushed_since_read = 0;
      const unsigned ref_count = w->match_count(sid);
      if (ref_count > 1) {
        x.ambiguous_bases = std::max<unsigned>(x.ambiguous_bases, p.range.s);
        if (!o().trace_ambiguous_ref) {
          p.cost += o().max_cost;
          ++st.prune_ambiguous_ref;
        }
      } else if (ref_count == 1) {
        p.cost += o().traverse_ref_cost;
      }
    }
    bool matched_pair = false;
    unsigned added_pairs = 0, added_reads = 0;
    for (uint32_t rid = reads.first; rid != reads.second; ++rid) {
      if (!w->has_mate(rid)) continue;
      if (added_reads < o().max_pairs_per_read) {
        p.seen_read_ids.push_back(rid);
        ++added_reads;
      }
      if (bool(w->is_forward[rid]) == bool(o().forward_pairs_face_inward)) {
        if (matched_pair) continue;
        int64_t ambiguous_in_ref = 0;
        if (x.ambiguous_bases > x.path_bases) ambiguous_in_ref = x.ambiguous_bases - x.path_bases;
        if (has_rc_mate_in_range(rid, int64_t(cur_offset) + x.path_bases - o().max_pair_distance,
                                 int64_t(cur_offset) - std::max<int64_t>(0, ambiguous_in_ref))) {
          matched_pair = true;
          continue;
        }
        if (path_has_read_in_range(x.path, rid, x.ambiguous_bases, o().max_pair_distance)) {
          matched_pair = true;
          continue;
        }
      } else {
        if (added_pairs < o().max_pairs_per_read) {
          p.seen_pairs.emplace_back(w->get_rev_comp(w->get_mate(rid)), int32_t(p.seq.size()));
          ++added_pairs;
          ++st.found_pairs;
        }
        if (matched_pair) continue;
        if (has_rc_mate_in_range(rid, cur_offset, int64_t(cur_offset) + x.path_bases + o().max_pair_distance)) {
          matched_pair = true;
          continue;
        }
      }
    }
    if (matched_pair) {
      x.pushed_since_pair = 0;
      x.branch_count_since_pair = 0;
      x.ambiguous_bases = 0;
      ++x.pairs_used;
      p.cost += o().pairs_used_cost;
      ++st.matched_pairs;
    }
  }

  // ---- queues (:879-941); the heap moves are those of libstdc++'s std::push_heap / std::pop_heap
  void push_next_path(NextPath x) {
    const Path& p = paths[x.path];
    if (p.cost > o().max_cost) { ++st.max_branch_cost; return; }
    if (!rejoins.empty() && rejoins.size() >= o().max_rejoins) {
      const unsigned worst = unsigned(paths[rejoins.front().path].cost + rejoins.front().rejoin_cost);  // "unsigned" (:888)
      if (p.cost > int64_t(worst)) { ++st.suboptimal_path_prune; return; }
    }
    next_paths.push_back(x);
    std::push_heap(next_paths.begin(), next_paths.end(),
                   [this](const NextPath& a, const NextPath& b) { return paths[a.path].cost > paths[b.path].cost; });
    if (next_paths.size() > o().max_next_paths) {
      next_paths.pop_back();
      ++st.next_paths_too_big;
    }
  }
  NextPath pop_next_path() {
    std::pop_heap(next_paths.begin(), next_paths.end(),
                  [this](const NextPath& a, const NextPath& b) { return paths[a.path].cost > paths[b.path].cost; });
    NextPath x = next_paths.back();
    next_paths.pop_back();
    return x;
  }
  int64_t total(const Rejoin& r) const { return paths[r.path].cost + r.rejoin_cost; }
  void push_rejoin(const Rejoin& r) {
    const int64_t t = total(r);
    if (t > o().max_cost) { ++st.max_branch_cost; return; }
    size_t i = 0;  // multiset with "a before b when a's total is larger": equal elements go after their equals
    while (i < rejoins.size() && total(rejoins[i]) >= t) ++i;
    rejoins.insert(rejoins.begin() + i, r);
    if (rejoins.size() > o().max_rejoins) rejoins.erase(rejoins.begin());
  }

  bool add_rejoins(const NextPath& x) {  // :675-724
    const Path& p = paths[x.path];
    if (!w->is_read(p.range)) return false;
    int64_t best_distance = -1;
    int32_t best_offset = -1;
    // candidates after cur_offset, inside the read-ahead window, in ascending order (the order they were inserted)
    const auto& v = w->entry_positions;
    for (auto it = std::upper_bound(v.begin(), v.end(), std::make_pair(uint64_t(p.range.b), cur_offset));
         it != v.end() && it->first == p.range.b && it->second <= win_hi; ++it) {
      const int64_t distance = std::llabs(int64_t(cur_offset) + x.path_bases - it->second);
      if (best_distance < 0 || distance < best_distance) { best_distance = distance; best_offset = it->second; }
    }
    if (best_distance < 0) return false;
    if (x.ambiguous_bases && best_distance > int64_t(x.ambiguous_bases)) { ++st.extend_ambiguous_rejoin; return false; }
    Rejoin r;
    r.path = x.path;
    r.rejoin_cost = o().rejoin_local_cost + best_distance * o().size_change_cost - o().traverse_ref_cost;
    r.right_offset = best_offset;
    r.right_anchor_len = int(p.range.s);
    push_rejoin(r);
    return true;
  }

  void add_dead_end_rejoin(const NextPath& x) {  // :726-735
    if (x.num_reads <= 1 || !o().trace_dead_ends) return;
    if (std::find(dead_ends.begin(), dead_ends.end(), x.path) == dead_ends.end()) dead_ends.push_back(x.path);
    const Path& p = paths[x.path];
    if (p.prev >= 0 && paths[p.prev].min_overlap == p.min_overlap) {
      auto it = std::find(dead_ends.begin(), dead_ends.end(), p.prev);
      if (it != dead_ends.end()) dead_ends.erase(it);
    }
  }

  // ---- the machine.  start() / resume() return true when a query is pending in `q`.
  bool start(bg_trace_query* q) {  // tracer::trace (:477-519)
    if (w->match_count(cur_range.b) > 1) {
      ++st.ambiguous_ref_reads;
      if (!o().trace_ambiguous_ref) return finish();
    } else {
      ++st.ref_reads;
    }
    NextPath x;
    x.path = alloc();
    paths[x.path].range = cur_range;
    paths[x.path].seq = left_anchor;
    x.ambiguous_bases = ref_ambiguous_bases;
    paths[x.path].min_overlap = uint8_t(left_anchor.size());
    push_next_path(x);
    return next_expansion(q);
  }

  bool finish() {
    output_rejoins();
    done = true;
    // give the search state back on the thread that allocated it (frees from another thread contend in malloc)
    std::vector<Path>().swap(paths);
    std::vector<NextPath>().swap(next_paths);
    std::vector<Rejoin>().swap(rejoins);
    std::vector<int>().swap(dead_ends);
    std::string().swap(left_anchor);
    return false;
  }

  void ask(bg_trace_query* q) {
    const Path& p = paths[np.path];
    q->begin = p.range.b;
    q->end = p.range.e;
    q->size = p.range.s;
    q->loop_check = np.loop_check_seqset_id;
    q->min_ctx = uint32_t(o().min_overlap) + np.pushed_since_read;
    q->flags = pushed_any ? 1u : 0u;
    q->max_steps = 0xffffffffu;
  }

  bool next_expansion(bg_trace_query* q) {  // trace_next_paths (:943-1021) up to the next expand_next_path
    while (!next_paths.empty()) {
      NextPath x = pop_next_path();
      if (!rejoins.empty() && rejoins.size() >= o().max_rejoins && paths[x.path].cost > total(rejoins.front())) {
        ++st.suboptimal_path_prune;
        continue;
      }
      if (search_steps > size_t(o().max_search_steps_per_read) * x.num_reads + o().initial_search_steps) {
        ++st.search_not_fast_enough;
        continue;
      }
      if (x.ambiguous_bases > o().max_ambiguous_bases) { ++st.too_many_ambiguous_bases; continue; }
      ++st.step_count;
      ++search_steps;
      if (search_steps > o().max_search_steps) { ++st.too_many_steps; return finish(); }
      if (ambiguous_steps > o().max_ambiguous_search_steps) { ++st.too_many_ambiguous_steps; return finish(); }
      np = x;
      pushed_any = false;
      expanding = true;
      ask(q);
      return true;
    }
    return finish();
  }

  // the reply to the pending query: the forced steps taken, then where the walk stopped
  bool resume(const bg_trace_reply& r, const uint32_t* steps, bg_trace_query* q) {  // expand_next_path (:528-673)
    for (uint32_t i = 0; i < r.n_steps; ++i) {
      const int base = int(steps[i] & 3);
      Path& p = paths[np.path];
      // the range after this step: only its size matters until the last one (no read in between)
      if (i + 1 == r.n_steps) { p.range.b = r.begin; p.range.e = r.end; p.range.s = r.size; }
      else { p.range.b = 0; p.range.e = 0; p.range.s = steps[i] >> 8; }
      add_base(np, base);
      pushed_any = true;
      if (np.pushed_since_pair > o().max_bases_between_pairs) { ++st.too_far_without_pair; return next_expansion(q); }
      if (i + 1 == r.n_steps && add_rejoins(np)) return next_expansion(q);
    }
    if (r.reason == BG_TRACE_STEPPED_ON_READ || r.reason == BG_TRACE_STEP_CAP) {
      ask(q);
      return true;
    }
    if (r.reason == BG_TRACE_AMBIGUOUS) ++ambiguous_steps;
    const int new_path = np.path;
    if (paths[new_path].prev >= 0) add_dead_end_rejoin(np);
    unsigned lens[4];
    int nl = 0;
    for (int b = 0; b < 4; ++b)
      if (r.valid_mask & (1u << b)) lens[nl++] = r.ps[b];
    if (!nl) return next_expansion(q);
    std::sort(lens, lens + nl);
    unsigned unambiguous_len = lens[nl - 1];
    if (nl > 1 && lens[nl - 2] == unambiguous_len) ++unambiguous_len;  // two with the same length: all ambiguous
    for (int b = 0; b < 4; ++b) {
      if (!(r.valid_mask & (1u << b))) continue;
      NextPath x = np;
      x.path = alloc();
      {
        Path& np_ = paths[x.path];
        const Path& old = paths[new_path];
        np_.prev = new_path;
        np_.min_overlap = old.min_overlap;
        np_.cost = old.cost;
        np_.range.b = r.pb[b];
        np_.range.e = r.pe[b];
        np_.range.s = r.ps[b];
      }
      add_base(x, b);
      if (r.ps[b] < unambiguous_len) {
        paths[x.path].cost += o().ambiguous_branch_cost;
        ++x.branch_count_since_pair;
        if (x.branch_count_since_pair > o().max_branches_between_pairs) { ++st.exceeded_branch_limit; continue; }
      }
      if (!add_rejoins(x)) push_next_path(x);
    }
    return next_expansion(q);
  }

  // ---- output (:1023-1250)
  void output_assembly(const Rejoin& r) {
    bg_trace_assembly_rec a;
    a.assembly_id = allocate_assembly_id();  // tracer.cpp:1166
    std::string seq;
    std::set<uint32_t> ids;
    for (int cur = r.path; cur >= 0; cur = paths[cur].prev) {
      Path& p = paths[cur];
      seq += rev_comp_str(p.seq);
      p.part_of_assembly = true;
      ids.insert(p.seen_read_ids.begin(), p.seen_read_ids.end());
      p.seen_read_ids.clear();
    }
    a.rc_read_ids.assign(ids.begin(), ids.end());
    a.left_offset = int64_t(cur_offset) - int64_t(left_anchor.size());
    a.left_anchor_len = int(left_anchor.size());
    if (r.anchor_drop) {
      a.anchor_drop = true;
      a.right_offset = int32_t(cur_offset + seq.size() * o().anchor_drop_size_multiplier);
      a.right_anchor_len = 0;
      a.score += o().anchor_drop_score;
    } else {
      a.right_offset = r.right_offset;
      a.right_anchor_len = r.right_anchor_len;
    }
    a.min_overlap = paths[r.path].min_overlap;
    a.trace_steps = unsigned(search_steps);
    a.left_anchor_ambiguous_bases = ref_ambiguous_bases;
    a.seq = rev_comp_str(seq);
    close_assembly(&a);
    ++st.output_count;
    out.push_back(std::move(a));
  }

  bool dead_end_less(int ia, int ib) const {  // dead_end_sorter (:1062-1097)
    for (;;) {
      const Path &a = paths[ia], &b = paths[ib];
      if (a.min_overlap != b.min_overlap) return a.min_overlap > b.min_overlap;
      if (a.cost != b.cost) return a.cost < b.cost;
      if (a.range.b != b.range.b || a.range.e != b.range.e || a.range.s != b.range.s) {
        if (a.range.s != b.range.s) return a.range.s > b.range.s;
        return a.range.b < b.range.b;
      }
      if (a.part_of_assembly != b.part_of_assembly) return b.part_of_assembly;
      if (a.seen_pairs != b.seen_pairs) return a.seen_pairs < b.seen_pairs;
      if (a.seen_read_ids != b.seen_read_ids) return a.seen_read_ids < b.seen_read_ids;
      if (a.seq != b.seq) {  // dna_sequence order = base codes; A < C < G < T as characters too
        return a.seq < b.seq;
      }
      if (a.prev == b.prev) return false;
      if ((a.prev >= 0) != (b.prev >= 0)) return (a.prev >= 0) < (b.prev >= 0);
      ia = a.prev;
      ib = b.prev;
    }
  }

  void output_rejoins() {
    unsigned output_count = 0;
    while (output_count < o().max_rejoins && !rejoins.empty()) {
      Rejoin r = rejoins.back();
      rejoins.pop_back();
      if (paths[r.path].part_of_assembly) continue;
      output_assembly(r);
      ++output_count;
    }
    if (output_count >= o().max_rejoins) return;
    const unsigned max_to_output = unsigned(o().max_rejoins) - output_count;
    std::vector<int> de = dead_ends;
    dead_ends.clear();
    std::stable_sort(de.begin(), de.end(), [this](int a, int b) { return dead_end_less(a, b); });
    output_count = 0;
    for (int p : de) {
      if (output_count >= max_to_output) return;
      if (paths[p].part_of_assembly) continue;
      Rejoin r;
      r.path = p;
      r.rejoin_cost = o().dead_end_cost + paths[p].cost;
      r.anchor_drop = true;
      output_assembly(r);
      ++output_count;
    }
  }
};

double now_seconds() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}

// every copy rides on the context's own stream and is waited for there: sessions on different streams do not meet on
// the null stream
int copy_sync(bg_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (!bytes) return BG_OK;
  BG_HIP(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

template <class T>
int download(bg_ctx* ctx, std::vector<T>& h, const T* d, size_t n) {
  h.resize(n);
  return copy_sync(ctx, h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
}

// The host side of a round is a microsecond or two per trace: threads only pay off on big rounds, and only a few of
// them (measured on a 256-core box, 1 Mb piece: host replay 0.29 s on one thread, 0.21 s on 16, 0.50 s on 64 when every
// round spawned its threads).  So: a persistent pool of at most 16 (BG_TRACER_THREADS overrides), each taking a
// contiguous share, and rounds below a few thousand traces per thread run on fewer threads or inline.
int g_tracer_threads = 0;

// BG_TRACER_POOLS=n: several pools, the sessions that trace side by side (bgbinary discovery runs 24) dealt over them.  One by
// default: with ten pools of 16 threads on the 256-thread host the chr20-shaped sample traced in 18.1 s instead of 12.0 --
// the process spends 112 s instead of 50 s in the kernel (its threads meet on the address-space lock: heap growth and page
// faults of the search states), so more host threads make the search slower, not faster.
thread_local int tl_pool = 0;
class HostPool {
 public:
  static int count() {
    static const int n = [] {
      const char* e = getenv("BG_TRACER_POOLS");
      int v = e ? atoi(e) : 1;
      return v < 1 ? 1 : (v > 12 ? 12 : v);
    }();
    return n;
  }
  static HostPool& get() {
    static HostPool* pools = new HostPool[size_t(count())];  // live as long as the process
    return pools[size_t(tl_pool) % size_t(count())];
  }
  int size() const { return int(workers_.size()) + 1; }
  // f(t) for t in [0, parts): part 0 on the calling thread, the rest on the workers
  void run(int parts, const std::function<void(int)>& f) {
    if (parts <= 1) { f(0); return; }
    // one job at a time: a second session tracing on another thread at the same moment runs its share inline
    // (callers that run several sessions side by side have their parallelism already)
    std::unique_lock<std::mutex> busy(job_mu_, std::try_to_lock);
    if (!busy.owns_lock()) {
      for (int t = 0; t < parts; ++t) f(t);
      return;
    }
    {
      std::unique_lock<std::mutex> l(mu_);
      fn_ = &f;
      parts_ = parts;
      left_ = parts - 1;
      ++gen_;
    }
    cv_work_.notify_all();
    f(0);
    std::unique_lock<std::mutex> l(mu_);
    cv_done_.wait(l, [&] { return left_ == 0; });
    fn_ = nullptr;
  }

 public:
  HostPool() {
    const char* e = getenv("BG_TRACER_THREADS");
    int nt = e ? atoi(e) : int(std::min(16u, std::thread::hardware_concurrency()));
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    g_tracer_threads = nt;
    for (int t = 1; t < nt; ++t) workers_.emplace_back([this, t] { loop(t); });
  }
  ~HostPool() {
    {
      std::unique_lock<std::mutex> l(mu_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (auto& th : workers_) th.join();
  }

 private:
  void loop(int t) {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      cv_work_.wait(l, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      if (t >= parts_) continue;  // this round uses fewer threads
      const std::function<void(int)>* f = fn_;
      l.unlock();
      (*f)(t);
      l.lock();
      if (--left_ == 0) cv_done_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, job_mu_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(int)>* fn_ = nullptr;
  int parts_ = 0, left_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

template <class F>
void parallel_over(size_t n, F f, size_t min_per_thread = 4096) {
  HostPool& pool = HostPool::get();
  int nt = pool.size();
  if (size_t(nt) > n / min_per_thread + 1) nt = int(n / min_per_thread + 1);
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
  const size_t per = (n + size_t(nt) - 1) / size_t(nt);
  pool.run(nt, [&](int t) {
    const size_t lo = size_t(t) * per, hi = std::min(n, lo + per);
    for (size_t i = lo; i < hi; ++i) f(i);
  });
}

void add_stats(bg_trace_stats* to, const bg_trace_stats& from) {
  const uint64_t* a = reinterpret_cast<const uint64_t*>(&from);
  uint64_t* b = reinterpret_cast<uint64_t*>(to);
  for (size_t i = 0; i < sizeof(bg_trace_stats) / 8; ++i) b[i] += a[i];
}

struct Extent { int64_t offset; std::string seq; };

int fail(const char* msg) { set_error("%s", msg); return BG_EINVAL; }

// tracer::assemble(start, limit) over an already oriented scaffold -> assemblies in output order
// A reference assembly that nobody is going to see (assemble_options::trace_reference_assemblies off): all pair_counter wants
// of it are its two offsets and its read ids -- no sequence, no record, no id, no trip through the anchor_dropper's queue.
// Twenty of twenty-one million assemblies of a chromosome are these.
struct LeanRef { int64_t left, right; uint32_t ids_at, n_ids; };
struct LeanOut {
  std::vector<LeanRef> refs;   // in trace order = by left offset
  std::vector<uint32_t> ids;
  void add(int64_t left, int64_t right, const uint32_t* p, size_t n) {
    refs.push_back(LeanRef{left, right, uint32_t(ids.size()), uint32_t(n)});
    ids.insert(ids.end(), p, p + n);
  }
};

int trace_chunk(World& w, const std::vector<Extent>& ext, int64_t start, int64_t limit,
                std::vector<bg_trace_assembly_rec>* out, LeanOut* lean = nullptr) {
  bg_ctx* ctx = w.ctx;
  const bg_trace_options& o = w.o;
  const int64_t max_read_len = w.ss->max_len;
  if (limit > start + int64_t(o.scaffold_split_size)) return fail("bg_push_trace: the range is longer than scaffold_split_size");
  const double t_tables = now_seconds();
  // ---- 1. read ahead over [skip, limit + read_ahead_distance]
  const int64_t skip = start - int64_t(o.max_pair_distance) - max_read_len;  // skip_ahead_to (:230)
  const int64_t ahead_limit = limit + int64_t(o.read_ahead_distance);         // no trace looks further
  w.position_entries.clear();
  w.position_info.clear();
  w.entry_positions.clear();
  w.rc_mate_read_positions.clear();
  bool tables_sorted = true;
  // the device's own copies of the two sorted tables, while ONE stretch made all of them (a piece inside one extent: the common
  // case): the never-branching traces read them where they were made instead of from a second upload
  DevBuf<uint64_t> kept_ep, kept_mp;
  int device_stretches = 0;
  int64_t prev_end = -1;
  for (const Extent& x : ext) {
    if (x.offset < prev_end) return fail("bg_push_trace: extents must be sorted and disjoint");
    prev_end = x.offset + int64_t(x.seq.size());
    int64_t lo = x.offset, hi = prev_end;  // bases [lo, hi) of this extent are walked
    bool dup_first = false;
    if (skip > lo && !ext.empty() && skip > ext.front().offset) {
      // skip_ahead_to only runs when the target lies beyond the scaffold's first base (:328-330)
      if (skip >= hi) continue;
      lo = skip;
      dup_first = true;  // landed inside the extent: the base is pushed twice (:341 and :362)
    }
    if (hi > ahead_limit) hi = ahead_limit;
    if (lo >= hi) continue;
    const uint64_t n = uint64_t(hi - lo);
    const double t_c0 = now_seconds();
    std::vector<uint8_t> codes(n);
    {
      static const std::array<uint8_t, 256> kCode = [] { std::array<uint8_t, 256> t; t.fill(3); t[uint8_t('A')] = 0; t[uint8_t('C')] = 1; t[uint8_t('G')] = 2; return t; }();
      const uint8_t* src = reinterpret_cast<const uint8_t*>(x.seq.data()) + size_t(lo - x.offset);
      for (uint64_t i = 0; i < n; ++i) codes[i] = kCode[src[i]];
    }
    DevBuf<uint8_t> d_codes;
    DevBuf<uint64_t> d_ent;
    BG_TRY(d_codes.alloc(ctx, n));
    BG_TRY(d_ent.alloc(ctx, n));
    BG_TRY(copy_sync(ctx, d_codes.p, codes.data(), n, hipMemcpyHostToDevice));
    const double t_c1 = now_seconds();
    BG_TRY(bg_ref_walk_entries(ctx, w.ss, d_codes.p, n, dup_first ? 1 : 0, d_ent.p));
    if (getenv("BG_TRACER_TIMING")) { BG_HIP(hipStreamSynchronize(ctx->stream)); }
    const double t_c2 = now_seconds();
    w.seconds_tab_codes += t_c1 - t_c0;
    w.seconds_tab_walk += t_c2 - t_c1;
    if (w.d_ref_map && n < (1ull << 24) && !getenv("BG_TRACER_HOST_TABLES")) {
      // the tables of this stretch on the device: the entries never leave it
      const size_t ep0 = w.entry_positions.size(), mp0 = w.rc_mate_read_positions.size();
      ++device_stretches;
      kept_ep.reset();
      kept_mp.reset();
      const bool first = device_stretches == 1 && ep0 == 0 && mp0 == 0;
      BG_TRY(tracer_read_tables(ctx, w.ss, w.rm, w.d_ref_map, o.ignore_ambiguous_ref_pairs != 0, d_ent.p, n, lo + 1, &w.position_entries,
                                &w.entry_positions, &w.rc_mate_read_positions, first ? &kept_ep : nullptr, first ? &kept_mp : nullptr,
                                &w.position_info));
      // (merging only while everything so far IS sorted: after a stretch that took the host path the final sort does it all --
      // the flag never goes back to true)
      if (tables_sorted) {
        if (ep0) std::inplace_merge(w.entry_positions.begin(), w.entry_positions.begin() + long(ep0), w.entry_positions.end());
        if (mp0) std::inplace_merge(w.rc_mate_read_positions.begin(), w.rc_mate_read_positions.begin() + long(mp0), w.rc_mate_read_positions.end());
      }
      w.seconds_tab_read += now_seconds() - t_c2;
      continue;
    }
    tables_sorted = false;
    kept_ep.reset();
    kept_mp.reset();
    std::vector<uint64_t> ent(n);
    BG_TRY(copy_sync(ctx, ent.data(), d_ent.p, n * 8, hipMemcpyDeviceToHost));
    // positions that are reads, and for those the (rc mate, offset) pairs: per-thread lists over slices of the extent
    {
      int nt = int(std::min<uint64_t>(uint64_t(HostPool::get().size()), n / 65536 + 1));
      std::vector<std::vector<std::pair<int32_t, uint64_t>>> pe(nt);
      std::vector<std::vector<std::pair<uint32_t, int32_t>>> mates(nt);
      parallel_over(size_t(nt), [&](size_t t) {
        const uint64_t a = n * t / uint64_t(nt), b = n * (t + 1) / uint64_t(nt);
        for (uint64_t i = a; i < b; ++i) {
          if (ent[i] == kNoEntry || !w.get_bit(ent[i])) continue;
          const int32_t off = int32_t(lo + int64_t(i) + 1);  // the offset after the base (:365)
          pe[t].emplace_back(off, ent[i]);
          if (w.pairing_skipped(ent[i])) continue;
          const auto reads = w.entry_to_index(ent[i]);
          for (uint32_t rid = reads.first; rid != reads.second; ++rid)
            if (w.has_mate(rid)) mates[t].emplace_back(w.get_rev_comp(w.get_mate(rid)), off);
        }
      }, 1);
      for (int t = 0; t < nt; ++t) {
        for (auto& y : pe[t]) {
          w.position_entries.push_back(y);
          w.position_info.push_back(uint16_t(w.sizes[y.second]) | uint16_t(w.match_count(y.second) > 1 ? 256 : 0));
          w.entry_positions.emplace_back(y.second, y.first);
        }
        w.rc_mate_read_positions.insert(w.rc_mate_read_positions.end(), mates[t].begin(), mates[t].end());
      }
    }
  }
  if (!tables_sorted) {
    std::sort(w.entry_positions.begin(), w.entry_positions.end());
    std::sort(w.rc_mate_read_positions.begin(), w.rc_mate_read_positions.end());
  }
  // ---- 2. the sequential state of the sweep (:236-296): which offsets trace, with which anchor
  struct SeedInfo { uint32_t ext; uint32_t anchor_from; };  // where the left anchor lies: ext[ext].seq[anchor_from, + size)
  std::vector<TraceSeed> seeds;
  std::vector<SeedInfo> infos;
  size_t pe = 0;
  int64_t cur_offset = 0;
  bool stop = false;
  for (size_t xi = 0; xi < ext.size(); ++xi) {
    const Extent& x = ext[xi];
    if (stop) break;
    cur_offset = x.offset;
    if (x.offset + int64_t(x.seq.size()) <= start) continue;
    if (x.offset > limit) break;
    size_t from = 0;
    if (x.offset + max_read_len < start) {
      from = size_t(start - x.offset - max_read_len);
      cur_offset += int64_t(from);
    }
    size_t anchor_from = from;  // left anchor = x.seq[anchor_from, i]
    unsigned ambiguous = 0;
    for (size_t i = from; i < x.seq.size(); ++i) {
      ++ambiguous;
      if (cur_offset >= limit) { stop = true; break; }
      ++cur_offset;
      if (cur_offset < start) continue;
      while (pe < w.position_entries.size() && w.position_entries[pe].first < cur_offset) ++pe;
      if (pe >= w.position_entries.size() || w.position_entries[pe].first != cur_offset) continue;
      const uint64_t e = w.position_entries[pe].second;
      const uint32_t size = w.position_info[pe] & 255u;
      if (i + 1 - anchor_from < size) continue;
      anchor_from = i + 1 - size;
      // tracer::trace (:477-503): the window check is always true here (start <= cur_offset < limit is implied by the
      // loop above except at cur_offset == limit)
      if (cur_offset < start || cur_offset >= limit) continue;
      const bool ref_ambiguous = (w.position_info[pe] & 256u) != 0;
      if (!ref_ambiguous) ambiguous = 0;
      TraceSeed sd;
      sd.entry = e;
      sd.cur_offset = int32_t(cur_offset);
      sd.size = size;
      sd.ref_ambiguous = ambiguous;
      sd.win_lo = int32_t(cur_offset - 1 - int64_t(o.max_pair_distance));
      sd.win_hi = int32_t(cur_offset - 1 + int64_t(o.read_ahead_distance));
      sd.pad = 0;
      seeds.push_back(sd);
      infos.push_back(SeedInfo{uint32_t(xi), uint32_t(anchor_from)});
    }
  }
  w.seconds_tables += now_seconds() - t_tables;
  if (getenv("BG_TRACER_TIMING"))
    fprintf(stderr, "bg_tracer_tables: codes + upload %.4f s, read-ahead walk %.4f s, tables %.4f s of %.4f s so far (%zu positions with reads, %zu mate positions)\n",
            w.seconds_tab_codes, w.seconds_tab_walk, w.seconds_tab_read, w.seconds_tables, w.position_entries.size(), w.rc_mate_read_positions.size());
  const size_t n_seeds = seeds.size();
  // ---- 2b. the traces that never branch, whole on the device (walk.hip k_trace_simple): all but six in a thousand follow the reference to the
  // next read and rejoin there.  What the kernel hands back as kSimpleHost runs on the state machine below, from its start.
  static const bool simple_off = [] { const char* e = getenv("BG_TRACER_DEVICE_SIMPLE"); return e && e[0] == '0'; }();
  const bool simple = !simple_off && w.dev_tables && w.d_ref_map && n_seeds && o.max_pairs_per_read <= uint64_t(kSimpleIds) && o.max_rejoins >= 1 &&
                      w.n_reads < (size_t(1) << 32);
  std::vector<TraceSimpleOut> simple_out;
  if (simple) {
    const double t_dev = now_seconds();
    DevBuf<TraceSeed> d_seeds;
    DevBuf<TraceSimpleOut> d_out;
    DevBuf<uint8_t> d_ep, d_mp;
    BG_TRY(d_seeds.alloc(ctx, n_seeds));
    BG_TRY(d_out.alloc(ctx, n_seeds));
    static_assert(sizeof(std::pair<uint64_t, int32_t>) == 16 && sizeof(std::pair<uint32_t, int32_t>) == 8, "table rows as the kernel reads them");
    // (kept: the one stretch's device tables are the host tables, row for row)
    const bool kept = device_stretches == 1 && tables_sorted && kept_ep.p && kept_ep.n >= w.entry_positions.size() * 2 &&
                      (w.rc_mate_read_positions.empty() || (kept_mp.p && kept_mp.n >= w.rc_mate_read_positions.size()));
    BG_HIP(hipMemcpyAsync(d_seeds.p, seeds.data(), n_seeds * sizeof(TraceSeed), hipMemcpyHostToDevice, ctx->stream));
    if (!kept) {
      BG_TRY(d_ep.alloc(ctx, w.entry_positions.size() * 16));
      BG_TRY(d_mp.alloc(ctx, w.rc_mate_read_positions.size() * 8));
      if (!w.entry_positions.empty())
        BG_HIP(hipMemcpyAsync(d_ep.p, w.entry_positions.data(), w.entry_positions.size() * 16, hipMemcpyHostToDevice, ctx->stream));
      if (!w.rc_mate_read_positions.empty())
        BG_HIP(hipMemcpyAsync(d_mp.p, w.rc_mate_read_positions.data(), w.rc_mate_read_positions.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    TraceSimpleTables tb;
    tb.ref_map = w.d_ref_map;
    tb.word_rank = w.dev_tables->word_rank;
    tb.group_start = w.dev_tables->group_start;
    tb.mate_loop = w.dev_tables->mate_loop;
    tb.is_forward = w.dev_tables->is_forward;
    tb.entry_positions = kept ? reinterpret_cast<const uint8_t*>(kept_ep.p) : d_ep.p;
    tb.n_entry_positions = w.entry_positions.size();
    tb.mate_positions = kept ? reinterpret_cast<const uint8_t*>(kept_mp.p) : d_mp.p;
    tb.n_mate_positions = w.rc_mate_read_positions.size();
    BG_TRY(tracer_simple_batch(ctx, w.ss, w.rm, tb, o, d_seeds.p, n_seeds, d_out.p));
    simple_out.resize(n_seeds);
    BG_TRY(copy_sync(ctx, simple_out.data(), d_out.p, n_seeds * sizeof(TraceSimpleOut), hipMemcpyDeviceToHost));
    w.seconds_device += now_seconds() - t_dev;
    ++w.stats.device_rounds;
    w.stats.device_queries += n_seeds;
  }
  // the traces left to the host, in seed order
  std::vector<uint32_t> host_seed;
  for (size_t i = 0; i < n_seeds; ++i)
    if (!simple || simple_out[i].status == kSimpleHost) host_seed.push_back(uint32_t(i));
  if (simple && getenv("BG_TRACER_TIMING")) {
    size_t why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n_seeds; ++i)
      if (simple_out[i].status == kSimpleHost) ++why[(simple_out[i].counters >> 20) & 7];
    fprintf(stderr, "trace_chunk: %zu of %zu traces to the host: a branch %zu, more than %d steps %zu, more than %d read ids %zu, the read lies nowhere ahead %zu, "
                    "ambiguous rejoin %zu\n", host_seed.size(), n_seeds, why[1], kSimpleSteps, why[2], kSimpleIds, why[3], why[4], why[5]);
  }
  std::vector<std::unique_ptr<Trace>> traces(host_seed.size());
  {
    const double t_mk = now_seconds();
    parallel_over(host_seed.size(), [&](size_t k) {
      const TraceSeed& sd = seeds[host_seed[k]];
      const SeedInfo& in = infos[host_seed[k]];
      std::unique_ptr<Trace> t(new Trace);
      t->w = &w;
      t->cur_offset = sd.cur_offset;
      t->cur_range.b = sd.entry; t->cur_range.e = sd.entry + 1; t->cur_range.s = sd.size;
      t->left_anchor = ext[in.ext].seq.substr(in.anchor_from, sd.size);
      t->ref_ambiguous_bases = sd.ref_ambiguous;
      t->win_lo = sd.win_lo;
      t->win_hi = sd.win_hi;
      traces[k] = std::move(t);
    });
    w.seconds_start += now_seconds() - t_mk;
  }
  std::vector<std::vector<bg_trace_assembly_rec>> host_out(host_seed.size());
  // ---- 3. all traces in lock step
  const uint32_t step_cap = 64;
  const size_t wave = 1 << 16;
  for (size_t w0 = 0; w0 < traces.size(); w0 += wave) {
    const size_t w1 = std::min(traces.size(), w0 + wave);
    // start / resume of different traces are independent: they run on the host's threads, each over a contiguous share
    std::vector<Trace*> live;
    std::vector<bg_trace_query> q;
    {
      const size_t n = w1 - w0;
      std::vector<bg_trace_query> qs(n);
      std::vector<uint8_t> pending(n, 0);
      const double t_start = now_seconds();
      parallel_over(n, [&](size_t i) { pending[i] = traces[w0 + i]->start(&qs[i]) ? 1 : 0; });
      w.seconds_start += now_seconds() - t_start;
      for (size_t i = 0; i < n; ++i)
        if (pending[i]) { live.push_back(traces[w0 + i].get()); q.push_back(qs[i]); }
    }
    DevBuf<uint8_t> dq, drs;
    // device buffers of the first (largest) round serve every later one; a round's replies and step lists lie side by side
    // (one download), behind one launch that is not waited for on its own
    const size_t reply_bytes_max = (live.size() * sizeof(bg_trace_reply) + 7) & ~size_t(7);
    BG_TRY(dq.alloc(ctx, live.size() * sizeof(bg_trace_query)));
    BG_TRY(drs.alloc(ctx, reply_bytes_max + live.size() * step_cap * 4));
    while (!live.empty()) {
      const size_t n = live.size();
      const double t_dev = now_seconds();
      const size_t reply_bytes = (n * sizeof(bg_trace_reply) + 7) & ~size_t(7);
      BG_TRY(w.pin_q.ensure(n * sizeof(bg_trace_query)));
      BG_TRY(w.pin_r.ensure(reply_bytes + n * step_cap * 4));
      memcpy(w.pin_q.p, q.data(), n * sizeof(bg_trace_query));
      const bg_trace_reply* replies = static_cast<const bg_trace_reply*>(w.pin_r.p);
      const uint32_t* steps = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(w.pin_r.p) + reply_bytes);
      // A round is a few hundred traces: the kernel reads its queries from and writes its replies to the page-locked host
      // buffers themselves (a launch and a wait) instead of a copy either way around it (four calls into a runtime that
      // sixteen sessions share).  BG_TRACER_ZEROCOPY=0, or a round of more than 4096 traces: through device buffers.
      static const bool zc_off = [] { const char* e = getenv("BG_TRACER_ZEROCOPY"); return e && e[0] == '0'; }();
      if (!zc_off && n <= 4096) {
        BG_TRY(tracer_extend_launch(ctx, w.ss, w.rm, n, static_cast<const bg_trace_query*>(w.pin_q.p), step_cap,
                                    static_cast<bg_trace_reply*>(w.pin_r.p),
                                    reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(w.pin_r.p) + reply_bytes)));
      } else {
        BG_HIP(hipMemcpyAsync(dq.p, w.pin_q.p, n * sizeof(bg_trace_query), hipMemcpyHostToDevice, ctx->stream));
        BG_TRY(tracer_extend_launch(ctx, w.ss, w.rm, n, reinterpret_cast<const bg_trace_query*>(dq.p), step_cap,
                                    reinterpret_cast<bg_trace_reply*>(drs.p), reinterpret_cast<uint32_t*>(drs.p + reply_bytes)));
        BG_HIP(hipMemcpyAsync(w.pin_r.p, drs.p, reply_bytes + n * step_cap * 4, hipMemcpyDeviceToHost, ctx->stream));
      }
      BG_HIP(hipStreamSynchronize(ctx->stream));
      w.seconds_device += now_seconds() - t_dev;
      ++w.stats.device_rounds;
      w.stats.device_queries += n;
      const double t_host = now_seconds();
      std::vector<bg_trace_query> qs(n);
      std::vector<uint8_t> pending(n, 0);
      parallel_over(n, [&](size_t i) { pending[i] = live[i]->resume(replies[i], steps + i * step_cap, &qs[i]) ? 1 : 0; });
      std::vector<Trace*> next_live;
      std::vector<bg_trace_query> next_q;
      for (size_t i = 0; i < n; ++i)
        if (pending[i]) { next_live.push_back(live[i]); next_q.push_back(qs[i]); }
      live.swap(next_live);
      q.swap(next_q);
      w.seconds_host += now_seconds() - t_host;
    }
    for (size_t i = w0; i < w1; ++i) {
      add_stats(&w.stats, traces[i]->st);
      host_out[i] = std::move(traces[i]->out);
    }
    const double t_down = now_seconds();
    for (size_t i = w0; i < w1; ++i) traces[i].reset();  // (in order on this thread: frees from other threads contend in malloc)
    w.seconds_teardown += now_seconds() - t_down;
  }
  // ---- 4. the assemblies in the order of their traces: the device's (tracer::output_assembly, :1161-1215, from the fields
  // k_trace_simple left) and the state machine's
  {
    const double t_out = now_seconds();
    size_t hk = 0;
    for (size_t i = 0; i < n_seeds; ++i) {
      if (hk < host_seed.size() && host_seed[hk] == i) {
        for (auto& a : host_out[hk]) {
          if (lean && a.matches_reference && !a.anchor_drop) lean->add(a.left_offset, a.right_offset, a.rc_read_ids.data(), a.rc_read_ids.size());
          else out->push_back(std::move(a));
        }
        ++hk;
        continue;
      }
      const TraceSimpleOut& so = simple_out[i];
      const uint32_t c = so.counters;
      w.stats.ref_reads += (c & kSimRefReads) ? 1 : 0;
      w.stats.ambiguous_ref_reads += (c & kSimAmbiguousRefReads) ? 1 : 0;
      w.stats.step_count += (c & kSimStepCount) ? 1 : 0;
      w.stats.output_count += (c & kSimOutput) ? 1 : 0;
      w.stats.matched_pairs += (c & kSimMatchedPairs) ? 1 : 0;
      w.stats.max_branch_cost += (c & kSimMaxBranchCost) ? 1 : 0;
      w.stats.too_many_ambiguous_bases += (c & kSimTooManyAmbiguousBases) ? 1 : 0;
      w.stats.too_many_steps += (c & kSimTooManySteps) ? 1 : 0;
      w.stats.too_far_without_pair += (c & kSimTooFarWithoutPair) ? 1 : 0;
      w.stats.prune_ambiguous_ref += (c & kSimPruneAmbiguousRef) ? 1 : 0;
      w.stats.found_pairs += so.found_pairs;
      if (so.status != kSimpleAssembly) continue;
      const TraceSeed& sd = seeds[i];
      const SeedInfo& in = infos[i];
      if (lean) {  // close_assembly's matches_reference from the lengths alone
        const int64_t lo = int64_t(sd.cur_offset) - int64_t(sd.size), ro = so.right_offset;
        if (int64_t(sd.size) + int64_t(so.right_anchor_len) >= ro - lo && ro - lo == int64_t(sd.size) + int64_t(so.n_steps)) {
          lean->add(lo, ro, so.ids, so.n_ids);
          continue;
        }
      }
      bg_trace_assembly_rec a;
      a.assembly_id = allocate_assembly_id();  // tracer.cpp:1166
      a.seq.reserve(size_t(sd.size) + so.n_steps);
      a.seq.assign(ext[in.ext].seq, in.anchor_from, sd.size);
      for (uint32_t k = 0; k < so.n_steps; ++k) a.seq.push_back("ACGT"[so.bases[k] & 3]);
      a.rc_read_ids.assign(so.ids, so.ids + so.n_ids);  // ascending read ids of one entry: the std::set of :1168-1177
      a.left_offset = int64_t(sd.cur_offset) - int64_t(sd.size);
      a.left_anchor_len = int(sd.size);
      a.right_offset = so.right_offset;
      a.right_anchor_len = int(so.right_anchor_len);
      a.min_overlap = so.min_overlap;
      a.trace_steps = 1;
      a.left_anchor_ambiguous_bases = sd.ref_ambiguous;
      close_assembly(&a);
      out->push_back(std::move(a));
    }
    w.seconds_host += now_seconds() - t_out;
  }
  w.stats.traces += n_seeds;
  return BG_OK;
}

// pair_counter.cpp (calc_score without the coverage terms: the tracer computes no coverage)
// read id -> how many live assemblies carry it: the ids in flight are the reads of a couple of thousand bases, so a
// small open-addressing table that lives in cache (a std::unordered_map spent 0.2 s per 1 Mb piece on its nodes)
struct IdCounter {
  std::vector<uint32_t> key, cnt;
  size_t mask = 0, used = 0;
  IdCounter() { reset(1 << 12); }
  void reset(size_t cap) {
    key.assign(cap, ~0u);
    cnt.assign(cap, 0);
    mask = cap - 1;
    used = 0;
  }
  size_t slot_of(uint32_t k) const {
    size_t s = (size_t(k) * 0x9E3779B1u) & mask;
    while (key[s] != ~0u && key[s] != k) s = (s + 1) & mask;
    return s;
  }
  void inc(uint32_t k) {
    size_t s = slot_of(k);
    if (key[s] == ~0u) {
      if ((used + 1) * 2 > key.size()) { regrow(); s = slot_of(k); }
      key[s] = k;
      ++used;
    }
    ++cnt[s];
  }
  void dec(uint32_t k) {
    const size_t s = slot_of(k);
    if (key[s] == k && cnt[s]) --cnt[s];
  }
  bool has(uint32_t k) const {
    const size_t s = slot_of(k);
    return key[s] == k && cnt[s] != 0;
  }
  void regrow() {  // drop the ids that went back to zero; double when the live ones fill a quarter
    std::vector<std::pair<uint32_t, uint32_t>> live;
    for (size_t i = 0; i < key.size(); ++i)
      if (key[i] != ~0u && cnt[i]) live.emplace_back(key[i], cnt[i]);
    size_t cap = key.size();
    while (live.size() * 4 > cap) cap *= 2;
    reset(cap);
    for (auto& kv : live) {
      const size_t s = slot_of(kv.first);
      key[s] = kv.first;
      cnt[s] = kv.second;
      ++used;
    }
  }
};

struct PairCounter {
  const World& w;
  std::vector<bg_trace_assembly_rec>* sink;
  int64_t cur_offset = 0;
  // pair_counter.h: m_left / m_right are multimaps that only ever receive non-decreasing keys here (assemblies
  // arrive by left offset, leave `active` by right offset), so they are queues
  std::deque<std::pair<int64_t, std::vector<uint32_t>>> left;
  std::deque<std::pair<int64_t, bg_trace_assembly_rec>> right;
  std::multimap<int64_t, bg_trace_assembly_rec> active;
  IdCounter left_ids, right_ids;
  int64_t last_left = INT64_MIN;
  // the reference assemblies that only lend their read ids (LeanRef), in queues of their own beside the three above: what
  // happens to an assembly at an offset does not depend on the order among the assemblies that share the offset (entering
  // `active` reads left_ids and takes from right_ids, leaving it reads right_ids and adds to left_ids, expiry takes from
  // left_ids: three passes one after the other), so the two kinds only have to agree on the offsets
  struct Lean { int64_t left, right; const uint32_t* ids; uint32_t n; };
  struct LaterRight { bool operator()(const Lean& a, const Lean& b) const { return a.right > b.right; } };
  std::deque<Lean> right_lean, left_lean;
  std::priority_queue<Lean, std::vector<Lean>, LaterRight> active_lean;
  bool add_lean(int64_t left, int64_t right_offset, const uint32_t* ids, uint32_t n) {
    if (left < last_left) return false;
    last_left = left;
    while (advance_some(left - int64_t(w.o.max_pair_distance))) {}
    for (uint32_t k = 0; k < n; ++k) right_ids.inc(ids[k]);
    right_lean.push_back(Lean{left, right_offset, ids, n});
    return true;
  }

  std::vector<uint32_t> matches(const bg_trace_assembly_rec& a, const IdCounter& ids, bool forward) const {
    std::vector<uint32_t> out;
    for (uint32_t rid : a.rc_read_ids) {
      if (!w.has_mate(rid) || bool(w.is_forward[rid]) != forward) continue;
      if (ids.has(w.get_rev_comp(w.get_mate(rid)))) out.push_back(rid);
    }
    return out;
  }
  bool add(bg_trace_assembly_rec a) {
    if (a.left_offset < last_left) return false;  // pair_counter expects left_offset order (pair_counter.cpp: set_expected_order)
    last_left = a.left_offset;
    while (advance_some(a.left_offset - int64_t(w.o.max_pair_distance))) {}
    for (uint32_t rid : a.rc_read_ids) right_ids.inc(rid);
    const int64_t lo = a.left_offset;
    right.emplace_back(lo, std::move(a));
    return true;
  }
  void flush() { while (advance_some(0x7fffffff)) {} }
  bool advance_some(int64_t target) {
    if (cur_offset >= target) return false;
    int64_t amount = target - cur_offset;
    if (!left.empty()) amount = std::min(amount, left.front().first + int64_t(w.o.max_pair_distance) - cur_offset + 1);
    if (!active.empty()) amount = std::min(amount, active.begin()->first - cur_offset + 1);
    if (!right.empty()) amount = std::min(amount, right.front().first - cur_offset + 1);
    if (!left_lean.empty()) amount = std::min(amount, left_lean.front().right + int64_t(w.o.max_pair_distance) - cur_offset + 1);
    if (!active_lean.empty()) amount = std::min(amount, active_lean.top().right - cur_offset + 1);
    if (!right_lean.empty()) amount = std::min(amount, right_lean.front().left - cur_offset + 1);
    cur_offset += amount;
    while (!right.empty() && right.front().first < cur_offset) {
      bg_trace_assembly_rec a = std::move(right.front().second);
      right.pop_front();
      for (uint32_t rid : a.rc_read_ids) right_ids.dec(rid);
      // (a reference assembly that is not going to be output only lends its read ids to the others: its own
      // matches would be thrown away with it below)
      const bool unused = a.matches_reference && !w.o.trace_reference_assemblies;
      if (!unused) a.left_pair_matches = matches(a, left_ids, w.o.forward_pairs_face_inward != 0);
      const int64_t ro = a.right_offset;
      active.emplace(ro, std::move(a));
    }
    while (!right_lean.empty() && right_lean.front().left < cur_offset) {
      const Lean x = right_lean.front();
      right_lean.pop_front();
      for (uint32_t k = 0; k < x.n; ++k) right_ids.dec(x.ids[k]);
      active_lean.push(x);
    }
    while (!active.empty() && active.begin()->first < cur_offset) {
      bg_trace_assembly_rec a = std::move(active.begin()->second);
      active.erase(active.begin());
      const bool unused = a.matches_reference && !w.o.trace_reference_assemblies;
      if (!unused) a.right_pair_matches = matches(a, right_ids, w.o.forward_pairs_face_inward == 0);
      std::vector<uint32_t> ids = std::move(a.rc_read_ids);
      a.rc_read_ids.clear();
      for (uint32_t rid : ids) left_ids.inc(rid);
      left.emplace_back(a.right_offset, std::move(ids));
      if (a.matches_reference && !w.o.trace_reference_assemblies) continue;
      a.score += int64_t(a.min_overlap) * w.o.min_overlap_score;
      a.score += int64_t(a.left_pair_matches.size() + a.right_pair_matches.size()) * w.o.pair_match_score;
      sink->push_back(std::move(a));
    }
    while (!active_lean.empty() && active_lean.top().right < cur_offset) {
      const Lean x = active_lean.top();
      active_lean.pop();
      for (uint32_t k = 0; k < x.n; ++k) left_ids.inc(x.ids[k]);
      left_lean.push_back(x);
    }
    while (!left.empty() && left.front().first < cur_offset - int64_t(w.o.max_pair_distance)) {
      for (uint32_t rid : left.front().second) left_ids.dec(rid);
      left.pop_front();
    }
    while (!left_lean.empty() && left_lean.front().right < cur_offset - int64_t(w.o.max_pair_distance)) {
      const Lean& x = left_lean.front();
      for (uint32_t k = 0; k < x.n; ++k) left_ids.dec(x.ids[k]);
      left_lean.pop_front();
    }
    return true;
  }
};

// reverse_assembly_in_place (assemble.cpp:933-984): with_read_ids for an assembly that still carries its rc_read_ids
// (an anchor drop on its way to the pop tracer; pair_counter has emptied them in everything it lets out)
void reverse_rec(const World& w, bg_trace_assembly_rec* a, int64_t end_pos, bool with_read_ids) {
  a->seq = rev_comp_str(a->seq);
  std::swap(a->left_anchor_len, a->right_anchor_len);
  const int64_t lo = a->left_offset, ro = a->right_offset;
  a->left_offset = end_pos - ro;
  a->right_offset = end_pos - lo;
  std::swap(a->left_pair_matches, a->right_pair_matches);
  for (auto* m : {&a->left_pair_matches, &a->right_pair_matches})
    for (uint32_t& rid : *m) rid = w.get_rev_comp(rid);
  if (with_read_ids) {
    for (uint32_t& rid : a->rc_read_ids) rid = w.get_rev_comp(rid);
    std::sort(a->rc_read_ids.begin(), a->rc_read_ids.end());  // read_id_set: ascending
  }
}

}  // namespace

// reversable_tracer::assemble(start, limit) (reversable_tracer.cpp:76-145) with the push tracer.
static std::atomic<int> g_next_pool{0};
// The pieces of a scaffold come with the same extents, to every session: the checked copy and its reverse complement are made
// once per process and recognised by a hash of the bases (a whole chromosome per call otherwise: 35 ms to copy and check,
// 100 ms to reverse -- per work item, and again in the pop tracer).
struct ExtSet {
  std::vector<uint64_t> hashes;
  int64_t end_pos = -1;
  std::vector<Extent> ext;
  // where the caller's copy lay and what a sample of it looked like (session_extents' quick way)
  std::vector<const char*> src_ptr;
  std::vector<uint64_t> src_len;
  std::vector<int64_t> src_off;
  uint64_t sample = 0;
  std::atomic<bool> src_noted{false};  // (the set is shared between sessions: written once under g_ext_mu, read after this says so)
  const std::vector<Extent>& rc() {  // scaffold::rev_comp (scaffold.cpp:163-175)
    std::call_once(rc_once_, [&] {
      for (auto it = ext.rbegin(); it != ext.rend(); ++it)
        ext_rc_.push_back(Extent{end_pos - it->offset - int64_t(it->seq.size()), rev_comp_str(it->seq)});
    });
    return ext_rc_;
  }
 private:
  std::once_flag rc_once_;
  std::vector<Extent> ext_rc_;
};
std::mutex g_ext_mu;
std::shared_ptr<ExtSet> g_ext_last;  // the scaffold the process is working on

// The session's extents for this call: its own from the last call, the process's, or a fresh checked copy.
// 32 bytes out of every 4096 of the extents, their ends whole
static uint64_t extents_sample(uint64_t n_extents, const char* const* ext_seq, const uint64_t* ext_len) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix = [&](const char* p, uint64_t n) {
    uint64_t j = 0;
    for (; j + 8 <= n; j += 8) { uint64_t x; memcpy(&x, p + j, 8); h = (h ^ x) * 0x100000001b3ull; h ^= h >> 29; }
    for (; j < n; ++j) h = (h ^ uint64_t(uint8_t(p[j]))) * 0x100000001b3ull;
  };
  for (uint64_t i = 0; i < n_extents; ++i) {
    const uint64_t n = ext_len[i];
    if (n <= 8192) { mix(ext_seq[i], n); continue; }
    mix(ext_seq[i], 4096);
    for (uint64_t j = 4096; j + 4096 < n; j += 4096) mix(ext_seq[i] + j, 32);
    mix(ext_seq[i] + n - 4096, 4096);
  }
  return h;
}

int session_extents(std::shared_ptr<ExtSet>* mine, const char* who, uint64_t n_extents, const int64_t* ext_offset,
                    const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos) {
  // The quick way (every work item of a scaffold passes the same extents: hashing a chromosome is 14-18 ms a call, twice a work
  // item): the caller's extents lie where the session's copy was made from, as long, at the same offsets, and a sample of their
  // bytes -- 32 of every 4096, the ends whole -- is what it was.  The extents a session is given must not change while it uses
  // them; anything else about them is checked in full below, as before.
  if (*mine && (*mine)->src_noted.load(std::memory_order_acquire) && (*mine)->end_pos == end_pos && (*mine)->src_ptr.size() == n_extents) {
    const ExtSet& e = **mine;
    bool same_place = true;
    for (uint64_t i = 0; i < n_extents && same_place; ++i)
      same_place = e.src_ptr[i] == ext_seq[i] && e.src_len[i] == ext_len[i] && e.src_off[i] == ext_offset[i];
    if (same_place && e.sample == extents_sample(n_extents, ext_seq, ext_len)) return BG_OK;
  }
  std::vector<uint64_t> hashes(n_extents);
  int64_t calc_end = 0, prev_end = INT64_MIN;
  for (uint64_t i = 0; i < n_extents; ++i) {
    // four interleaved chains (one multiply chain alone runs at its latency: 13 ms for a chromosome, per call)
    uint64_t hh[4] = {0xcbf29ce484222325ull ^ uint64_t(ext_offset[i]) ^ (ext_len[i] << 20), 0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full,
                      0x165667b19e3779f9ull};
    const char* p = ext_seq[i];
    uint64_t j = 0;
    for (; j + 32 <= ext_len[i]; j += 32) {
      uint64_t x[4];
      memcpy(x, p + j, 32);
      for (int k = 0; k < 4; ++k) {
        hh[k] = (hh[k] ^ x[k]) * 0x100000001b3ull;
        hh[k] ^= hh[k] >> 29;
      }
    }
    uint64_t h = hh[0];
    for (int k = 1; k < 4; ++k) h = (h ^ hh[k]) * 0x100000001b3ull + (h >> 31);
    for (; j < ext_len[i]; ++j) h = (h ^ uint64_t(uint8_t(p[j]))) * 0x100000001b3ull;
    hashes[i] = h;
    if (ext_offset[i] < prev_end) { set_error("%s: extents must be sorted and disjoint", who); return BG_EINVAL; }
    prev_end = ext_offset[i] + int64_t(ext_len[i]);
    calc_end = std::max<int64_t>(calc_end, prev_end);
  }
  if (end_pos < calc_end) { set_error("%s: end_pos lies before the end of the last extent", who); return BG_EINVAL; }
  auto same = [&](const std::shared_ptr<ExtSet>& e) { return e && e->end_pos == end_pos && e->hashes == hashes; };
  // (the sets are shared between sessions: where THIS caller's copy lies is noted under the lock, and only a session whose
  // caller passes the same place takes the quick way)
  auto note_source = [&](ExtSet* e) {
    std::lock_guard<std::mutex> l(g_ext_mu);
    if (!e->src_noted.load(std::memory_order_relaxed) && n_extents) {
      e->src_ptr.assign(ext_seq, ext_seq + n_extents);
      e->src_len.assign(ext_len, ext_len + n_extents);
      e->src_off.assign(ext_offset, ext_offset + n_extents);
      e->sample = extents_sample(n_extents, ext_seq, ext_len);
      e->src_noted.store(true, std::memory_order_release);
    }
  };
  if (same(*mine)) { note_source(mine->get()); return BG_OK; }
  {
    std::unique_lock<std::mutex> l(g_ext_mu);
    if (same(g_ext_last)) { *mine = g_ext_last; l.unlock(); note_source(mine->get()); return BG_OK; }
  }
  std::shared_ptr<ExtSet> fresh(new ExtSet);
  for (uint64_t i = 0; i < n_extents; ++i) {
    fresh->ext.push_back(Extent{ext_offset[i], std::string(ext_seq[i], ext_len[i])});
    if (!all_acgt(ext_seq[i], ext_len[i])) { set_error("%s: extents hold ACGT only", who); return BG_EINVAL; }
  }
  fresh->hashes = hashes;
  fresh->end_pos = end_pos;
  {
    std::lock_guard<std::mutex> l(g_ext_mu);
    if (same(g_ext_last)) fresh = g_ext_last;  // (another session was quicker)
    else g_ext_last = fresh;
  }
  *mine = fresh;
  note_source(mine->get());
  return BG_OK;
}

// The session's helper thread: runs the host steps behind the tracer of work item i (bg_tracer_trace_async) while the session's
// own thread is in the tracer of item i + 1.  Jobs run in the order they were handed in.
struct Helper {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::packaged_task<int()>> jobs;
  bool stop = false;
  std::future<int> run(std::packaged_task<int()> job) {
    std::future<int> f = job.get_future();
    std::lock_guard<std::mutex> l(mu);
    if (!th.joinable())
      th = std::thread([this] {
        for (;;) {
          std::packaged_task<int()> j;
          {
            std::unique_lock<std::mutex> l2(mu);
            cv.wait(l2, [&] { return stop || !jobs.empty(); });
            if (jobs.empty()) return;
            j = std::move(jobs.front());
            jobs.pop_front();
          }
          j();
        }
      });
    jobs.push_back(std::move(job));
    cv.notify_one();
    return f;
  }
  ~Helper() {
    { std::lock_guard<std::mutex> l(mu); stop = true; }
    cv.notify_all();
    if (th.joinable()) th.join();  // (after the jobs that are still queued)
  }
};

struct bg_tracer {
  World w;
  int pool = g_next_pool.fetch_add(1);  // which of the host pools this session's big rounds use
  // the scaffold of the last work item (see bg_tracer_trace)
  std::shared_ptr<ExtSet> exts;
  Helper helper;  // (destroyed first: its jobs read w)
};

// The host copies a tracer needs of the seqset, the readmap and the ref_map are made once per session
// (assemble_options::seqset / readmap / rmap of a trace_ref object); bg_tracer_trace is one work item of trace_ref.
extern "C" int bg_tracer_open(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const uint8_t* d_ref_map,
                              const bg_trace_options* options, bg_tracer** out) {
  if (!ctx || !ss || !rm || !out) return fail("bg_tracer_open: null argument");
  if (ss->n && !d_ref_map) return fail("bg_tracer_open: the ref_map is required (bg_ref_map_build)");
  if (rm->n_entries != ss->n) return fail("bg_tracer_open: the readmap belongs to another seqset");
  if (rm->pair_mode) return fail("bg_tracer_open: the readmap has no mate loop table");
  if (ss->max_len > 255) return fail("bg_tracer_open: reads longer than 255 bases");
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_tracer> t(new bg_tracer);
  World& w = t->w;
  w.ctx = ctx;
  w.ss = ss;
  w.rm = rm;
  if (options) w.o = *options; else bg_trace_options_default(&w.o);
  memset(&w.stats, 0, sizeof(w.stats));
  w.n_entries = ss->n;
  w.d_ref_map = d_ref_map;
  {
    // The readmap's columns come narrow and shared (readmap_tables: made on the device, downloaded once, the same object the
    // coverage session reads); the seqset's sizes and the reference map are this session's, through the pinned double buffer.
    std::shared_ptr<HostTables> h(new HostTables);
    BG_TRY(readmap_tables(ctx, ss, rm, &h->rt));
    h->sizes.resize(ss->n);
    h->ref_map.resize(ss->n);
    if (ss->n) {
      BG_TRY(staged_download(ctx, h->sizes.data(), ss->sizes.p, ss->n, 0));
      BG_TRY(staged_download(ctx, h->ref_map.data(), d_ref_map, ss->n, 0));
    }
    w.tables = h;
    // the device copies serve the never-branching traces (k_trace_simple); BG_TRACER_DEVICE_SIMPLE=0: not used
    const char* off = getenv("BG_TRACER_DEVICE_SIMPLE");
    if (!(off && off[0] == '0')) {
      std::shared_ptr<DeviceTables> d(new DeviceTables);
      d->word_rank = h->rt->d_word_rank;
      d->group_start = h->rt->d_group_start;
      d->mate_loop = h->rt->d_mate_loop;
      d->is_forward = h->rt->d_is_forward;
      w.dev_tables = d;
    }
  }
  w.bind_tables();
  *out = t.release();
  return BG_OK;
}

// Another session on the same seqset / readmap / ref_map for another host thread: its own context (stream, allocator)
// and scratch, the read-only host tables shared with `from` (no second copy: less memory, warmer caches).
extern "C" int bg_tracer_clone(const bg_tracer* from, bg_ctx* ctx, bg_tracer** out) {
  if (!from || !ctx || !out) return fail("bg_tracer_clone: null argument");
  std::unique_ptr<bg_tracer> t(new bg_tracer);
  World& w = t->w;
  w.ctx = ctx;
  w.ss = from->w.ss;
  w.rm = from->w.rm;
  w.o = from->w.o;
  memset(&w.stats, 0, sizeof(w.stats));
  w.n_entries = from->w.n_entries;
  w.d_ref_map = from->w.d_ref_map;
  w.tables = from->w.tables;
  w.dev_tables = from->w.dev_tables;
  w.bind_tables();
  *out = t.release();
  return BG_OK;
}

extern "C" void bg_tracer_close(bg_tracer* t) { delete t; }

static int tracer_trace(bg_tracer* t, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                        const uint64_t* ext_len, int64_t end_pos, int64_t start_offset, int64_t limit_offset,
                        int rev_comp, int pair_count, bool async, bg_trace_result** out) {
  if (t) tl_pool = t->pool;
  if (!t || !out || (n_extents && (!ext_offset || !ext_seq || !ext_len))) return fail("bg_tracer_trace: null argument");
  World& w = t->w;
  const double t_call = now_seconds();
  BG_HIP(hipSetDevice(w.ctx->device));
  memset(&w.stats, 0, sizeof(w.stats));
  w.seconds_device = w.seconds_host = w.seconds_tables = w.seconds_start = w.seconds_teardown = 0;
  w.seconds_tab_codes = w.seconds_tab_walk = w.seconds_tab_read = 0;
  const double t_ext = now_seconds();
  BG_TRY(session_extents(&t->exts, "bg_tracer_trace", n_extents, ext_offset, ext_seq, ext_len, end_pos));
  const std::vector<Extent>& ext = t->exts->ext;
  w.scaffold_end = end_pos;
  if (limit_offset > end_pos) limit_offset = end_pos;
  std::unique_ptr<bg_trace_result> res(new bg_trace_result);
  if (rev_comp && w.o.only_trace_forward) { memset(&res->stats, 0, sizeof(res->stats)); *out = res.release(); return BG_OK; }
  std::vector<bg_trace_assembly_rec> raw;
  // the unseen reference assemblies as lean records (BG_TRACER_LEAN_REFS=0: as full records through every step, as before)
  static const bool lean_allowed = [] { const char* e = getenv("BG_TRACER_LEAN_REFS"); return !(e && e[0] == '0'); }();
  LeanOut lean_out;
  LeanOut* lean = (lean_allowed && pair_count && w.o.anchor_drop && !w.o.trace_reference_assemblies) ? &lean_out : nullptr;
  const double s_ext = now_seconds() - t_ext;
  const double t_chunk = now_seconds();
  static const std::vector<Extent> no_extents;
  const std::vector<Extent>& rc = rev_comp ? t->exts->rc() : no_extents;
  if (rev_comp) {
    BG_TRY(trace_chunk(w, rc, end_pos - limit_offset, end_pos - start_offset, &raw, lean));
  } else {
    BG_TRY(trace_chunk(w, ext, start_offset, limit_offset, &raw, lean));
  }
  const double s_chunk = now_seconds() - t_chunk;
  // report_anchor_drop_func (tracer.cpp:1233-1235 through reversable_tracer.cpp:28-35): the raw assemblies that end in an
  // anchor drop, in output order, turned into forward coordinates with their read ids -- kept for bg_pop_trace
  for (const auto& a : raw) {
    if (!a.anchor_drop) continue;
    bg_anchor_drop_rec d;
    d.a = a;
    d.right_anchor = false;
    if (rev_comp) {
      reverse_rec(w, &d.a, end_pos, true);
      d.right_anchor = true;
    }
    res->anchor_drops.push_back(std::move(d));
  }
  // ---- the host steps behind the tracer: anchor_dropper, pair_counter, back to forward coordinates.  They read nothing of the
  // session that the next work item's tracer changes (the options, the shared host tables, the scaffold the result keeps alive),
  // so bg_tracer_trace_async leaves them to the session's helper thread.
  res->stats = w.stats;
  bg_trace_result* const rp = res.get();
  const World* const wp = &w;
  std::shared_ptr<ExtSet> exts = t->exts;
  struct Behind {  // (moved into the job)
    std::vector<bg_trace_assembly_rec> raw;
    LeanOut lean_out;
    double s_drop = 0, s_pair = 0;
  };
  std::shared_ptr<Behind> bh(new Behind);
  bh->raw = std::move(raw);
  bh->lean_out = std::move(lean_out);
  auto behind = [rp, wp, exts, bh, rev_comp, end_pos, pair_count]() -> int {
    const World& w = *wp;
    std::vector<bg_trace_assembly_rec>& raw = bh->raw;
    LeanOut& lean_out = bh->lean_out;
    const std::vector<Extent>& run_ext = rev_comp ? exts->rc() : exts->ext;
    const double t_drop = now_seconds();
    if (w.o.anchor_drop) {  // anchor_dropper over the scaffold the tracer ran on (reversable_tracer.cpp:124-133)
      ScaffoldView sc;
      for (const Extent& e : run_ext) sc.ext.push_back(ScaffoldView::Ext{e.offset, e.seq.data(), int64_t(e.seq.size())});
      sc.end_pos = end_pos;
      std::vector<bg_trace_assembly_rec> dropped;
      try {
        AnchorDropper ad(AlignParams::from(w.o), sc, [&](bg_trace_assembly_rec&& a) { dropped.push_back(std::move(a)); });
        // report_half_aligned_func through reversable_tracer.cpp:20-27 (reverse_half_aligned, assemble.cpp:1002-1013)
        ad.report_half_aligned = [&](int64_t offset, const std::string& seq, uint64_t id) {
          bg_half_aligned_rec h;
          h.assembly_id = id;
          if (rev_comp) { h.offset = end_pos - offset; h.right_anchor = true; h.seq = rev_comp_str(seq); }
          else { h.offset = offset; h.right_anchor = false; h.seq = seq; }
          rp->half_aligned.push_back(std::move(h));
        };
        for (auto& a : raw) ad.add(std::move(a));
        ad.flush();
      } catch (const CheckFailed& f) {
        rp->pending_error = "bg_tracer_trace: anchor_dropper: " + f.what + " (the reference aborts on this input)";
        return BG_EINVAL;
      }
      raw = std::move(dropped);
    }
    bh->s_drop = now_seconds() - t_drop;
    const double t_pair = now_seconds();
    std::vector<bg_trace_assembly_rec> counted;
    if (pair_count) {
      PairCounter pc{w, &counted};
      // (traces run by ascending offset and a context grows by at most one base per step: left offsets never go back)
      auto by_left = [](const LeanRef& a, const LeanRef& b) { return a.left < b.left; };
      if (!std::is_sorted(lean_out.refs.begin(), lean_out.refs.end(), by_left)) std::stable_sort(lean_out.refs.begin(), lean_out.refs.end(), by_left);
      size_t li = 0;
      auto lean_up_to = [&](int64_t left) {
        for (; li < lean_out.refs.size() && lean_out.refs[li].left <= left; ++li) {
          const LeanRef& x = lean_out.refs[li];
          if (!pc.add_lean(x.left, x.right, lean_out.ids.data() + x.ids_at, x.n_ids)) return false;
        }
        return true;
      };
      bool ordered = true;
      for (auto& a : raw)
        if (!lean_up_to(a.left_offset) || !pc.add(std::move(a))) { ordered = false; break; }
      if (ordered && !lean_up_to(INT64_MAX)) ordered = false;
      if (!ordered) { rp->pending_error = "bg_tracer_trace: pair_counter: assemblies out of left-offset order"; return BG_EINVAL; }
      pc.flush();
    } else {
      counted = std::move(raw);
    }
    if (rev_comp)
      for (auto& a : counted) reverse_rec(w, &a, end_pos, false);
    rp->a = std::move(counted);
    bh->s_pair = now_seconds() - t_pair;
    return BG_OK;
  };
  int rc_behind = BG_OK;
  if (async) {
    res->pending = t->helper.run(std::packaged_task<int()>(behind));
  } else {
    rc_behind = behind();
    if (rc_behind != BG_OK) set_error("%s", res->pending_error.c_str());
  }
  if (getenv("BG_TRACER_TIMING"))
    fprintf(stderr, "bg_tracer_trace: %.3f s in all (extents %.3f, chunk %.3f, anchor drop %.3f, pair counter + reverse %.3f): tables %.3f s, start %.3f s, device rounds %.3f s, host replay %.3f s, teardown %.3f s (%d threads), %llu traces\n",
            now_seconds() - t_call, s_ext, s_chunk, async ? 0.0 : bh->s_drop, async ? 0.0 : bh->s_pair, w.seconds_tables, w.seconds_start, w.seconds_device, w.seconds_host, w.seconds_teardown, g_tracer_threads, (unsigned long long)w.stats.traces);
  // the chunk's tables are not kept between work items
  std::vector<std::pair<int32_t, uint64_t>>().swap(w.position_entries);
  std::vector<uint16_t>().swap(w.position_info);
  std::vector<std::pair<uint64_t, int32_t>>().swap(w.entry_positions);
  std::vector<std::pair<uint32_t, int32_t>>().swap(w.rc_mate_read_positions);
  if (rc_behind != BG_OK) return rc_behind;
  *out = res.release();
  return BG_OK;
}

extern "C" int bg_tracer_trace(bg_tracer* t, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                               const uint64_t* ext_len, int64_t end_pos, int64_t start_offset, int64_t limit_offset,
                               int rev_comp, int pair_count, bg_trace_result** out) {
  return tracer_trace(t, n_extents, ext_offset, ext_seq, ext_len, end_pos, start_offset, limit_offset, rev_comp, pair_count, false, out);
}

// The same with the anchor_dropper and the pair_counter of this work item left to the session's helper thread: the call returns when
// the tracer is done with the device, and the session can start its next work item.  *out must be handed to bg_trace_result_wait
// before anything else looks at it (bg_trace_result_free waits by itself); the session must stay open until then.
extern "C" int bg_tracer_trace_async(bg_tracer* t, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                                     const uint64_t* ext_len, int64_t end_pos, int64_t start_offset, int64_t limit_offset,
                                     int rev_comp, int pair_count, bg_trace_result** out) {
  return tracer_trace(t, n_extents, ext_offset, ext_seq, ext_len, end_pos, start_offset, limit_offset, rev_comp, pair_count, true, out);
}

extern "C" int bg_trace_result_wait(bg_trace_result* r) {
  if (!r) return fail("bg_trace_result_wait: null argument");
  if (!r->pending.valid()) return BG_OK;
  const int rc = r->pending.get();
  if (rc != BG_OK) set_error("%s", r->pending_error.c_str());
  return rc;
}

// reversable_tracer::assemble(start, limit) (reversable_tracer.cpp:76-145) with the push tracer: one session, one work item.
extern "C" int bg_push_trace(bg_ctx* ctx, const bg_seqset* ss, const bg_readmap* rm, const uint8_t* d_ref_map,
                             uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                             const uint64_t* ext_len, int64_t end_pos, int64_t start_offset, int64_t limit_offset,
                             int rev_comp, int pair_count, const bg_trace_options* options, bg_trace_result** out) {
  bg_tracer* t = nullptr;
  BG_TRY(bg_tracer_open(ctx, ss, rm, d_ref_map, options, &t));
  const int rc = bg_tracer_trace(t, n_extents, ext_offset, ext_seq, ext_len, end_pos, start_offset, limit_offset, rev_comp,
                                 pair_count, out);
  bg_tracer_close(t);
  return rc;
}

// =====================================================================================================================
// The pop tracer (variants/pop_tracer.cpp:55-653) behind reversable_tracer's direction wrapper
// (reversable_tracer.cpp:49-74,118-126), one work item per call as trace_ref::execute_work runs it (trace_ref.cpp:141-221).
// Host state machine over batched device walks: everything the reference asks of the seqset / readmap entry by entry is
// collected per phase and answered by one launch --
//   * add_reference: the push_front_drop chain along the (reverse-complemented) scaffold in 1024-base chains with a
//     max_read_len warm-up (bg_walk_chains_batch, as bg_ref_walk_entries does for the push tracer), then ONE
//     get_prefix_reads for all the ranges add_reference_at keeps, ONE get_seqset_entry for all their mates, ONE
//     get_prefix_reads for those;
//   * a pop pass: ONE pop_front for all poppers, ONE get_prefix_reads for the survivors.
// m_fronts only ever grows before assemble() and shrinks during it, so it is a vector sorted once (by key, equal keys in
// insertion order: std::multimap::emplace) with an erased flag; the popper heap and the two sorts are libstdc++'s own
// (std::push_heap / pop_heap / make_heap, std::sort), as in the reference's binary.
namespace {

inline unsigned code_of(char c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; }

struct PopEntry {  // pop_tracer::entry (pop_tracer.h:58-84)
  bool has_left = false, has_right = false;
  int64_t left_offset = 0, right_offset = 0, start_limit = 0, end_limit = 0;
  TRange orig_r, popped_r;
  std::string seq;
  // seen_read_ids = the ids the entry was made with -- ids_n words at ids_at of the pop tracer's pool: a piece has 350 000 fronts and
  // a few hundred of them ever meet a popper, a vector each was two allocations a front -- followed by what it has gathered since
  uint32_t ids_at = 0, ids_n = 0;
  std::vector<uint32_t> seen_read_ids;
  // where this popper stands in the passes the device ran ahead (PopTracer::pop_pass): its row in the batch, or -1 once a merge
  // has given it another range
  int32_t ahead_slot = -1;
  uint32_t ahead_batch = 0;
};

struct PopTracer {
  World& w;
  bg_ctx* ctx;
  const std::vector<Extent>& sc;  // the scaffold in the direction this pop tracer runs
  const std::vector<Extent>& sc_rc;  // ... and reversed: entries face right while push_front_drop runs through it
  int64_t end_pos;
  std::deque<PopEntry> store;     // stable addresses
  std::vector<uint32_t> id_pool;  // the ids the entries were made with (PopEntry::ids_at)
  void made_with(PopEntry* e, const uint32_t* first, const uint32_t* last) {
    e->ids_at = uint32_t(id_pool.size());
    e->ids_n = uint32_t(last - first);
    id_pool.insert(id_pool.end(), first, last);
  }
  template <class V>
  void append_ids(V* to, const PopEntry& e) const {
    to->insert(to->end(), id_pool.begin() + e.ids_at, id_pool.begin() + e.ids_at + e.ids_n);
    to->insert(to->end(), e.seen_read_ids.begin(), e.seen_read_ids.end());
  }
  struct Front { uint64_t b; uint32_t s; uint32_t seq; PopEntry* e; bool erased; };
  std::vector<Front> fronts;      // insertion order until sort_fronts()
  std::vector<PopEntry*> poppers;
  std::vector<bg_trace_assembly_rec> out;
  uint64_t device_rounds = 0, device_queries = 0;

  static bool popper_less(const PopEntry* a, const PopEntry* b) { return a->popped_r.s < b->popped_r.s; }  // :285-290

  // ---- device batches (host vectors in and out): one upload, one launch, one download per batch -- a pass of the pop
  // tracer is two of these, and there are hundreds of passes per work item
  static size_t up8(size_t x) { return (x + 7) & ~size_t(7); }
  int upload_ranges(const std::vector<TRange>& r, DevBuf<uint8_t>* d, size_t extra_bytes, uint64_t** db, uint64_t** de, uint32_t** ds,
                    uint8_t** rest) {
    const size_t n = r.size();
    const size_t in_bytes = n * 16 + up8(n * 4);
    std::vector<uint8_t> h(in_bytes);
    uint64_t* hb = reinterpret_cast<uint64_t*>(h.data());
    uint64_t* he = hb + n;
    uint32_t* hs = reinterpret_cast<uint32_t*>(he + n);
    for (size_t i = 0; i < n; ++i) { hb[i] = r[i].b; he[i] = r[i].e; hs[i] = r[i].s; }
    BG_TRY(d->alloc(ctx, in_bytes + extra_bytes));
    BG_TRY(copy_sync(ctx, d->p, h.data(), in_bytes, hipMemcpyHostToDevice));
    *db = reinterpret_cast<uint64_t*>(d->p);
    *de = *db + n;
    *ds = reinterpret_cast<uint32_t*>(*de + n);
    *rest = d->p + in_bytes;
    return BG_OK;
  }
  int prefix_reads(const std::vector<TRange>& r, uint32_t cap, std::vector<uint32_t>* ids, std::vector<uint32_t>* count) {
    const size_t n = r.size();
    ids->assign(n * cap, 0);
    count->assign(n, 0);
    if (!n) return BG_OK;
    DevBuf<uint8_t> d;
    uint64_t *db, *de;
    uint32_t* ds;
    uint8_t* rest;
    BG_TRY(upload_ranges(r, &d, (n * cap + n) * 4, &db, &de, &ds, &rest));
    uint32_t* di = reinterpret_cast<uint32_t*>(rest);
    uint32_t* dc = di + n * cap;
    BG_TRY(bg_prefix_reads_batch(ctx, w.ss, w.rm, n, db, de, ds, int(w.o.min_overlap), cap, di, dc));
    std::vector<uint32_t> h(n * cap + n);
    BG_TRY(copy_sync(ctx, h.data(), di, h.size() * 4, hipMemcpyDeviceToHost));
    std::copy(h.begin(), h.begin() + long(n * cap), ids->begin());
    std::copy(h.begin() + long(n * cap), h.end(), count->begin());
    ++device_rounds;
    device_queries += n;
    return BG_OK;
  }
  // the same ranges' prefix reads, packed, with each read's reverse complement / mate / length / direction from the device
  int prefix_reads_attr(const std::vector<TRange>& r, uint32_t cap, PackedPrefixReads* out) {
    const size_t n = r.size();
    DevBuf<uint8_t> d;
    uint64_t *db = nullptr, *de = nullptr;
    uint32_t* ds = nullptr;
    uint8_t* rest = nullptr;
    if (n) BG_TRY(upload_ranges(r, &d, 0, &db, &de, &ds, &rest));
    BG_TRY(prefix_reads_packed(ctx, w.ss, w.rm, n, db, de, ds, int(w.o.min_overlap), cap, out));
    if (n) { ++device_rounds; device_queries += n; }
    return BG_OK;
  }
  int read_entries(const std::vector<uint32_t>& ids, std::vector<TRange>* r) {  // readmap::read::get_seqset_entry
    const size_t n = ids.size();
    r->assign(n, TRange());
    if (!n) return BG_OK;
    // in: ids; out: begin, end (u64), size, mate_rc, rev_comp (u32), has_mate (u8)
    DevBuf<uint8_t> d;
    const size_t in_bytes = up8(n * 4), out_bytes = n * 16 + n * 12 + up8(n);
    BG_TRY(d.alloc(ctx, in_bytes + out_bytes));
    BG_TRY(copy_sync(ctx, d.p, ids.data(), n * 4, hipMemcpyHostToDevice));
    uint64_t* db = reinterpret_cast<uint64_t*>(d.p + in_bytes);
    uint64_t* de = db + n;
    uint32_t* ds = reinterpret_cast<uint32_t*>(de + n);
    uint32_t* dm = ds + n;
    uint32_t* drc = dm + n;
    uint8_t* dh = reinterpret_cast<uint8_t*>(drc + n);
    BG_TRY(bg_read_info_batch(ctx, w.ss, w.rm, n, reinterpret_cast<const uint32_t*>(d.p), dh, dm, drc, db, de, ds));
    std::vector<uint8_t> h(n * 16 + n * 4);
    BG_TRY(copy_sync(ctx, h.data(), db, h.size(), hipMemcpyDeviceToHost));
    const uint64_t* hb = reinterpret_cast<const uint64_t*>(h.data());
    const uint64_t* he = hb + n;
    const uint32_t* hs = reinterpret_cast<const uint32_t*>(he + n);
    for (size_t i = 0; i < n; ++i) (*r)[i] = TRange{hb[i], he[i], hs[i]};
    ++device_rounds;
    device_queries += n;
    return BG_OK;
  }
  int pop_fronts(const std::vector<TRange>& r, std::vector<TRange>* o) {
    const size_t n = r.size();
    o->assign(n, TRange());
    if (!n) return BG_OK;
    DevBuf<uint8_t> d;
    uint64_t *db, *de;
    uint32_t* ds;
    uint8_t* rest;
    BG_TRY(upload_ranges(r, &d, n * 16 + up8(n * 4), &db, &de, &ds, &rest));
    uint64_t* ob = reinterpret_cast<uint64_t*>(rest);
    uint64_t* oe = ob + n;
    uint32_t* os = reinterpret_cast<uint32_t*>(oe + n);
    BG_TRY(bg_pop_front_batch(ctx, w.ss, n, db, de, ds, ob, oe, os));
    std::vector<uint8_t> h(n * 16 + n * 4);
    BG_TRY(copy_sync(ctx, h.data(), ob, h.size(), hipMemcpyDeviceToHost));
    const uint64_t* hb = reinterpret_cast<const uint64_t*>(h.data());
    const uint64_t* he = hb + n;
    const uint32_t* hs = reinterpret_cast<const uint32_t*>(he + n);
    for (size_t i = 0; i < n; ++i) (*o)[i] = TRange{hb[i], he[i], hs[i]};
    ++device_rounds;
    device_queries += n;
    return BG_OK;
  }

  char front_base(const TRange& r) const {  // seqset_range::front (seqset.cpp:663): the block of `fixed` begin lies in
    for (int b = 3; b > 0; --b)
      if (r.b >= w.ss->fixed[b]) return "ACGT"[b];
    return 'A';
  }

  // add_entry_reads (:458-467) from the answer of one prefix_reads query
  void take_entry_reads(PopEntry* p, const uint32_t* ids, uint32_t count, uint32_t cap) const {
    const uint32_t n = std::min<uint32_t>(std::min(count, cap), uint32_t(w.o.max_pairs_per_read));
    for (uint32_t i = 0; i < n; ++i) p->seen_read_ids.push_back(w.get_rev_comp(ids[i]));
  }

  // ---- the reads that become fronts are collected first (their entries and their prefix reads come from two launches)
  struct PendingRead { uint32_t read_id; int64_t start, limit; };

  // the fronts a list of items adds, in the reference's insertion order: item = one add_read, or an entry that exists
  struct Item { PopEntry* ready; PendingRead rd; };
  int materialize(std::vector<Item>& items) {
    std::vector<uint32_t> ids;
    for (auto& it : items) if (!it.ready) ids.push_back(it.rd.read_id);
    std::vector<TRange> ranges;
    BG_TRY(read_entries(ids, &ranges));
    const uint32_t cap = uint32_t(std::max<uint64_t>(1, w.o.max_pairs_per_read));
    PackedPrefixReads pr;
    BG_TRY(prefix_reads_attr(ranges, cap, &pr));
    size_t k = 0;
    for (auto& it : items) {
      PopEntry* e = it.ready;
      if (!e) {  // add_read (:76-95)
        store.emplace_back();
        e = &store.back();
        e->start_limit = it.rd.start;
        e->end_limit = it.rd.limit;
        e->orig_r = e->popped_r = ranges[k];
        // add_entry_reads (:97-119): the reverse complements of the first max_pairs_per_read prefix reads
        made_with(e, pr.rc.data() + pr.first[k], pr.rc.data() + pr.first[k + 1]);
        ++k;
      }
      fronts.push_back(Front{e->orig_r.b, e->orig_r.s, uint32_t(fronts.size()), e, false});
    }
    return BG_OK;
  }

  // ---- add_anchor_drop (:208-283) for all drops of the work item
  int add_anchor_drops(const std::vector<const bg_anchor_drop_rec*>& drops, bool rev_comp) {
    if (drops.empty()) return BG_OK;
    // orientation of this pop tracer (wrap_report_anchor_drop_for_pop_tracer, reversable_tracer.cpp:55-73)
    std::vector<bg_trace_assembly_rec> as;
    std::vector<bool> right;
    for (const bg_anchor_drop_rec* d : drops) {
      as.push_back(d->a);
      right.push_back(d->right_anchor);
      if (rev_comp) {
        reverse_rec(w, &as.back(), end_pos, true);
        right.back() = !d->right_anchor;
      }
    }
    const size_t n = as.size();
    // orig_r: push_front_drop of the assembly's bases, last base first (one chain each); popped_r of a left-anchored
    // drop: push_front of the same bases for as long as an entry is left
    std::vector<bg_walk_chain> chains(n);
    std::vector<uint8_t> codes;
    std::vector<uint64_t> offs(n + 1, 0);
    for (size_t i = 0; i < n; ++i) {
      const std::string& s = as[i].seq;
      chains[i].begin = 0; chains[i].end = w.n_entries; chains[i].size = 0;
      chains[i].code_offset = codes.size();
      chains[i].len = uint32_t(s.size());
      chains[i].out_offset = uint64_t(i) * 3;
      chains[i].out_stride = 0;  // every step overwrites the same three words: the last one stays
      // k_walk_chains pushes the COMPLEMENT of the code it is given: codes = the reverse complement of the sequence
      for (size_t j = s.size(); j-- > 0;) codes.push_back(uint8_t(3 - code_of(s[j])));
      offs[i + 1] = codes.size();
    }
    std::vector<uint64_t> orig(n * 3, 0), popped(n * 3, 0);
    // out_stride 0 is not what bg_walk_chains_batch promises (>= 3): give every chain a private strip instead
    uint64_t words = 0;
    for (size_t i = 0; i < n; ++i) { chains[i].out_offset = words; chains[i].out_stride = 3; words += uint64_t(chains[i].len) * 3; }
    std::vector<uint64_t> strip(words ? words : 1, 0);
    BG_TRY(bg_walk_chains_batch(ctx, w.ss, uint32_t(n), chains.data(), codes.data(), codes.size(), strip.data(), words));
    for (size_t i = 0; i < n; ++i) {
      if (chains[i].len) memcpy(&orig[i * 3], &strip[chains[i].out_offset + uint64_t(chains[i].len - 1) * 3], 24);
      else { orig[i * 3] = 0; orig[i * 3 + 1] = w.n_entries; orig[i * 3 + 2] = 0; }
    }
    // push_front takes the base itself: the sequence backwards
    std::vector<uint8_t> fcodes;
    for (size_t i = 0; i < n; ++i) {
      const std::string& s = as[i].seq;
      offs[i] = fcodes.size();
      for (size_t j = s.size(); j-- > 0;) fcodes.push_back(uint8_t(code_of(s[j])));
    }
    offs[n] = fcodes.size();
    BG_TRY(bg_push_front_chains(ctx, w.ss, uint32_t(n), offs.data(), fcodes.data(), popped.data()));
    device_rounds += 2;
    device_queries += 2 * n;
    std::vector<Item> items;
    std::vector<PopEntry*> new_poppers;
    for (size_t i = 0; i < n; ++i) {
      const bg_trace_assembly_rec& a = as[i];
      const bool right_anchor = right[i];
      const int64_t offset = right_anchor ? a.right_offset : a.left_offset, len = int64_t(a.seq.size());
      for (uint32_t rc_read_id : a.rc_read_ids) {  // mates of the assembly's reads (:216-238)
        if (!w.has_mate(rc_read_id)) continue;
        int64_t start_offset = offset - len, limit_offset = offset + len;
        if (right_anchor) start_offset -= len; else limit_offset += len;
        const int64_t size = w.tables->rt->read_len[rc_read_id];
        if (bool(w.is_forward[rc_read_id]) == (w.o.forward_pairs_face_inward != 0)) start_offset -= size + int64_t(w.o.max_pair_distance);
        else limit_offset += size + int64_t(w.o.max_pair_distance);
        items.push_back(Item{nullptr, PendingRead{w.get_mate(rc_read_id), start_offset, limit_offset}});
      }
      store.emplace_back();
      PopEntry* e = &store.back();
      e->start_limit = offset - (len + int64_t(w.o.max_pair_distance));
      e->end_limit = offset + (len + int64_t(w.o.max_pair_distance));
      e->orig_r = TRange{orig[i * 3], orig[i * 3 + 1], uint32_t(orig[i * 3 + 2])};
      if (e->orig_r.b >= e->orig_r.e) return fail("bg_pop_trace: an anchor drop's sequence is not in the seqset");
      if (right_anchor) {
        e->has_right = true;
        e->right_offset = a.right_offset;
        e->popped_r = TRange{0, w.n_entries, 0};
        e->seq = a.seq;
      } else {
        e->has_left = true;
        e->left_offset = a.left_offset;
        e->popped_r = TRange{popped[i * 3], popped[i * 3 + 1], uint32_t(popped[i * 3 + 2])};
        if (int64_t(e->popped_r.s) > len) return fail("bg_pop_trace: popped range longer than the assembly");
        e->seq = a.seq.substr(0, size_t(len - int64_t(e->popped_r.s)));
        new_poppers.push_back(e);
      }
      items.push_back(Item{e, PendingRead{0, 0, 0}});
    }
    BG_TRY(materialize(items));
    for (PopEntry* p : new_poppers) add_popper(p);
    return BG_OK;
  }

  void add_popper(PopEntry* p) {  // :447-456
    poppers.push_back(p);
    std::push_heap(poppers.begin(), poppers.end(), popper_less);
  }

  // ---- add_reference (:162-206) with add_reference_at (:97-160)
  int add_reference(int64_t start, int64_t limit) {
    double t_mark = now_seconds();
    // the scaffold reversed: entries face right while push_front_drop runs through it (:167-169)
    const std::vector<Extent>& rc = sc_rc;
    const int64_t scaffold_end = end_pos;
    const int64_t skip_pos = scaffold_end - limit - int64_t(w.rm->max_read_len);
    // positions walked, per extent: [first, last) indices into the extent; the walk of an extent starts from the empty
    // context (first_in_extent) -- or, where skip_to lands inside an extent, from wherever the iterator stands with the
    // range still empty (r = ctx_begin before the loop)
    struct Piece { size_t ext; int64_t first, count; };
    std::vector<Piece> pieces;
    std::vector<bg_walk_chain> chains;
    std::vector<uint8_t> codes;
    uint64_t words = 0;
    const int64_t chain_len = 512, warm = int64_t(w.ss->max_len) + 1;  // (a lane walks chain_len + warm steps: 2.8 ms; 1024: 5 ms)
    bool stopped = false;
    for (size_t x = 0; x < rc.size() && !stopped; ++x) {
      const Extent& e = rc[x];
      const int64_t n = int64_t(e.seq.size());
      int64_t first = 0;
      if (skip_pos > e.offset) {
        if (skip_pos >= e.offset + n) continue;  // skip_to passes the whole extent
        first = skip_pos - e.offset;
      }
      // the loop runs while offset >= start, offset = scaffold_end - it.offset() before the step
      int64_t last = n;
      const int64_t max_pos = scaffold_end - start;  // it.offset() <= this
      if (e.offset + first > max_pos) { stopped = true; break; }
      if (e.offset + last - 1 > max_pos) { last = max_pos - e.offset + 1; stopped = true; }
      if (last <= first) continue;
      pieces.push_back(Piece{x, first, last - first});
      const size_t code0 = codes.size();
      for (int64_t j = first; j < last; ++j) codes.push_back(uint8_t(code_of(e.seq[size_t(j)])));
      // chains of chain_len positions; a chain that does not start the piece warms up over the `warm` bases before it
      for (int64_t c0 = 0; c0 < last - first; c0 += chain_len) {
        const int64_t c1 = std::min<int64_t>(c0 + chain_len, last - first);
        const int64_t w0 = std::max<int64_t>(0, c0 - warm);
        bg_walk_chain ch;
        ch.begin = 0; ch.end = w.n_entries; ch.size = 0;
        ch.code_offset = code0 + uint64_t(w0);
        ch.len = uint32_t(c1 - w0);
        ch.out_stride = 3;
        // the chain's strip starts `c0 - w0` steps before the part that is kept: point it so that step (c0 - w0) lands at
        // position c0 of the piece's output; the warm-up steps then fall on the tail of the chain before (written first
        // or last, they hold values nobody reads: every chain's kept part is written by that chain alone ... they are
        // NOT, so every chain gets a private strip and the kept part is copied)
        ch.out_offset = words;
        words += uint64_t(ch.len) * 3;
        chains.push_back(ch);
      }
    }
    // (30-40 MB of ranges per piece, a third of them the chains' warm-up rows, into pageable memory: page-locked staging of that
    // size was tried and lost -- sixteen sessions growing 50 MB pinned buffers cost more than the copies: pop_reference 6.8 -> 9.3
    // session-seconds, A/B on one box)
    std::vector<uint64_t> strip_v(words ? words : 1);
    uint64_t* strip = strip_v.data();
    if (!chains.empty()) {
      BG_TRY(bg_walk_chains_batch(ctx, w.ss, uint32_t(chains.size()), chains.data(), codes.data(), codes.size(), strip, words));
      ++device_rounds;
      device_queries += codes.size();
    }
    ref_t[0] += now_seconds() - t_mark; t_mark = now_seconds();
    // replay the loop of :180-199 over the ranges
    std::vector<std::pair<int64_t, TRange>> refs;  // add_reference_at(offset, r) in call order
    size_t ci = 0;
    TRange r{0, w.n_entries, 0};
    int64_t offset = scaffold_end - (rc.empty() ? scaffold_end : rc[0].offset);
    {
      // offset before the loop: scaffold_end - next_it.offset() after the optional skip_to
      int64_t it_off = scaffold_end;
      if (!pieces.empty()) it_off = rc[pieces[0].ext].offset + pieces[0].first;
      else if (!rc.empty() && skip_pos <= rc[0].offset) it_off = rc[0].offset;
      else {  // skip_to ran off the walked part: where the iterator stands decides nothing but the last add_reference_at
        it_off = scaffold_end;
        for (const Extent& e : rc) {
          if (skip_pos < e.offset + int64_t(e.seq.size())) { it_off = std::max<int64_t>(skip_pos, e.offset); break; }
        }
      }
      offset = scaffold_end - it_off;
    }
    for (const Piece& pc : pieces) {
      const Extent& e = rc[pc.ext];
      for (int64_t c0 = 0; c0 < pc.count; c0 += chain_len, ++ci) {
        const int64_t c1 = std::min<int64_t>(c0 + chain_len, pc.count);
        const int64_t w0 = std::max<int64_t>(0, c0 - warm);
        const uint64_t* o = &strip[chains[ci].out_offset + uint64_t(c0 - w0) * 3];
        for (int64_t j = c0; j < c1; ++j, o += 3) {
          const int64_t pos = pc.first + j;          // index in the extent
          const bool first_in_extent = pos == 0;
          if (first_in_extent) {
            refs.emplace_back(offset, r);
            r = TRange{0, w.n_entries, 0};
          }
          const TRange next{o[0], o[1], uint32_t(o[2])};
          if (next.b >= next.e) return fail("bg_pop_trace: a reference base is missing from the seqset");
          if (next.s <= r.s) refs.emplace_back(offset, r);
          r = next;
          if (first_in_extent) offset = scaffold_end - (e.offset + pos) - 1;
          else --offset;
        }
      }
    }
    refs.emplace_back(offset, r);
    ref_t[1] += now_seconds() - t_mark; t_mark = now_seconds();
    // add_reference_at for all of them: ranges shorter than min_pop_overlap add nothing
    std::vector<std::pair<int64_t, TRange>> kept;
    for (auto& x : refs)
      if (x.second.s >= w.o.min_pop_overlap) kept.push_back(x);
    std::vector<TRange> ranges;
    for (auto& x : kept) ranges.push_back(x.second);
    // their prefix reads: the first max_pairs_per_read feed add_entry_reads, the first max_pairs_per_read WITH a mate
    // supply the mates; a range with more reads than the buffer holds and too few mates among them is asked again
    const uint32_t cap = uint32_t(std::max<uint64_t>(w.o.max_pairs_per_read, 1)) * 2;
    PackedPrefixReads pr;
    ref_t[2] += now_seconds() - t_mark; t_mark = now_seconds();
    BG_TRY(prefix_reads_attr(ranges, cap, &pr));
    std::vector<std::vector<uint32_t>> wide(ranges.size());
    {
      std::vector<size_t> again;
      for (size_t i = 0; i < ranges.size(); ++i) {
        if (pr.count[i] <= cap) continue;
        uint32_t mated = 0;
        for (uint32_t p = pr.first[i]; p < pr.first[i + 1]; ++p) mated += pr.flags[p] & 1u;
        if (mated <= w.o.max_pairs_per_read) again.push_back(i);
      }
      uint32_t big = cap;
      while (!again.empty()) {
        big *= 8;
        std::vector<TRange> rr;
        for (size_t i : again) rr.push_back(ranges[i]);
        std::vector<uint32_t> id2, c2;
        BG_TRY(prefix_reads(rr, big, &id2, &c2));
        std::vector<size_t> still;
        for (size_t k = 0; k < again.size(); ++k) {
          const size_t i = again[k];
          const uint32_t have = std::min(c2[k], big);
          uint32_t mated = 0;
          for (uint32_t j = 0; j < have; ++j) mated += w.has_mate(id2[k * big + j]) ? 1u : 0u;
          if (c2[k] > big && mated <= w.o.max_pairs_per_read) { still.push_back(i); continue; }
          wide[i].assign(id2.begin() + long(k * big), id2.begin() + long(k * big + have));
        }
        again.swap(still);
      }
    }
    ref_t[3] += now_seconds() - t_mark; t_mark = now_seconds();
    std::vector<Item> items;
    for (size_t i = 0; i < kept.size(); ++i) {
      const int64_t off = kept[i].first;
      store.emplace_back();
      PopEntry* e = &store.back();
      e->has_left = true;
      e->left_offset = off;
      e->start_limit = e->end_limit = off;
      e->orig_r = e->popped_r = kept[i].second;
      // one mated read of the entry (:121-159): its length, its direction, the reverse complement of its mate
      auto add_mate = [&](int64_t size, bool forward, uint32_t mate_rc) {
        if (size < int64_t(w.o.min_pop_overlap)) return;
        int64_t start_offset = off, limit_offset = off;
        if (forward == (w.o.forward_pairs_face_inward != 0)) {  // faces inward
          start_offset += size + w.o.min_pair_distance;
          limit_offset += size + int64_t(w.o.max_pair_distance);
        } else {  // faces outwards
          start_offset -= size + int64_t(w.o.max_pair_distance);
          limit_offset -= size + w.o.min_pair_distance;
        }
        items.push_back(Item{nullptr, PendingRead{mate_rc, start_offset, limit_offset}});
      };
      if (wide[i].empty()) {
        const uint32_t p0 = pr.first[i], p1 = pr.first[i + 1];
        const uint32_t n_seen = std::min<uint32_t>(p1 - p0, uint32_t(w.o.max_pairs_per_read));
        made_with(e, pr.rc.data() + p0, pr.rc.data() + p0 + n_seen);
        items.push_back(Item{e, PendingRead{0, 0, 0}});
        unsigned read_count = 0;
        for (uint32_t p = p0; p < p1; ++p) {
          if (!(pr.flags[p] & 1)) continue;
          if (++read_count > w.o.max_pairs_per_read) break;
          add_mate(int64_t(pr.len[p]), (pr.flags[p] & 2) != 0, pr.mate_rc[p]);
        }
        continue;
      }
      // (an entry with more prefix reads than the packed buffer holds and too few mates among them: the long list)
      const uint32_t* ids = wide[i].data();
      const uint32_t have = uint32_t(wide[i].size());
      take_entry_reads(e, ids, have, have);
      items.push_back(Item{e, PendingRead{0, 0, 0}});
      unsigned read_count = 0;
      for (uint32_t j = 0; j < have; ++j) {
        const uint32_t rid = ids[j];
        if (!w.has_mate(rid)) continue;
        if (++read_count > w.o.max_pairs_per_read) break;
        add_mate(int64_t(w.tables->rt->read_len[rid]), bool(w.is_forward[rid]), w.get_rev_comp(w.get_mate(rid)));
      }
    }
    ref_t[4] += now_seconds() - t_mark; t_mark = now_seconds();
    const int rc_mat = materialize(items);
    ref_t[5] += now_seconds() - t_mark;
    return rc_mat;
  }
  double ref_t[6] = {0, 0, 0, 0, 0, 0};  // walk, replay, kept ranges, prefix reads, entries and items, materialize

  void sort_fronts() {  // std::multimap<seqset_range, ..., seqset_range_comparer>: (begin, size), equal keys in insertion order
    // A stable LSD radix sort of (begin, size) keys (350 000 fronts per piece: std::stable_sort's merges of 32-byte structs were a
    // tenth of the pop tracer's time): 16-bit digits over as many bits as the largest key has, then one permutation.
    const size_t n = fronts.size();
    bool fits = n < (size_t(1) << 32);
    uint64_t top = 0;
    for (const Front& f : fronts) { fits = fits && f.s < 256 && f.b < (uint64_t(1) << 56); top |= (f.b << 8) | f.s; }
    if (!fits || n < 2048) {
      std::stable_sort(fronts.begin(), fronts.end(), [](const Front& a, const Front& b) { return a.b != b.b ? a.b < b.b : a.s < b.s; });
      return;
    }
    std::vector<uint64_t> key(n), key2(n);
    std::vector<uint32_t> idx(n), idx2(n);
    for (size_t i = 0; i < n; ++i) { key[i] = (fronts[i].b << 8) | fronts[i].s; idx[i] = uint32_t(i); }
    std::vector<uint32_t> count(size_t(1) << 16);
    for (int shift = 0; shift < 64 && (top >> shift); shift += 16) {
      std::fill(count.begin(), count.end(), 0u);
      for (size_t i = 0; i < n; ++i) ++count[(key[i] >> shift) & 0xffff];
      uint32_t run = 0;
      for (auto& c : count) { const uint32_t k = c; c = run; run += k; }
      for (size_t i = 0; i < n; ++i) {
        const uint32_t at = count[(key[i] >> shift) & 0xffff]++;
        key2[at] = key[i];
        idx2[at] = idx[i];
      }
      key.swap(key2);
      idx.swap(idx2);
    }
    std::vector<Front> sorted(n);
    for (size_t i = 0; i < n; ++i) sorted[i] = fronts[idx[i]];
    fronts.swap(sorted);
  }

  // ---- assemble (:292-311)
  int assemble() {
    sort_fronts();
    std::make_heap(poppers.begin(), poppers.end(), popper_less);
    match_and_output_pass();
    while (!poppers.empty()) {
      BG_TRY(pop_pass());
      match_and_output_pass();
    }
    return BG_OK;
  }

  int pop_pass() {  // :364-445
    std::vector<PopEntry*> old;
    old.swap(poppers);
    std::vector<PopEntry*> order;
    while (!old.empty()) {
      std::pop_heap(old.begin(), old.end(), popper_less);
      order.push_back(old.back());
      old.pop_back();
    }
    for (PopEntry* p : order)
      if (p->has_right) return fail("bg_pop_trace: a popper that is aligned on the right is popped again");
    const uint32_t cap = uint32_t(std::max<uint64_t>(1, w.o.max_pairs_per_read));
    static const bool ahead_off = [] { const char* e = getenv("BG_POP_PASSES_AHEAD"); return e && e[0] == '0'; }();
    if (ahead_off || cap > kPopPassMaxIds) return pop_pass_one(order, cap);
    // The passes the device ran ahead serve while every popper still is where the device left it: a merge
    // (match_and_output_pass) gives a popper another range, and the batch ends there -- for everybody: a batch per popper was
    // tried and lost (poppers then run out of passes at different times and nearly every pass needs a launch; a launch per
    // merge is the floor either way, and merges come every other pass).
    bool fresh = ahead_used < ahead_passes;
    for (PopEntry* p : order)
      if (p->ahead_slot < 0 || p->ahead_batch != ahead_batch) { fresh = false; break; }
    if (!fresh) {
      // as many passes as the last batch got through, doubled when it was used up
      if (ahead_batch) ahead_passes = ahead_used >= ahead_passes ? std::min<uint32_t>(ahead_passes * 2, 64) : std::max<uint32_t>(ahead_used, 2);
      ++ahead_batch;
      ahead_used = 0;
      const size_t n = order.size();
      std::vector<uint64_t> hb(n), he(n);
      std::vector<uint32_t> hs(n);
      for (size_t i = 0; i < n; ++i) {
        hb[i] = order[i]->popped_r.b; he[i] = order[i]->popped_r.e; hs[i] = order[i]->popped_r.s;
        order[i]->ahead_slot = int32_t(i);
        order[i]->ahead_batch = ahead_batch;
      }
      BG_TRY(w.pin_q.ensure(pop_passes_in_bytes(n)));
      BG_TRY(w.pin_r.ensure(pop_passes_out_bytes(n, ahead_passes, cap)));
      BG_TRY(pop_passes(ctx, w.ss, w.rm, n, hb.data(), he.data(), hs.data(), ahead_passes, uint32_t(w.o.min_pop_overlap), int(w.o.min_overlap),
                        cap, w.pin_q.p, w.pin_r.p, &ahead_recs, &ahead_ids));
      ++device_rounds;
      device_queries += n * ahead_passes;
    }
    for (PopEntry* p : order) {
      const PopPassRec& rec = ahead_recs[size_t(p->ahead_slot) * ahead_passes + ahead_used];
      if (rec.s < w.o.min_pop_overlap) continue;  // popped too much: discarded
      p->seq.push_back(front_base(p->popped_r));
      p->popped_r = TRange{rec.b, rec.e, rec.s};
      p->seen_read_ids.insert(p->seen_read_ids.end(), ahead_ids.begin() + rec.ids_at, ahead_ids.begin() + rec.ids_at + rec.n_ids);
      p->end_limit++;
      add_popper(p);
    }
    ++ahead_used;
    ++passes_run;
    return BG_OK;
  }
  uint64_t passes_run = 0, match_seconds_us = 0;
  uint32_t ahead_passes = 4, ahead_used = 0, ahead_batch = 0;
  std::vector<PopPassRec> ahead_recs;
  std::vector<uint32_t> ahead_ids;

  // one pass, two launches (BG_POP_PASSES_AHEAD=0, or more pairs per read than the batch kernel holds)
  int pop_pass_one(const std::vector<PopEntry*>& order, uint32_t cap) {
    std::vector<TRange> in;
    for (PopEntry* p : order) in.push_back(p->popped_r);
    std::vector<TRange> popped;
    BG_TRY(pop_fronts(in, &popped));
    std::vector<PopEntry*> alive;
    std::vector<TRange> ranges;
    for (size_t i = 0; i < order.size(); ++i) {
      if (popped[i].s < w.o.min_pop_overlap) continue;  // popped too much: discarded
      PopEntry* p = order[i];
      p->seq.push_back(front_base(p->popped_r));
      p->popped_r = popped[i];
      alive.push_back(p);
      ranges.push_back(popped[i]);
    }
    std::vector<uint32_t> pid, pcount;
    BG_TRY(prefix_reads(ranges, cap, &pid, &pcount));
    for (size_t i = 0; i < alive.size(); ++i) {
      take_entry_reads(alive[i], &pid[i * cap], pcount[i], cap);
      alive[i]->end_limit++;
      add_popper(alive[i]);
    }
    return BG_OK;
  }

  void match_and_output_pass() {  // :469-613
    const double t_match = now_seconds();
    struct Tick { PopTracer* t; double t0; ~Tick() { t->match_seconds_us += uint64_t((now_seconds() - t0) * 1e6); } } tick{this, t_match};
    std::vector<PopEntry*> old;
    old.swap(poppers);
    const int64_t slop = int64_t(w.o.pop_tracer_offset_slop);
    for (PopEntry* p : old) {
      std::vector<Front*> matches;
      const auto key_less = [](const Front& f, const TRange& r) { return f.b != r.b ? f.b < r.b : f.s < r.s; };
      for (auto it = std::lower_bound(fronts.begin(), fronts.end(), p->popped_r, key_less);
           it != fronts.end() && it->e->orig_r.b < p->popped_r.e; ++it) {
        if (it->erased) continue;
        const PopEntry& front = *it->e;
        if (&front == p) continue;                         // no loops
        if (front.orig_r.e > p->popped_r.e) continue;      // not a prefix
        const int64_t plen = int64_t(p->seq.size());
        if (p->end_limit + (plen + slop) < front.start_limit) continue;  // out of range (1)
        if (front.end_limit + (plen + slop) < p->start_limit) continue;  // out of range (2)
        if (front.has_right && p->has_left && front.right_offset < p->left_offset) continue;  // misordered
        if (front.has_left && p->has_left && front.left_offset < p->left_offset) continue;    // misordered left offset
        matches.push_back(&*it);
      }
      const int64_t p_middle = (p->start_limit + p->end_limit) / 2 + int64_t(p->seq.size());
      std::sort(matches.begin(), matches.end(), [&](const Front* li, const Front* ri) {  // match_sorter (:313-362)
        const PopEntry& lhs = *li->e;
        const PopEntry& rhs = *ri->e;
        if (lhs.has_left != rhs.has_left) return lhs.has_left > rhs.has_left;  // reanchor to reference if possible
        if (lhs.orig_r.s != rhs.orig_r.s) return lhs.orig_r.s > rhs.orig_r.s;  // more specific is better
        const int64_t ld = std::llabs(p_middle - (lhs.start_limit + lhs.end_limit) / 2);
        const int64_t rd = std::llabs(p_middle - (rhs.start_limit + rhs.end_limit) / 2);
        if (ld != rd) return ld < rd;
        // (:347-349 compares rhs.seq.size() with itself)
        const int64_t ls = lhs.end_limit - lhs.start_limit, rs = rhs.end_limit - rhs.start_limit;
        if (ls != rs) return ls < rs;
        return false;
      });
      bool found = false;
      for (Front* fi : matches) {
        const PopEntry& front = *fi->e;
        found = true;
        if (front.has_left) {  // the front is aligned to reference
          p->has_right = true;
          p->right_offset = front.left_offset;
          if (p->has_left) output_assembly(*p);
        } else {  // merge in
          const int64_t plen = int64_t(p->seq.size());
          p->start_limit = std::max(p->start_limit, front.start_limit - plen);
          p->end_limit = std::min(p->end_limit, front.end_limit - plen);
          p->popped_r = front.popped_r;
          p->ahead_slot = -1;  // (the passes run ahead were for the old range)
          p->seq += front.seq;
          append_ids(&p->seen_read_ids, front);
          if (front.has_right) {
            p->has_right = true;
            p->right_offset = front.right_offset;
            if (p->has_left) output_assembly(*p);
          } else {
            poppers.push_back(p);  // keeps searching for something to align to on the right
          }
        }
        if (!front.has_left) fi->erased = true;  // consumed
        break;
      }
      if (!found) poppers.push_back(p);  // try again after popping more
    }
  }

  void output_assembly(const PopEntry& p) {  // :615-651
    bg_trace_assembly_rec a;
    a.from_pop = true;
    a.assembly_id = allocate_assembly_id();
    a.left_offset = p.left_offset;
    a.right_offset = p.right_offset;
    a.seq = p.seq;
    append_ids(&a.rc_read_ids, p);
    std::sort(a.rc_read_ids.begin(), a.rc_read_ids.end());
    a.rc_read_ids.erase(std::unique(a.rc_read_ids.begin(), a.rc_read_ids.end()), a.rc_read_ids.end());
    if (a.left_offset == a.right_offset && a.seq.empty()) return;
    // (:631-636: matches_reference is only ever set inside `if (k_trace_dbg)`)
    out.push_back(std::move(a));
  }
};

}  // namespace

extern "C" uint64_t bg_trace_result_anchor_drops(const bg_trace_result* r) { return r ? r->anchor_drops.size() : 0; }

extern "C" int bg_pop_trace(bg_tracer* t, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq,
                            const uint64_t* ext_len, int64_t end_pos, int64_t start_offset, int64_t limit_offset, int rev_comp,
                            const bg_trace_result* const* push, uint32_t n_push, bg_trace_result** out) {
  if (t) tl_pool = t->pool;
  if (!t || !out || (n_extents && (!ext_offset || !ext_seq || !ext_len)) || (n_push && !push)) return fail("bg_pop_trace: null argument");
  World& w = t->w;
  BG_HIP(hipSetDevice(w.ctx->device));
  BG_TRY(session_extents(&t->exts, "bg_pop_trace", n_extents, ext_offset, ext_seq, ext_len, end_pos));
  const std::vector<Extent>& ext = t->exts->ext;
  const std::vector<Extent>& rc = t->exts->rc();  // (either direction's add_reference walks the other one's bases)
  if (limit_offset > end_pos) limit_offset = end_pos;
  std::unique_ptr<bg_trace_result> res(new bg_trace_result);
  memset(&res->stats, 0, sizeof(res->stats));
  PopTracer pt{w, w.ctx, rev_comp ? rc : ext, rev_comp ? ext : rc, end_pos};
  std::vector<const bg_anchor_drop_rec*> drops;
  for (uint32_t i = 0; i < n_push; ++i) {
    if (!push[i]) return fail("bg_pop_trace: null push result");
    for (const auto& d : push[i]->anchor_drops) drops.push_back(&d);
  }
  const double t0 = now_seconds();
  BG_TRY(pt.add_anchor_drops(drops, rev_comp != 0));
  const double t1 = now_seconds();
  if (rev_comp) BG_TRY(pt.add_reference(end_pos - limit_offset, end_pos - start_offset));
  else BG_TRY(pt.add_reference(start_offset, limit_offset));
  const double t2 = now_seconds();
  const size_t n_fronts = pt.fronts.size(), n_poppers = pt.poppers.size();
  BG_TRY(pt.assemble());
  if (getenv("BG_TRACER_TIMING"))
    fprintf(stderr, "bg_pop_trace: %zu anchor drops %.3f s, reference %.3f s (%zu fronts), assemble %.3f s (%zu poppers, %llu device rounds), %zu assemblies; %llu passes in %u batches, matching %.3f s; reference: walk %.3f replay %.3f kept %.3f prefix reads %.3f items %.3f materialize %.3f\n",
            drops.size(), t1 - t0, t2 - t1, n_fronts, now_seconds() - t2, n_poppers, (unsigned long long)pt.device_rounds, pt.out.size(),
            (unsigned long long)pt.passes_run, pt.ahead_batch, double(pt.match_seconds_us) * 1e-6, pt.ref_t[0], pt.ref_t[1], pt.ref_t[2], pt.ref_t[3],
            pt.ref_t[4], pt.ref_t[5]);
  // sorter(left_offset_less_than) -> pair_counter -> (reversed back) (reversable_tracer.cpp:118-126)
  std::sort(pt.out.begin(), pt.out.end(), [](const bg_trace_assembly_rec& a, const bg_trace_assembly_rec& b) { return a.left_offset < b.left_offset; });
  std::vector<bg_trace_assembly_rec> counted;
  {
    PairCounter pc{w, &counted};
    for (auto& a : pt.out)
      if (!pc.add(std::move(a))) return fail("bg_pop_trace: pair_counter: assemblies out of left-offset order");
    pc.flush();
  }
  if (rev_comp)
    for (auto& a : counted) reverse_rec(w, &a, end_pos, false);
  res->a = std::move(counted);
  res->stats.traces = drops.size();
  res->stats.output_count = res->a.size();
  res->stats.device_rounds = pt.device_rounds;
  res->stats.device_queries = pt.device_queries;
  *out = res.release();
  return BG_OK;
}
