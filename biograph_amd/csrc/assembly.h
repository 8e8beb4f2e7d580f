// assembly.h — what tracer.cpp and align.cpp share: the assembly record behind bg_trace_result, the scaffold view and
// the steps of align.cpp that the tracer session runs itself (the anchor_dropper sits between tracer and pair_counter,
// reversable_tracer.cpp:124-125).  Internal; the C ABI is in include/bgseq.h.
#pragma once
#include <stdint.h>

#include <functional>
#include <future>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/bgseq.h"

struct bg_aligned_var_rec {  // variants::aligned_var (assemble.h:124-152)
  int64_t left_offset = 0, right_offset = 0;
  std::string seq;
  bool operator<(const bg_aligned_var_rec& r) const {
    if (left_offset != r.left_offset) return left_offset < r.left_offset;
    if (right_offset != r.right_offset) return right_offset < r.right_offset;
    return seq < r.seq;
  }
  bool empty() const { return left_offset == right_offset && seq.empty(); }
};

struct bg_trace_assembly_rec {  // variants::assembly (assemble.h:285-380), the fields this path fills
  int64_t left_offset = 0, right_offset = 0;
  int left_anchor_len = 0, right_anchor_len = 0;
  std::string seq;
  unsigned min_overlap = 0, trace_steps = 0, left_anchor_ambiguous_bases = 0;
  bool matches_reference = false;
  int64_t score = 0;
  std::vector<uint32_t> rc_read_ids, left_pair_matches, right_pair_matches;
  std::vector<bg_aligned_var_rec> aligned_variants;
  // calc_coverage / simple_genotype_filter (csrc/coverage.cpp)
  bool has_coverage = false;
  std::vector<int> coverage, pair_coverage;  // interbase: seq.size() + 1 values
  int other_depth = 0, other_pair_depth = 0, ref_depth = 0, strand_count = 0;
  double genotype_quality = 0;
  // assembly_ml_features (assemble.h:191-199) and the number of assemblies merged into this one (merged_assembly_ids)
  bool has_ml = false;
  int ml_score = 0, ml_refspan = 0, ml_lanch = 0, ml_ranch = 0;
  float ml_refgc = 0, ml_altgc = 0;
  uint64_t ml_alt_len = 0;
  uint32_t merged = 0;
  // assembly::assembly_id (assemble.h:295): one id per traced assembly, inherited by everything cut or copied from it
  // (split_assembly, the anchor_dropper's anchored copy); rvg_exclude groups sections by it
  uint64_t assembly_id = 0;
  // assembly::tags (assemble.h:377): {"PUSH"} from the push tracer (tracer.cpp:1165,1237), {"POP"} from the pop tracer
  // (pop_tracer.cpp:618); nothing downstream merges them.  As string sets {"POP"} < {"PUSH"} (canon_assembly_order).
  bool from_pop = false;
  // set by the push tracer on an assembly that ends in an anchor drop (rejoin::anchor_drop, tracer.cpp:1199-1203)
  bool anchor_drop = false;
};

struct bg_anchor_drop_rec {  // what tracer::output_assembly hands to report_anchor_drop_func (tracer.cpp:1233-1235),
  bg_trace_assembly_rec a;    // in FORWARD coordinates (reversable_tracer.cpp:28-35), rc_read_ids still on board
  bool right_anchor = false;
};

struct bg_half_aligned_rec {  // variants::half_aligned_assembly (assemble.h:449-465), in FORWARD coordinates
  int64_t offset = 0;
  bool right_anchor = false;  // false: anchored on the left at `offset`, the sequence runs on from there; true: ends there
  std::string seq;
  uint64_t assembly_id = 0;
};

struct bg_trace_result {
  std::vector<bg_trace_assembly_rec> a;
  bg_trace_stats stats;
  std::vector<bg_anchor_drop_rec> anchor_drops;  // bg_tracer_trace -> bg_pop_trace
  std::vector<bg_half_aligned_rec> half_aligned;  // what the anchor_dropper could not anchor on the right (report_half_aligned_func)
  // bg_tracer_trace_async: the host steps behind the tracer (anchor_dropper, pair_counter) still running on the session's helper
  // thread fill `a` and `half_aligned`; bg_trace_result_wait takes their verdict.  Invalid for every other result.
  std::future<int> pending;
  std::string pending_error;
  ~bg_trace_result() { if (pending.valid()) pending.wait(); }
};

namespace bg {

// Every byte one of A C G T?  Without an early exit, so that the loop is a few vector compares per 16 bytes: the checks of
// whole chromosomes at the entry points (a branch per base took 50 ms per 64 Mb call).
inline bool all_acgt(const char* s, uint64_t n) {
  unsigned bad = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const char c = s[i];
    bad |= unsigned(!((c == 'A') | (c == 'C') | (c == 'G') | (c == 'T')));
  }
  return bad == 0;
}

// variants::scaffold (scaffold.h) as a view: extents point into storage the caller keeps alive.
struct ScaffoldView {
  struct Ext { int64_t offset; const char* p; int64_t n; };
  std::vector<Ext> ext;
  int64_t end_pos = 0;
  ScaffoldView sub(int64_t start, int64_t len) const;           // scaffold::subscaffold (scaffold.cpp:46-80)
  int64_t shared_prefix_length(const char* s, int64_t n) const;  // scaffold.cpp:150-161
  int64_t shared_suffix_length(const char* s, int64_t n) const;  // rev_comp().shared_prefix_length(rev_comp(seq)), align.cpp:34-35
};

bool canon_assembly_less(const bg_trace_assembly_rec& a, const bg_trace_assembly_rec& b);
uint64_t allocate_assembly_id();  // assemble.cpp:873-877: process-wide, starts at 1
// what deduper::combine / exact_deduper keep of the assembly they swallow (dedup.cpp:51-53,71-78,171-183)
void merge_bookkeeping(bg_trace_assembly_rec& target, const bg_trace_assembly_rec& a);  // canon_assembly_order (align.cpp)

struct CheckFailed { std::string what; };  // a CHECK(...) of the reference: it aborts there; here the ABI call fails

using AssemblySink = std::function<void(bg_trace_assembly_rec&&)>;

// sorted_output_pipeline_step (assemble.cpp:291-346): a queue in canon order drained up to a left offset.
class SortedOutput {
 public:
  explicit SortedOutput(AssemblySink out) : out_(std::move(out)) {}
  void flush() { flush_sorted_to(0x7fffffff); }

 protected:
  void flush_sorted_to(int64_t offset);
  void sort_and_output(bg_trace_assembly_rec&& a);
  // left offsets of assemblies a step still holds back: nothing at or beyond the first of them leaves the queue
  void track_left_offset(int64_t offset);
  void untrack_left_offset(int64_t offset);

 private:
  AssemblySink out_;
  std::vector<bg_trace_assembly_rec> queue_;  // ascending canon order, equal keys in arrival order
  int64_t flush_point_ = INT32_MIN;
  std::multiset<int64_t> tracked_;
};

struct AlignParams {  // the assemble_options fields the steps of align.cpp read
  uint64_t min_overlap = 100, read_ahead_distance = 100000, min_anchor_drop_overlap = 15;
  int ref_align_factor = 4, max_ref_align_bases = 30, max_kmer_size = 31;
  bool trace_reference_assemblies = false;
  static AlignParams from(const bg_trace_options& o);
};

// anchor_dropper (anchor_drop.cpp): gives dead-end assemblies (right_anchor_len == 0) a right anchor on the
// reference ahead of them, or drops them.
class AnchorDropper : public SortedOutput {
 public:
  AnchorDropper(const AlignParams& p, const ScaffoldView& sc, AssemblySink out);
  void add(bg_trace_assembly_rec&& a);
  // assemble_options::report_half_aligned_func (anchor_drop.cpp:301-329): offset of the left anchor's end, the bases from there
  // on that found no place on the reference, the assembly's id
  std::function<void(int64_t, const std::string&, uint64_t)> report_half_aligned;

 private:
  void advance_range_to(int64_t trail_behind, int64_t read_ahead);
  bool try_long_rejoin(bg_trace_assembly_rec& a);
  AlignParams p_;
  const ScaffoldView& sc_;
  int kmer_size_;
  int64_t kmer_skip_len_;
  uint64_t kmer_mask_;
  // scaffold::iterator of m_read_ahead (scaffold.h:14-60)
  size_t it_ext_ = 0;
  int64_t it_pos_ = 0, it_offset_ = 0;
  int64_t kmer_needed_ = 0;
  uint64_t kmer_ = 0;
  std::unordered_map<uint64_t, std::vector<int64_t>> kmers_;  // values in insertion order; equal_range order = backwards
};

}  // namespace bg
