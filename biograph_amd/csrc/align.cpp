// align.cpp — row f.2, second slice: what the reference does with the tracers' assemblies before anything is counted
// or genotyped, as host steps over bg_trace_result:
//
//   anchor_dropper   modules/variants/anchor_drop.cpp      between tracer and pair_counter (reversable_tracer.cpp:124-125)
//   sorter           modules/variants/sort.cpp             head of the serial pipeline (pipeline.cpp:48)
//   ref_trimmer      modules/variants/trim_ref.cpp         pipeline.cpp:52
//   deduper          modules/variants/dedup.cpp:7-144      pipeline.cpp:53
//   aligner          modules/variants/align.cpp:11-385     add_standard_variants_pipeline, pipeline.cpp:92-95
//   align_splitter   modules/variants/align.cpp:387-460
//   normalizer       modules/variants/normalize.cpp:7-126  pipeline.cpp:96
//   exact_deduper    modules/variants/dedup.cpp:146-216    pipeline.cpp:97
//   vcf_padder       modules/variants/normalize.cpp:128-162 pipeline.cpp:98
//
// These are per-assembly string work on a few thousand assemblies per megabase, orders of magnitude below the
// tracer's device walks; they run on the host, single threaded, in the reference's order.  Where the reference's
// result depends on a container's iteration order it is written out: std::upper_bound's probes over the match sizes
// (align.cpp:262-278) and unordered_multimap::equal_range returning a key's values newest first (libstdc++).
//
// Half-anchored assemblies do not occur behind the push tracer (it reports them through report_half_aligned_func,
// anchor_drop.cpp:303-339, which this library drops) and are not modelled.  The sorter is a stable sort (std::sort in
// the reference: the order of assemblies that compare equal in canon order is unspecified there).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>

#include "assembly.h"
#include "bg_internal.h"

namespace bg {

namespace {

void check(bool cond, const char* what) {
  if (!cond) throw CheckFailed{what};
}

int64_t shared_prefix(const char* a, int64_t na, const char* b, int64_t nb) {
  const int64_t n = std::min(na, nb);
  int64_t i = 0;
  while (i < n && a[i] == b[i]) ++i;
  return i;
}
int64_t shared_suffix(const char* a, int64_t na, const char* b, int64_t nb) {
  const int64_t n = std::min(na, nb);
  int64_t i = 0;
  while (i < n && a[na - 1 - i] == b[nb - 1 - i]) ++i;
  return i;
}

inline unsigned code_of(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }

// canon_assembly_order, GRAPH_DISCOVER flavour (assemble.cpp:1046-1082) for fully anchored, untagged assemblies.
bool canon_less(const bg_trace_assembly_rec& a, const bg_trace_assembly_rec& b) {
  const int64_t amin = std::min(a.left_offset, a.right_offset), bmin = std::min(b.left_offset, b.right_offset);
  if (amin != bmin) return amin < bmin;
  if (a.matches_reference != b.matches_reference) return a.matches_reference < b.matches_reference;
  const int64_t amax = std::max(a.left_offset, a.right_offset), bmax = std::max(b.left_offset, b.right_offset);
  if (amax != bmax) return amax > bmax;
  if (a.seq != b.seq) return a.seq < b.seq;
  if (a.from_pop != b.from_pop) return a.from_pop;  // COMPARE(tags): {"POP"} < {"PUSH"}
  if (a.left_anchor_len != b.left_anchor_len) return a.left_anchor_len < b.left_anchor_len;
  if (a.right_anchor_len != b.right_anchor_len) return a.right_anchor_len < b.right_anchor_len;
  if (a.score != b.score) return a.score < b.score;
  if (a.left_pair_matches.size() != b.left_pair_matches.size()) return a.left_pair_matches.size() < b.left_pair_matches.size();
  if (a.right_pair_matches.size() != b.right_pair_matches.size()) return a.right_pair_matches.size() < b.right_pair_matches.size();
  if (a.rc_read_ids.size() != b.rc_read_ids.size()) return a.rc_read_ids.size() < b.rc_read_ids.size();
  return false;
}

// split_assembly (assemble.cpp:128-231)
std::pair<bg_trace_assembly_rec, bg_trace_assembly_rec> split_assembly(const bg_trace_assembly_rec& a, int64_t split_pos,
                                                                      int64_t offset_split_pos) {
  check(offset_split_pos >= 0 && offset_split_pos <= a.right_offset - a.left_offset, "split_assembly: reference offset outside the assembly");
  check(split_pos >= 0 && split_pos <= int64_t(a.seq.size()), "split_assembly: position outside the sequence");
  bg_trace_assembly_rec left = a, right = a;
  left.right_offset = right.left_offset = a.left_offset + offset_split_pos;
  left.seq = a.seq.substr(0, split_pos);
  right.seq = a.seq.substr(split_pos);
  if (a.left_anchor_len > split_pos) {
    right.left_anchor_len = int(a.left_anchor_len - split_pos);
    left.left_anchor_len = int(split_pos);
  } else {
    right.left_anchor_len = 0;
  }
  const int64_t rest = int64_t(a.seq.size()) - split_pos;
  if (a.right_anchor_len > rest) {
    left.right_anchor_len = int(a.right_anchor_len - rest);
    right.right_anchor_len = int(rest);
  } else {
    left.right_anchor_len = 0;
  }
  for (const bg_trace_assembly_rec* part : {&left, &right}) {
    check(part->left_anchor_len + part->right_anchor_len <= int64_t(part->seq.size()), "split_assembly: anchors longer than a part's sequence");
    check(part->left_anchor_len + part->right_anchor_len <= part->right_offset - part->left_offset, "split_assembly: anchors longer than a part's reference span");
  }
  left.aligned_variants.clear();
  right.aligned_variants.clear();
  int64_t seq_offset = 0, ref_offset = a.left_offset;
  for (const auto& v : a.aligned_variants) {
    seq_offset += v.left_offset - ref_offset;
    ref_offset = v.left_offset;
    if (seq_offset >= int64_t(left.seq.size()) && v.left_offset >= right.left_offset) {
      right.aligned_variants.push_back(v);
    } else {
      check(seq_offset + int64_t(v.seq.size()) <= int64_t(left.seq.size()), "split_assembly: a variant crosses the split (sequence)");
      check(v.right_offset <= left.right_offset, "split_assembly: a variant crosses the split (reference)");
      left.aligned_variants.push_back(v);
    }
    seq_offset += int64_t(v.seq.size());
    ref_offset = v.right_offset;
  }
  return {std::move(left), std::move(right)};
}

// ---- aligner::find_match and friends (align.cpp:156-385) ------------------------------------------------------------
enum Anchor { ANCHORED_TO_BOTH, ANCHORED_TO_LEFT, ANCHORED_TO_RIGHT };

struct Slice { const char* p; int64_t n; };

bool find_match(Slice seq, const ScaffoldView& s, int64_t match_len, int64_t* seq_match_start, int64_t* scaffold_match_start,
                Anchor anchor) {
  if (match_len > s.end_pos || match_len > seq.n) return false;
  int64_t seq_anchor = 0, scaffold_anchor = 0;
  if (anchor == ANCHORED_TO_BOTH) {
    seq_anchor = (seq.n - match_len) / 2;
    scaffold_anchor = (s.end_pos - match_len) / 2;
  } else if (anchor == ANCHORED_TO_RIGHT) {
    seq_anchor = seq.n;
    scaffold_anchor = s.end_pos;
  }
  const int k = int(std::min<int64_t>(match_len, 30));
  const uint64_t mask = ~((~uint64_t(0)) << (k * 2));
  std::unordered_map<uint64_t, std::vector<int64_t>> seq_kmers;  // offsets ascending; equal_range visits them descending
  {
    uint64_t km = 0;
    for (int64_t i = 0; i < seq.n; ++i) {
      km = ((km << 2) | code_of(seq.p[i])) & mask;
      const int64_t off = i + 1 - k;
      if (off < 0) continue;
      if (off + match_len > seq.n) break;
      seq_kmers[km].push_back(off);
    }
  }
  int64_t best = INT64_MAX;
  for (const auto& e : s.ext) {
    uint64_t km = 0;
    for (int64_t i = 0; i < e.n; ++i) {
      km = ((km << 2) | code_of(e.p[i])) & mask;
      const int64_t in_ext = i + 1 - k;
      if (in_ext < 0) continue;
      if (in_ext + match_len > e.n) break;
      auto it = seq_kmers.find(km);
      if (it == seq_kmers.end()) continue;
      const int64_t scaffold_offset = e.offset + in_ext;
      const int64_t scaffold_distance = scaffold_offset - scaffold_anchor;
      for (auto so = it->second.rbegin(); so != it->second.rend(); ++so) {
        const int64_t distance = llabs((*so - seq_anchor) - scaffold_distance);
        if (distance > best) continue;
        if (match_len > k && memcmp(e.p + in_ext + k, seq.p + *so + k, size_t(match_len - k)) != 0) continue;
        best = distance;
        *seq_match_start = *so;
        *scaffold_match_start = scaffold_offset;
      }
    }
  }
  return best != INT64_MAX;
}

bool find_biggest_match(const AlignParams& o, Slice seq, const ScaffoldView& s, int64_t* match_len, int64_t* seq_match_start,
                        int64_t* scaffold_match_start, int64_t* min_match_size, Anchor anchor) {
  *min_match_size = std::max<int64_t>(seq.n, s.end_pos) / o.ref_align_factor;
  if (*min_match_size < 1) *min_match_size = 1;
  if (*min_match_size > o.max_ref_align_bases) *min_match_size = o.max_ref_align_bases;
  if (anchor != ANCHORED_TO_BOTH && *min_match_size < int64_t(o.min_anchor_drop_overlap)) *min_match_size = int64_t(o.min_anchor_drop_overlap);
  const int64_t max_match_size = std::min<int64_t>(seq.n, s.end_pos);
  if (max_match_size < *min_match_size) return false;
  *match_len = 0;
  // std::upper_bound(irange(min, max + 1), false, comp): comp(false, size) is true when `size` has no match
  int64_t first = *min_match_size, n = max_match_size + 1 - *min_match_size;
  while (n > 0) {
    const int64_t half = n >> 1, size = first + half;
    int64_t so, ro;
    bool go_left;
    if (find_match(seq, s, size, &so, &ro, anchor)) {
      if (size > *match_len) { *match_len = size; *seq_match_start = so; *scaffold_match_start = ro; }
      go_left = false;
    } else {
      go_left = true;
    }
    if (go_left) {
      n = half;
    } else {
      first = size + 1;
      n = n - half - 1;
    }
  }
  if (!*match_len) {
    check(first == *min_match_size, "find_biggest_match: no match, yet the search moved");
    return false;
  }
  check(first == *match_len + 1, "find_biggest_match: match sizes are not monotone");
  return true;
}

bool find_end_matches(const AlignParams& o, Slice seq, const ScaffoldView& s, int64_t* match_len, int64_t* seq_match_start,
                      int64_t* scaffold_match_start, int64_t max_match_size) {
  const int64_t shortest = std::min<int64_t>(seq.n, s.end_pos);
  max_match_size = std::min(shortest, max_match_size);
  for (*match_len = max_match_size; *match_len >= 1; --*match_len) {
    int64_t search_len;
    switch (*match_len) {
      case 1: search_len = 2; break;
      case 2: search_len = 3; break;
      case 3: search_len = 5; break;
      case 4: search_len = 7; break;
      default: search_len = *match_len * o.ref_align_factor;
    }
    search_len = std::min(shortest, search_len);
    if (find_match(Slice{seq.p, search_len}, s.sub(0, search_len), *match_len, seq_match_start, scaffold_match_start, ANCHORED_TO_LEFT))
      return true;
    // as written there: match_len bases (not search_len) from search_len before the ends
    if (find_match(Slice{seq.p + (seq.n - search_len), *match_len}, s.sub(s.end_pos - search_len, *match_len), *match_len,
                   seq_match_start, scaffold_match_start, ANCHORED_TO_RIGHT)) {
      *seq_match_start += seq.n - search_len;
      *scaffold_match_start += s.end_pos - search_len;
      return true;
    }
  }
  return false;
}

bool find_biggest_match_with_ends(const AlignParams& o, Slice seq, const ScaffoldView& s, int64_t* match_len,
                                  int64_t* seq_match_start, int64_t* scaffold_match_start) {
  int64_t min_match_size;
  if (find_biggest_match(o, seq, s, match_len, seq_match_start, scaffold_match_start, &min_match_size, ANCHORED_TO_BOTH)) return true;
  return find_end_matches(o, seq, s, match_len, seq_match_start, scaffold_match_start, min_match_size - 1);
}

// ---- aligner (align.cpp:11-154) ---------------------------------------------------------------------------------------
class Aligner : public SortedOutput {
 public:
  Aligner(const AlignParams& p, const ScaffoldView& sc, AssemblySink out) : SortedOutput(std::move(out)), p_(p), sc_(sc) {}
  void add(bg_trace_assembly_rec&& a) {
    flush_sorted_to(a.left_offset);
    if (a.matches_reference) { sort_and_output(std::move(a)); return; }
    check(a.left_anchor_len >= 0 && a.right_anchor_len >= 0, "aligner: negative anchor length");
    check(a.right_offset >= a.left_offset, "aligner: right offset before the left one");
    const ScaffoldView s = sc_.sub(a.left_offset, a.right_offset - a.left_offset);
    const int64_t max_len = std::min<int64_t>(int64_t(a.seq.size()), a.right_offset - a.left_offset);
    int64_t shared_left = s.shared_prefix_length(a.seq.data(), int64_t(a.seq.size()));
    check(shared_left >= a.left_anchor_len, "aligner: the left anchor does not match the reference");
    const int64_t shared_right = s.shared_suffix_length(a.seq.data(), int64_t(a.seq.size()));
    check(shared_right >= a.right_anchor_len, "aligner: the right anchor does not match the reference");
    if (shared_left + shared_right > max_len) shared_left = max_len - shared_right;
    a.left_anchor_len = int(shared_left);
    a.right_anchor_len = int(shared_right);
    a.aligned_variants.clear();
    struct Work { int64_t left, right; Slice seq; ScaffoldView s; };
    std::vector<Work> work;
    work.push_back(Work{a.left_offset + shared_left, a.right_offset - shared_right,
                        Slice{a.seq.data() + shared_left, int64_t(a.seq.size()) - shared_right - shared_left},
                        s.sub(shared_left, a.right_offset - shared_right - shared_left - a.left_offset)});
    while (!work.empty()) {
      Work w = std::move(work.back());
      work.pop_back();
      int64_t match_len, so, ro;
      if (find_biggest_match_with_ends(p_, w.seq, w.s, &match_len, &so, &ro)) {
        check(match_len >= 0 && ro >= 0, "aligner: negative match");
        check(ro + match_len <= w.s.end_pos, "aligner: match past the end of the scaffold");
        Work left{w.left, w.left + ro, Slice{w.seq.p, so}, w.s.sub(0, ro)};
        Work right{w.left + ro + match_len, w.right, Slice{w.seq.p + so + match_len, w.seq.n - (so + match_len)},
                   w.s.sub(ro + match_len, w.s.end_pos - (ro + match_len))};
        if (!(left.left == left.right && left.seq.n == 0)) work.push_back(std::move(left));
        if (!(right.left == right.right && right.seq.n == 0)) work.push_back(std::move(right));
      } else if (!(w.left == w.right && w.seq.n == 0)) {
        a.aligned_variants.push_back(bg_aligned_var_rec{w.left, w.right, std::string(w.seq.p, size_t(w.seq.n))});
      }
    }
    std::sort(a.aligned_variants.begin(), a.aligned_variants.end());
    if (a.aligned_variants.empty()) {
      check(int64_t(a.seq.size()) == a.right_offset - a.left_offset, "aligner: no variants, yet the lengths differ");
      if (!p_.trace_reference_assemblies) return;
      a.matches_reference = true;
      a.left_anchor_len = a.right_anchor_len = 0;
    }
    sort_and_output(std::move(a));
  }

 private:
  AlignParams p_;
  const ScaffoldView& sc_;
};

// ---- align_splitter (align.cpp:387-460) -------------------------------------------------------------------------------
class AlignSplitter : public SortedOutput {
 public:
  explicit AlignSplitter(AssemblySink out) : SortedOutput(std::move(out)) {}
  void add(bg_trace_assembly_rec&& a) {
    flush_sorted_to(a.left_offset);
    int64_t ref_offset = a.left_offset;
    const int64_t orig_left = a.left_offset, orig_right = a.right_offset;
    const std::vector<bg_aligned_var_rec> vars = a.aligned_variants;
    for (const auto& v : vars) {
      check(v.left_offset >= orig_left && v.right_offset <= orig_right && v.left_offset >= ref_offset, "align_splitter: a variant lies outside its assembly");
      if (v.left_offset != ref_offset) {
        const int64_t chunk = v.left_offset - ref_offset;
        check(int64_t(a.seq.size()) >= chunk, "align_splitter: reference section longer than the sequence");
        auto parts = split_assembly(a, chunk, chunk);
        set_matches_reference(parts.first);
        sort_and_output(std::move(parts.first));
        a = std::move(parts.second);
        ref_offset = v.left_offset;
      }
      auto parts = split_assembly(a, int64_t(v.seq.size()), v.right_offset - v.left_offset);
      check(parts.first.seq == v.seq, "align_splitter: the split-off variant differs from the aligned one");
      sort_and_output(std::move(parts.first));
      a = std::move(parts.second);
      ref_offset = a.left_offset;
    }
    check(ref_offset == a.left_offset, "align_splitter: lost its place");
    if (a.left_offset == a.right_offset && a.seq.empty()) return;
    set_matches_reference(a);
    sort_and_output(std::move(a));
  }

 private:
  static void set_matches_reference(bg_trace_assembly_rec& a) {
    check(int64_t(a.seq.size()) == a.right_offset - a.left_offset, "align_splitter: a reference section changes length");
    a.matches_reference = true;
    a.left_anchor_len = a.right_anchor_len = 0;
  }
};

// ---- ref_trimmer (trim_ref.cpp:10-113) ----------------------------------------------------------------------------------
class RefTrimmer : public SortedOutput {
 public:
  RefTrimmer(const ScaffoldView& sc, AssemblySink out) : SortedOutput(std::move(out)), sc_(sc) {}
  void add(bg_trace_assembly_rec&& a) {
    flush_sorted_to(std::min(a.left_offset, a.right_offset) - 1000 /* k_max_backtrack_len, trim_ref.h:31 */);
    if (a.matches_reference) return;
    check(a.right_offset >= a.left_offset, "ref_trimmer: right offset before the left one");
    const ScaffoldView s = sc_.sub(a.left_offset, a.right_offset - a.left_offset);
    if (s.ext.empty()) { sort_and_output(std::move(a)); return; }
    int64_t shared_left = s.shared_prefix_length(a.seq.data(), int64_t(a.seq.size()));
    check(shared_left >= a.left_anchor_len, "ref_trimmer: the left anchor does not match the reference");
    int64_t shared_right = s.shared_suffix_length(a.seq.data(), int64_t(a.seq.size()));
    check(shared_right >= a.right_anchor_len, "ref_trimmer: the right anchor does not match the reference");
    const int64_t max_anchor = std::min<int64_t>(int64_t(a.seq.size()), a.right_offset - a.left_offset);
    if (shared_right + shared_left > max_anchor && shared_right + a.left_anchor_len <= max_anchor) shared_left = max_anchor - shared_right;
    if (shared_right + shared_left > max_anchor && a.right_anchor_len + shared_left <= max_anchor) shared_right = max_anchor - shared_left;
    if (shared_right + shared_left <= max_anchor) {
      a.left_anchor_len = int(shared_left);
      a.right_anchor_len = int(shared_right);
    }
    const int64_t new_size = int64_t(a.seq.size()) - a.left_anchor_len - a.right_anchor_len;
    check(new_size >= 0, "ref_trimmer: anchors longer than the sequence");
    a.seq = a.seq.substr(size_t(a.left_anchor_len), size_t(new_size));
    a.left_offset += a.left_anchor_len;
    a.right_offset -= a.right_anchor_len;
    a.left_anchor_len = a.right_anchor_len = 0;
    check(a.right_offset >= a.left_offset, "ref_trimmer: trimmed past itself");
    if (new_size == 0 && a.left_offset == a.right_offset) return;
    sort_and_output(std::move(a));
  }

 private:
  const ScaffoldView& sc_;
};

// ---- deduper (dedup.cpp:7-144) ------------------------------------------------------------------------------------------
class Deduper {
 public:
  explicit Deduper(AssemblySink out) : out_(std::move(out)) {}
  void add(bg_trace_assembly_rec&& a) {
    advance_to(a.left_offset);
    for (auto it = queued_.begin(); it != queued_.end(); ++it) {
      const bg_trace_assembly_rec& q = it->second;
      if (q.left_offset > a.left_offset + a.left_anchor_len) break;
      if (a.matches_reference || q.matches_reference) continue;
      if (a.left_offset + a.left_anchor_len != q.left_offset + q.left_anchor_len) continue;
      if (a.right_offset - a.right_anchor_len != q.right_offset - q.right_anchor_len) continue;
      check(int64_t(a.seq.size()) >= a.left_anchor_len + a.right_anchor_len && int64_t(q.seq.size()) >= q.left_anchor_len + q.right_anchor_len,
            "deduper: anchors longer than the sequence");
      const size_t an = a.seq.size() - size_t(a.left_anchor_len + a.right_anchor_len), qn = q.seq.size() - size_t(q.left_anchor_len + q.right_anchor_len);
      if (an != qn || a.seq.compare(size_t(a.left_anchor_len), an, q.seq, size_t(q.left_anchor_len), qn) != 0) continue;
      bg_trace_assembly_rec old = std::move(it->second);
      queued_.erase(it);
      a = combine(std::move(old), std::move(a));
      break;
    }
    const int64_t lo = a.left_offset;
    queued_.emplace(lo, std::move(a));
  }
  void flush() {
    advance_to(0x7fffffff);
    check(queued_.empty(), "deduper: assemblies left after the flush");
  }

 private:
  static bg_trace_assembly_rec combine(bg_trace_assembly_rec a, bg_trace_assembly_rec target) {  // :7-80
    if (a.left_offset < target.left_offset) std::swap(a, target);
    const int64_t add = a.right_offset - target.right_offset;
    if (add > 0) {
      check(add < int64_t(a.seq.size()), "deduper: combining would add the whole sequence");
      target.seq += a.seq.substr(a.seq.size() - size_t(add));
      target.right_anchor_len += int(add);
      target.right_offset += add;
    }
    for (auto m : {&bg_trace_assembly_rec::left_pair_matches, &bg_trace_assembly_rec::right_pair_matches}) {
      auto& pm = target.*m;
      pm.insert(pm.end(), (a.*m).begin(), (a.*m).end());
      std::sort(pm.begin(), pm.end());
      pm.erase(std::unique(pm.begin(), pm.end()), pm.end());
    }
    check(target.left_anchor_len >= a.left_anchor_len && target.right_anchor_len >= a.right_anchor_len, "deduper: combined anchors shrink");
    merge_bookkeeping(target, a);
    target.trace_steps = std::max(target.trace_steps, a.trace_steps);
    target.min_overlap = std::max(target.min_overlap, a.min_overlap);
    target.left_anchor_ambiguous_bases = std::max(target.left_anchor_ambiguous_bases, a.left_anchor_ambiguous_bases);
    return target;
  }
  void advance_to(int64_t offset) {  // :131-141
    while (!queued_.empty()) {
      auto it = queued_.begin();
      if (it->second.left_offset + it->second.left_anchor_len >= offset) break;
      out_(std::move(it->second));
      queued_.erase(it);
    }
  }
  AssemblySink out_;
  std::multimap<int64_t, bg_trace_assembly_rec> queued_;
};

// ---- normalizer (normalize.cpp:7-126) -----------------------------------------------------------------------------------
// (release_early, for the pipeline fed piece by piece: the reference's normalizer keeps everything of an extent until the extent
// is through -- a variant may move left through a repeat of any length -- so a chromosome in one extent leaves the step only at
// the flush.  With a margin, what lies more than `margin` bases before the variant coming in leaves at once; a later variant
// that moves to before what has left breaks that, says so (broke()) and the caller runs the steps over everything instead.)
struct NormalizerBroke {};
class Normalizer : public SortedOutput {
 public:
  Normalizer(const ScaffoldView& sc, AssemblySink out) : SortedOutput(std::move(out)), sc_(sc) {}
  void release_early(int64_t margin) { early_ = margin; }
  void add(bg_trace_assembly_rec&& a) {
    const int64_t came_at = std::min(a.left_offset, a.right_offset);
    if (early_ >= 0 && came_at - early_ > released_) {
      released_ = came_at - early_;
      flush_sorted_to(released_);
    }
    if (a.matches_reference) { sort_and_output(std::move(a)); return; }
    a.aligned_variants.clear();
    for (;;) {
      // split_extent_at(a.left_offset) (scaffold.cpp:189-203)
      const char *left_p = nullptr, *after_p = nullptr;
      int64_t left_n = 0, after_n = 0;
      for (const auto& e : sc_.ext) {
        if (e.offset > a.left_offset) break;
        if (e.offset + e.n < a.left_offset) continue;
        left_p = e.p;
        left_n = a.left_offset - e.offset;
        after_p = e.p + left_n;
        after_n = e.n - left_n;
        break;
      }
      advance_to(a.left_offset - left_n - 1);
      const int64_t span = a.right_offset - a.left_offset;
      if (left_n == 0 || after_n < span) break;
      check(span != 0 || !a.seq.empty(), "normalizer: empty variant");
      const char left_base = left_p[left_n - 1];
      if (span != 0 && !a.seq.empty() && after_p[span - 1] != a.seq.back()) break;
      const char right_base = span != 0 ? after_p[span - 1] : a.seq.back();
      if (left_base != right_base) break;
      --a.left_offset;
      --a.right_offset;
      if (!a.seq.empty()) {
        a.seq.pop_back();
        a.seq.insert(a.seq.begin(), left_base);
      }
    }
    if (early_ >= 0 && std::min(a.left_offset, a.right_offset) < released_) throw NormalizerBroke{};
    sort_and_output(std::move(a));
  }

 private:
  void advance_to(int64_t new_offset) {
    if (new_offset < cur_offset_) return;
    cur_offset_ = new_offset;
    if (early_ >= 0) {  // (what an early release has let go is gone already)
      if (new_offset <= released_) return;
      released_ = new_offset;
    }
    flush_sorted_to(new_offset);
  }
  const ScaffoldView& sc_;
  int64_t cur_offset_ = 0;
  int64_t early_ = -1, released_ = INT64_MIN;
};

// ---- exact_deduper (dedup.cpp:146-216) ------------------------------------------------------------------------------------
class ExactDeduper {
 public:
  explicit ExactDeduper(AssemblySink out) : out_(std::move(out)) {}
  void add(bg_trace_assembly_rec&& a) {
    advance_to(a.left_offset);
    for (auto& kv : queued_) {
      bg_trace_assembly_rec& q = kv.second;
      if (a.left_offset != q.left_offset || a.right_offset != q.right_offset || a.seq != q.seq || a.matches_reference != q.matches_reference) continue;
      for (auto m : {&bg_trace_assembly_rec::left_pair_matches, &bg_trace_assembly_rec::right_pair_matches}) {
        auto& pm = q.*m;
        pm.insert(pm.end(), (a.*m).begin(), (a.*m).end());
        std::sort(pm.begin(), pm.end());
        pm.erase(std::unique(pm.begin(), pm.end()), pm.end());
      }
      merge_bookkeeping(q, a);
      return;
    }
    const int64_t lo = a.left_offset;
    queued_.emplace(lo, std::move(a));
  }
  void flush() {
    advance_to(0x7fffffff);
    check(queued_.empty(), "exact_deduper: assemblies left after the flush");
  }

 private:
  void advance_to(int64_t offset) {
    while (!queued_.empty() && queued_.begin()->first < offset) {
      out_(std::move(queued_.begin()->second));
      queued_.erase(queued_.begin());
    }
  }
  AssemblySink out_;
  std::multimap<int64_t, bg_trace_assembly_rec> queued_;
};

// ---- vcf_padder (normalize.cpp:128-162) -----------------------------------------------------------------------------------
class VcfPadder : public SortedOutput {
 public:
  VcfPadder(const ScaffoldView& sc, AssemblySink out) : SortedOutput(std::move(out)), sc_(sc) {}
  void add(bg_trace_assembly_rec&& a) {
    if (a.left_offset > 3) flush_sorted_to(a.left_offset - 2);
    if ((a.left_offset < a.right_offset && !a.seq.empty()) || a.left_offset <= 0) { sort_and_output(std::move(a)); return; }
    const ScaffoldView s = sc_.sub(a.left_offset - 1, 1);
    if (!(s.ext.size() == 1 && s.ext[0].offset == 0)) { sort_and_output(std::move(a)); return; }  // is_simple
    check(s.ext[0].n == 1, "vcf_padder: reference base missing");
    --a.left_offset;
    a.seq.insert(a.seq.begin(), s.ext[0].p[0]);
    sort_and_output(std::move(a));
  }

 private:
  const ScaffoldView& sc_;
};

}  // namespace

bool canon_assembly_less(const bg_trace_assembly_rec& a, const bg_trace_assembly_rec& b) { return canon_less(a, b); }
uint64_t allocate_assembly_id() {
  // Process-wide and unique like the reference's counter; a thread takes 1024 ids at a time (twenty million assemblies of a
  // chromosome, sixteen sessions and their pool threads on ONE counter: its cache line was a sixth of the tracing stage).
  static std::atomic<uint64_t> next{1};
  static thread_local uint64_t mine = 0, mine_end = 0;
  if (mine == mine_end) {
    mine = next.fetch_add(1024);
    mine_end = mine + 1024;
  }
  return mine++;
}

void merge_bookkeeping(bg_trace_assembly_rec& target, const bg_trace_assembly_rec& a) {
  target.merged += 1 + a.merged;
  if (target.has_ml || a.has_ml) {
    check(target.has_ml && a.has_ml, "ML features on one side of a merge only");
    if (target.ml_alt_len < a.ml_alt_len) {
      target.ml_score = a.ml_score;
      target.ml_refspan = a.ml_refspan;
      target.ml_lanch = a.ml_lanch;
      target.ml_ranch = a.ml_ranch;
      target.ml_refgc = a.ml_refgc;
      target.ml_altgc = a.ml_altgc;
      target.ml_alt_len = a.ml_alt_len;
    }
  }
}

// add_ml_features of biograph_variants.cpp:746-779: on every discovered assembly, right behind the sorter
static void add_ml_features(const ScaffoldView& sc, bg_trace_assembly_rec& a) {
  a.has_ml = true;
  a.ml_score = int(a.score);
  a.ml_refspan = int(a.right_offset - a.left_offset);
  a.ml_lanch = a.left_anchor_len;
  a.ml_ranch = a.right_anchor_len;
  a.ml_refgc = 0;
  if (a.ml_refspan > 0) {
    int gc = 0;
    const ScaffoldView sub = sc.sub(a.left_offset, a.right_offset - a.left_offset);
    for (const auto& e : sub.ext)
      for (int64_t i = 0; i < e.n; ++i) gc += e.p[i] == 'G' || e.p[i] == 'C';
    a.ml_refgc = float(gc) / a.ml_refspan;
  }
  a.ml_altgc = 0;
  if (!a.seq.empty()) {
    int gc = 0;
    for (char c : a.seq) gc += c == 'G' || c == 'C';
    a.ml_altgc = float(gc) / int(a.seq.size());
  }
  a.ml_alt_len = a.seq.size();
}

// ---- scaffold.cpp -----------------------------------------------------------------------------------------------------
ScaffoldView ScaffoldView::sub(int64_t start, int64_t len) const {
  const int64_t limit = start + len;
  check(start >= 0 && len >= 0 && limit >= 0, "subscaffold: negative range");
  ScaffoldView r;
  for (const Ext& e : ext) {
    if (e.offset >= limit) continue;
    if (e.offset + e.n <= start) continue;
    const int64_t rs = std::max<int64_t>(start - e.offset, 0), rl = std::min<int64_t>(limit - e.offset, e.n);
    r.ext.push_back(Ext{e.offset + rs - start, e.p + rs, rl - rs});
  }
  r.end_pos = len;
  return r;
}
int64_t ScaffoldView::shared_prefix_length(const char* s, int64_t n) const {
  if (ext.empty() || ext[0].offset != 0) return 0;
  return shared_prefix(s, n, ext[0].p, ext[0].n);
}
int64_t ScaffoldView::shared_suffix_length(const char* s, int64_t n) const {
  if (ext.empty() || ext.back().offset + ext.back().n != end_pos) return 0;
  return shared_suffix(s, n, ext.back().p, ext.back().n);
}

// ---- sorted_output_pipeline_step ----------------------------------------------------------------------------------------
void SortedOutput::track_left_offset(int64_t offset) {
  check(offset >= flush_point_, "sorted output: tracked offset behind the flush point");
  tracked_.insert(offset);
}
void SortedOutput::untrack_left_offset(int64_t offset) {
  auto it = tracked_.find(offset);
  check(it != tracked_.end(), "sorted output: offset is not tracked");
  tracked_.erase(it);
}
void SortedOutput::flush_sorted_to(int64_t offset) {
  check(offset >= flush_point_, "sorted output: flush goes backwards (input not sorted by left offset)");
  if (!tracked_.empty() && offset > *tracked_.begin()) offset = *tracked_.begin();
  size_t n = 0;
  while (n < queue_.size() && std::min(queue_[n].left_offset, queue_[n].right_offset) < offset) ++n;
  for (size_t i = 0; i < n; ++i) out_(std::move(queue_[i]));
  queue_.erase(queue_.begin(), queue_.begin() + n);
  if (queue_.empty()) flush_point_ = offset;  // the reference returns before this when something stays queued
}
void SortedOutput::sort_and_output(bg_trace_assembly_rec&& a) {
  check(std::min(a.left_offset, a.right_offset) >= flush_point_, "sorted output: assembly behind the flush point");
  auto at = std::upper_bound(queue_.begin(), queue_.end(), a, canon_less);
  queue_.insert(at, std::move(a));
}

AlignParams AlignParams::from(const bg_trace_options& o) {
  AlignParams p;
  p.min_overlap = o.min_overlap;
  p.read_ahead_distance = o.read_ahead_distance;
  p.min_anchor_drop_overlap = o.min_anchor_drop_overlap;
  p.ref_align_factor = o.ref_align_factor;
  p.max_ref_align_bases = o.max_ref_align_bases;
  p.max_kmer_size = o.max_anchor_kmer_size;
  p.trace_reference_assemblies = o.trace_reference_assemblies != 0;
  return p;
}

// ---- anchor_dropper (anchor_drop.cpp) -----------------------------------------------------------------------------------
AnchorDropper::AnchorDropper(const AlignParams& p, const ScaffoldView& sc, AssemblySink out)
    : SortedOutput(std::move(out)), p_(p), sc_(sc) {
  kmer_size_ = int(std::min<uint64_t>(p.min_overlap, uint64_t(p.max_kmer_size)));  // :21
  check(kmer_size_ >= 1 && kmer_size_ <= 31, "anchor_dropper: k-mer size outside 1..31");
  kmer_mask_ = ~((~uint64_t(0)) << (kmer_size_ * 2));
  kmer_skip_len_ = int64_t(p.min_overlap) - kmer_size_;
  it_offset_ = sc.ext.empty() ? sc.end_pos : sc.ext[0].offset;
}

void AnchorDropper::advance_range_to(int64_t trail_behind, int64_t read_ahead) {  // :76-84 with :30-63
  auto at_end = [&] { return it_ext_ >= sc_.ext.size(); };
  auto next_extent = [&] {
    ++it_ext_;
    it_pos_ = 0;
    it_offset_ = at_end() ? sc_.end_pos : sc_.ext[it_ext_].offset;
  };
  const int64_t target = trail_behind - kmer_size_;
  if (it_offset_ < target && !at_end()) {  // skip_to: scaffold.cpp:205-247
    while (it_offset_ < target) {
      const int64_t adv = std::min(target - it_offset_, sc_.ext[it_ext_].n - it_pos_);
      it_pos_ += adv;
      it_offset_ += adv;
      if (it_pos_ == sc_.ext[it_ext_].n) {
        next_extent();
        if (at_end()) break;
      }
    }
    kmer_needed_ = kmer_size_ - 1;
  }
  while (it_offset_ < read_ahead) {  // advance_read_ahead_to
    if (at_end()) break;
    if (it_pos_ == 0) kmer_needed_ = kmer_size_ - 1;
    kmer_ = (kmer_ << 2) | code_of(sc_.ext[it_ext_].p[it_pos_]);
    if (kmer_needed_) {
      --kmer_needed_;
    } else {
      kmers_[kmer_ & kmer_mask_].push_back(it_offset_ + 1 - kmer_size_);
      kmer_needed_ = kmer_skip_len_;
    }
    ++it_pos_;
    ++it_offset_;
    if (it_pos_ == sc_.ext[it_ext_].n) next_extent();
  }
  // advance_trail_behind_to(trail_behind) advances the read-ahead iterator there (:86), which is already past it:
  // no k-mer is ever taken out of the table; candidates left of the assembly are filtered where they are used.
}

bool AnchorDropper::try_long_rejoin(bg_trace_assembly_rec& a) {  // :107-192
  if (a.seq.size() < p_.min_overlap + uint64_t(a.left_anchor_len)) return false;
  int64_t best_len = 0, best_seq = 0, best_ref = 0;
  const int64_t n = int64_t(a.seq.size());
  uint64_t km = 0;
  for (int64_t i = 0; i < n; ++i) {
    km = ((km << 2) | code_of(a.seq[size_t(i)])) & kmer_mask_;
    const int64_t seq_offset = i + 1 - kmer_size_;
    if (seq_offset < a.left_anchor_len) continue;
    auto found = kmers_.find(km);
    if (found == kmers_.end()) continue;
    for (auto at = found->second.rbegin(); at != found->second.rend(); ++at) {
      const int64_t ref_at = *at;
      if (ref_at < a.left_offset + a.left_anchor_len) continue;
      const int64_t max_before = std::min<int64_t>(seq_offset - a.left_anchor_len, ref_at - a.left_offset - a.left_anchor_len);
      // split_extent_at (scaffold.cpp:189-203)
      const char *before_p = nullptr, *after_p = nullptr;
      int64_t before_n = 0, after_n = 0;
      for (const auto& e : sc_.ext) {
        if (e.offset > ref_at) break;
        if (e.offset + e.n < ref_at) continue;
        before_p = e.p;
        before_n = ref_at - e.offset;
        after_p = e.p + before_n;
        after_n = e.n - before_n;
        break;
      }
      int64_t before = shared_suffix(before_p, before_n, a.seq.data() + a.left_anchor_len, seq_offset - a.left_anchor_len);
      if (before > max_before) before = max_before;
      const int64_t shared = before + shared_prefix(after_p, after_n, a.seq.data() + seq_offset, n - seq_offset);
      if (shared <= best_len || shared < int64_t(p_.min_overlap)) continue;
      best_len = shared;
      best_seq = seq_offset - before;
      best_ref = ref_at - before;
    }
  }
  if (best_len == 0) return false;
  check(best_ref > a.left_offset, "anchor_dropper: long rejoin at the left anchor");
  a.right_offset = best_ref + best_len;
  bg_trace_assembly_rec anchored = split_assembly(a, best_seq + best_len, a.right_offset - a.left_offset).first;
  anchored.right_anchor_len = int(best_len);
  if (anchored.right_anchor_len + anchored.left_anchor_len > anchored.right_offset - anchored.left_offset)
    anchored.right_anchor_len = int(anchored.right_offset - anchored.left_offset - anchored.left_anchor_len);
  sort_and_output(std::move(anchored));
  return true;
}

void AnchorDropper::add(bg_trace_assembly_rec&& a) {  // on_assembly :194-341
  flush_sorted_to(a.left_offset);
  if (a.right_anchor_len || a.matches_reference) {
    if (!a.matches_reference) check(a.left_anchor_len != 0, "anchor_dropper: right anchor without a left one");
    sort_and_output(std::move(a));
    return;
  }
  check(a.left_anchor_len != 0, "anchor_dropper: dead end without a left anchor");
  advance_range_to(a.left_offset, a.left_offset + int64_t(p_.read_ahead_distance));
  if (a.left_anchor_len >= int64_t(a.seq.size())) return;
  if (try_long_rejoin(a)) return;
  bool have = false;
  int64_t new_seq_len = 0, new_ref_len = 0, new_match_len = 0;
  int64_t seq_start = 0, ref_start = a.left_offset;
  const int64_t n = int64_t(a.seq.size());
  for (;;) {
    const int64_t shared = sc_.sub(ref_start, n - seq_start).shared_prefix_length(a.seq.data() + seq_start, n - seq_start);
    seq_start += shared;
    check(seq_start >= a.left_anchor_len, "anchor_dropper: the left anchor does not match the reference");
    ref_start += shared;
    const Slice variant{a.seq.data() + seq_start, n - seq_start};
    if (variant.n == 0) break;
    const ScaffoldView variant_s = sc_.sub(ref_start, variant.n * 5 / 4 + 3);
    int64_t match_len, so, ro, min_match;
    if (!find_biggest_match(p_, variant, variant_s, &match_len, &so, &ro, &min_match, ANCHORED_TO_LEFT)) break;
    seq_start += so + match_len;
    ref_start += ro + match_len;
    have = true;
    new_seq_len = seq_start;
    new_ref_len = ref_start - a.left_offset;
    new_match_len = match_len;
  }
  if (!have) {  // nothing goes down the pipeline; the unanchored bases are reported (:316-329)
    if (report_half_aligned) report_half_aligned(a.left_offset + a.left_anchor_len, a.seq.substr(size_t(a.left_anchor_len)), a.assembly_id);
    return;
  }
  a.right_offset = a.left_offset + new_ref_len;
  auto parts = split_assembly(a, new_seq_len, new_ref_len);
  bg_trace_assembly_rec& anchored = parts.first;
  anchored.right_anchor_len = int(new_match_len);
  if (report_half_aligned && !parts.second.seq.empty()) {  // what lies behind the match is discarded (:301-311)
    check(parts.second.left_anchor_len == 0, "anchor_dropper: the discarded part has a left anchor");
    report_half_aligned(parts.second.left_offset, parts.second.seq, parts.second.assembly_id);
  }
  sort_and_output(std::move(anchored));
}

}  // namespace bg

using namespace bg;

// ---- C ABI ---------------------------------------------------------------------------------------------------------------
extern "C" int bg_trace_result_new(bg_trace_result** out) {
  if (!out) { set_error("bg_trace_result_new: null argument"); return BG_EINVAL; }
  *out = new bg_trace_result;
  memset(&(*out)->stats, 0, sizeof((*out)->stats));
  return BG_OK;
}

extern "C" int bg_trace_result_add(bg_trace_result* r, const bg_trace_assembly* a) {
  if (!r || !a || (a->seq_len && !a->seq)) { set_error("bg_trace_result_add: null argument"); return BG_EINVAL; }
  for (uint64_t i = 0; i < a->seq_len; ++i)
    if (a->seq[i] != 'A' && a->seq[i] != 'C' && a->seq[i] != 'G' && a->seq[i] != 'T') { set_error("bg_trace_result_add: sequences hold ACGT only"); return BG_EINVAL; }
  bg_trace_assembly_rec x;
  x.left_offset = a->left_offset;
  x.right_offset = a->right_offset;
  x.left_anchor_len = a->left_anchor_len;
  x.right_anchor_len = a->right_anchor_len;
  x.seq.assign(a->seq ? a->seq : "", size_t(a->seq_len));
  x.min_overlap = a->min_overlap;
  x.trace_steps = a->trace_steps;
  x.left_anchor_ambiguous_bases = a->left_anchor_ambiguous_bases;
  x.matches_reference = a->matches_reference != 0;
  x.score = a->score;
  if (a->n_rc_read_ids) x.rc_read_ids.assign(a->rc_read_ids, a->rc_read_ids + a->n_rc_read_ids);
  if (a->n_left_pair_matches) x.left_pair_matches.assign(a->left_pair_matches, a->left_pair_matches + a->n_left_pair_matches);
  if (a->n_right_pair_matches) x.right_pair_matches.assign(a->right_pair_matches, a->right_pair_matches + a->n_right_pair_matches);
  x.assembly_id = a->assembly_id ? a->assembly_id : allocate_assembly_id();  // assembly::assembly(), assemble.cpp:1248
  x.from_pop = a->from_pop != 0;
  r->a.push_back(std::move(x));
  return BG_OK;
}

extern "C" uint64_t bg_trace_result_half_aligned_count(const bg_trace_result* r) { return r ? r->half_aligned.size() : 0; }

extern "C" int bg_trace_result_half_aligned(const bg_trace_result* r, uint64_t i, int64_t* offset, int* right_anchor, const char** seq,
                                            uint64_t* seq_len, uint64_t* assembly_id) {
  if (!r || i >= r->half_aligned.size() || !offset || !right_anchor || !seq || !seq_len || !assembly_id) {
    set_error("bg_trace_result_half_aligned: bad argument");
    return BG_EINVAL;
  }
  const bg_half_aligned_rec& h = r->half_aligned[i];
  *offset = h.offset;
  *right_anchor = h.right_anchor ? 1 : 0;
  *seq = h.seq.data();
  *seq_len = h.seq.size();
  *assembly_id = h.assembly_id;
  return BG_OK;
}

extern "C" uint64_t bg_trace_result_var_count(const bg_trace_result* r, uint64_t i) {
  return (r && i < r->a.size()) ? r->a[i].aligned_variants.size() : 0;
}

extern "C" int bg_trace_result_var(const bg_trace_result* r, uint64_t i, uint64_t j, int64_t* left_offset, int64_t* right_offset,
                                   const char** seq, uint64_t* seq_len) {
  if (!r || i >= r->a.size() || j >= r->a[i].aligned_variants.size() || !left_offset || !right_offset || !seq || !seq_len) {
    set_error("bg_trace_result_var: bad argument");
    return BG_EINVAL;
  }
  const bg_aligned_var_rec& v = r->a[i].aligned_variants[j];
  *left_offset = v.left_offset;
  *right_offset = v.right_offset;
  *seq = v.seq.data();
  *seq_len = v.seq.size();
  return BG_OK;
}

// the steps behind the sorter, built back to front like assemble_pipeline::make_pipeline (pipeline.cpp:81-90)
namespace {
struct Chain {
  std::unique_ptr<VcfPadder> pad;
  std::unique_ptr<ExactDeduper> xdedup;
  std::unique_ptr<Normalizer> norm;
  std::unique_ptr<AlignSplitter> split;
  std::unique_ptr<Aligner> align;
  std::unique_ptr<Deduper> dedup;
  std::unique_ptr<RefTrimmer> trim;
  AssemblySink head;  // where an assembly enters
  // sc and p are read by the steps for as long as the chain lives
  Chain(uint32_t steps, const AlignParams& p, const ScaffoldView& sc, AssemblySink out) {
    AssemblySink sink = std::move(out);
    if (steps & BG_STEP_VCF_PAD) { pad.reset(new VcfPadder(sc, sink)); VcfPadder* q = pad.get(); sink = [q](bg_trace_assembly_rec&& a) { q->add(std::move(a)); }; }
    if (steps & BG_STEP_EXACT_DEDUP) { xdedup.reset(new ExactDeduper(sink)); ExactDeduper* q = xdedup.get(); sink = [q](bg_trace_assembly_rec&& a) { q->add(std::move(a)); }; }
    if (steps & BG_STEP_NORMALIZE) { norm.reset(new Normalizer(sc, sink)); Normalizer* q = norm.get(); sink = [q](bg_trace_assembly_rec&& a) { q->add(std::move(a)); }; }
    if (steps & BG_STEP_SPLIT) { split.reset(new AlignSplitter(sink)); AlignSplitter* q = split.get(); sink = [q](bg_trace_assembly_rec&& a) { q->add(std::move(a)); }; }
    if (steps & BG_STEP_ALIGN) { align.reset(new Aligner(p, sc, sink)); Aligner* q = align.get(); sink = [q](bg_trace_assembly_rec&& a) { q->add(std::move(a)); }; }
    if (steps & BG_STEP_DEDUP) { dedup.reset(new Deduper(sink)); Deduper* q = dedup.get(); sink = [q](bg_trace_assembly_rec&& a) { q->add(std::move(a)); }; }
    if (steps & BG_STEP_TRIM_REF) { trim.reset(new RefTrimmer(sc, sink)); RefTrimmer* q = trim.get(); sink = [q](bg_trace_assembly_rec&& a) { q->add(std::move(a)); }; }
    head = std::move(sink);
  }
  void flush() {
    if (trim) trim->flush();
    if (dedup) dedup->flush();
    if (align) align->flush();
    if (split) split->flush();
    if (norm) norm->flush();
    if (xdedup) xdedup->flush();
    if (pad) pad->flush();
  }
};

int scaffold_view(const char* who, uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq, const uint64_t* ext_len,
                  int64_t end_pos, ScaffoldView* sc) {
  int64_t calc_end = 0;
  for (uint64_t i = 0; i < n_extents; ++i) {
    if (ext_offset[i] < calc_end) { set_error("%s: extents must ascend without overlap", who); return BG_EINVAL; }
    if (!all_acgt(ext_seq[i], ext_len[i])) { set_error("%s: extents hold ACGT only", who); return BG_EINVAL; }
    sc->ext.push_back(ScaffoldView::Ext{ext_offset[i], ext_seq[i], int64_t(ext_len[i])});
    calc_end = ext_offset[i] + int64_t(ext_len[i]);
  }
  if (end_pos < calc_end) { set_error("%s: end_pos lies before the end of the last extent", who); return BG_EINVAL; }
  sc->end_pos = end_pos;
  return BG_OK;
}
}  // namespace

extern "C" int bg_assemblies_process(const bg_trace_result* const* in, uint32_t n_in, uint64_t n_extents, const int64_t* ext_offset,
                                     const char* const* ext_seq, const uint64_t* ext_len, int64_t end_pos,
                                     const bg_trace_options* options, uint32_t steps, bg_trace_result** out) {
  if (!out || (n_in && !in) || (n_extents && (!ext_offset || !ext_seq || !ext_len))) { set_error("bg_assemblies_process: null argument"); return BG_EINVAL; }
  if (steps & ~uint32_t(BG_STEP_ALL)) { set_error("bg_assemblies_process: unknown step bits"); return BG_EINVAL; }
  bg_trace_options o;
  if (options) o = *options; else bg_trace_options_default(&o);
  if (o.ref_align_factor < 1) { set_error("bg_assemblies_process: ref_align_factor must be at least 1"); return BG_EINVAL; }
  ScaffoldView sc;
  BG_TRY(scaffold_view("bg_assemblies_process", n_extents, ext_offset, ext_seq, ext_len, end_pos, &sc));
  const AlignParams p = AlignParams::from(o);
  std::unique_ptr<bg_trace_result> res(new bg_trace_result);
  memset(&res->stats, 0, sizeof(res->stats));
  try {
    bg_trace_result* r = res.get();
    Chain chain(steps, p, sc, [r](bg_trace_assembly_rec&& a) { r->a.push_back(std::move(a)); });
    std::unique_ptr<AnchorDropper> drop;
    std::vector<bg_trace_assembly_rec> all;
    AssemblySink collect = [&](bg_trace_assembly_rec&& a) { all.push_back(std::move(a)); };
    if (steps & BG_STEP_ANCHOR_DROP) drop.reset(new AnchorDropper(p, sc, collect));
    for (uint32_t i = 0; i < n_in; ++i) {
      if (!in[i]) { set_error("bg_assemblies_process: null input"); return BG_EINVAL; }
      for (const auto& a : in[i]->a) {
        bg_trace_assembly_rec copy = a;
        if (drop) drop->add(std::move(copy)); else all.push_back(std::move(copy));
      }
    }
    if (drop) drop->flush();
    if (steps & BG_STEP_SORT) std::stable_sort(all.begin(), all.end(), canon_less);
    if (o.output_ml_features)  // report_discovered_assemblies_func (pipeline.cpp:49-51): once per discovered assembly
      for (auto& a : all)
        if (!a.has_ml) add_ml_features(sc, a);
    for (auto& a : all) chain.head(std::move(a));
    chain.flush();
  } catch (const CheckFailed& f) {
    set_error("bg_assemblies_process: %s (the reference aborts on this input)", f.what.c_str());
    return BG_EINVAL;
  }
  *out = res.release();
  return BG_OK;
}

// ---- the same pipeline fed piece by piece (bgbinary discovery: behind the tracers of a scaffold instead of after them) ----------
// The sorter orders a scaffold's assemblies by min(left, right) offset first (canon_less), and a piece's assemblies cannot lie
// far before the piece (the tracers read the reference from start - max_pair_distance - a read's length, the reverse direction
// from limit + read_ahead_distance back): what lies before a mark no later piece reaches is final.  bg_pipeline_feed takes the
// results of the next pieces -- in the order bg_assemblies_process would be given them -- and a mark: every assembly held that
// starts before the mark is sorted (stable: equal keys stay in arrival order) and goes through the steps.  The caller's mark is a
// promise; an assembly that arrives behind a mark already given breaks it, and bg_pipeline_feed says so (*behind = 1, nothing
// else changes: the caller runs bg_assemblies_process over everything instead).  Kept promises give bg_assemblies_process's
// output, assembly for assembly: equal sort keys share their first offset and so their batch.
struct bg_pipeline {
  bg_trace_options o;
  AlignParams p;
  ScaffoldView sc;
  uint32_t steps = 0;
  std::unique_ptr<bg_trace_result> res;
  std::unique_ptr<Chain> chain;
  std::vector<bg_trace_assembly_rec> held;
  int64_t mark = INT64_MIN;
  bool broken = false, finished = false, early_broke = false, fed = false;
};

extern "C" int bg_pipeline_open(uint64_t n_extents, const int64_t* ext_offset, const char* const* ext_seq, const uint64_t* ext_len,
                                int64_t end_pos, const bg_trace_options* options, uint32_t steps, bg_pipeline** out) {
  if (!out || (n_extents && (!ext_offset || !ext_seq || !ext_len))) { set_error("bg_pipeline_open: null argument"); return BG_EINVAL; }
  if ((steps & ~uint32_t(BG_STEP_ALL)) || (steps & BG_STEP_ANCHOR_DROP) || !(steps & BG_STEP_SORT)) {
    set_error("bg_pipeline_open: the steps must hold the sorter and not the anchor_dropper (it works on the tracers' own order)");
    return BG_EINVAL;
  }
  std::unique_ptr<bg_pipeline> pl(new bg_pipeline);
  if (options) pl->o = *options; else bg_trace_options_default(&pl->o);
  if (pl->o.ref_align_factor < 1) { set_error("bg_pipeline_open: ref_align_factor must be at least 1"); return BG_EINVAL; }
  BG_TRY(scaffold_view("bg_pipeline_open", n_extents, ext_offset, ext_seq, ext_len, end_pos, &pl->sc));  // (the caller keeps the extents)
  pl->p = AlignParams::from(pl->o);
  pl->steps = steps;
  pl->res.reset(new bg_trace_result);
  memset(&pl->res->stats, 0, sizeof(pl->res->stats));
  bg_trace_result* r = pl->res.get();
  pl->chain.reset(new Chain(steps, pl->p, pl->sc, [r](bg_trace_assembly_rec&& a) { r->a.push_back(std::move(a)); }));
  *out = pl.release();
  return BG_OK;
}

static int pipeline_run_to(bg_pipeline* pl, int64_t mark, const char* who) {
  // the assemblies before the mark, in the sorter's order, through the steps; the rest stays
  auto first_offset = [](const bg_trace_assembly_rec& a) { return std::min(a.left_offset, a.right_offset); };
  auto mid = std::stable_partition(pl->held.begin(), pl->held.end(), [&](const bg_trace_assembly_rec& a) { return first_offset(a) < mark; });
  std::stable_sort(pl->held.begin(), mid, canon_less);
  try {
    for (auto it = pl->held.begin(); it != mid; ++it) {
      if (pl->o.output_ml_features && !it->has_ml) add_ml_features(pl->sc, *it);
      pl->chain->head(std::move(*it));
    }
  } catch (const NormalizerBroke&) {
    pl->broken = true;
    pl->early_broke = true;
    return BG_OK;
  } catch (const CheckFailed& f) {
    pl->broken = true;
    set_error("%s: %s (the reference aborts on this input)", who, f.what.c_str());
    return BG_EINVAL;
  }
  pl->held.erase(pl->held.begin(), mid);
  if (mark > pl->mark) pl->mark = mark;
  return BG_OK;
}

extern "C" int bg_pipeline_feed(bg_pipeline* pl, const bg_trace_result* const* in, uint32_t n_in, int64_t mark, int* behind) {
  if (!pl || !behind || (n_in && !in)) { set_error("bg_pipeline_feed: null argument"); return BG_EINVAL; }
  *behind = 0;
  if (pl->broken || pl->finished) { set_error("bg_pipeline_feed: the pipeline is finished or failed"); return BG_EINVAL; }
  if (mark < pl->mark) { set_error("bg_pipeline_feed: the mark goes backwards"); return BG_EINVAL; }
  for (uint32_t i = 0; i < n_in; ++i) {
    if (!in[i]) { set_error("bg_pipeline_feed: null input"); return BG_EINVAL; }
    for (const auto& a : in[i]->a)
      if (std::min(a.left_offset, a.right_offset) < pl->mark) { *behind = 1; pl->broken = true; return BG_OK; }
  }
  for (uint32_t i = 0; i < n_in; ++i)
    for (const auto& a : in[i]->a) pl->held.push_back(a);
  pl->fed = true;
  BG_TRY(pipeline_run_to(pl, mark, "bg_pipeline_feed"));
  if (pl->early_broke) *behind = 1;  // (a variant moved left past what the normalizer had released early)
  return BG_OK;
}

extern "C" int bg_pipeline_release_early(bg_pipeline* pl, int64_t margin) {
  if (!pl) { set_error("bg_pipeline_release_early: null argument"); return BG_EINVAL; }
  if (pl->fed) { set_error("bg_pipeline_release_early: before the first feed"); return BG_EINVAL; }
  if (pl->chain->norm) pl->chain->norm->release_early(margin);
  return BG_OK;
}

extern "C" uint64_t bg_pipeline_ready(const bg_pipeline* pl) { return pl && pl->res ? pl->res->a.size() : 0; }

extern "C" int bg_pipeline_take(bg_pipeline* pl, bg_trace_result** out) {
  if (!pl || !out) { set_error("bg_pipeline_take: null argument"); return BG_EINVAL; }
  if (pl->broken || pl->finished) { set_error("bg_pipeline_take: the pipeline is finished or failed"); return BG_EINVAL; }
  std::unique_ptr<bg_trace_result> r(new bg_trace_result);
  memset(&r->stats, 0, sizeof(r->stats));
  r->a.swap(pl->res->a);
  *out = r.release();
  return BG_OK;
}

extern "C" int bg_pipeline_finish(bg_pipeline* pl, bg_trace_result** out) {
  if (!pl || !out) { set_error("bg_pipeline_finish: null argument"); return BG_EINVAL; }
  if (pl->broken || pl->finished) { set_error("bg_pipeline_finish: the pipeline is finished or failed"); return BG_EINVAL; }
  BG_TRY(pipeline_run_to(pl, INT64_MAX, "bg_pipeline_finish"));
  if (pl->early_broke) { *out = nullptr; return BG_OK; }  // (a variant of the tail moved left past what the normalizer had released early)
  try {
    pl->chain->flush();
  } catch (const NormalizerBroke&) {
    pl->broken = true;
    *out = nullptr;
    return BG_OK;
  } catch (const CheckFailed& f) {
    pl->broken = true;
    set_error("bg_pipeline_finish: %s (the reference aborts on this input)", f.what.c_str());
    return BG_EINVAL;
  }
  pl->finished = true;
  *out = pl->res.release();
  return BG_OK;
}

extern "C" void bg_pipeline_close(bg_pipeline* pl) { delete pl; }

extern "C" int bg_trace_result_ml(const bg_trace_result* r, uint64_t i, bg_assembly_ml* out) {
  if (!r || !out || i >= r->a.size()) { set_error("bg_trace_result_ml: bad argument"); return BG_EINVAL; }
  const bg_trace_assembly_rec& a = r->a[i];
  out->has_ml = a.has_ml ? 1 : 0;
  out->score = a.ml_score;
  out->refspan = a.ml_refspan;
  out->lanch = a.ml_lanch;
  out->ranch = a.ml_ranch;
  out->refgc = a.ml_refgc;
  out->altgc = a.ml_altgc;
  out->alt_len = a.ml_alt_len;
  out->merged = a.merged;
  return BG_OK;
}

extern "C" int bg_trace_result_set_ml(bg_trace_result* r, uint64_t i, const bg_assembly_ml* in) {
  if (!r || !in || i >= r->a.size()) { set_error("bg_trace_result_set_ml: bad argument"); return BG_EINVAL; }
  bg_trace_assembly_rec& a = r->a[i];
  a.has_ml = in->has_ml != 0;
  a.ml_score = in->score;
  a.ml_refspan = in->refspan;
  a.ml_lanch = in->lanch;
  a.ml_ranch = in->ranch;
  a.ml_refgc = in->refgc;
  a.ml_altgc = in->altgc;
  a.ml_alt_len = in->alt_len;
  a.merged = in->merged;
  return BG_OK;
}

// ---- the records of a bg_trace_result (host only) --------------------------------------------------------------------------
extern "C" uint64_t bg_trace_result_size(const bg_trace_result* r) { return r ? r->a.size() : 0; }
extern "C" void bg_trace_result_free(bg_trace_result* r) { delete r; }
extern "C" int bg_trace_result_stats(const bg_trace_result* r, bg_trace_stats* out) {
  if (!r || !out) { set_error("bg_trace_result_stats: null argument"); return BG_EINVAL; }
  *out = r->stats;
  return BG_OK;
}
extern "C" int bg_trace_result_get(const bg_trace_result* r, uint64_t i, bg_trace_assembly* out) {
  if (!r || !out || i >= r->a.size()) { set_error("bg_trace_result_get: bad argument"); return BG_EINVAL; }
  const bg_trace_assembly_rec& a = r->a[i];
  out->left_offset = a.left_offset;
  out->right_offset = a.right_offset;
  out->left_anchor_len = a.left_anchor_len;
  out->right_anchor_len = a.right_anchor_len;
  out->seq = a.seq.data();
  out->seq_len = a.seq.size();
  out->min_overlap = a.min_overlap;
  out->trace_steps = a.trace_steps;
  out->left_anchor_ambiguous_bases = a.left_anchor_ambiguous_bases;
  out->matches_reference = a.matches_reference ? 1 : 0;
  out->score = a.score;
  out->rc_read_ids = a.rc_read_ids.data();
  out->n_rc_read_ids = a.rc_read_ids.size();
  out->left_pair_matches = a.left_pair_matches.data();
  out->n_left_pair_matches = a.left_pair_matches.size();
  out->right_pair_matches = a.right_pair_matches.data();
  out->n_right_pair_matches = a.right_pair_matches.size();
  out->assembly_id = a.assembly_id;
  out->from_pop = a.from_pop ? 1 : 0;
  return BG_OK;
}
