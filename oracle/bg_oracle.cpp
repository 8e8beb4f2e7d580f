// bg_oracle.cpp — CPU restatement of the BioGraph seqset-build + overlap-walk hot path.
//
// *** TEST INFRASTRUCTURE ONLY ***  Nothing in the product path (biograph_amd/, include/)
// may link, import or call this file.  It is used by tests/, by __graft_entry__.smoke() as the
// checker, and by bench.py's `cpu_baseline` leg (kind = "port", single thread).
//
// Parity pinning: this restatement is checked by tests/test_oracle_golden.py against the
// reference's own committed fixtures (copied as data into tests/golden/):
//   * golden/e_coli_10000snp.fq -> golden/e_coli_10000snp.bg/seqset (7108 kept k-mers,
//     8444 corrected reads / 288464 bases, 19935 entries, all arrays byte-equal),
//   * datasets/lambdaToyData/benchmark/father_lambda.bg (row E closed loop, 98006 entries),
//   * builder_test.cpp:52-123 literal answers, seqset_test.py:17-104 walk answers.
//
// Every function cites the reference file:line (relative to the reference tree) it follows.
// Plain single-threaded C++17, no dependencies.

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <queue>
#include <set>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace orc {

using kmer_t = uint64_t;

// modules/bio_base/dna_base.h:38-56 — A0 C1 G2 T3 (case-insensitive); anything else is not a base.
static inline int base_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}
static const char kBases[5] = "ACGT";

// modules/bio_base/dna_sequence.cpp:330-333 — reverse complement of a k-mer (first base in high bits).
static inline kmer_t rev_comp(kmer_t in, unsigned size) {
  kmer_t out = 0;
  for (unsigned i = 0; i < size; ++i) {
    out = (out << 2) | (3 - (in & 3));
    in >>= 2;
  }
  return out;
}

// modules/bio_base/dna_sequence.cpp:341-351 — canonical = min(kmer, rc); flipped iff rc < kmer.
static inline kmer_t canonicalize(kmer_t in, unsigned size, bool& flipped) {
  kmer_t c = rev_comp(in, size);
  if (c < in) { flipped = true; return c; }
  flipped = false;
  return in;
}

// modules/bio_base/fast_read_correct.cpp:5-11
static inline kmer_t kmer_shift_left(kmer_t orig, unsigned k, int b) {
  kmer_t r = (orig << 2) | kmer_t(b);
  if (k < 32) r &= ~(~kmer_t(0) << (2 * k));
  return r;
}

static std::string revcomp_str(const std::string& s) {
  std::string r(s.size(), 'A');
  for (size_t i = 0; i < s.size(); ++i) {
    int b = base_code(s[s.size() - 1 - i]);
    r[i] = kBases[3 - b];
  }
  return r;
}

// ------------------------------------------------------------------------------------------
// k-mer counting (rows B1-B4).
// modules/build_seqset/kmer_counter.h:297-326 (rolling window, N resets and clears first flag),
// modules/build_seqset/kmer_count_table.h:54-103 (fwd/rev counts, flags swapped when flipped),
// modules/bio_mapred/kmerize_bf.cpp:290-318 (keep iff fwd+rev >= min_count; skew/overrep inactive
// at default flags).  The probabilistic pre-pass (kmer_counter.cpp:579-697) only prunes k-mers
// whose true count is < min(min_count,3) and therefore never changes the kept set.
struct KmerElem {
  uint32_t fwd = 0, rev = 0;
  uint8_t flags = 0;  // bit0 = fwd_starts_read, bit1 = rev_starts_read (kmer_set.h:20-21)
};

struct KmerSet {
  unsigned k = 30;
  std::vector<kmer_t> kmers;  // sorted canonical
  std::vector<uint8_t> flags;
  std::vector<uint32_t> fwd, rev;
  std::unordered_map<kmer_t, uint32_t> index;
  void finish() {
    index.reserve(kmers.size() * 2);
    for (size_t i = 0; i < kmers.size(); ++i) index.emplace(kmers[i], uint32_t(i));
  }
  // correct_reads.cpp:163-171 kmer_lookup_f
  bool lookup(kmer_t kmer, bool* flipped, uint32_t* idx) const {
    kmer_t canon = canonicalize(kmer, k, *flipped);
    auto it = index.find(canon);
    if (it == index.end()) return false;
    *idx = it->second;
    return true;
  }
};

static int count_kmers(const char* seq, const uint64_t* offs, uint64_t n_reads, unsigned k,
                       std::unordered_map<kmer_t, KmerElem>& table) {
  for (uint64_t r = 0; r < n_reads; ++r) {
    const char* it = seq + offs[r];
    const char* end = seq + offs[r + 1];
    bool is_first = true;
    unsigned left = k;
    kmer_t kmer = 0;
    while (it != end) {
      if (*it == 'N') {  // kmer_counter.h:306-311
        ++it;
        left = k;
        is_first = false;
        continue;
      }
      int b = base_code(*it);
      if (b < 0) return -1;  // dna_base.h:54 throws
      kmer = kmer_shift_left(kmer, k, b);
      ++it;
      if (left) --left;
      if (!left) {
        bool is_last = (it == end);
        bool flipped;
        kmer_t canon = canonicalize(kmer, k, flipped);
        KmerElem& e = table[canon];
        bool sf = is_first, sr = is_last;
        if (flipped) std::swap(sf, sr);  // kmer_count_table.h:82-84
        if (sf) e.flags |= 1;
        if (sr) e.flags |= 2;
        if (flipped) e.rev++; else e.fwd++;
        is_first = false;
      }
    }
  }
  return 0;
}

static KmerSet* make_kmer_set(const std::unordered_map<kmer_t, KmerElem>& table, unsigned k,
                              uint32_t min_count, bool keep_all) {
  std::vector<std::pair<kmer_t, KmerElem>> v;
  v.reserve(table.size());
  for (auto& kv : table) {
    uint64_t tot = uint64_t(kv.second.fwd) + kv.second.rev;
    if (keep_all || tot >= min_count) v.push_back(kv);  // kmerize_bf.cpp:291-294
  }
  std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first < b.first; });
  auto* ks = new KmerSet;
  ks->k = k;
  for (auto& kv : v) {
    ks->kmers.push_back(kv.first);
    ks->flags.push_back(kv.second.flags);
    ks->fwd.push_back(kv.second.fwd);
    ks->rev.push_back(kv.second.rev);
  }
  ks->finish();
  return ks;
}

// ------------------------------------------------------------------------------------------
// Read correction (rows C1, C2).
struct FrcParams {
  unsigned max_corrections = 8;
  unsigned min_good_run = 2;
  unsigned kmer_size = 30;
  const KmerSet* ks = nullptr;
};

struct FrcKmer { bool flipped; uint32_t index; };

struct FrcOut {
  std::string corrected;
  unsigned corrections = 0;
  std::vector<FrcKmer> kmers;
};

// modules/bio_base/fast_read_correct.cpp:16-90 — literal port (std::string for dna_sequence).
static void correct_internal(FrcOut* result, const char* in, size_t n, kmer_t kmer,
                             const FrcParams& p, unsigned min_good_run_here,
                             unsigned max_corrections, bool require_run_at_end) {
  const char* it = in;
  const char* end = in + n;
  FrcKmer ki;
  if (*it != 'N') {
    kmer_t next = kmer_shift_left(kmer, p.kmer_size, base_code(*it));
    while (p.ks->lookup(next, &ki.flipped, &ki.index)) {
      result->corrected.push_back(kBases[base_code(*it)]);
      result->kmers.push_back(ki);
      ++it;
      if (it == end) return;
      kmer = next;
      if (*it == 'N') break;
      next = kmer_shift_left(kmer, p.kmer_size, base_code(*it));
    }
  }
  if (result->corrected.size() < min_good_run_here) return;
  if (max_corrections == 0) return;
  ++it;
  FrcOut tries[4];
  unsigned best_size = 0;
  int best_b = 0;
  size_t input_left = end - it;
  for (int b = 0; b < 4; ++b) {
    kmer_t tk = kmer_shift_left(kmer, p.kmer_size, b);
    if (!p.ks->lookup(tk, &ki.flipped, &ki.index)) continue;
    tries[b].kmers.push_back(ki);
    if (it != end) {
      correct_internal(&tries[b], it, input_left, tk, p, p.min_good_run, max_corrections - 1,
                       require_run_at_end);
    }
    if (require_run_at_end && tries[b].corrected.size() < p.min_good_run) continue;
    if (tries[b].corrected.size() >= best_size) {  // :74-77 — size+1 so only strictly longer wins later
      best_size = unsigned(tries[b].corrected.size()) + 1;
      best_b = b;
    }
  }
  if (best_size) {
    result->corrected.push_back(kBases[best_b]);
    result->corrections++;
    result->corrected += tries[best_b].corrected;
    result->corrections += tries[best_b].corrections;
    result->kmers.insert(result->kmers.end(), tries[best_b].kmers.begin(), tries[best_b].kmers.end());
  }
}

// modules/bio_base/fast_read_correct.cpp:94-182
static FrcOut fast_read_correct(const char* in, size_t n, const FrcParams& p) {
  FrcOut result;
  const unsigned k = p.kmer_size;
  if (n < k) return FrcOut{};
  unsigned max_corrections = p.max_corrections;
  const char* begin = in;
  const char* end = in + n;
  const char* it = begin;
  kmer_t kmer = 0;
  unsigned left = k;
  FrcKmer ki{false, 0};
  while (left || !p.ks->lookup(kmer, &ki.flipped, &ki.index)) {
    if (it == end) return FrcOut{};
    if (*it == 'N') { ++it; left = k; continue; }
    kmer = kmer_shift_left(kmer, k, base_code(*it));
    ++it;
    if (left) --left;
  }
  FrcOut right;
  if (it == begin + k) {
    for (unsigned i = 0; i < k; ++i) result.corrected.push_back(kBases[base_code(begin[i])]);
    if (it == end) { result.kmers.push_back(ki); return result; }
    right.kmers.push_back(ki);
  } else {
    const char* kmer_start = it - k;
    std::string left_to_correct(begin, kmer_start - begin);
    std::string left_ok;
    for (unsigned i = 0; i < k; ++i) left_ok.push_back(kBases[base_code(kmer_start[i])]);
    std::reverse(left_to_correct.begin(), left_to_correct.end());
    for (char& c : left_to_correct)
      if (c != 'N') c = kBases[3 - base_code(c)];
    FrcOut lc;
    correct_internal(&lc, left_to_correct.data(), left_to_correct.size(), rev_comp(kmer, k), p, 0,
                     max_corrections, false);
    if (lc.corrected.size() != left_to_correct.size()) return FrcOut{};
    result.corrected = revcomp_str(lc.corrected);
    result.corrected += left_ok;
    for (auto& x : lc.kmers) x.flipped = !x.flipped;
    result.kmers = std::move(lc.kmers);
    std::reverse(result.kmers.begin(), result.kmers.end());
    result.kmers.push_back(ki);
    max_corrections -= lc.corrections;
    result.corrections += lc.corrections;
  }
  if (it != end) correct_internal(&right, it, end - it, kmer, p, 0, max_corrections, true);
  result.corrected += right.corrected;
  result.corrections += right.corrections;
  result.kmers.insert(result.kmers.end(), right.kmers.begin(), right.kmers.end());
  return result;
}

// modules/build_seqset/correct_reads.cpp:154-231.  trim_after_portion is parsed as float
// (biograph_create.cpp:489-490), stored in a double field (read_correction.h:42) and multiplied
// by size_t -> double, truncated to unsigned (correct_reads.cpp:174): 0.7f*150 -> 104.
struct CorrectStats {
  uint64_t corrected_reads = 0, corrected_bases = 0, failed = 0, modified = 0, modified_bases = 0,
           truncated = 0, dropped_bases = 0, ref_records = 0;
};

static bool correct_read(const char* in, size_t n, const FrcParams& p, float trim_after_portion,
                         std::string* out, CorrectStats* st) {
  if (n < p.kmer_size) return false;
  for (size_t i = 0; i < n; ++i)
    if (in[i] != 'N' && base_code(in[i]) < 0) return false;
  FrcOut res = fast_read_correct(in, n, p);
  unsigned needed = unsigned(double(trim_after_portion) * double(n));
  if (res.corrected.size() < needed || res.corrected.size() < p.kmer_size) {
    st->failed++;
    return false;
  }
  if (res.corrected.size() < n) { st->dropped_bases += n - res.corrected.size(); st->truncated++; }
  if (res.corrections) { st->modified++; st->modified_bases += res.corrections; }
  // correct_reads.cpp:195-213 — number of suffix records the REFERENCE would emit (stat only; the
  // final entry set is independent of it).
  auto starts = [&](const FrcKmer& k) {
    return (p.ks->flags[k.index] & (k.flipped ? 2 : 1)) != 0;
  };
  unsigned nf = 0, nr = 0;
  for (auto& k : res.kmers) { if (nf > 0 && starts(k)) break; ++nf; }
  for (auto it = res.kmers.rbegin(); it != res.kmers.rend(); ++it) {
    FrcKmer k{!it->flipped, it->index};
    if (nr > 0 && starts(k)) break;
    ++nr;
  }
  st->ref_records += nf + nr;
  st->corrected_reads++;
  st->corrected_bases += res.corrected.size();
  *out = std::move(res.corrected);
  return true;
}

// ------------------------------------------------------------------------------------------
// Entry set (row E): prefix-free maxima of every suffix of every read and of its reverse
// complement, sorted A<C<G<T with a prefix before its extensions
// (expand.cpp:19-48,285-380; expand_test.cpp:43-75; builder.cpp:95-117).
static void dedup_sorted(std::vector<std::string>& v) {
  // expand.cpp:19-48 skip_dups: drop equal entries and entries that are a prefix of the successor.
  std::vector<std::string> out;
  out.reserve(v.size());
  for (size_t i = 0; i < v.size(); ++i) {
    if (i + 1 < v.size()) {
      const std::string& a = v[i];
      const std::string& b = v[i + 1];
      if (a.size() <= b.size() && std::memcmp(a.data(), b.data(), a.size()) == 0) continue;
    }
    out.push_back(std::move(v[i]));
  }
  v.swap(out);
}

static std::vector<std::string> build_entries_simple(const std::vector<std::string>& reads) {
  std::vector<std::string> all;
  for (auto& r : reads) {
    std::string rc = revcomp_str(r);
    for (size_t j = 0; j < r.size(); ++j) {
      all.push_back(r.substr(j));
      all.push_back(rc.substr(j));
    }
  }
  std::sort(all.begin(), all.end());
  dedup_sorted(all);
  return all;
}

// Same set, computed with the exact-cover emission rule the GPU path uses: for an
// oriented read s, suffix j>=1 and all later ones are skipped as soon as s[j:] is a prefix of some
// oriented read (found through an index of first-K-mers).  Used as the CPU baseline ("port") and
// cross-checked against build_entries_simple in tests.
static std::vector<std::string> build_entries_cover(const std::vector<std::string>& reads, unsigned K,
                                                   uint64_t* n_records) {
  std::vector<std::string> o;
  o.reserve(reads.size() * 2);
  for (auto& r : reads) { o.push_back(r); o.push_back(revcomp_str(r)); }
  auto key_at = [&](const std::string& s, size_t j) {
    kmer_t x = 0;
    for (unsigned i = 0; i < K; ++i) x = (x << 2) | kmer_t(base_code(s[j + i]));
    return x;
  };
  std::unordered_map<kmer_t, std::vector<uint32_t>> idx;
  for (size_t i = 0; i < o.size(); ++i)
    if (o[i].size() >= K) idx[key_at(o[i], 0)].push_back(uint32_t(i));
  std::vector<std::string> rec;
  for (size_t i = 0; i < o.size(); ++i) {
    const std::string& s = o[i];
    rec.push_back(s);
    for (size_t j = 1; j < s.size(); ++j) {
      bool covered = false;
      size_t rem = s.size() - j;
      if (rem >= K) {
        auto it = idx.find(key_at(s, j));
        if (it != idx.end()) {
          for (uint32_t c : it->second) {
            const std::string& t = o[c];
            if (t.size() >= rem && std::memcmp(t.data(), s.data() + j, rem) == 0) { covered = true; break; }
          }
        }
      }
      if (covered) break;
      rec.push_back(s.substr(j));
    }
  }
  if (n_records) *n_records = rec.size();
  std::sort(rec.begin(), rec.end());
  dedup_sorted(rec);
  return rec;
}

// ------------------------------------------------------------------------------------------
// Seqset arrays (rows F1, F2) and the walk primitive (row G1).
struct Seqset {
  uint64_t n = 0;
  unsigned max_len = 0;
  std::vector<uint32_t> sizes, shared;
  std::vector<uint64_t> bits[4], subaccum[4], accum[4];
  uint64_t fixed[5] = {0, 0, 0, 0, 0};
  std::string err;

  // modules/io/bitcount.cpp:84-124 finalize (sizes: :66-78)
  uint64_t finalize_bitcount(int b) {
    size_t nw = (n + 63) / 64;
    bits[b].resize(nw, 0);
    subaccum[b].assign((n + 511) / 512, 0);
    accum[b].assign((n + 1 + 511) / 512, 0);
    if (n == 0) { accum[b].assign(1, 0); return 0; }
    uint64_t sub = 0, total = 0;
    for (size_t i = 0; i < nw; ++i) {
      if (i % 8 == 0) {
        accum[b][i / 8] = total;
        if (i != 0) subaccum[b][i / 8 - 1] = sub;
        sub = 0;
      }
      sub <<= 8;
      uint64_t st = __builtin_popcountll(bits[b][i]);
      sub |= st;
      total += st;
    }
    size_t left_over = nw % 8;
    while (left_over != 0) { sub <<= 8; left_over = (left_over + 1) % 8; }
    subaccum[b][(n + 511) / 512 - 1] = sub;
    if (n % 512 == 0) accum[b][n / 512] = total;
    return total;
  }
  // modules/bio_base/seqset.cpp:113-129
  bool finalize() {
    uint64_t off = 0;
    for (int b = 0; b < 4; ++b) { fixed[b] = off; off += finalize_bitcount(b); }
    fixed[4] = off;
    if (off != n) { err = "Invalid seqset in finalize"; return false; }
    return true;
  }
  // modules/io/bitcount.h:218-270
  uint64_t count(int b, uint64_t i) const {
    uint64_t x = (i / 64 < bits[b].size()) ? bits[b][i / 64] : 0;
    x &= (uint64_t(1) << (i & 63)) - 1;
    uint64_t bc = __builtin_popcountll(x);
    size_t group = i / 512, subgroup = (i / 64) & 7;
    uint64_t sa = (group < subaccum[b].size()) ? subaccum[b][group] : 0;
    sa >>= 56 - (8 * subgroup);
    sa &= 0xffffffffffffff00ull;
    sa = (sa >> 32) + (sa & 0xffffffffull);
    sa = (sa >> 16) + (sa & 0xffff);
    sa = (sa >> 8) + (sa & 0xff);
    return accum[b][group] + sa + bc;
  }
  // int_map_interface.cpp:128-197 (semantics only: first pos>=start with shared<v, else n)
  uint64_t next_forward_lt(uint64_t pos, uint32_t v) const {
    while (pos < n && shared[pos] >= v) ++pos;
    return pos;
  }
  uint64_t next_backward_lt(uint64_t pos, uint32_t v) const {
    while (pos > 0 && shared[pos] >= v) --pos;
    return pos;
  }
  struct Range { uint64_t begin, end; uint32_t size; bool valid() const { return begin < end; } };
  // seqset.cpp:319-338
  Range push_front(Range r, int b) const {
    uint64_t nb = fixed[b] + count(b, r.begin), ne = fixed[b] + count(b, r.end);
    if (nb < ne && sizes[nb] < r.size + 1) nb++;
    return Range{nb, ne, r.size + 1};
  }
  // seqset.cpp:350-429
  Range push_front_drop(Range r, int b, uint32_t min_ctx) const {
    uint64_t fx = fixed[b];
    uint64_t ob = r.begin, oe = r.end;
    uint32_t octx = r.size;
    uint64_t sb = count(b, ob), se = count(b, oe);
    if (octx < min_ctx) return Range{0, 0, 0};
    while (sb == se || (sb + 1 == se && sizes[fx + sb] < octx + 1)) {
      // DEVIATION from the reference: when no entry starts with base b and min_ctx == 0, its loop reaches the whole
      // range and never leaves it (seqset.cpp:368-421).  The extension does not exist: invalid range.
      if (ob == 0 && oe == n && sb == se) return Range{0, 0, 0};
      uint32_t drop = std::max(shared[ob], (oe == n ? 0u : shared[oe]));
      if (sb != se) drop = std::max(drop, sizes[fx + sb] - 1);
      if (drop < min_ctx) return Range{0, 0, 0};
      bool ub = false, ue = false;
      if (ob > 0 && shared[ob] >= drop) { ob = next_backward_lt(ob, drop); ub = true; }
      if (oe < n && shared[oe] >= drop) { oe = next_forward_lt(oe, drop); ue = true; }
      if (ub) sb = count(b, ob);
      if (ue) se = count(b, oe);
      octx = drop;
    }
    uint64_t nb = fx + sb, ne = fx + se;
    if (nb < ne && sizes[nb] < octx + 1) nb++;
    return Range{nb, ne, octx + 1};
  }
  // seqset.cpp:146-155
  Range find(const std::string& s) const {
    Range r{0, n, 0};
    for (size_t i = 0; i < s.size(); ++i) {
      if (!r.valid()) break;
      r = push_front(r, base_code(s[s.size() - 1 - i]));
    }
    return r;
  }
};

static unsigned lcp(const std::string& a, const std::string& b) {
  size_t m = std::min(a.size(), b.size()), i = 0;
  while (i < m && a[i] == b[i]) ++i;
  return unsigned(i);
}

// builder.cpp:8-164 via the closed form verified in SURVEY row F1: for every entry s set
// prev_{s[0]}[lower_bound(entries, s[1:])]; the target must have s[1:] as a prefix
// ("Missing expansion?" builder.cpp:95-96).
static Seqset* seqset_from_entries(const std::vector<std::string>& e) {
  auto* ss = new Seqset;
  ss->n = e.size();
  ss->sizes.resize(ss->n);
  ss->shared.resize(ss->n);
  for (int b = 0; b < 4; ++b) ss->bits[b].assign((ss->n + 63) / 64, 0);
  for (size_t i = 0; i < e.size(); ++i) {
    ss->sizes[i] = uint32_t(e[i].size());
    ss->max_len = std::max<unsigned>(ss->max_len, unsigned(e[i].size()));
    ss->shared[i] = i ? lcp(e[i - 1], e[i]) : 0;
  }
  for (size_t i = 0; i < e.size(); ++i) {
    std::string tail = e[i].substr(1);
    size_t t = std::lower_bound(e.begin(), e.end(), tail) - e.begin();
    if (t >= e.size() || e[t].compare(0, tail.size(), tail) != 0) {
      ss->err = "Missing expansion: " + e[i];
      return ss;
    }
    int b = base_code(e[i][0]);
    ss->bits[b][t / 64] |= uint64_t(1) << (t & 63);
  }
  ss->finalize();
  return ss;
}

// modules/io/packed_varbit_vector.cpp:174-188,80-147: values LSB-first, bits_for_value(max) bits each.
static unsigned bits_for_value(uint64_t v) { unsigned b = 0; while (v) { ++b; v >>= 1; } return b; }
static std::vector<uint64_t> pack_varbit(const std::vector<uint32_t>& vals, uint64_t max_value) {
  unsigned bpv = bits_for_value(max_value);
  std::vector<uint64_t> out((vals.size() * bpv + 63) / 64, 0);
  for (size_t i = 0; i < vals.size() && bpv; ++i) {
    size_t bit = i * bpv;
    uint64_t v = vals[i];
    out[bit / 64] |= v << (bit % 64);
    if (bit % 64 + bpv > 64) out[bit / 64 + 1] |= v >> (64 - bit % 64);
  }
  return out;
}


// ------------------------------------------------------------------------------------------
// Overlap-walk drivers (rows G2, G3, G7).
using Range = Seqset::Range;
struct RangeLess {  // seqset_range::operator< (bio_base/seqset.h:211-218)
  bool operator()(const Range& a, const Range& b) const {
    if (a.begin != b.begin) return a.begin < b.begin;
    if (a.end != b.end) return a.end > b.end;
    return a.size < b.size;
  }
};
using RangeSet = std::set<Range, RangeLess>;

// variants/assemble.cpp:1473-1498 — literal
static void dedup_prefixes(RangeSet& rs) {
  if (rs.empty()) return;
  bool again = true;
  while (again) {
    again = false;
    for (auto it = rs.begin(); it != rs.end(); ++it) {
      auto next = it;
      ++next;
      for (; next != rs.end(); ++next) {
        if (next->begin >= it->end) break;
        if (it->begin <= next->begin && it->end >= next->end) {
          it = rs.erase(it);
          again = true;
          break;
        }
      }
      if (it == rs.end()) break;
    }
  }
}

// graph_discover/push_to_pair.cpp:159-240 without the per-trace state (m_seen_ranges, readmap;
// k_check_reads=false :14).  Returns the chosen rc base (-1 = none); cands = best_rs U best_rs_drop
// before prefix-dedup (set order) with their drop flag; next = the returned frontier.
static int trace_one_base(const Seqset& ss, const RangeSet& rs, uint32_t min_ctx,
                          std::vector<std::pair<Range, bool>>* cands, RangeSet* next) {
  RangeSet nrs[4], nrs_drop[4];
  for (int b = 0; b < 4; ++b)
    for (const Range& r : rs) {
      Range n = ss.push_front_drop(r, b, min_ctx);
      if (!n.valid()) continue;
      bool drop = n.size != r.size + 1;
      (drop ? nrs_drop[b] : nrs[b]).insert(n);
    }
  unsigned best_len = 0;
  int best = -1;
  for (int b = 0; b < 4; ++b) {
    unsigned nl = 0;
    for (auto& r : nrs[b]) nl = std::max(nl, r.size);
    for (auto& r : nrs_drop[b]) nl = std::max(nl, r.size);
    if (nl > best_len) { best_len = nl; best = b; }
  }
  cands->clear();
  next->clear();
  if (!best_len) return -1;
  std::map<Range, bool, RangeLess> all;
  for (auto& r : nrs[best]) all[r] = false;
  for (auto& r : nrs_drop[best]) {
    auto it = all.find(r);
    if (it == all.end()) all[r] = true; else it->second = true;  // same range reached both ways: in both sets
  }
  for (auto& kv : all) cands->push_back(kv);
  *next = nrs[best];
  next->insert(nrs_drop[best].begin(), nrs_drop[best].end());
  dedup_prefixes(*next);
  return best;
}

// The extension loop of push_to_pair_discover::on_trace (graph_discover/push_to_pair.cpp:271-343) as far
// as it depends on the seqset alone: repeat trace_one_base from `rs` (k_check_reads=false => the
// min context is always min_overlap), stop when nothing extends (reason 0), when a dropped range of the
// chosen base was already seen in this trace (m_seen_ranges, :221-225; reason 1) or after max_steps
// (reason 2).  The mate / anchor checks between the steps (:289-342) only decide where the host cuts
// the path; they do not alter it.
static int push_to_pair_trace(const Seqset& ss, RangeSet rs, uint32_t min_overlap, size_t max_steps,
                              std::vector<uint8_t>* bases, std::vector<RangeSet>* fronts) {
  std::set<Range, RangeLess> seen;
  while (!rs.empty()) {
    if (bases->size() >= max_steps) return 2;
    std::vector<std::pair<Range, bool>> cands;
    RangeSet next;
    int best = trace_one_base(ss, rs, min_overlap, &cands, &next);
    if (best < 0) return 0;
    for (auto& c : cands)
      if (c.second && !seen.insert(c.first).second) return 1;
    bases->push_back(uint8_t(3 - best));  // the appended base is the complement of the pushed one (:236)
    fronts->push_back(next);
    rs = std::move(next);
  }
  return 0;
}

// graph_discover/branch.cpp:20-80: walk `seq` (codes 0..3) from frontier `cur`; at each offset the first
// non-reference base (order A,C,G,T over b, pushing its complement) with any valid
// push_front_drop(rc_b, min_overlap) is reported; then cur advances along the reference base.
static void branch_trace(const Seqset& ss, const uint8_t* seq, size_t n, RangeSet cur, uint32_t min_overlap,
                         std::vector<std::pair<uint32_t, uint8_t>>* events, RangeSet* final_cur) {
  for (size_t off = 0; off < n; ++off) {
    int ref_b = seq[off], rc_ref_b = 3 - ref_b;
    for (int b = 0; b < 4; ++b) {
      int rc_b = 3 - b;
      if (rc_b == rc_ref_b) continue;
      bool extend = false;
      for (const Range& r : cur)
        if (ss.push_front_drop(r, rc_b, min_overlap).valid()) extend = true;
      if (extend) { events->emplace_back(uint32_t(off), uint8_t(b)); break; }
    }
    RangeSet nc;
    // (an invalid result can only come from the deviation in push_front_drop — a base no entry starts with — where
    // the reference never returns; such a range leaves the frontier)
    for (const Range& r : cur) {
      const Range x = ss.push_front_drop(r, rc_ref_b, 0);
      if (x.valid()) nc.insert(x);
    }
    cur = std::move(nc);
    dedup_prefixes(cur);
  }
  *final_cur = cur;
}

// variants/ref_map.cpp:72-185 + flush_updates :44-70 — literal chunking (max(extent/100, 25600) bases,
// 256-base warm-up), both strands; NB the forward slice sets k_rev_flag and the rc slice k_fwd_flag.
static void ref_map_build(const Seqset& ss, const uint8_t* ref, const uint64_t* ext_start, const uint64_t* ext_size,
                          size_t n_ext, uint8_t* out) {
  const unsigned k_fwd_flag = 1 << 7, k_rev_flag = 1 << 6, k_count_mask = 63;
  const size_t k_min_chunk = 25600;
  std::memset(out, 0, ss.n);
  auto mark = [&](uint64_t id, bool is_rc) {
    uint8_t v = out[id];
    unsigned c = v & k_count_mask;
    if (c != k_count_mask) ++c;
    v = uint8_t((v & ~k_count_mask) | c);
    v |= is_rc ? k_fwd_flag : k_rev_flag;
    out[id] = v;
  };
  auto walk = [&](const std::vector<uint8_t>& slice, size_t prestart, bool is_rc) {
    Range r{0, ss.n, 0};
    for (size_t i = 0; i < slice.size(); ++i) {
      r = ss.push_front_drop(r, 3 - slice[i], 0);
      if (i < prestart) continue;
      if (r.begin + 1 != r.end) continue;
      if (r.size != ss.sizes[r.begin]) continue;
      mark(r.begin, is_rc);
    }
  };
  for (size_t e = 0; e < n_ext; ++e) {
    const uint8_t* seq = ref + ext_start[e];
    size_t len = ext_size[e];
    size_t chunk = std::max(len / 100, k_min_chunk);
    for (size_t cs = 0; cs < len; cs += chunk) {
      size_t ce = std::min(cs + chunk, len);
      size_t pre = cs >= 256 ? cs - 256 : 0;
      size_t pre_rc = std::min(ce + 256, len);
      walk(std::vector<uint8_t>(seq + pre, seq + ce), cs - pre, false);
      std::vector<uint8_t> rc;
      for (size_t i = pre_rc; i > cs; --i) rc.push_back(uint8_t(3 - seq[i - 1]));
      walk(rc, pre_rc - ce, true);
    }
  }
}


// ------------------------------------------------------------------------------------------
// Readmap: bio_mapred/make_readmap.cpp:120-366.
struct Readmap {
  uint64_t n_reads = 0;        // read ids (4 per pair, 2 per single)
  unsigned max_read_len = 0;
  std::vector<uint64_t> entry_id;      // per read id (sorted), not stored directly
  std::vector<uint32_t> read_len;
  std::vector<uint64_t> mate_loop_ptr;
  std::vector<uint8_t> is_forward;
  Seqset s2m, d2m;  // only bits[0]/subaccum[0]/accum[0] used as plain bitcounts
};

struct LoopRow {  // make_readmap.h:56-90
  uint64_t entry_id;
  int type;  // LOOP_START 0, RC 1, MATE 2, MATE_RC 3
  uint32_t read_length, mate_read_length;
  uint64_t loop_entry_id;
};
static const uint64_t kNoLoop = (1ull << 37) - 1;
static bool row_less(const LoopRow& a, const LoopRow& b) {  // make_readmap.h:187-213
  if (a.entry_id != b.entry_id) return a.entry_id < b.entry_id;
  if (a.type != b.type) return a.type < b.type;
  if (a.read_length != b.read_length) return a.read_length < b.read_length;
  if (a.mate_read_length != b.mate_read_length) return a.mate_read_length < b.mate_read_length;
  return a.loop_entry_id < b.loop_entry_id;
}

// records: record r holds reads [rec_first[r], rec_first[r+1]) (1 or 2 corrected sequences).
template <class F>
static void rm_parallel(uint64_t n, unsigned T, F f) {  // f(thread, lo, hi) on T threads (T <= 1: inline)
  if (T <= 1 || n < 4096) { f(0u, uint64_t(0), n); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t) th.emplace_back([=, &f] { f(t, n * t / T, n * (t + 1) / T); });
  for (auto& x : th) x.join();
}

// threads > 1 (the CPU baseline of bench.py): the entry look-ups of the records, the row sort (stable sorts of equal shares,
// merged in order: the result of ONE stable sort) and the first pass run on `threads` threads; the claim pass stays
// sequential as in the reference (make_readmap.cpp:330-392).  Same readmap as threads = 1 (tests/test_oracle_mt.py).
static Readmap* readmap_build(const Seqset& ss, const std::vector<std::string>& entries,
                              const std::vector<std::string>& reads, const std::vector<uint64_t>& rec_first,
                              bool is_paired, bool canonicalize, std::string* err, unsigned threads = 1) {
  std::mutex err_mu;
  auto find_existing = [&](const std::string& s) -> uint64_t {  // seqset.cpp:173-188 == first entry with prefix s
    size_t t = std::lower_bound(entries.begin(), entries.end(), s) - entries.begin();
    if (t >= entries.size() || entries[t].compare(0, s.size(), s) != 0) {
      std::lock_guard<std::mutex> l(err_mu);
      *err = "read not found in seqset: " + s;
      return 0;
    }
    return t;
  };
  std::vector<LoopRow> rows;
  unsigned max_len = 0;
  const size_t n_rec = rec_first.empty() ? 0 : rec_first.size() - 1;
  if (threads > 1) {
    // a record's rows have a place of their own: 4 for a pair, 2 for a single read
    std::vector<uint64_t> slot(n_rec + 1, 0);
    for (size_t r = 0; r < n_rec; ++r) {
      const size_t cnt = rec_first[r + 1] - rec_first[r];
      if (cnt != 1 && !(cnt == 2 && is_paired)) { *err = "Unexpected read pairing"; return nullptr; }
      slot[r + 1] = slot[r] + (cnt == 2 ? 4 : 2);
    }
    rows.resize(slot[n_rec]);
    std::vector<unsigned> tmax(threads, 0);
    rm_parallel(n_rec, threads, [&](unsigned t, uint64_t lo, uint64_t hi) {
      for (uint64_t r = lo; r < hi; ++r) {
        const size_t cnt = rec_first[r + 1] - rec_first[r];
        const std::string& seq = reads[rec_first[r]];
        unsigned read_len = unsigned(seq.size());
        uint64_t e = find_existing(seq), erc = find_existing(revcomp_str(seq));
        LoopRow* o = rows.data() + slot[r];
        if (cnt == 2) {
          const std::string& mate = reads[rec_first[r] + 1];
          unsigned mate_len = unsigned(mate.size());
          tmax[t] = std::max(tmax[t], std::max(read_len, mate_len));
          uint64_t me = find_existing(mate), merc = find_existing(revcomp_str(mate));
          if (canonicalize && seq > mate) { std::swap(me, e); std::swap(mate_len, read_len); std::swap(merc, erc); }  // :171-176
          o[0] = {e, 0, read_len, 0, erc};
          o[1] = {erc, 1, read_len, mate_len, me};
          o[2] = {me, 2, mate_len, 0, merc};
          o[3] = {merc, 3, mate_len, 0, kNoLoop};
        } else {
          tmax[t] = std::max(tmax[t], read_len);
          o[0] = {e, 0, read_len, 0, erc};
          o[1] = {erc, 1, read_len, 0, kNoLoop};
        }
      }
    });
    if (!err->empty()) return nullptr;
    for (unsigned x : tmax) max_len = std::max(max_len, x);
    // stable sort = stable sorts of consecutive shares merged in order (std::inplace_merge keeps the left share's rows first)
    const unsigned T = threads;
    const size_t nr = rows.size();
    std::vector<size_t> cut(T + 1);
    for (unsigned t = 0; t <= T; ++t) cut[t] = nr * t / T;
    rm_parallel(T, T, [&](unsigned, uint64_t lo, uint64_t hi) {
      for (uint64_t t = lo; t < hi; ++t) std::stable_sort(rows.begin() + cut[t], rows.begin() + cut[t + 1], row_less);
    });
    for (unsigned width = 1; width < T; width *= 2) {
      std::vector<std::thread> th;
      for (unsigned a = 0; a + width < T; a += 2 * width) {
        const size_t lo = cut[a], mid = cut[a + width], hi = cut[std::min(T, a + 2 * width)];
        th.emplace_back([&rows, lo, mid, hi] { std::inplace_merge(rows.begin() + lo, rows.begin() + mid, rows.begin() + hi, row_less); });
      }
      for (auto& x : th) x.join();
    }
  }
  for (size_t r = 0; threads <= 1 && r + 1 < rec_first.size(); ++r) {
    size_t cnt = rec_first[r + 1] - rec_first[r];
    std::string seq = reads[rec_first[r]];
    std::string mate = (cnt == 2 && is_paired) ? reads[rec_first[r] + 1] : std::string();
    if (cnt != 1 && !(cnt == 2 && is_paired)) { *err = "Unexpected read pairing"; return nullptr; }
    unsigned read_len = unsigned(seq.size());
    uint64_t e = find_existing(seq), erc = find_existing(revcomp_str(seq));
    unsigned mate_len = unsigned(mate.size());
    max_len = std::max(max_len, std::max(read_len, mate_len));
    if (mate_len) {
      uint64_t me = find_existing(mate), merc = find_existing(revcomp_str(mate));
      if (canonicalize && seq > mate) { std::swap(me, e); std::swap(mate_len, read_len); std::swap(merc, erc); }  // :171-176
      rows.push_back({e, 0, read_len, 0, erc});
      rows.push_back({erc, 1, read_len, mate_len, me});
      rows.push_back({me, 2, mate_len, 0, merc});
      rows.push_back({merc, 3, mate_len, 0, kNoLoop});
    } else {
      rows.push_back({e, 0, read_len, 0, erc});
      rows.push_back({erc, 1, read_len, 0, kNoLoop});
    }
    if (!err->empty()) return nullptr;
  }
  if (threads <= 1) std::stable_sort(rows.begin(), rows.end(), row_less);
  auto* rm = new Readmap;
  const size_t n = rows.size();
  rm->n_reads = n;
  for (const auto& e : entries) max_len = std::max<unsigned>(max_len, unsigned(e.size()));  // seqset::max_read_len() (biograph_create.cpp:825)
  rm->max_read_len = max_len;
  rm->read_len.resize(n);
  rm->entry_id.resize(n);
  rm->mate_loop_ptr.assign(n, 0);
  rm->is_forward.assign(n, 0);
  // sparse_multi (io/sparse_multi.h): bit per seqset entry with reads; bit per read id starting a group
  rm->s2m.n = ss.n;
  rm->s2m.bits[0].assign((ss.n + 63) / 64, 0);
  rm->d2m.n = n;
  rm->d2m.bits[0].assign((n + 63) / 64, 0);
  for (size_t i = 0; i < n; ++i) {
    rm->read_len[i] = rows[i].read_length;
    rm->entry_id[i] = rows[i].entry_id;
    rm->s2m.bits[0][rows[i].entry_id / 64] |= 1ull << (rows[i].entry_id & 63);
    if (i == 0 || rows[i].entry_id != rows[i - 1].entry_id) rm->d2m.bits[0][i / 64] |= 1ull << (i & 63);
  }
  rm->s2m.finalize_bitcount(0);
  rm->d2m.finalize_bitcount(0);
  auto find_first_of = [&](int type, uint64_t entry_id, unsigned len) -> uint64_t {  // :283-289
    LoopRow probe{entry_id, type, len, 0, 0};
    return std::lower_bound(rows.begin(), rows.end(), probe, row_less) - rows.begin();
  };
  rm_parallel(n, threads, [&](unsigned, uint64_t lo, uint64_t hi) {  // first pass :293-326 (every row for itself)
    for (uint64_t i = lo; i < hi; ++i) {
      const LoopRow& r = rows[i];
      switch (r.type) {
        case 0: rm->is_forward[i] = 1; rm->mate_loop_ptr[i] = find_first_of(1, r.loop_entry_id, r.read_length); break;
        case 1: if (r.loop_entry_id != kNoLoop) rm->mate_loop_ptr[i] = find_first_of(2, r.loop_entry_id, r.mate_read_length); break;
        case 2: rm->is_forward[i] = 1; rm->mate_loop_ptr[i] = find_first_of(3, r.loop_entry_id, r.read_length); break;
        default: break;
      }
    }
  });
  std::vector<uint8_t> claimed(n, 0);  // second pass, sequential on purpose :330-392
  auto claim_next = [&](uint64_t t, int type, uint64_t entry_id, unsigned len) -> uint64_t {
    while (t < n && claimed[t]) ++t;
    if (t >= n || rows[t].entry_id != entry_id || rows[t].read_length != len || rows[t].type != type) {
      *err = "mate loop claim failed";
      return 0;
    }
    claimed[t] = 1;
    return t;
  };
  for (size_t i = 0; i < n && err->empty(); ++i) {
    const LoopRow& r = rows[i];
    if (r.type != 0) continue;
    uint64_t rc = claim_next(rm->mate_loop_ptr[i], 1, r.loop_entry_id, r.read_length);
    rm->mate_loop_ptr[i] = rc;
    const LoopRow& rr = rows[rc];
    if (rr.loop_entry_id == kNoLoop) { rm->mate_loop_ptr[rc] = i; continue; }
    uint64_t mi = claim_next(rm->mate_loop_ptr[rc], 2, rr.loop_entry_id, rr.mate_read_length);
    rm->mate_loop_ptr[rc] = mi;
    const LoopRow& mr = rows[mi];
    uint64_t mrc = claim_next(rm->mate_loop_ptr[mi], 3, mr.loop_entry_id, rr.mate_read_length);
    rm->mate_loop_ptr[mi] = mrc;
    rm->mate_loop_ptr[mrc] = i;
  }
  if (!err->empty()) { delete rm; return nullptr; }
  return rm;
}

}  // namespace orc

// ============================================================================================
// C interface (ctypes).
using namespace orc;

static std::vector<std::string> split_reads(const char* seq, const uint64_t* offs, uint64_t n) {
  std::vector<std::string> v(n);
  for (uint64_t i = 0; i < n; ++i) v[i].assign(seq + offs[i], seq + offs[i + 1]);
  return v;
}

// ------------------------------------------------------------------------------------------
// Readmap queries and seqset_path::propagate_from_end (row G4).
// bio_base/readmap.cpp:207-218 (has_mate), :248-290 (get_rev_comp / get_mate_rc), :434-447 +
// :538-649 (get_prefix_reads: the read_iterator state machine, literal), readmap.h:249-251
// (read::get_seqset_entry), bio_base/seqset.cpp:630-661 (pop_back / truncate).
static bool rm_has_mate(const Readmap& rm, uint64_t id) { return rm.mate_loop_ptr[rm.mate_loop_ptr[id]] != id; }
static uint64_t rm_follow(const Readmap& rm, uint64_t id, int n) {
  for (int c = 0; c < n; ++c) id = rm.mate_loop_ptr[id];
  return id;
}
static uint64_t rm_get_rev_comp(const Readmap& rm, uint64_t id) { return rm_follow(rm, id, rm.is_forward[id] ? 1 : 3); }
static uint64_t rm_get_mate_rc(const Readmap& rm, uint64_t id) { return rm_follow(rm, id, rm.is_forward[id] ? 3 : 1); }
static unsigned rm_min_read_len(const Readmap& rm) {
  unsigned m = ~0u;
  for (uint32_t l : rm.read_len) m = std::min(m, l);
  return rm.read_len.empty() ? 0 : m;
}
static Range ss_truncate(const Seqset& ss, Range r, uint32_t count) {  // seqset.cpp:630-661
  if (r.size <= count) return r;
  uint64_t nb = r.begin, ne = r.end;
  while (nb && ss.shared[nb] >= count) nb--;
  while (ne < ss.n && ss.shared[ne] >= count) ne++;
  return Range{nb, ne, count};
}
static Range rm_read_entry(const Seqset& ss, const Readmap& rm, uint64_t id) {  // readmap.h:249-251
  uint64_t e = rm.entry_id[id];
  return ss_truncate(ss, Range{e, e + 1, ss.sizes[e]}, rm.read_len[id]);
}
static bool rm_d2m_bit(const Readmap& rm, uint64_t id) { return (rm.d2m.bits[0][id / 64] >> (id & 63)) & 1; }
static std::vector<uint64_t> rm_prefix_reads(const Seqset& ss, const Readmap& rm, Range r, int read_len_limit) {
  std::vector<uint64_t> out;
  read_len_limit = std::max<int>(read_len_limit, int(rm_min_read_len(rm)));
  if (int(r.size) < read_len_limit || rm.n_reads == 0) return out;
  // sparse_multi::lookup_lower_bound (io/sparse_multi.cpp:32-38): first read id of the first group at or
  // after entry r.begin
  const uint64_t n = rm.n_reads;
  uint64_t id0 = std::lower_bound(rm.entry_id.begin(), rm.entry_id.end(), r.begin) - rm.entry_id.begin();
  enum { FORWARD, BACKWARD, DONE } phase = FORWARD;
  uint64_t read_id = id0, seqset_id = r.begin;
  const int min_size = read_len_limit;
  int max_len = int(r.size);
  auto advance = [&](auto& done_direction) {
    if (phase == FORWARD) { ++read_id; return; }
    if (read_id == 0) { done_direction(); return; }
    --read_id;
  };
  std::function<void()> done_direction = [&]() {
    if (phase == FORWARD) {
      phase = BACKWARD;
      read_id = id0;
      seqset_id = r.begin;
      max_len = int(r.size);
      advance(done_direction);
    } else {
      phase = DONE;
    }
  };
  auto skip_once = [&]() -> bool {
    bool forward = phase == FORWARD;
    bool first_in_group;
    if (forward) {
      if (read_id == n) { done_direction(); return false; }
      first_in_group = read_id == 0 || rm_d2m_bit(rm, read_id);
    } else {
      uint64_t prev = read_id + 1;
      first_in_group = prev == n || rm_d2m_bit(rm, prev);
    }
    if (first_in_group) {
      uint64_t new_sid = rm.entry_id[read_id];  // sparse_multi::reverse_lookup
      while (seqset_id != new_sid) {
        int shared;
        if (forward) { ++seqset_id; shared = ss.shared[seqset_id]; }
        else { shared = ss.shared[seqset_id]; --seqset_id; }
        if (shared < min_size) { done_direction(); return false; }
        if (shared < max_len) max_len = shared;
      }
    }
    if (int(rm.read_len[read_id]) > max_len) { advance(done_direction); return false; }
    return true;
  };
  for (;;) {
    while (phase != DONE && !skip_once()) {}
    if (phase == DONE) break;
    out.push_back(read_id);
    advance(done_direction);
  }
  return out;
}

struct PathOut {
  std::map<uint32_t, RangeSet> entries;  // offset -> ranges (0 and seq size always present)
  RangeSet mates;
};
// variants/assemble.cpp:1345-1418.  marks = offsets already present in m_entries (refreshed on the way).
static void propagate_from_end(const Seqset& ss, const Readmap* rm, RangeSet new_ends, const uint8_t* seq, size_t n,
                               unsigned max_pairs_per_read, const std::set<uint32_t>& marks, PathOut* out) {
  RangeSet cur = new_ends;
  dedup_prefixes(cur);
  out->entries[uint32_t(n)] = cur;
  out->entries[0];
  size_t offset = n;
  while (offset) {
    --offset;
    RangeSet nc;
    for (const Range& r : cur) {
      const Range x = ss.push_front_drop(r, seq[offset], 0);
      if (x.valid()) nc.insert(x);  // see branch_trace
    }
    dedup_prefixes(nc);
    cur = std::move(nc);
    if (offset == 0 || marks.count(uint32_t(offset))) out->entries[uint32_t(offset)] = cur;
    if (rm) {
      for (const Range& r : cur) {
        unsigned nreads = 0;
        std::vector<Range> found;
        for (uint64_t id : rm_prefix_reads(ss, *rm, r, 0)) {
          if (!rm_has_mate(*rm, id)) continue;
          ++nreads;
          if (nreads > max_pairs_per_read) break;
          found.push_back(rm_read_entry(ss, *rm, rm_get_mate_rc(*rm, id)));
        }
        if (nreads > max_pairs_per_read) continue;
        out->mates.insert(found.begin(), found.end());
      }
    }
  }
}


// ------------------------------------------------------------------------------------------
// Merge: make_mergemap::count_range (bio_base/make_mergemap.cpp:150-220) over the whole
// range — a k-way merge of the sorted entry lists in which an entry that equals or is a prefix of the
// next one in merged order joins its chain; a chain becomes ONE merged entry (its longest member) and
// sets the mergemap bit of every input that contributed.  make_readmap::create_from_fast_migrate
// (bio_mapred/make_readmap.cpp:455-519): read ids keep their order, lengths, loop pointers and
// orientation; the entry of a read id becomes mergemap.find_count(old entry) = the position of the
// old-entry-th set bit of that input's mergemap.
struct MergeOut {
  std::vector<std::string> entries;
  std::vector<std::vector<uint64_t>> maps;  // per input: merged index of every input entry (ascending)
};
static void merge_entries(const std::vector<const std::vector<std::string>*>& in, MergeOut* out) {
  struct QE {
    const std::string* s;
    unsigned flat;
    size_t id;
    bool operator<(const QE& r) const { return *s > *r.s; }  // make_mergemap.h:65-69
  };
  auto is_equal_or_prefix = [](const std::string& p, const std::string& l) {
    return p.size() <= l.size() && l.compare(0, p.size(), p) == 0;
  };
  std::priority_queue<QE> q;
  auto add = [&](unsigned f, size_t id) {
    if (id < in[f]->size()) q.push(QE{&(*in[f])[id], f, id});
  };
  out->maps.assign(in.size(), {});
  for (unsigned f = 0; f < in.size(); ++f) add(f, 0);
  while (!q.empty()) {
    QE e = q.top();
    q.pop();
    const std::string* slice = e.s;
    const uint64_t merged = out->entries.size();
    out->maps[e.flat].push_back(merged);
    add(e.flat, e.id + 1);
    while (!q.empty() && is_equal_or_prefix(*slice, *q.top().s)) {
      e = q.top();
      q.pop();
      slice = e.s;
      out->maps[e.flat].push_back(merged);
      add(e.flat, e.id + 1);
    }
    out->entries.push_back(*slice);
  }
}


extern "C" {

// ---- k-mers
void* orc_kmer_count(const char* seq, const uint64_t* offs, uint64_t n_reads, int k, uint32_t min_count,
                     int keep_all) {
  std::unordered_map<kmer_t, KmerElem> table;
  if (count_kmers(seq, offs, n_reads, unsigned(k), table) != 0) return nullptr;
  return make_kmer_set(table, unsigned(k), min_count, keep_all != 0);
}
uint64_t orc_kmerset_size(void* h) { return static_cast<KmerSet*>(h)->kmers.size(); }
void orc_kmerset_get(void* h, uint64_t* kmers, uint8_t* flags, uint32_t* fwd, uint32_t* rev) {
  auto* ks = static_cast<KmerSet*>(h);
  size_t n = ks->kmers.size();
  if (kmers) std::memcpy(kmers, ks->kmers.data(), n * 8);
  if (flags) std::memcpy(flags, ks->flags.data(), n);
  if (fwd) std::memcpy(fwd, ks->fwd.data(), n * 4);
  if (rev) std::memcpy(rev, ks->rev.data(), n * 4);
}
void* orc_kmerset_from_arrays(const uint64_t* kmers, const uint8_t* flags, uint64_t n, int k) {
  auto* ks = new KmerSet;
  ks->k = unsigned(k);
  ks->kmers.assign(kmers, kmers + n);
  ks->flags.assign(flags, flags + n);
  ks->fwd.assign(n, 0);
  ks->rev.assign(n, 0);
  ks->finish();
  return ks;
}
void orc_kmerset_free(void* h) { delete static_cast<KmerSet*>(h); }

// ---- correction.  out_seq must hold offs[n_reads] bytes; out_offs n_reads+1; stats 8 x u64.
int orc_correct_reads(void* ksh, const char* seq, const uint64_t* offs, uint64_t n_reads,
                      uint32_t max_corrections, uint32_t min_good_run, float trim_after_portion,
                      char* out_seq, uint64_t* out_offs, uint8_t* out_keep, uint64_t* stats) {
  auto* ks = static_cast<KmerSet*>(ksh);
  FrcParams p;
  p.max_corrections = max_corrections;
  p.min_good_run = min_good_run;
  p.kmer_size = ks->k;
  p.ks = ks;
  CorrectStats st;
  uint64_t pos = 0;
  for (uint64_t r = 0; r < n_reads; ++r) {
    out_offs[r] = pos;
    std::string c;
    bool ok = correct_read(seq + offs[r], offs[r + 1] - offs[r], p, trim_after_portion, &c, &st);
    out_keep[r] = ok ? 1 : 0;
    if (ok) { std::memcpy(out_seq + pos, c.data(), c.size()); pos += c.size(); }
  }
  out_offs[n_reads] = pos;
  if (stats) {
    stats[0] = st.corrected_reads; stats[1] = st.corrected_bases; stats[2] = st.failed;
    stats[3] = st.modified; stats[4] = st.modified_bases; stats[5] = st.truncated;
    stats[6] = st.dropped_bases; stats[7] = st.ref_records;
  }
  return 0;
}

// ---- the same stages on a std::thread pool: the CPU baseline SURVEY.md section 8(d) asks for (the reference's own
// parallel_for over reads / partitions, biograph_create.cpp:553-693).  Same algorithms, same results as the
// single-threaded functions above (tests/test_oracle_mt.py compares them); only the loop structure differs:
// k-mers are routed to hash partitions (kmer_counter.h:167-178) and each partition is counted by one thread;
// reads are corrected in ranges; suffix records are bucketed by their leading bases and the buckets sorted
// in parallel (part_repo.cpp:32-45).
}  // extern "C"

#include <atomic>
#include <thread>

namespace orc {

template <class F>
static void parallel_ranges(uint64_t n, unsigned nthreads, F f) {  // f(thread, lo, hi)
  if (nthreads < 1) nthreads = 1;
  std::vector<std::thread> pool;
  const uint64_t per = (n + nthreads - 1) / nthreads;
  for (unsigned t = 0; t < nthreads; ++t) {
    const uint64_t lo = std::min<uint64_t>(n, uint64_t(t) * per), hi = std::min<uint64_t>(n, lo + per);
    pool.emplace_back([=] { f(t, lo, hi); });
  }
  for (auto& th : pool) th.join();
}

struct KmerInst { kmer_t canon; uint8_t bits; };  // bits: 1 fwd_starts, 2 rev_starts, 4 flipped

static KmerSet* kmer_count_mt(const char* seq, const uint64_t* offs, uint64_t n_reads, unsigned k, uint32_t min_count,
                              bool keep_all, unsigned T) {
  const unsigned NP = 256;  // kmer_counter.h:167-178
  std::vector<std::vector<std::vector<KmerInst>>> routed(T, std::vector<std::vector<KmerInst>>(NP));
  std::atomic<int> bad{0};
  parallel_ranges(n_reads, T, [&](unsigned t, uint64_t lo, uint64_t hi) {
    auto& mine = routed[t];
    for (uint64_t r = lo; r < hi; ++r) {
      const char* it = seq + offs[r];
      const char* end = seq + offs[r + 1];
      bool is_first = true;
      unsigned left = k;
      kmer_t kmer = 0;
      while (it != end) {
        if (*it == 'N') { ++it; left = k; is_first = false; continue; }
        int b = base_code(*it);
        if (b < 0) { bad = 1; return; }
        kmer = kmer_shift_left(kmer, k, b);
        ++it;
        if (left) --left;
        if (!left) {
          bool is_last = (it == end), flipped;
          kmer_t canon = canonicalize(kmer, k, flipped);
          bool sf = is_first, sr = is_last;
          if (flipped) std::swap(sf, sr);
          mine[(canon * 0x9E3779B97F4A7C15ull) >> 56].push_back(KmerInst{canon, uint8_t((sf ? 1 : 0) | (sr ? 2 : 0) | (flipped ? 4 : 0))});
          is_first = false;
        }
      }
    }
  });
  if (bad) return nullptr;
  std::vector<std::vector<std::pair<kmer_t, KmerElem>>> kept(NP);
  std::atomic<unsigned> next{0};
  parallel_ranges(T, T, [&](unsigned, uint64_t, uint64_t) {
    for (;;) {
      const unsigned p = next++;
      if (p >= NP) break;
      // the partition's instances sorted by k-mer, then one pass over the runs (a hash map per partition spent most of
      // its time on allocation and cache misses: this baseline is a port, but it should not misstate what a CPU does)
      size_t total = 0;
      for (unsigned t = 0; t < T; ++t) total += routed[t][p].size();
      std::vector<KmerInst> all;
      all.reserve(total);
      for (unsigned t = 0; t < T; ++t) {
        all.insert(all.end(), routed[t][p].begin(), routed[t][p].end());
        std::vector<KmerInst>().swap(routed[t][p]);
      }
      std::sort(all.begin(), all.end(), [](const KmerInst& a, const KmerInst& b) { return a.canon < b.canon; });
      for (size_t i = 0; i < all.size();) {
        KmerElem e{};
        size_t j = i;
        for (; j < all.size() && all[j].canon == all[i].canon; ++j) {
          e.flags |= all[j].bits & 3;
          if (all[j].bits & 4) e.rev++; else e.fwd++;
        }
        if (keep_all || uint64_t(e.fwd) + e.rev >= min_count) kept[p].emplace_back(all[i].canon, e);
        i = j;
      }
    }
  });
  std::vector<std::pair<kmer_t, KmerElem>> v;
  for (auto& part : kept) v.insert(v.end(), part.begin(), part.end());
  std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first < b.first; });
  auto* ks = new KmerSet;
  ks->k = k;
  for (auto& kv : v) {
    ks->kmers.push_back(kv.first);
    ks->flags.push_back(kv.second.flags);
    ks->fwd.push_back(kv.second.fwd);
    ks->rev.push_back(kv.second.rev);
  }
  ks->finish();
  return ks;
}

static std::vector<std::string> build_entries_cover_mt(const std::vector<std::string>& reads, unsigned K, uint64_t* n_records,
                                                      unsigned T) {
  std::vector<std::string> o(reads.size() * 2);
  parallel_ranges(reads.size(), T, [&](unsigned, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) { o[2 * i] = reads[i]; o[2 * i + 1] = revcomp_str(reads[i]); }
  });
  auto key_at = [&](const std::string& s, size_t j) {
    kmer_t x = 0;
    for (unsigned i = 0; i < K; ++i) x = (x << 2) | kmer_t(base_code(s[j + i]));
    return x;
  };
  const unsigned NP = 256;
  std::vector<std::unordered_map<kmer_t, std::vector<uint32_t>>> idx(NP);
  {
    // first K-mers routed to index partitions by the threads that own the reads, each partition built by one thread
    std::vector<std::vector<std::vector<std::pair<kmer_t, uint32_t>>>> routed(T, std::vector<std::vector<std::pair<kmer_t, uint32_t>>>(NP));
    parallel_ranges(o.size(), T, [&](unsigned t, uint64_t lo, uint64_t hi) {
      for (uint64_t i = lo; i < hi; ++i)
        if (o[i].size() >= K) {
          const kmer_t key = key_at(o[i], 0);
          routed[t][(key * 0x9E3779B97F4A7C15ull) >> 56].emplace_back(key, uint32_t(i));
        }
    });
    std::atomic<unsigned> next{0};
    parallel_ranges(T, T, [&](unsigned, uint64_t, uint64_t) {
      for (;;) {
        const unsigned p = next++;
        if (p >= NP) break;
        for (unsigned t = 0; t < T; ++t)  // thread order = read order: candidates stay in ascending id order
          for (auto& kv : routed[t][p]) idx[p][kv.first].push_back(kv.second);
      }
    });
  }
  // records bucketed by their leading 4 bases (a shorter record goes by its zero-padded prefix)
  const unsigned NB = 256;
  std::vector<std::vector<std::vector<std::string>>> rec(T, std::vector<std::vector<std::string>>(NB));
  auto bucket_of = [](const char* p, size_t n) {
    unsigned b = 0;
    for (unsigned i = 0; i < 4; ++i) b = (b << 2) | (i < n ? unsigned(base_code(p[i])) : 0u);
    return b;
  };
  parallel_ranges(o.size(), T, [&](unsigned t, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      const std::string& s = o[i];
      if (s.empty()) continue;
      rec[t][bucket_of(s.data(), s.size())].push_back(s);
      for (size_t j = 1; j < s.size(); ++j) {
        bool covered = false;
        const size_t rem = s.size() - j;
        if (rem >= K) {
          const kmer_t key = key_at(s, j);
          auto& part = idx[(key * 0x9E3779B97F4A7C15ull) >> 56];
          auto it = part.find(key);
          if (it != part.end())
            for (uint32_t c : it->second) {
              const std::string& tt = o[c];
              if (tt.size() >= rem && std::memcmp(tt.data(), s.data() + j, rem) == 0) { covered = true; break; }
            }
        }
        if (covered) break;
        rec[t][bucket_of(s.data() + j, rem)].push_back(s.substr(j));
      }
    }
  });
  std::vector<std::vector<std::string>> sorted(NB);
  std::atomic<uint64_t> total{0};
  {
    std::atomic<unsigned> next{0};
    parallel_ranges(T, T, [&](unsigned, uint64_t, uint64_t) {
      for (;;) {
        const unsigned b = next++;
        if (b >= NB) break;
        auto& v = sorted[b];
        for (unsigned t = 0; t < T; ++t) {
          for (auto& x : rec[t][b]) v.push_back(std::move(x));
          std::vector<std::string>().swap(rec[t][b]);
        }
        total += v.size();
        std::sort(v.begin(), v.end());
      }
    });
  }
  if (n_records) *n_records = total;
  // buckets are lexicographic ranges: concatenation is sorted (a record shorter than 4 bases sorts before the longer
  // ones of its padded bucket only if it is their prefix or smaller, which the bucket's own sort has settled).
  // skip_dups (expand.cpp:19-48) drops a record that is a prefix of (or equal to) its successor in the sorted order: every
  // bucket for itself, the successor of a bucket's last record being the first record of the next bucket that has one
  // (part_repo.cpp:294-303: min_next_entry) -- no serial pass over all records (it was half of make_seqset's time on 256 threads).
  std::vector<std::string> next_first(NB);  // first record of the next non-empty bucket, "" when there is none (a record is never empty)
  {
    std::string carry;
    bool have = false;
    for (unsigned b = NB; b-- > 0;) {
      if (have) next_first[b] = carry;
      if (!sorted[b].empty()) { carry = sorted[b].front(); have = true; }
    }
  }
  bool boundaries_sorted = true;
  for (unsigned b = 0; b < NB; ++b)
    if (!sorted[b].empty() && !next_first[b].empty() && !(sorted[b].back() <= next_first[b])) boundaries_sorted = false;
  if (!boundaries_sorted) {  // (cannot happen; the serial path of round 3 kept as the answer to it)
    std::vector<std::string> all;
    all.reserve(total);
    for (auto& v : sorted) { for (auto& x : v) all.push_back(std::move(x)); std::vector<std::string>().swap(v); }
    std::sort(all.begin(), all.end());
    dedup_sorted(all);
    return all;
  }
  std::vector<uint64_t> kept(NB + 1, 0);
  {
    std::atomic<unsigned> next{0};
    parallel_ranges(T, T, [&](unsigned, uint64_t, uint64_t) {
      for (;;) {
        const unsigned b = next++;
        if (b >= NB) break;
        auto& v = sorted[b];
        size_t w = 0;
        for (size_t i = 0; i < v.size(); ++i) {
          const std::string& a = v[i];
          const std::string* succ = i + 1 < v.size() ? &v[i + 1] : (next_first[b].empty() ? nullptr : &next_first[b]);
          if (succ && a.size() <= succ->size() && std::memcmp(a.data(), succ->data(), a.size()) == 0) continue;
          if (w != i) v[w] = std::move(v[i]);
          ++w;
        }
        v.resize(w);
        kept[b + 1] = w;
      }
    });
  }
  for (unsigned b = 0; b < NB; ++b) kept[b + 1] += kept[b];
  std::vector<std::string> all(kept[NB]);
  {
    std::atomic<unsigned> next{0};
    parallel_ranges(T, T, [&](unsigned, uint64_t, uint64_t) {
      for (;;) {
        const unsigned b = next++;
        if (b >= NB) break;
        auto& v = sorted[b];
        for (size_t i = 0; i < v.size(); ++i) all[kept[b] + i] = std::move(v[i]);
        std::vector<std::string>().swap(v);
      }
    });
  }
  return all;
}

static Seqset* seqset_from_entries_mt(const std::vector<std::string>& e, unsigned T) {
  auto* ss = new Seqset;
  ss->n = e.size();
  ss->sizes.resize(ss->n);
  ss->shared.resize(ss->n);
  for (int b = 0; b < 4; ++b) ss->bits[b].assign((ss->n + 63) / 64, 0);
  std::vector<unsigned> maxes(T, 0);
  std::atomic<int> missing{0};
  parallel_ranges(e.size(), T, [&](unsigned t, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      ss->sizes[i] = uint32_t(e[i].size());
      maxes[t] = std::max<unsigned>(maxes[t], unsigned(e[i].size()));
      ss->shared[i] = i ? lcp(e[i - 1], e[i]) : 0;
      std::string tail = e[i].substr(1);
      size_t tg = std::lower_bound(e.begin(), e.end(), tail) - e.begin();
      if (tg >= e.size() || e[tg].compare(0, tail.size(), tail) != 0) { missing = 1; continue; }
      int b = base_code(e[i][0]);
      __atomic_fetch_or(&ss->bits[b][tg / 64], uint64_t(1) << (tg & 63), __ATOMIC_RELAXED);
    }
  });
  for (unsigned m : maxes) ss->max_len = std::max(ss->max_len, m);
  if (missing) { ss->err = "Missing expansion"; return ss; }
  ss->finalize();
  return ss;
}

}  // namespace orc

extern "C" {

unsigned orc_hardware_threads() { unsigned n = std::thread::hardware_concurrency(); return n ? n : 1; }

void* orc_kmer_count_mt(const char* seq, const uint64_t* offs, uint64_t n_reads, int k, uint32_t min_count, int keep_all,
                        unsigned nthreads) {
  return kmer_count_mt(seq, offs, n_reads, unsigned(k), min_count, keep_all != 0, nthreads ? nthreads : 1);
}

int orc_correct_reads_mt(void* ksh, const char* seq, const uint64_t* offs, uint64_t n_reads, uint32_t max_corrections,
                         uint32_t min_good_run, float trim_after_portion, char* out_seq, uint64_t* out_offs, uint8_t* out_keep,
                         uint64_t* stats, unsigned nthreads) {
  auto* ks = static_cast<KmerSet*>(ksh);
  FrcParams p;
  p.max_corrections = max_corrections;
  p.min_good_run = min_good_run;
  p.kmer_size = ks->k;
  p.ks = ks;
  const unsigned T = nthreads ? nthreads : 1;
  std::vector<CorrectStats> sts(T);
  std::vector<std::string> outs(T);
  std::vector<std::vector<uint32_t>> lens(T);
  std::vector<uint64_t> los(T, 0);
  parallel_ranges(n_reads, T, [&](unsigned t, uint64_t lo, uint64_t hi) {
    los[t] = lo;
    lens[t].reserve(hi - lo);
    for (uint64_t r = lo; r < hi; ++r) {
      std::string c;
      bool ok = correct_read(seq + offs[r], offs[r + 1] - offs[r], p, trim_after_portion, &c, &sts[t]);
      out_keep[r] = ok ? 1 : 0;
      lens[t].push_back(ok ? uint32_t(c.size()) : 0u);
      if (ok) outs[t] += c;
    }
  });
  uint64_t pos = 0;
  CorrectStats st;
  for (unsigned t = 0; t < T; ++t) {
    uint64_t r = los[t];
    for (uint32_t l : lens[t]) { out_offs[r++] = pos; pos += l; }
    st.corrected_reads += sts[t].corrected_reads; st.corrected_bases += sts[t].corrected_bases; st.failed += sts[t].failed;
    st.modified += sts[t].modified; st.modified_bases += sts[t].modified_bases; st.truncated += sts[t].truncated;
    st.dropped_bases += sts[t].dropped_bases; st.ref_records += sts[t].ref_records;
  }
  out_offs[n_reads] = pos;
  pos = 0;
  for (unsigned t = 0; t < T; ++t) { std::memcpy(out_seq + pos, outs[t].data(), outs[t].size()); pos += outs[t].size(); }
  if (stats) {
    stats[0] = st.corrected_reads; stats[1] = st.corrected_bases; stats[2] = st.failed;
    stats[3] = st.modified; stats[4] = st.modified_bases; stats[5] = st.truncated;
    stats[6] = st.dropped_bases; stats[7] = st.ref_records;
  }
  return 0;
}

// ---- entries
struct Entries { std::vector<std::string> e; uint64_t n_records = 0; };
void* orc_build_entries(const char* seq, const uint64_t* offs, uint64_t n_reads, int mode, int cover_k) {
  auto reads = split_reads(seq, offs, n_reads);
  reads.erase(std::remove_if(reads.begin(), reads.end(), [](const std::string& s) { return s.empty(); }),
              reads.end());
  auto* en = new Entries;
  if (mode == 0) {
    en->e = build_entries_simple(reads);
    for (auto& r : reads) en->n_records += 2 * r.size();
  } else {
    en->e = build_entries_cover(reads, unsigned(cover_k), &en->n_records);
  }
  return en;
}
void* orc_build_entries_mt(const char* seq, const uint64_t* offs, uint64_t n_reads, int cover_k, unsigned nthreads) {
  auto reads = split_reads(seq, offs, n_reads);
  reads.erase(std::remove_if(reads.begin(), reads.end(), [](const std::string& s) { return s.empty(); }), reads.end());
  auto* en = new Entries;
  en->e = build_entries_cover_mt(reads, unsigned(cover_k), &en->n_records, nthreads ? nthreads : 1);
  return en;
}
void* orc_entries_from_strings(const char* seq, const uint64_t* offs, uint64_t n) {
  auto* en = new Entries;
  en->e = split_reads(seq, offs, n);
  return en;
}
uint64_t orc_entries_count(void* h) { return static_cast<Entries*>(h)->e.size(); }
uint64_t orc_entries_records(void* h) { return static_cast<Entries*>(h)->n_records; }
uint64_t orc_entries_total_bases(void* h) {
  uint64_t t = 0;
  for (auto& s : static_cast<Entries*>(h)->e) t += s.size();
  return t;
}
void orc_entries_get(void* h, char* seq, uint64_t* offs) {
  auto* en = static_cast<Entries*>(h);
  uint64_t pos = 0;
  for (size_t i = 0; i < en->e.size(); ++i) {
    offs[i] = pos;
    std::memcpy(seq + pos, en->e[i].data(), en->e[i].size());
    pos += en->e[i].size();
  }
  offs[en->e.size()] = pos;
}
void orc_entries_free(void* h) { delete static_cast<Entries*>(h); }

// ---- seqset
void* orc_seqset_from_entries(void* eh) { return seqset_from_entries(static_cast<Entries*>(eh)->e); }
void* orc_seqset_from_entries_mt(void* eh, unsigned nthreads) {
  return seqset_from_entries_mt(static_cast<Entries*>(eh)->e, nthreads ? nthreads : 1);
}
// From decoded arrays (golden files): sizes/shared u32[n], bits[b] = ceil(n/64) words each.
void* orc_seqset_from_arrays(uint64_t n, const uint32_t* sizes, const uint32_t* shared, const uint64_t* bA,
                             const uint64_t* bC, const uint64_t* bG, const uint64_t* bT) {
  auto* ss = new Seqset;
  ss->n = n;
  ss->sizes.assign(sizes, sizes + n);
  ss->shared.assign(shared, shared + n);
  const uint64_t* src[4] = {bA, bC, bG, bT};
  for (int b = 0; b < 4; ++b) ss->bits[b].assign(src[b], src[b] + (n + 63) / 64);
  for (uint64_t i = 0; i < n; ++i) ss->max_len = std::max<unsigned>(ss->max_len, sizes[i]);
  ss->finalize();
  return ss;
}
const char* orc_seqset_error(void* h) {
  auto* ss = static_cast<Seqset*>(h);
  return ss->err.empty() ? nullptr : ss->err.c_str();
}
uint64_t orc_seqset_size(void* h) { return static_cast<Seqset*>(h)->n; }
uint32_t orc_seqset_max_len(void* h) { return static_cast<Seqset*>(h)->max_len; }
void orc_seqset_fixed(void* h, uint64_t* out5) { std::memcpy(out5, static_cast<Seqset*>(h)->fixed, 40); }
void orc_seqset_arrays(void* h, uint32_t* sizes, uint32_t* shared) {
  auto* ss = static_cast<Seqset*>(h);
  std::memcpy(sizes, ss->sizes.data(), ss->n * 4);
  std::memcpy(shared, ss->shared.data(), ss->n * 4);
}
// which: 0 bits, 1 subaccum, 2 accum. Returns word count; copies if out != null.
uint64_t orc_seqset_prev(void* h, int b, int which, uint64_t* out) {
  auto* ss = static_cast<Seqset*>(h);
  auto& v = which == 0 ? ss->bits[b] : which == 1 ? ss->subaccum[b] : ss->accum[b];
  if (out) std::memcpy(out, v.data(), v.size() * 8);
  return v.size();
}
// which: 0 entry_sizes (max=max_len), 1 shared (max=max_len-1). Returns word count.
uint64_t orc_seqset_packed(void* h, int which, uint64_t* out) {
  auto* ss = static_cast<Seqset*>(h);
  auto v = pack_varbit(which == 0 ? ss->sizes : ss->shared, which == 0 ? ss->max_len : ss->max_len - 1);
  if (out) std::memcpy(out, v.data(), v.size() * 8);
  return v.size();
}
uint64_t orc_seqset_count(void* h, int b, uint64_t i) { return static_cast<Seqset*>(h)->count(b, i); }
void orc_seqset_find(void* h, const char* s, uint64_t len, uint64_t* out3) {
  auto r = static_cast<Seqset*>(h)->find(std::string(s, len));
  out3[0] = r.begin; out3[1] = r.end; out3[2] = r.size;
}
// Batched walk primitives; invalid result = (0,0,0) for push_front_drop (seqset.cpp:366,385).
void orc_push_front_drop(void* h, uint64_t n, const uint64_t* begin, const uint64_t* end, const uint32_t* size,
                         const uint8_t* base, const uint32_t* min_ctx, uint64_t* obegin, uint64_t* oend,
                         uint32_t* osize) {
  auto* ss = static_cast<Seqset*>(h);
  for (uint64_t i = 0; i < n; ++i) {
    auto r = ss->push_front_drop(Seqset::Range{begin[i], end[i], size[i]}, base[i], min_ctx[i]);
    obegin[i] = r.begin; oend[i] = r.end; osize[i] = r.size;
  }
}
void orc_push_front(void* h, uint64_t n, const uint64_t* begin, const uint64_t* end, const uint32_t* size,
                    const uint8_t* base, uint64_t* obegin, uint64_t* oend, uint32_t* osize) {
  auto* ss = static_cast<Seqset*>(h);
  for (uint64_t i = 0; i < n; ++i) {
    auto r = ss->push_front(Seqset::Range{begin[i], end[i], size[i]}, base[i]);
    obegin[i] = r.begin; oend[i] = r.end; osize[i] = r.size;
  }
}
void orc_seqset_free(void* h) { delete static_cast<Seqset*>(h); }

// ---- walk drivers.  Ranges travel as (begin,end,size) triples of u64.
static RangeSet to_set(const uint64_t* tri, uint64_t n) {
  RangeSet rs;
  for (uint64_t i = 0; i < n; ++i) rs.insert(Range{tri[3 * i], tri[3 * i + 1], uint32_t(tri[3 * i + 2])});
  return rs;
}
// out_cands: up to cap triples + drop flags; out_next: up to cap triples.  Returns best rc base or -1.
int orc_trace_one_base(void* h, const uint64_t* tri, uint64_t n, uint32_t min_ctx, uint64_t cap, uint64_t* out_cands,
                       uint8_t* out_drop, uint64_t* n_cands, uint64_t* out_next, uint64_t* n_next) {
  auto* ss = static_cast<Seqset*>(h);
  std::vector<std::pair<Range, bool>> cands;
  RangeSet next;
  int best = trace_one_base(*ss, to_set(tri, n), min_ctx, &cands, &next);
  *n_cands = cands.size();
  *n_next = next.size();
  uint64_t i = 0;
  for (auto& c : cands) {
    if (i >= cap) break;
    out_cands[3 * i] = c.first.begin; out_cands[3 * i + 1] = c.first.end; out_cands[3 * i + 2] = c.first.size;
    out_drop[i] = c.second;
    ++i;
  }
  i = 0;
  for (auto& r : next) {
    if (i >= cap) break;
    out_next[3 * i] = r.begin; out_next[3 * i + 1] = r.end; out_next[3 * i + 2] = r.size;
    ++i;
  }
  return best;
}
// events: (offset, base) pairs up to cap; final frontier triples up to cap.
void orc_branch_trace(void* h, const uint8_t* seq, uint64_t n, const uint64_t* tri, uint64_t n_cur, uint32_t min_overlap,
                      uint64_t cap, uint32_t* ev_off, uint8_t* ev_base, uint64_t* n_events, uint64_t* out_cur,
                      uint64_t* n_out_cur) {
  auto* ss = static_cast<Seqset*>(h);
  std::vector<std::pair<uint32_t, uint8_t>> ev;
  RangeSet fin;
  branch_trace(*ss, seq, n, to_set(tri, n_cur), min_overlap, &ev, &fin);
  *n_events = ev.size();
  for (uint64_t i = 0; i < ev.size() && i < cap; ++i) { ev_off[i] = ev[i].first; ev_base[i] = ev[i].second; }
  *n_out_cur = fin.size();
  uint64_t i = 0;
  for (auto& r : fin) {
    if (i >= cap) break;
    out_cur[3 * i] = r.begin; out_cur[3 * i + 1] = r.end; out_cur[3 * i + 2] = r.size;
    ++i;
  }
}
// out_bases[cap_steps], out_counts[cap_steps], out_tri[cap_steps*cap*3]; returns the stop reason, *n_steps
int orc_push_to_pair_trace(void* h, const uint64_t* tri, uint64_t n, uint32_t min_overlap, uint64_t max_steps,
                           uint64_t cap, uint8_t* out_bases, uint64_t* out_counts, uint64_t* out_tri,
                           uint64_t* n_steps) {
  std::vector<uint8_t> bases;
  std::vector<RangeSet> fronts;
  int reason = push_to_pair_trace(*static_cast<Seqset*>(h), to_set(tri, n), min_overlap, max_steps, &bases, &fronts);
  *n_steps = bases.size();
  for (size_t st = 0; st < bases.size(); ++st) {
    out_bases[st] = bases[st];
    out_counts[st] = fronts[st].size();
    uint64_t i = 0;
    for (auto& r : fronts[st]) {
      if (i >= cap) break;
      uint64_t* o = out_tri + (st * cap + i) * 3;
      o[0] = r.begin; o[1] = r.end; o[2] = r.size;
      ++i;
    }
  }
  return reason;
}
void orc_ref_map_build(void* h, const uint8_t* ref_codes, const uint64_t* ext_start, const uint64_t* ext_size,
                       uint64_t n_ext, uint8_t* out) {
  ref_map_build(*static_cast<Seqset*>(h), ref_codes, ext_start, ext_size, n_ext, out);
}

// ---- readmap.  reads = concatenated corrected reads; rec_first[n_rec+1] delimits records (1 or 2 reads).
static std::string g_rm_err;
void* orc_readmap_build(void* ssh, void* eh, const char* seq, const uint64_t* offs, uint64_t n_reads,
                        const uint64_t* rec_first, uint64_t n_rec, int is_paired, int canonicalize) {
  auto reads = split_reads(seq, offs, n_reads);
  std::vector<uint64_t> rf(rec_first, rec_first + n_rec + 1);
  g_rm_err.clear();
  return readmap_build(*static_cast<Seqset*>(ssh), static_cast<Entries*>(eh)->e, reads, rf, is_paired != 0, canonicalize != 0, &g_rm_err);
}
void* orc_readmap_build_mt(void* ssh, void* eh, const char* seq, const uint64_t* offs, uint64_t n_reads, const uint64_t* rec_first,
                           uint64_t n_rec, int is_paired, int canonicalize, unsigned threads) {
  if (!threads) threads = std::max(1u, std::thread::hardware_concurrency());
  std::vector<std::string> reads(n_reads);
  rm_parallel(n_reads, threads, [&](unsigned, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) reads[i].assign(seq + offs[i], seq + offs[i + 1]);
  });
  std::vector<uint64_t> rf(rec_first, rec_first + n_rec + 1);
  g_rm_err.clear();
  return readmap_build(*static_cast<Seqset*>(ssh), static_cast<Entries*>(eh)->e, reads, rf, is_paired != 0, canonicalize != 0, &g_rm_err,
                       threads);
}
const char* orc_readmap_error() { return g_rm_err.c_str(); }
uint64_t orc_readmap_size(void* h) { return static_cast<Readmap*>(h)->n_reads; }
uint32_t orc_readmap_max_len(void* h) { return static_cast<Readmap*>(h)->max_read_len; }
// which: 0 read_len(u32->u64), 1 mate_loop_ptr, 2 is_forward, 3 entry_id ; out = n u64
void orc_readmap_get(void* h, int which, uint64_t* out) {
  auto* rm = static_cast<Readmap*>(h);
  for (uint64_t i = 0; i < rm->n_reads; ++i)
    out[i] = which == 0 ? rm->read_len[i] : which == 1 ? rm->mate_loop_ptr[i] : which == 2 ? rm->is_forward[i] : rm->entry_id[i];
}
// bitcounts: sel 0 = source_to_mid, 1 = dest_to_mid; which 0 bits / 1 subaccum / 2 accum
uint64_t orc_readmap_bitcount(void* h, int sel, int which, uint64_t* out) {
  auto* rm = static_cast<Readmap*>(h);
  Seqset& b = sel == 0 ? rm->s2m : rm->d2m;
  auto& v = which == 0 ? b.bits[0] : which == 1 ? b.subaccum[0] : b.accum[0];
  if (out) std::memcpy(out, v.data(), v.size() * 8);
  return v.size();
}
// generic LSB-first packing of n values at `bits` bits each (packed_varbit_vector / packed_vector layout)
uint64_t orc_pack_bits(const uint64_t* vals, uint64_t n, int bits, uint64_t* out) {
  uint64_t nw = (n * bits + 63) / 64;
  if (!out) return nw;
  std::memset(out, 0, nw * 8);
  for (uint64_t i = 0; i < n && bits; ++i) {
    uint64_t bit = i * bits;
    out[bit / 64] |= vals[i] << (bit % 64);
    if (bit % 64 + bits > 64) out[bit / 64 + 1] |= vals[i] >> (64 - bit % 64);
  }
  return nw;
}
// readmap from decoded arrays (a readmap file read back): entry ids, lengths, loop pointers, orientation
void* orc_readmap_from_arrays(uint64_t n, const uint64_t* entry_id, const uint64_t* read_len, const uint64_t* ptr,
                              const uint64_t* is_fwd, uint32_t max_read_len) {
  auto* rm = new Readmap;
  rm->n_reads = n;
  rm->max_read_len = max_read_len;
  rm->entry_id.assign(entry_id, entry_id + n);
  rm->read_len.resize(n);
  rm->mate_loop_ptr.assign(ptr, ptr + n);
  rm->is_forward.resize(n);
  rm->d2m.n = n;
  rm->d2m.bits[0].assign((n + 63) / 64, 0);
  for (uint64_t i = 0; i < n; ++i) {
    rm->read_len[i] = uint32_t(read_len[i]);
    rm->is_forward[i] = uint8_t(is_fwd[i]);
    if (i == 0 || entry_id[i] != entry_id[i - 1]) rm->d2m.bits[0][i / 64] |= 1ull << (i & 63);
  }
  return rm;
}
uint64_t orc_readmap_prefix_reads(void* rmh, void* ssh, uint64_t begin, uint64_t end, uint32_t size, int limit,
                                  uint64_t cap, uint64_t* out) {
  auto ids = rm_prefix_reads(*static_cast<Seqset*>(ssh), *static_cast<Readmap*>(rmh), Range{begin, end, size}, limit);
  for (uint64_t i = 0; i < ids.size() && i < cap; ++i) out[i] = ids[i];
  return ids.size();
}
// out[6] = has_mate, mate_rc id (or ~0), rev_comp id, entry range begin/end/size
void orc_readmap_read_info(void* rmh, void* ssh, uint64_t id, uint64_t* out) {
  auto& rm = *static_cast<Readmap*>(rmh);
  auto& ss = *static_cast<Seqset*>(ssh);
  out[0] = rm_has_mate(rm, id);
  out[1] = out[0] ? rm_get_mate_rc(rm, id) : ~0ull;
  out[2] = rm_get_rev_comp(rm, id);
  Range r = rm_read_entry(ss, rm, id);
  out[3] = r.begin; out[4] = r.end; out[5] = r.size;
}
// marks: ascending offsets in (0, n).  Outputs: for offset 0, each mark, and n (in that order) a count
// in out_counts[] and up to cap triples in out_tri[slot*cap*3 ..]; mates likewise (count returned).
uint64_t orc_propagate_from_end(void* ssh, void* rmh, const uint8_t* seq, uint64_t n, const uint64_t* tri,
                                uint64_t n_ends, uint32_t max_pairs, const uint32_t* marks, uint64_t n_marks,
                                uint64_t cap, uint64_t* out_counts, uint64_t* out_tri, uint64_t mate_cap,
                                uint64_t* out_mates) {
  PathOut po;
  std::set<uint32_t> mk(marks, marks + n_marks);
  propagate_from_end(*static_cast<Seqset*>(ssh), static_cast<Readmap*>(rmh), to_set(tri, n_ends), seq, n, max_pairs, mk,
                     &po);
  std::vector<uint32_t> slots{0};
  for (uint64_t i = 0; i < n_marks; ++i) slots.push_back(marks[i]);
  slots.push_back(uint32_t(n));
  for (size_t sidx = 0; sidx < slots.size(); ++sidx) {
    const RangeSet& rs = po.entries[slots[sidx]];
    out_counts[sidx] = rs.size();
    uint64_t i = 0;
    for (auto& r : rs) {
      if (i >= cap) break;
      uint64_t* o = out_tri + (sidx * cap + i) * 3;
      o[0] = r.begin; o[1] = r.end; o[2] = r.size;
      ++i;
    }
  }
  uint64_t i = 0;
  for (auto& r : po.mates) {
    if (i >= mate_cap) break;
    out_mates[3 * i] = r.begin; out_mates[3 * i + 1] = r.end; out_mates[3 * i + 2] = r.size;
    ++i;
  }
  return po.mates.size();
}
// ---- merge: handles of Entries in, MergeOut out
void* orc_merge(void* const* entry_handles, uint64_t n_in) {
  std::vector<const std::vector<std::string>*> in;
  for (uint64_t i = 0; i < n_in; ++i) in.push_back(&static_cast<Entries*>(entry_handles[i])->e);
  auto* mo = new MergeOut;
  merge_entries(in, mo);
  return mo;
}
void* orc_merge_entries(void* h) {
  auto* en = new Entries;
  en->e = static_cast<MergeOut*>(h)->entries;
  return en;
}
uint64_t orc_merge_map(void* h, uint64_t input, uint64_t* out) {
  auto& m = static_cast<MergeOut*>(h)->maps[input];
  if (out) std::copy(m.begin(), m.end(), out);
  return m.size();
}
void orc_merge_free(void* h) { delete static_cast<MergeOut*>(h); }
// fast_migrate: same read ids, entry ids translated through map[]
void* orc_readmap_migrate(void* rmh, const uint64_t* map, uint64_t n_new_entries) {
  auto* o = static_cast<Readmap*>(rmh);
  auto* rm = new Readmap(*o);
  for (auto& e : rm->entry_id) e = map[e];
  rm->s2m = Seqset();
  rm->s2m.n = n_new_entries;
  rm->s2m.bits[0].assign((n_new_entries + 63) / 64, 0);
  rm->d2m = Seqset();
  rm->d2m.n = rm->n_reads;
  rm->d2m.bits[0].assign((rm->n_reads + 63) / 64, 0);
  for (uint64_t i = 0; i < rm->n_reads; ++i) {
    rm->s2m.bits[0][rm->entry_id[i] / 64] |= 1ull << (rm->entry_id[i] & 63);
    if (i == 0 || rm->entry_id[i] != rm->entry_id[i - 1]) rm->d2m.bits[0][i / 64] |= 1ull << (i & 63);
  }
  rm->s2m.finalize_bitcount(0);
  rm->d2m.finalize_bitcount(0);
  return rm;
}
void orc_readmap_free(void* h) { delete static_cast<Readmap*>(h); }

}  // extern "C"
