// pycapi.cpp — pybind11 module biograph_amd._capi: the classes and functions of the reference's Python C API
// (module biograph._capi_<maj><min>, python/biograph/module.cpp:30) for the seqset / readmap / graph-discover
// path, over the C ABI of libbgseq.so (include/bgseq.h).  Same names, argument meaning and error behaviour:
//
//   Seqset, SeqsetEntry      python/biograph/seqset.cpp:46-183      (invalid results are None, valid_or_none :20-42)
//   Readmap, ReadmapRead     python/biograph/readmap.cpp:27-116
//   BioGraph                 python/biograph/biograph.cpp:39-79
//   Reference, ReferenceRange, ReferenceContext   python/biograph/reference.cpp:14-185 (the BWT search on the host:
//                            bg_bwt_*)
//   Sequence                 python/biograph/dna_sequence.cpp:69-158;  CacheStrategy  biograph.cpp:9-37 (accepted and
//                            ignored: the files are read once into HBM);  version, build_revision  version.cpp:15-18
//   Assembly + make_ref_assemblies, graph_trim_ref, update_rc_seqset_entries, discover_branch,
//   discover_push_to_pair    python/biograph/variants/graph_discover.cpp:109-133, variants/assemble.h:285-420
//
// C++ exceptions surface as RuntimeError like the reference's io_exception.  Every query is a call into the
// library (batch of one for the single-entry methods); there is no CPU implementation behind this module.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <limits.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bgseq.h"

namespace py = pybind11;

namespace {

void check(int rc, const char* what) {
  if (rc != BG_OK) throw std::runtime_error(std::string(what) + ": " + bg_last_error());
}

// one context per process (device 0, the null stream), created on first use
struct Ctx {
  bg_ctx* h = nullptr;
  Ctx() { check(bg_ctx_create(0, nullptr, &h), "bg_ctx_create"); }
  ~Ctx() {}  // objects may outlive it at interpreter exit: the driver reclaims the memory
};
std::shared_ptr<Ctx> ctx() {
  static std::shared_ptr<Ctx> c = std::make_shared<Ctx>();
  return c;
}

// a few device words for batch-of-one calls
struct DevScratch {
  std::shared_ptr<Ctx> c;
  void* p = nullptr;
  explicit DevScratch(uint64_t bytes) : c(ctx()) { check(bg_dev_alloc(c->h, bytes, &p), "bg_dev_alloc"); }
  ~DevScratch() { if (p) bg_dev_free(c->h, p); }
  template <class T> T* at(uint64_t byte_off) { return reinterpret_cast<T*>(static_cast<char*>(p) + byte_off); }
  void up(uint64_t off, const void* src, uint64_t n) { check(bg_dev_upload(c->h, at<char>(off), src, n), "upload"); }
  void down(void* dst, uint64_t off, uint64_t n) { check(bg_dev_download(c->h, dst, at<char>(off), n), "download"); }
};

int base_code(char b) {
  switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
  }
  throw std::runtime_error(std::string("Invalid base: ") + b);  // dna_base.h:54
}

struct SeqsetObj : std::enable_shared_from_this<SeqsetObj> {
  std::shared_ptr<Ctx> c;
  bg_seqset* h = nullptr;
  std::string uuid;
  std::vector<uint32_t> sizes, shared;  // host copies for truncate / get_entry_by_id (fetched on first use)
  explicit SeqsetObj(const std::string& path) : c(ctx()) {
    check(bg_seqset_open(c->h, path.c_str(), &h), "Seqset");
    char u[64] = "";
    bg_seqset_uuid(h, u, sizeof u);
    uuid = u;
  }
  ~SeqsetObj() { if (h) bg_seqset_free(h); }
  uint64_t size() const { return bg_seqset_size(h); }
  void need_arrays() {
    if (!sizes.empty() || !size()) return;
    sizes.resize(size());
    shared.resize(size());
    check(bg_seqset_arrays(c->h, h, sizes.data(), shared.data()), "bg_seqset_arrays");
  }
};

struct Entry {  // seqset_range
  std::shared_ptr<SeqsetObj> ss;
  uint64_t begin = 0, end = 0;
  uint32_t size = 0;
  bool valid() const { return begin < end; }
  bool operator==(const Entry& o) const { return ss == o.ss && begin == o.begin && end == o.end && size == o.size; }
};

py::object entry_or_none(const Entry& e) { return e.valid() ? py::cast(e) : py::none(); }

Entry seqset_find(const std::shared_ptr<SeqsetObj>& ss, const std::string& seq) {
  for (char b : seq) base_code(b);
  DevScratch d(seq.size() + 64);
  uint64_t offs[2] = {0, seq.size()};
  const uint64_t o_off = (seq.size() + 7) / 8 * 8;
  if (!seq.empty()) d.up(0, seq.data(), seq.size());
  d.up(o_off, offs, 16);
  check(bg_find_batch(ss->c->h, ss->h, 1, d.at<uint8_t>(0), d.at<uint64_t>(o_off), d.at<uint64_t>(o_off + 16),
                      d.at<uint64_t>(o_off + 24)), "find");
  uint64_t be[2];
  d.down(be, o_off + 16, 16);
  return Entry{ss, be[0], be[1], uint32_t(seq.size())};
}

std::string entry_sequence(const Entry& e, int bases) {
  if (!e.valid() || e.size == 0 || bases <= 0) return "";
  uint64_t offs[2] = {0, 0};
  check(bg_seqset_entries(e.ss->c->h, e.ss->h, e.begin, 1, nullptr, offs), "sequence");
  std::string s(offs[1], '\0');
  check(bg_seqset_entries(e.ss->c->h, e.ss->h, e.begin, 1, reinterpret_cast<uint8_t*>(&s[0]), offs), "sequence");
  s.resize(std::min<uint64_t>(e.size, s.size()));
  if (uint64_t(bases) < s.size()) s.resize(size_t(bases));
  return s;
}

Entry entry_walk(const Entry& e, char base, bool drop, int min_ctx) {
  const uint8_t b = uint8_t(base_code(base));
  DevScratch d(128);
  const uint32_t mc = uint32_t(min_ctx < 0 ? 0 : min_ctx);
  d.up(0, &e.begin, 8);
  d.up(8, &e.end, 8);
  d.up(16, &e.size, 4);
  d.up(20, &b, 1);
  d.up(24, &mc, 4);
  if (drop)
    check(bg_push_front_drop_batch(e.ss->c->h, e.ss->h, 1, d.at<uint64_t>(0), d.at<uint64_t>(8), d.at<uint32_t>(16), d.at<uint8_t>(20),
                                   d.at<uint32_t>(24), d.at<uint64_t>(32), d.at<uint64_t>(40), d.at<uint32_t>(48)), "push_front_drop");
  else
    check(bg_push_front_batch(e.ss->c->h, e.ss->h, 1, d.at<uint64_t>(0), d.at<uint64_t>(8), d.at<uint32_t>(16), d.at<uint8_t>(20),
                              d.at<uint64_t>(32), d.at<uint64_t>(40), d.at<uint32_t>(48)), "push_front");
  Entry r{e.ss, 0, 0, 0};
  d.down(&r.begin, 32, 8);
  d.down(&r.end, 40, 8);
  d.down(&r.size, 48, 4);
  return r;
}

// seqset_range::truncate (bio_base/seqset.cpp:630-661): widen to the entries sharing the first new_size bases
Entry entry_truncate(const Entry& e, uint32_t new_size) {
  if (e.size <= new_size) return e;
  e.ss->need_arrays();
  const auto& sh = e.ss->shared;
  const uint64_t n = e.ss->size();
  uint64_t b = e.begin, en = e.end;
  while (b > 0 && sh[b] >= new_size) --b;
  while (en < n && sh[en] >= new_size) ++en;
  return Entry{e.ss, b, en, new_size};
}

Entry entry_pop_front(const Entry& e) {  // seqset_range::pop_front: the entries prefixed by sequence()[1:]
  if (e.size == 0) throw std::runtime_error("pop_front of an empty sequence");
  return seqset_find(e.ss, entry_sequence(e, INT_MAX).substr(1));
}

std::string entry_repr(const Entry& e) {
  std::ostringstream o;
  o << "<SeqsetEntry " << e.begin << "-" << e.end << ": " << entry_sequence(e, INT_MAX) << ">";
  return o.str();
}

struct ReadmapObj : std::enable_shared_from_this<ReadmapObj> {
  std::shared_ptr<Ctx> c;
  std::shared_ptr<SeqsetObj> ss;
  bg_readmap* h = nullptr;
  std::string path;
  std::vector<uint64_t> entry_id;
  std::vector<uint32_t> read_len, loop;
  std::vector<uint8_t> fwd;
  bool has_loops = true;  // false: `loop` holds the legacy pair table (readmap.cpp:48-51)
  ReadmapObj(const std::shared_ptr<SeqsetObj>& s, const std::string& p) : c(ctx()), ss(s), path(p) {
    check(bg_readmap_open(c->h, p.c_str(), &h), "Readmap");
    const uint64_t n = bg_readmap_size(h);
    entry_id.resize(n); read_len.resize(n); loop.resize(n); fwd.resize(n);
    if (n) check(bg_readmap_arrays(c->h, h, entry_id.data(), read_len.data(), loop.data(), fwd.data()), "bg_readmap_arrays");
    has_loops = bg_readmap_has_mate_loop(h) != 0;
  }
  ~ReadmapObj() { if (h) bg_readmap_free(h); }
  uint64_t size() const { return entry_id.size(); }
  uint32_t min_read_len() const {
    uint32_t m = UINT32_MAX;
    for (uint32_t l : read_len) m = std::min(m, l);
    return read_len.empty() ? 0 : m;
  }
};

struct Read {  // readmap::read
  std::shared_ptr<ReadmapObj> rm;
  uint32_t id = 0;
  bool operator==(const Read& o) const { return rm == o.rm && id == o.id; }
};

Read read_by_id(const std::shared_ptr<ReadmapObj>& rm, int64_t id) {
  if (id < 0 || uint64_t(id) >= rm->size()) throw std::runtime_error("Read id out of range");
  return Read{rm, uint32_t(id)};
}
bool read_has_mate(const Read& r) {  // readmap.cpp:207-218
  if (!r.rm->has_loops) return r.rm->loop[r.id] != 0xffffffffu;
  return r.rm->loop[r.rm->loop[r.id]] != r.id;
}
Read read_mate(const Read& r) {  // readmap.cpp:220-239
  if (!read_has_mate(r)) throw std::runtime_error("Read has no mate");
  return Read{r.rm, r.rm->has_loops ? r.rm->loop[r.rm->loop[r.id]] : r.rm->loop[r.id]};
}
Read read_rev_comp(const Read& r) {  // readmap.cpp:248-268: one hop forward for a forward read, three for its reverse
  if (!r.rm->has_loops)
    throw std::runtime_error("Readmap has no mate loop table; use \"biograph upgrade\" to construct it");
  uint32_t x = r.id;
  for (int i = 0, n = r.rm->fwd[r.id] ? 1 : 3; i < n; ++i) x = r.rm->loop[x];
  return Read{r.rm, x};
}
Entry read_entry(const Read& r) {  // readmap.h:249-251: ctx_entry(id).truncate(length)
  const uint64_t e = r.rm->entry_id[r.id];
  r.rm->ss->need_arrays();
  return entry_truncate(Entry{r.rm->ss, e, e + 1, r.rm->ss->sizes[e]}, r.rm->read_len[r.id]);
}
std::string read_repr(const Read& r) {
  std::ostringstream o;
  o << "<ReadmapRead " << r.id << ": " << entry_sequence(read_entry(r), INT_MAX) << ">";
  return o.str();
}

std::vector<Read> prefix_reads(const std::shared_ptr<ReadmapObj>& rm, const Entry& e, int min_read_len) {
  if (e.ss->uuid != rm->ss->uuid) throw std::runtime_error("Readmap does not belong to the seqset of this entry");  // readmap.cpp:436-438
  uint32_t cap = 256;
  for (;;) {
    DevScratch d(64 + uint64_t(cap) * 4);
    d.up(0, &e.begin, 8);
    d.up(8, &e.end, 8);
    d.up(16, &e.size, 4);
    check(bg_prefix_reads_batch(rm->c->h, rm->ss->h, rm->h, 1, d.at<uint64_t>(0), d.at<uint64_t>(8), d.at<uint32_t>(16), min_read_len, cap,
                                d.at<uint32_t>(64), d.at<uint32_t>(24)), "get_prefix_reads");
    uint32_t count = 0;
    d.down(&count, 24, 4);
    if (count > cap) { cap = count; continue; }
    std::vector<uint32_t> ids(count);
    if (count) d.down(ids.data(), 64, uint64_t(count) * 4);
    std::vector<Read> out;
    for (uint32_t i : ids) out.push_back(Read{rm, i});
    return out;
  }
}

// readmap::get_pair_stats (bio_base/readmap.cpp:300-319)
struct PairStats { uint64_t paired_reads = 0, paired_bases = 0, unpaired_reads = 0, unpaired_bases = 0; };
PairStats pair_stats(const std::shared_ptr<ReadmapObj>& rm) {
  PairStats r;
  for (uint32_t i = 0; i < rm->size(); ++i) {
    if (read_has_mate(Read{rm, i})) { ++r.paired_reads; r.paired_bases += rm->read_len[i]; }
    else { ++r.unpaired_reads; r.unpaired_bases += rm->read_len[i]; }
  }
  r.paired_reads /= 2; r.unpaired_reads /= 2; r.paired_bases /= 2; r.unpaired_bases /= 2;  // counted in both orientations
  return r;
}

// read ids of the entries [begin, end) (readmap::entry_to_index_range): the read ids are grouped by ascending entry
std::pair<uint32_t, uint32_t> entry_reads(const ReadmapObj& rm, uint64_t begin, uint64_t end) {
  const auto lo = std::lower_bound(rm.entry_id.begin(), rm.entry_id.end(), begin);
  const auto hi = std::lower_bound(lo, rm.entry_id.end(), end);
  return {uint32_t(lo - rm.entry_id.begin()), uint32_t(hi - rm.entry_id.begin())};
}

// readmap::approx_coverage (bio_base/readmap.cpp:350-411): the push_front_drop walk along the sequence is one device
// chain (bg_walk_chains), the counting is the reference's.
std::vector<int> approx_seq_coverage(const std::shared_ptr<ReadmapObj>& rm, const std::string& seq) {
  const size_t n = seq.size();
  std::vector<int> out(n, 0);
  if (!n) return out;
  std::vector<uint8_t> codes(n);
  for (size_t i = 0; i < n; ++i) codes[i] = uint8_t(base_code(seq[i]));
  const uint64_t start[3] = {0, rm->ss->size(), 0};
  std::vector<uint64_t> ranges(n * 3);
  check(bg_walk_chains(rm->c->h, rm->ss->h, 1, start, codes.data(), uint32_t(n), ranges.data()), "get_approx_seq_coverage");
  std::vector<int> rstart[2] = {std::vector<int>(n, 0), std::vector<int>(n, 0)}, rend[2] = {std::vector<int>(n, 0), std::vector<int>(n, 0)};
  for (size_t pos = 0; pos < n; ++pos) {
    const uint64_t b = ranges[pos * 3], e = ranges[pos * 3 + 1], sz = ranges[pos * 3 + 2];
    if (b + 1 != e) continue;
    const auto ids = entry_reads(*rm, b, b + 1);
    for (uint32_t id = ids.first; id < ids.second; ++id) {
      const int read_len = int(rm->read_len[id]);
      if (read_len > int(sz)) continue;
      const int st = int(pos) + 1 - read_len;
      if (st < 0) continue;
      const int strand = rm->fwd[id] ? 1 : 0;  // the walk builds the complement: the strand switches
      rstart[strand][size_t(st)]++;
      rend[strand][pos]++;
    }
  }
  int cur[2] = {0, 0};
  for (size_t i = 0; i < n; ++i)
    for (int s = 0; s < 2; ++s) {
      cur[s] += rstart[s][i];
      out[i] += cur[s];
      cur[s] -= rend[s][i];
    }
  return out;
}

// readmap::get_reads_containing (bio_base/readmap.cpp:485-495, containing_read_iterator :675-746): a depth-first walk
// over everything that can stand in front of the entry's sequence (push_front of the four bases, one device call per
// node), reporting the reads of every node that are at least as long as it: (bases before the sequence, read).
std::vector<std::pair<uint32_t, Read>> reads_containing(const std::shared_ptr<ReadmapObj>& rm, const Entry& e) {
  if (e.ss->uuid != rm->ss->uuid) throw std::runtime_error("Cannot use a readmap with a seqset it doesn't belong to.");  // :486-488
  std::vector<std::pair<uint32_t, Read>> out;
  if (!e.valid()) return out;
  const uint32_t orig_len = e.size;
  auto report = [&](const Entry& r) {  // start_entry + skip_non_matching
    const auto ids = entry_reads(*rm, r.begin, r.end);
    for (uint32_t id = ids.first; id < ids.second; ++id)
      if (rm->read_len[id] >= r.size) out.emplace_back(r.size - orig_len, Read{rm, id});
  };
  auto push4 = [&](const Entry& r, Entry* next) {  // the four push_fronts of a node in one call
    DevScratch d(256);
    uint64_t b[4] = {r.begin, r.begin, r.begin, r.begin}, en[4] = {r.end, r.end, r.end, r.end};
    uint32_t sz[4] = {r.size, r.size, r.size, r.size};
    uint8_t base[4] = {0, 1, 2, 3};
    d.up(0, b, 32); d.up(32, en, 32); d.up(64, sz, 16); d.up(80, base, 4);
    check(bg_push_front_batch(rm->c->h, rm->ss->h, 4, d.at<uint64_t>(0), d.at<uint64_t>(32), d.at<uint32_t>(64), d.at<uint8_t>(80),
                              d.at<uint64_t>(96), d.at<uint64_t>(128), d.at<uint32_t>(160)), "get_reads_containing");
    uint64_t ob[4], oe[4];
    uint32_t os[4];
    d.down(ob, 96, 32); d.down(oe, 128, 32); d.down(os, 160, 16);
    for (int i = 0; i < 4; ++i) next[i] = Entry{r.ss, ob[i], oe[i], os[i]};
  };
  Entry cur = e;
  report(cur);
  for (;;) {  // advance_entry (:714-746)
    Entry next[4];
    push4(cur, next);
    int down = -1;
    for (int b = 0; b < 4 && down < 0; ++b) if (next[b].valid()) down = b;
    if (down >= 0) { cur = next[down]; report(cur); continue; }
    bool moved = false;
    while (cur.size > orig_len && !moved) {
      const int prev_base = base_code(entry_sequence(cur, 1)[0]);
      const Entry popped = entry_pop_front(cur);
      push4(popped, next);
      for (int b = prev_base + 1; b < 4 && !moved; ++b)
        if (next[b].valid()) { cur = next[b]; report(cur); moved = true; }
      if (!moved) cur = popped;
    }
    if (!moved) return out;
  }
}

// ---- BioGraph directory (bio_base/biograph_dir.h:48-88): seqset, coverage/<sha>.readmap, metadata/bg_info.json
std::string read_file(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open " + p);
  std::ostringstream o;
  o << f.rdbuf();
  return o.str();
}
std::string json_take(const std::string& js, size_t p, size_t* end) {
  std::string v;
  size_t i = p + 1;
  for (; i < js.size() && js[i] != '"'; ++i) {
    if (js[i] == '\\' && i + 1 < js.size()) { ++i; v.push_back(js[i] == 'n' ? '\n' : js[i] == 't' ? '\t' : js[i]); }
    else v.push_back(js[i]);
  }
  *end = i + 1;
  return v;
}
struct Metadata {
  std::string accession_id, biograph_id, version;
  std::map<std::string, std::string> samples;
};
Metadata parse_metadata(const std::string& js) {
  Metadata m;
  auto str_of = [&](const char* key) {
    std::string pat = std::string("\"") + key + "\":\"";
    size_t p = js.find(pat), e;
    return p == std::string::npos ? std::string() : json_take(js, p + pat.size() - 1, &e);
  };
  m.accession_id = str_of("accession_id");
  m.biograph_id = str_of("biograph_id");
  m.version = str_of("version");
  size_t p = js.find("\"samples\":{");
  if (p != std::string::npos) {
    p += 11;
    while (p < js.size()) {
      while (p < js.size() && (js[p] == ',' || js[p] == ' ' || js[p] == '\n')) ++p;
      if (p >= js.size() || js[p] != '"') break;
      size_t e;
      std::string k = json_take(js, p, &e);
      size_t v0 = js.find('"', e);
      if (v0 == std::string::npos) break;
      m.samples[k] = json_take(js, v0, &e);
      p = e;
    }
  }
  return m;
}

struct BioGraphObj {
  std::string dir;
  std::shared_ptr<SeqsetObj> seqset;
  Metadata meta;
  explicit BioGraphObj(const std::string& d) : dir(d) {
    seqset = std::make_shared<SeqsetObj>(dir + "/seqset");
    meta = parse_metadata(read_file(dir + "/metadata/bg_info.json"));
  }
  std::shared_ptr<ReadmapObj> open_readmap(const std::string& id) {  // biograph_dir.cpp:97-130 find_readmap
    if (meta.samples.empty()) throw std::runtime_error("No samples found in BioGraph metadata.");
    std::string sha;
    if (id.empty()) {
      if (meta.samples.size() > 1) throw std::runtime_error("Multiple samples are present. You must specify a readmap or accession ID.");
      sha = meta.samples.begin()->second;
    } else if (meta.samples.count(id)) {
      sha = meta.samples[id];
    } else {
      for (auto& kv : meta.samples) if (kv.second == id) sha = id;
      if (sha.empty()) throw std::runtime_error("Sample " + id + " not found in " + dir);
    }
    return std::make_shared<ReadmapObj>(seqset, dir + "/coverage/" + sha + ".readmap");
  }
};

enum class CacheStrategy { MMAP, MMAPCACHE, RAM };  // biograph::cache_strategy
#ifndef BG_AMD_VERSION
#define BG_AMD_VERSION "biograph_amd-r5"
#endif
#ifndef BG_AMD_BUILD_REVISION
#define BG_AMD_BUILD_REVISION "unknown"
#endif

// ---- reference + assemblies
// dna_sequence as the SDK shows it (python/biograph/dna_sequence.cpp:69-158): upper-case ACGT, anything else throws
struct Seq {
  std::string s;
  Seq() {}
  explicit Seq(const std::string& in) : s(in) {
    for (char& c : s) {
      const char u = char(c & ~0x20);
      if (u != 'A' && u != 'C' && u != 'G' && u != 'T') throw std::runtime_error(std::string("Converting invalid base to dna_base: '") + c + "'");
      c = u;
    }
  }
  Seq rev_comp() const {
    Seq r;
    r.s.resize(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
      const char c = s[s.size() - 1 - i];
      r.s[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    }
    return r;
  }
};
// what the SDK's functions take where the reference converts a str to a Sequence implicitly
std::string bases_of(const py::object& o) { return py::isinstance<Seq>(o) ? o.cast<const Seq&>().s : Seq(py::str(o).cast<std::string>()).s; }
void slice_of(const py::slice& sl, size_t n, ssize_t* start, ssize_t* stop, ssize_t* len) {
  ssize_t step = 0;
  if (!sl.compute(n, reinterpret_cast<size_t*>(start), reinterpret_cast<size_t*>(stop), reinterpret_cast<size_t*>(&step), reinterpret_cast<size_t*>(len))) throw py::error_already_set();
  if (step != 1) throw py::index_error("Slicing a DNA sequence with steps other than 1 not supported");
}

struct ReferenceObj {
  bg_reference* h = nullptr;
  bg_bwt* bwt = nullptr;  // opened by the first find()
  std::string path;
  struct Super { uint32_t sc; uint64_t off, len, flat; };
  std::vector<Super> supers;  // in flat order (scaffold order, then offset: reference_assembly.cpp:146-163)
  explicit ReferenceObj(const std::string& p) : path(p) {
    check(bg_reference_open(path.c_str(), &h), "Reference");
    for (uint32_t i = 0; i < bg_reference_supercontig_count(h); ++i) {
      Super s{};
      check(bg_reference_supercontig(h, i, &s.sc, &s.off, &s.len, &s.flat), "bg_reference_supercontig");
      supers.push_back(s);
    }
    std::sort(supers.begin(), supers.end(), [](const Super& a, const Super& b) { return a.flat < b.flat; });
  }
  ~ReferenceObj() {
    if (bwt) bg_bwt_free(bwt);
    if (h) bg_reference_free(h);
  }
  std::map<std::string, uint64_t> scaffold_lens() const {
    std::map<std::string, uint64_t> m;
    for (uint32_t i = 0; i < bg_reference_scaffold_count(h); ++i) m[bg_reference_scaffold_name(h, i)] = bg_reference_scaffold_len(h, i);
    return m;
  }
  int32_t scaffold_index(const std::string& name) const {
    for (uint32_t i = 0; i < bg_reference_scaffold_count(h); ++i) if (name == bg_reference_scaffold_name(h, i)) return int32_t(i);
    throw std::runtime_error("Unknown scaffold " + name);
  }
  // the N-free extents of one scaffold: (offset inside the scaffold, bases)
  std::vector<std::pair<int64_t, std::string>> extents(const std::string& name) const {
    const int32_t want = scaffold_index(name);
    std::vector<std::pair<int64_t, std::string>> out;
    for (const Super& sc : supers) {
      if (int32_t(sc.sc) != want) continue;
      std::string s(sc.len, '\0');
      check(bg_reference_bases(h, sc.flat, sc.len, &s[0]), "bg_reference_bases");
      out.emplace_back(int64_t(sc.off), std::move(s));
    }
    return out;
  }
  const Super& super_at(uint64_t flat) const {  // reference_assembly::get_supercontig(flat position)
    size_t a = 0, b = supers.size();
    while (b - a > 1) {
      const size_t m = (a + b) / 2;
      if (supers[m].flat <= flat) a = m; else b = m;
    }
    if (supers.empty() || flat < supers[a].flat || flat > supers[a].flat + supers[a].len) throw std::runtime_error("flat position outside the reference");  // (== the end: only after the last extent)
    return supers[a];
  }
  // reference_assembly::flatten (reference_assembly.cpp:100-135); false + why when the locus holds no sequence data
  bool flatten(int32_t sc, const std::string& name, uint64_t pos, uint64_t* flat, std::string* why) const {
    if (supers.empty()) throw std::runtime_error("The reference does not appear to contain any sequence data");
    const Super* best = nullptr;  // the last extent of the scaffold that starts at or before pos
    for (const Super& s : supers) if (int32_t(s.sc) == sc && s.off <= pos && (!best || s.off > best->off)) best = &s;
    std::ostringstream o;
    if (!best) {
      o << name << ":" << pos << " comes before the first non-N sequence in the reference";
    } else if (pos - best->off > best->len) {
      o << name << ":" << pos << " is in a part of the reference (offset " << best->off << ", length " << best->len
        << ") with no sequence data (only N bases)";
    } else {
      *flat = best->flat + (pos - best->off);
      return true;
    }
    *why = o.str();
    return false;
  }
  // reference_assembly::flatten_range (reference_assembly.cpp:172-243)
  std::pair<uint64_t, uint64_t> flatten_range(const std::string& name, uint64_t start, uint64_t end, bool exact) const {
    const int32_t sc = scaffold_index(name);
    uint64_t fs = 0, fe = 0;
    std::string why_s, why_e;
    const bool ok_s = flatten(sc, name, start, &fs, &why_s), ok_e = flatten(sc, name, end, &fe, &why_e);
    if (exact && !ok_s) throw std::runtime_error(why_s);
    if (exact && !ok_e) throw std::runtime_error(why_e);
    std::ostringstream o;
    if (!ok_s && !ok_e) {
      o << "Both the start (" << name << ":" << start << ") and end (" << name << ":" << end
        << ") loci are in reference areas with no data (N bases).\n\nStart: " << why_s << "\n\nEnd: " << why_e;
      throw std::runtime_error(o.str());
    }
    if (start >= end) {
      o << "The start location \"" << name << ":" << start << "\" must come before the end location \"" << name << ":" << end
        << "\" on the reference.";
      throw std::runtime_error(o.str());
    }
    if (!ok_s) fs = super_at(fe - 1).flat;                                  // forward to the start of the end's extent
    else if (!ok_e) { const Super& s = super_at(fs); fe = s.flat + s.len; }  // back to the end of the start's extent
    if (&super_at(fs) != &super_at(fe - 1)) {
      o << "The start location \"" << name << ":" << start << "\" and end location \"" << name << ":" << end
        << "\" are not in the same contiguous region.";
      throw std::runtime_error(o.str());
    }
    return {fs, fe};
  }
};

// reference_range (python/biograph/reference.cpp:120-166): [start, end) of one extent, in flat positions
struct RefRange {
  std::shared_ptr<ReferenceObj> ref;
  uint64_t fs = 0, fe = 0;
  // reference_assembly::get_seq_position (reference_assembly.cpp:85-93).  As there, the end of a range that stops where its
  // extent stops is the first position of the NEXT extent; past the last extent it is the position after its last base.
  std::pair<uint32_t, uint64_t> seq_pos(uint64_t flat, bool) const {
    const ReferenceObj::Super& s = ref->super_at(flat);
    return {s.sc, s.off + (flat - s.flat)};
  }
  std::string scaffold() const { return bg_reference_scaffold_name(ref->h, seq_pos(fs, false).first); }
  uint64_t start() const { return seq_pos(fs, false).second; }
  uint64_t end() const { return seq_pos(fe, true).second; }
  uint64_t size() const { return fe - fs; }
  Seq sequence() const {
    Seq q;
    q.s.resize(fe - fs);
    if (fe > fs) check(bg_reference_bases(ref->h, fs, fe - fs, &q.s[0]), "bg_reference_bases");
    return q;
  }
  std::string str() const {
    std::ostringstream o;
    o << scaffold() << ":[" << start() << "-" << end() << ")";
    return o.str();
  }
};
RefRange make_range(const std::shared_ptr<ReferenceObj>& r, const std::string& name, uint64_t start, uint64_t end, bool exact) {
  const auto fr = r->flatten_range(name, start, end, exact);
  return RefRange{r, fr.first, fr.second};
}
// reference_wrapper::get_supercontig (reference.cpp:56-66): the extent that holds a position, as a range.  (The reference
// asks make_range for [offset, offset + len - 1]: the range it returns is one base short of the extent; so is this one.)
RefRange get_supercontig(const std::shared_ptr<ReferenceObj>& r, const std::string& name, uint64_t pos) {
  const int32_t sc = r->scaffold_index(name);
  uint64_t flat = 0;
  std::string why;
  if (!r->flatten(sc, name, pos, &flat, &why)) throw std::runtime_error(why);
  const ReferenceObj::Super& s = r->super_at(flat);
  return make_range(r, name, s.off, s.off + s.len - 1, false);
}
std::vector<RefRange> supercontigs(const std::shared_ptr<ReferenceObj>& r) {
  std::vector<RefRange> out;
  for (const auto& s : r->supers) out.push_back(get_supercontig(r, bg_reference_scaffold_name(r->h, s.sc), s.off));
  return out;
}
std::vector<RefRange> find_ranges(const std::shared_ptr<ReferenceObj>& r, const std::string& name, uint64_t start, uint64_t end) {
  const int32_t sc = r->scaffold_index(name);  // reference.cpp:68-88
  std::vector<RefRange> out;
  for (const auto& s : r->supers) {
    if (int32_t(s.sc) != sc) continue;
    const RefRange whole = get_supercontig(r, name, s.off);
    if (whole.start() > end || whole.end() < start) continue;
    out.push_back(make_range(r, name, std::max<uint64_t>(whole.start(), start), std::min<uint64_t>(whole.end(), end), false));
  }
  return out;
}

// bwt_wrapper (python/biograph/reference.cpp:168-185, reference.h): the sorted suffixes of the reference that start with
// the query
struct RefContext {
  std::shared_ptr<ReferenceObj> ref;
  Seq query;
  uint64_t begin = 0, end = 0;
  bool valid() const { return begin < end; }
  uint64_t matches() const { return valid() ? end - begin : 0; }
  RefRange get_match(uint64_t which) const {
    if (!valid()) throw std::runtime_error("Called get_match() on an invalid bwt object");
    if (which >= matches()) throw std::runtime_error("Called get_match() on invalid match " + std::to_string(which));
    uint64_t flat = 0;
    check(bg_bwt_match(ref->bwt, begin + which, &flat), "bg_bwt_match");
    return RefRange{ref, flat, flat + query.s.size()};
  }
  std::string str() const { return ref->path + "{" + query.s + "}"; }
};
RefContext reference_find(const std::shared_ptr<ReferenceObj>& r, const py::object& seq) {
  if (!r->bwt) check(bg_bwt_open(r->path.c_str(), &r->bwt), "Reference.find");
  RefContext c{r, Seq(bases_of(seq))};
  check(bg_bwt_find(r->bwt, c.query.s.data(), c.query.s.size(), &c.begin, &c.end), "bg_bwt_find");
  return c;
}

struct Assembly {  // variants::assembly, the fields this path touches (variants/assemble.h:285-420)
  py::object left_offset = py::none(), right_offset = py::none();  // optional_aoffset
  std::string seq;
  bool matches_reference = false;
  std::vector<std::string> tags;
  uint64_t assembly_id = 0;
};
std::string asm_repr(const Assembly& a) {
  std::ostringstream o;
  o << "Assembly(" << py::str(a.left_offset).cast<std::string>() << ", " << py::str(a.right_offset).cast<std::string>() << ", "
    << (a.seq.size() < 40 ? a.seq : a.seq.substr(0, 37) + "...") << (a.matches_reference ? ", ref" : "") << ")";
  return o.str();
}

struct AsmList {
  bg_asm_list* h = nullptr;
  AsmList() { check(bg_asm_list_create(&h), "bg_asm_list_create"); }
  ~AsmList() { if (h) bg_asm_list_free(h); }
  AsmList(const AsmList&) = delete;
  void fill(const py::iterable& input) {
    bool any = false;
    for (py::handle o : input) {
      const Assembly& a = o.cast<const Assembly&>();
      std::string tags;
      for (size_t i = 0; i < a.tags.size(); ++i) tags += (i ? "," : "") + a.tags[i];
      const int64_t lo = a.left_offset.is_none() ? BG_OFFSET_NONE : a.left_offset.cast<int64_t>();
      const int64_t ro = a.right_offset.is_none() ? BG_OFFSET_NONE : a.right_offset.cast<int64_t>();
      check(bg_asm_list_add(h, lo, ro, a.seq.data(), a.seq.size(), a.matches_reference ? 1 : 0, tags.c_str()), "bg_asm_list_add");
      any = true;
    }
    if (any) check(bg_asm_list_sort(h), "bg_asm_list_sort");
  }
  std::vector<Assembly> take() const {
    static uint64_t next_id = 1;
    std::vector<Assembly> out;
    for (uint64_t i = 0; i < bg_asm_list_size(h); ++i) {
      int64_t lo, ro;
      const char *sq, *tg;
      uint64_t ln;
      int mr;
      check(bg_asm_list_get(h, i, &lo, &ro, &sq, &ln, &mr, &tg), "bg_asm_list_get");
      Assembly a;
      if (lo != BG_OFFSET_NONE) a.left_offset = py::int_(lo);
      if (ro != BG_OFFSET_NONE) a.right_offset = py::int_(ro);
      a.seq.assign(sq, ln);
      a.matches_reference = mr != 0;
      std::string t = tg ? tg : "";
      size_t p = 0;
      while (p < t.size()) {
        size_t e = t.find(',', p);
        if (e == std::string::npos) e = t.size();
        if (e > p) a.tags.push_back(t.substr(p, e - p));
        p = e + 1;
      }
      a.assembly_id = next_id++;
      out.push_back(std::move(a));
    }
    return out;
  }
};

struct Extents {  // argument arrays of the graph entry points
  std::vector<std::pair<int64_t, std::string>> ext;
  std::vector<int64_t> off;
  std::vector<const char*> ptr;
  std::vector<uint64_t> len;
  Extents(const ReferenceObj& ref, const std::string& scaffold) : ext(ref.extents(scaffold)) {
    for (auto& e : ext) { off.push_back(e.first); ptr.push_back(e.second.c_str()); len.push_back(e.second.size()); }
  }
};

std::string join(const std::vector<std::string>& v) {
  std::string s;
  for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + v[i];
  return s;
}

// the pipeline steps are generators in the reference (define_pipeline_generator): an iterator over the step's output
py::object iterate(std::vector<Assembly> v) {
  auto holder = std::make_shared<std::vector<Assembly>>(std::move(v));
  py::list l;
  for (auto& a : *holder) l.append(py::cast(a));
  return l.attr("__iter__")();
}

}  // namespace

// The reference names its module after the interpreter (python/biograph/module.cpp:30: _capi_<major><minor>, imported by
// python/biograph/__init__.py:24 as biograph._capi_310 ...): built a second time with -DBG_CAPI_MODULE=_capi_310 the same
// source is that module, to be dropped into the reference's python/biograph/ package.
#ifndef BG_CAPI_MODULE
#define BG_CAPI_MODULE _capi
#endif
PYBIND11_MODULE(BG_CAPI_MODULE, m) {
  m.attr("__doc__") = "Internal C API calls for the BioGraph SDK surface of biograph_amd (MI355X build of the seqset path)";

  py::class_<SeqsetObj, std::shared_ptr<SeqsetObj>>(m, "Seqset", "Load a Seqset (python/biograph/seqset.cpp:46).  Raises RuntimeError if the file cannot be opened.")
      .def(py::init<const std::string&>(), py::arg("path"))
      .def("size", &SeqsetObj::size, "Total number of entries in the Seqset.")
      .def("find", [](const std::shared_ptr<SeqsetObj>& s, const py::object& seq) { return entry_or_none(seqset_find(s, py::str(seq).cast<std::string>())); },
           py::arg("seq"), "Search the Seqset for this sequence: a SeqsetEntry, or None when it is absent.")
      .def("empty_entry", [](const std::shared_ptr<SeqsetObj>& s) { return entry_or_none(Entry{s, 0, s->size(), 0}); })
      .def("get_entry_by_id", [](const std::shared_ptr<SeqsetObj>& s, py::int_ id) {
             if (PyObject_RichCompareBool(id.ptr(), py::int_(0).ptr(), Py_LT) == 1) throw py::type_error("entry id must be non-negative");
             const uint64_t i = id.cast<uint64_t>();
             if (i >= s->size()) throw std::runtime_error("Entry id out of range");
             s->need_arrays();
             return entry_or_none(Entry{s, i, i + 1, s->sizes[i]});
           }, py::arg("id"))
      .def("max_sequence_length", [](const SeqsetObj& s) { return bg_seqset_max_len(s.h); })
      .def_property_readonly("uuid", [](const SeqsetObj& s) { return s.uuid; });

  py::class_<Entry>(m, "SeqsetEntry", "A sequence contained in the Seqset, as a [start, end) range of entries.")
      .def("sequence", &entry_sequence, py::arg("bases") = INT_MAX)
      .def("__len__", [](const Entry& e) { return e.size; })
      .def("truncate", [](const Entry& e, uint32_t n) { return entry_or_none(entry_truncate(e, n)); }, py::arg("new_size"))
      .def("push_front", [](const Entry& e, const std::string& b) {
             if (b.size() != 1) throw std::runtime_error("push_front takes one base");
             return entry_or_none(entry_walk(e, b[0], false, 0));
           }, py::arg("base"))
      .def("push_front_drop", [](const Entry& e, const std::string& b, int min_ctx) {
             if (b.size() != 1) throw std::runtime_error("push_front_drop takes one base");
             return entry_or_none(entry_walk(e, b[0], true, min_ctx));
           }, py::arg("base"), py::arg("min_ctx") = 0)
      .def("pop_front", [](const Entry& e) { return entry_or_none(entry_pop_front(e)); })
      .def("get_begin_entry_id", [](const Entry& e) { return e.begin; })
      .def("get_end_entry_id", [](const Entry& e) { return e.end; })
      .def("__eq__", [](const Entry& a, const py::object& o) { return py::isinstance<Entry>(o) && a == o.cast<const Entry&>(); })
      .def("__ne__", [](const Entry& a, const py::object& o) { return !(py::isinstance<Entry>(o) && a == o.cast<const Entry&>()); })
      .def("__hash__", [](const Entry& e) { return py::hash(py::make_tuple(e.begin, e.end, e.size)); })
      .def("__str__", &entry_repr)
      .def("__repr__", &entry_repr);

  py::class_<ReadmapObj, std::shared_ptr<ReadmapObj>>(m, "Readmap", "The reads of one sample over a Seqset (python/biograph/readmap.cpp:27).")
      .def("__repr__", [](const ReadmapObj& r) { return "<Readmap " + r.path + ">"; })
      .def_property_readonly("seqset", [](const ReadmapObj& r) { return r.ss; })
      .def("max_read_len", [](const ReadmapObj& r) { return bg_readmap_max_read_len(r.h); })
      .def("min_read_len", &ReadmapObj::min_read_len)
      .def("size", &ReadmapObj::size)
      .def("get_read_by_id", [](const std::shared_ptr<ReadmapObj>& r, int64_t id) { return read_by_id(r, id); }, py::arg("id"))
      .def("get_prefix_reads", [](const std::shared_ptr<ReadmapObj>& r, const Entry& e, int min_read_len) {
             py::list l;
             for (const Read& x : prefix_reads(r, e, min_read_len)) l.append(py::cast(x));
             return l.attr("__iter__")();
           }, py::arg("entry"), py::arg("min_read_len") = 0)
      .def("get_reads_containing", [](const std::shared_ptr<ReadmapObj>& r, const Entry& e) {
             py::list l;
             for (const auto& x : reads_containing(r, e)) l.append(py::make_tuple(x.first, py::cast(x.second)));
             return l.attr("__iter__")();
           }, py::arg("entry"),
           "Reads that contain the entry's sequence somewhere inside: (bases before it in the read, ReadmapRead) (python/biograph/readmap.cpp:54-58)")
      .def("get_approx_seq_coverage", &approx_seq_coverage, py::arg("seq"), "Coverage of a sequence (python/biograph/readmap.cpp:59-62)")
      .def("get_pair_stats", &pair_stats, "The number of paired / unpaired reads / bases (python/biograph/readmap.cpp:64)")
      .def("get_read_count", [](const ReadmapObj& r) { return r.size() / 2; })
      .def("get_num_bases", [](const ReadmapObj& r) {
             uint64_t t = 0;
             for (size_t i = 0; i < r.read_len.size(); ++i) if (r.fwd[i]) t += r.read_len[i];
             return t;
           });

  py::class_<PairStats>(m, "ReadmapPairStats", "Statistics on pairing data within a readmap (python/biograph/readmap.cpp:66-78)")
      .def_readwrite("paired_reads", &PairStats::paired_reads)
      .def_readwrite("paired_bases", &PairStats::paired_bases)
      .def_readwrite("unpaired_reads", &PairStats::unpaired_reads)
      .def_readwrite("unpaired_bases", &PairStats::unpaired_bases);

  py::class_<Read>(m, "ReadmapRead", "One read of a Readmap.")
      .def("__len__", [](const Read& r) { return r.rm->read_len[r.id]; })
      .def("__eq__", [](const Read& a, const py::object& o) { return py::isinstance<Read>(o) && a == o.cast<const Read&>(); })
      .def("__ne__", [](const Read& a, const py::object& o) { return !(py::isinstance<Read>(o) && a == o.cast<const Read&>()); })
      .def("__hash__", [](const Read& r) { return py::hash(py::int_(r.id)); })
      .def("__repr__", &read_repr)
      .def("__str__", &read_repr)
      .def("has_mate", &read_has_mate)
      .def("get_mate", &read_mate)
      .def("get_rev_comp", &read_rev_comp)
      .def("is_original_orientation", [](const Read& r) { return r.rm->fwd[r.id] != 0; })
      .def("get_seqset_entry", &read_entry)
      .def("get_read_id", [](const Read& r) { return r.id; });

  py::class_<Metadata>(m, "Metadata")
      .def_readonly("accession_id", &Metadata::accession_id)
      .def_readonly("biograph_id", &Metadata::biograph_id)
      .def_readonly("version", &Metadata::version)
      .def_readonly("samples", &Metadata::samples);

  py::enum_<CacheStrategy>(m, "CacheStrategy", "How the reference library keeps a BioGraph's files (python/biograph/biograph.cpp:9-37).  Accepted for "
                           "compatibility: this build reads the files once and keeps the tables in HBM whatever the mode.")
      .value("MMAP", CacheStrategy::MMAP)
      .value("MMAPCACHE", CacheStrategy::MMAPCACHE)
      .value("RAM", CacheStrategy::RAM);

  py::class_<BioGraphObj, std::shared_ptr<BioGraphObj>>(m, "BioGraph", "A BioGraph directory: Seqset + Readmaps (python/biograph/biograph.cpp:39).")
      .def(py::init<const std::string&>(), py::arg("dirname"))
      .def(py::init([](const std::string& d, CacheStrategy) { return std::make_shared<BioGraphObj>(d); }), py::arg("dirname"), py::arg("mode"))
      .def_property_readonly("seqset", [](const BioGraphObj& b) { return b.seqset; })
      .def("open_readmap", &BioGraphObj::open_readmap, py::arg("accession_id") = "")
      .def_property_readonly("metadata", [](const BioGraphObj& b) { return b.meta; });

  py::class_<Seq>(m, "Sequence", "A DNA sequence, compared and sliced like a string (python/biograph/dna_sequence.cpp:69).")
      .def(py::init<const std::string&>())
      .def(py::init<>())
      .def("__str__", [](const Seq& q) { return q.s; })
      .def("__repr__", [](const Seq& q) { return "biograph.Sequence('" + q.s + "')"; })
      .def("__len__", [](const Seq& q) { return q.s.size(); })
      .def("rev_comp", &Seq::rev_comp, "A new Sequence: the reverse complement of this one.")
      .def("__getitem__", [](const Seq& q, const py::slice& sl) {
        ssize_t start, stop, len;
        slice_of(sl, q.s.size(), &start, &stop, &len);
        Seq r;
        r.s = q.s.substr(size_t(start), size_t(len));
        return r;
      })
      .def("__getitem__", [](const Seq& q, int pos) {
        if (pos < 0) pos += int(q.s.size());
        if (pos < 0 || pos >= int(q.s.size())) throw py::index_error();
        Seq r;
        r.s = q.s.substr(size_t(pos), 1);
        return r;
      })
      .def("__setitem__", [](Seq& q, const py::slice& sl, const py::object& with) {
        ssize_t start, stop, len;
        slice_of(sl, q.s.size(), &start, &stop, &len);
        q.s = q.s.substr(0, size_t(start)) + bases_of(with) + q.s.substr(size_t(stop));
      })
      .def("__setitem__", [](Seq& q, int pos, char with) {
        if (pos < 0) pos += int(q.s.size());
        if (pos < 0 || pos >= int(q.s.size())) throw py::index_error();
        q.s[size_t(pos)] = Seq(std::string(1, with)).s[0];
      })
      .def("__lt__", [](const Seq& a, const Seq& b) { return a.s < b.s; })
      .def("__le__", [](const Seq& a, const Seq& b) { return a.s <= b.s; })
      .def("__gt__", [](const Seq& a, const Seq& b) { return a.s > b.s; })
      .def("__ge__", [](const Seq& a, const Seq& b) { return a.s >= b.s; })
      .def("__eq__", [](const Seq& a, const py::object& o) { return (py::isinstance<Seq>(o) || py::isinstance<py::str>(o)) && a.s == bases_of(o); })
      .def("__ne__", [](const Seq& a, const py::object& o) { return !((py::isinstance<Seq>(o) || py::isinstance<py::str>(o)) && a.s == bases_of(o)); })
      .def("__iadd__", [](Seq& a, const py::object& o) { a.s += bases_of(o); return a; })
      .def("__add__", [](const Seq& a, const py::object& o) { Seq r; r.s = a.s + bases_of(o); return r; })
      .def("__radd__", [](const Seq& a, const std::string& str) { Seq r; r.s = Seq(str).s + a.s; return r; })
      .def("__hash__", [](const Seq& q) { return std::hash<std::string>{}(q.s); });
  py::implicitly_convertible<std::string, Seq>();

  py::class_<ReferenceObj, std::shared_ptr<ReferenceObj>>(m, "Reference", "A reference directory, .ref file or FASTA (python/biograph/reference.cpp:190).")
      .def(py::init<const std::string&>(), py::arg("path"))
      .def("__str__", [](const ReferenceObj& r) { return r.path; })
      .def("__repr__", [](const ReferenceObj& r) { return "<biograph.Reference " + r.path + ">"; })
      .def("make_range", &make_range, py::arg("scaffold_name"), py::arg("start"), py::arg("end"), py::arg("use_exact_loci") = true,
           "A ReferenceRange for 0-based [start, end) of a scaffold; with use_exact_loci=False a start or end inside an N region "
           "moves to the edge of the sequence data.  RuntimeError if the two are not in one contiguous region.")
      .def_property_readonly("size", [](const ReferenceObj& r) { return bg_reference_size(r.h); }, "The number of bases of the reference that are not N.")
      .def_property_readonly("scaffolds", [](const ReferenceObj& r) {
        std::vector<std::string> v;
        for (const auto& kv : r.scaffold_lens()) v.push_back(kv.first);  // (a std::set ordered by name in the reference too)
        return v;
      })
      .def_property_readonly("chromosomes", [](const ReferenceObj& r) {
        std::vector<std::string> v;
        for (const auto& kv : r.scaffold_lens()) v.push_back(kv.first);
        return v;
      })
      .def_property_readonly("supercontigs", &supercontigs, "The contiguous (N-free) regions of the reference, in order, as ReferenceRanges.")
      .def_property_readonly("scaffold_lens", &ReferenceObj::scaffold_lens)
      .def("get_supercontig", &get_supercontig, py::arg("scaffold_name"), py::arg("pos"))
      .def("find_ranges", &find_ranges, py::arg("scaffold_name"), py::arg("start"), py::arg("end"))
      .def("find", &reference_find, py::arg("seq"), "The ReferenceContext of a sequence: where it occurs in the reference (needs reference.bwt).");

  py::class_<RefRange>(m, "ReferenceRange", "A region of one contiguous stretch of the reference, 0-based (python/biograph/reference.cpp:282).")
      .def("__str__", &RefRange::str)
      .def("__repr__", [](const RefRange& r) { return "<biograph.ReferenceRange " + r.str() + ">"; })
      .def_property_readonly("scaffold", &RefRange::scaffold)
      .def_property_readonly("chromosome", &RefRange::scaffold)
      .def_property_readonly("start", &RefRange::start)
      .def_property_readonly("end", &RefRange::end)
      .def_property_readonly("sequence", &RefRange::sequence)
      .def_property_readonly("size", &RefRange::size);

  py::class_<RefContext>(m, "ReferenceContext", "Where a sequence occurs in the reference (python/biograph/reference.cpp:324).")
      .def("__str__", &RefContext::str)
      .def("__repr__", [](const RefContext& c) { return "<biograph.ReferenceContext " + c.str() + ">"; })
      .def_property_readonly("matches", &RefContext::matches)
      .def_property_readonly("valid", &RefContext::valid)
      .def_property_readonly("start", [](const RefContext& c) { return c.begin; })
      .def_property_readonly("end", [](const RefContext& c) { return c.end; })
      .def("get_match", &RefContext::get_match, py::arg("which"))
      .def_property_readonly("query", [](const RefContext& c) { return c.query; });

  m.def("version", []() { return std::string(BG_AMD_VERSION); }, "BioGraph SDK version.");
  m.def("build_revision", []() { return std::string(BG_AMD_BUILD_REVISION); }, "BioGraph build revision.");

  py::class_<Assembly>(m, "Assembly")
      .def(py::init([](py::object lo, py::object ro, std::string seq, bool mr, std::vector<std::string> tags) {
             Assembly a;
             a.left_offset = std::move(lo);
             a.right_offset = std::move(ro);
             a.seq = std::move(seq);
             a.matches_reference = mr;
             a.tags = std::move(tags);
             return a;
           }), py::arg("left_offset") = py::none(), py::arg("right_offset") = py::none(), py::arg("seq") = "",
           py::arg("matches_reference") = false, py::arg("tags") = std::vector<std::string>())
      .def_readwrite("left_offset", &Assembly::left_offset)
      .def_readwrite("right_offset", &Assembly::right_offset)
      .def_readwrite("seq", &Assembly::seq)
      .def_readwrite("matches_reference", &Assembly::matches_reference)
      .def_readwrite("tags", &Assembly::tags)
      .def_readonly("assembly_id", &Assembly::assembly_id)
      .def("__repr__", &asm_repr);

  m.def("make_ref_assemblies", [](const ReferenceObj& ref, const std::string& scaffold, int64_t start, int64_t limit, int64_t max_chunk) {
          Extents x(ref, scaffold);
          AsmList l;
          check(bg_make_ref_assemblies(l.h, x.off.size(), x.off.data(), x.ptr.data(), x.len.data(), start, limit, max_chunk), "make_ref_assemblies");
          return l.take();
        }, py::arg("ref"), py::arg("scaffold_name"), py::arg("start_offset"), py::arg("limit_offset"), py::arg("max_chunk_size") = 100);
  m.def("graph_trim_ref", [](const py::iterable& input, const ReferenceObj& ref, const std::string& scaffold, bool rev_comp) {
          if (rev_comp) throw std::runtime_error("graph_trim_ref: rev_comp is not supported");
          Extents x(ref, scaffold);
          AsmList l;
          l.fill(input);
          check(bg_graph_trim_ref(l.h, x.off.size(), x.off.data(), x.ptr.data(), x.len.data(), 0), "graph_trim_ref");
          return iterate(l.take());
        }, py::arg("input"), py::arg("ref"), py::arg("scaffold_name"), py::arg("rev_comp") = false);
  m.def("update_rc_seqset_entries", [](const py::iterable& input, const std::shared_ptr<SeqsetObj>& ss, bool) {
          AsmList l;
          l.fill(input);
          check(bg_update_rc_seqset_entries(ss->c->h, ss->h, nullptr, l.h, 20), "update_rc_seqset_entries");
          return iterate(l.take());
        }, py::arg("input"), py::arg("seqset"), py::arg("enable_self_test") = false);
  // NB the rc seqset entries live on the C++ side of a step and are rebuilt by the step that needs them: the
  // discover steps below run update_rc_seqset_entries on their input first, which is what the reference's pipeline
  // does right before them (graph_discover_test.py:36-39)
  m.def("discover_branch", [](const py::iterable& input, const std::shared_ptr<ReadmapObj>& rm, int min_overlap, const std::string& tag,
                              const std::vector<std::string>& discover_tags) {
          AsmList l;
          l.fill(input);
          check(bg_update_rc_seqset_entries(rm->c->h, rm->ss->h, nullptr, l.h, 20), "update_rc_seqset_entries");
          check(bg_discover_branch(rm->c->h, rm->ss->h, rm->h, l.h, uint32_t(min_overlap), tag.c_str(), join(discover_tags).c_str()), "discover_branch");
          return iterate(l.take());
        }, py::arg("input"), py::arg("readmap"), py::arg("min_overlap") = 70, py::arg("tag") = "GRAPH_BRANCH",
        py::arg("discover_tags") = std::vector<std::string>());
  m.def("discover_push_to_pair", [](const py::iterable& input, const std::shared_ptr<ReadmapObj>& rm, int min_overlap, int max_pair_distance,
                                    const std::string& tag, const std::vector<std::string>& discover_tags) {
          AsmList l;
          l.fill(input);
          check(bg_update_rc_seqset_entries(rm->c->h, rm->ss->h, nullptr, l.h, 20), "update_rc_seqset_entries");
          check(bg_discover_push_to_pair(rm->c->h, rm->ss->h, rm->h, l.h, uint32_t(min_overlap), max_pair_distance, tag.c_str(),
                                         join(discover_tags).c_str()), "discover_push_to_pair");
          return iterate(l.take());
        }, py::arg("input"), py::arg("readmap"), py::arg("min_overlap") = 70, py::arg("max_pair_distance") = 1000,
        py::arg("tag") = "GRAPH_PUSH_TO_PAIR", py::arg("discover_tags") = std::vector<std::string>());
}
