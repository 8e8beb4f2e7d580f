// refbuild.cpp — `biograph reference` (modules/main/make_ref.cpp -> pipeline/build_reference_task.cpp), host part:
//   IMPORT_FASTA  fasta_ref_importer (bio_format/fasta_ref_importer.cpp:30-167): scaffolds are cut into supercontigs at
//                 runs of >= min_n_run N (shorter runs are filled with the base that follows them, :87-93; every
//                 IUPAC code but ACGTU counts as N, :82), written as reference.fasta (">chr:offset", 80 bases per line,
//                 ordered by (scaffold index, offset)) and described in karyotype.json;
//   MAKE_FLAT     flat_ref_builder (bio_base/flat_ref.cpp:224-360) over reference.fasta: "SPR000", the bases 2 bits each
//                 (first base in the high bits), the msgpack index {1: scaffolds [{1: name, 2: md5 of the record's
//                 sequence lines, 3: size}], 2: extents [{1: scaffold, 2: offset, 3: size, 4: flat}], 3: fixups []} and
//                 its length as a big-endian u32.  reference.fasta holds ACGT only, so there are never fixups.
//   MAKE_BWT      on the GPU: csrc/bwt.hip.
// Host code only.  The written files are byte-identical to the reference's for the golden lambda reference
// (tests/test_refbuild.py).
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "bg_internal.h"

namespace {

using bg::set_error;

// ---- MD5 (RFC 1321) ---------------------------------------------------------------------------------------
struct Md5 {
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
  uint64_t n = 0;
  unsigned char buf[64];
  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const unsigned char* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
        0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
        0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
        0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
        0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
        0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t w[16];
    for (int i = 0; i < 16; ++i) w[i] = uint32_t(p[4 * i]) | uint32_t(p[4 * i + 1]) << 8 | uint32_t(p[4 * i + 2]) << 16 | uint32_t(p[4 * i + 3]) << 24;
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) { f = (B & C) | (~B & D); g = i; }
      else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { f = C ^ (B | ~D); g = (7 * i) & 15; }
      const uint32_t t = D;
      D = C;
      C = B;
      B = B + rol(A + f + K[i] + w[g], S[i]);
      A = t;
    }
    a += A; b += B; c += C; d += D;
  }
  void write(const char* p, size_t len) {
    for (size_t i = 0; i < len; ++i) {
      buf[n++ & 63] = static_cast<unsigned char>(p[i]);
      if ((n & 63) == 0) block(buf);
    }
  }
  std::string hex() {
    const uint64_t bits = n * 8;
    const char one = char(0x80), zero = 0;
    write(&one, 1);
    while ((n & 63) != 56) write(&zero, 1);
    char len[8];
    for (int i = 0; i < 8; ++i) len[i] = char(bits >> (8 * i));
    write(len, 8);
    char out[33];
    const uint32_t v[4] = {a, b, c, d};
    for (int i = 0; i < 16; ++i) snprintf(out + 2 * i, 3, "%02x", unsigned((v[i >> 2] >> (8 * (i & 3))) & 0xff));
    return std::string(out, 32);
  }
};

// ---- msgpack writer: the encodings msgpack-c 0.5 picks (fixraw / raw16 for strings, smallest unsigned) ------------
struct MpOut {
  std::string s;
  void byte(unsigned v) { s.push_back(char(v)); }
  void be(uint64_t v, int bytes) { for (int i = bytes - 1; i >= 0; --i) byte(unsigned(v >> (8 * i)) & 0xff); }
  void uint(uint64_t v) {
    if (v < 128) byte(unsigned(v));
    else if (v < 256) { byte(0xcc); be(v, 1); }
    else if (v < 65536) { byte(0xcd); be(v, 2); }
    else if (v < (1ull << 32)) { byte(0xce); be(v, 4); }
    else { byte(0xcf); be(v, 8); }
  }
  void map(uint32_t n) { if (n < 16) byte(0x80 | n); else if (n < 65536) { byte(0xde); be(n, 2); } else { byte(0xdf); be(n, 4); } }
  void array(uint64_t n) { if (n < 16) byte(0x90 | unsigned(n)); else if (n < 65536) { byte(0xdc); be(n, 2); } else { byte(0xdd); be(n, 4); } }
  void raw(const std::string& v) {
    if (v.size() < 32) byte(0xa0 | unsigned(v.size()));
    else if (v.size() < 65536) { byte(0xda); be(v.size(), 2); }
    else { byte(0xdb); be(v.size(), 4); }
    s.append(v);
  }
};

std::string json_escape(const std::string& v) {
  std::string o = "\"";
  for (char c : v) {
    if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(c); }
    else if (static_cast<unsigned char>(c) < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", unsigned(c)); o += b; }
    else o.push_back(c);
  }
  return o + "\"";
}

bool write_file(const std::string& path, const std::string& data) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
  return fclose(f) == 0 && ok;
}

struct Datum { uint64_t index; std::string name, seq; uint64_t start; };

}  // namespace

// source_fasta: plain FASTA text (the caller has inflated .gz).  Writes refdir/reference.fasta, karyotype.json and
// reference.ref.  refdir must exist.
extern "C" int bg_reference_build(const char* source_fasta, const char* refdir_c, uint64_t min_n_run) {
  if (!source_fasta || !refdir_c) { set_error("bg_reference_build: null argument"); return BG_EINVAL; }
  const std::string refdir = refdir_c;
  FILE* f = fopen(source_fasta, "rb");
  if (!f) { set_error("%s: %s", source_fasta, strerror(errno)); return BG_EIO; }
  // ---- fasta_ref_importer::run
  std::vector<std::string> scaffold_order;
  std::map<std::string, std::pair<uint64_t, int>> scaffolds;  // name -> (len, index): std::set<scaffold> ordered by name
  std::vector<Datum> data;
  std::string name, seq;
  uint64_t position = 0, start = 0, end = 0;
  bool had_contig = false;
  int rc = BG_OK;
  auto finish_supercontig = [&]() {
    if (seq.empty()) { start = end = position; return; }
    had_contig = true;
    const uint64_t idx = uint64_t(std::find(scaffold_order.begin(), scaffold_order.end(), name) - scaffold_order.begin());
    data.push_back(Datum{idx, name, seq, start});
    start = end = position;
    seq.clear();
  };
  auto finish_scaffold = [&]() {
    if (!name.empty()) {
      finish_supercontig();
      if (had_contig) {
        int index = -1;
        for (size_t i = 0; i < scaffold_order.size(); ++i)
          if (scaffold_order[i] == name) { index = int(i); break; }
        if (index < 0) { index = int(scaffold_order.size()); scaffold_order.push_back(name); }
        scaffolds.emplace(name, std::make_pair(position, index));  // std::set::insert: the first one of a name stays
      }
    }
    position = start = end = 0;
    had_contig = false;
  };
  static const char translate[] = "ANCN..GN..N.NN...NNTTNNNN.";
  std::string line;
  bool any_record = false;
  int ch;
  auto handle_line = [&]() {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) return;
    if (line[0] == '>') {
      finish_scaffold();
      size_t e = 1;
      while (e < line.size() && !isspace(static_cast<unsigned char>(line[e]))) ++e;
      if (e == 1) { set_error("Illegal fasta entry name '%s'", line.c_str()); rc = BG_EINVAL; return; }
      name = line.substr(1, e - 1);
      any_record = true;
      return;
    }
    for (char c0 : line) {
      if (!isalpha(static_cast<unsigned char>(c0))) { set_error("Non-alpha fasta: ascii value of %d", int(c0)); rc = BG_EINVAL; return; }
      char c = translate[toupper(c0) - 'A'];
      if (c == '.') { set_error("Invalid base in fasta: ascii value of %d", int(c0)); rc = BG_EINVAL; return; }
      if (c != 'N') {
        if (end != position) {
          if (position - end < min_n_run) { while (end != position) { seq.push_back(c); ++end; } }
          else finish_supercontig();
        }
        seq.push_back(c);
        ++end;
      }
      ++position;
    }
  };
  while (rc == BG_OK && (ch = fgetc(f)) != EOF) {
    if (ch == '\n') { handle_line(); line.clear(); }
    else line.push_back(char(ch));
  }
  if (rc == BG_OK && !line.empty()) handle_line();
  fclose(f);
  if (rc != BG_OK) return rc;
  if (!any_record) { set_error("%s: no FASTA records", source_fasta); return BG_EINVAL; }
  finish_scaffold();
  // the names of supercontigs registered before their scaffold got its index (the first supercontig of a new scaffold is
  // stored while scaffold_order does not hold the name yet: std::find gives order.size(), the index it is about to get)
  // ---- karyotype.json (json_serialize(kt_compat): keys in alphabetical order, no white space)
  struct Sc { std::string chr; uint64_t offset, len; };
  std::vector<Sc> scs;
  for (const Datum& d : data) scs.push_back(Sc{d.name, d.start, d.seq.size()});
  std::sort(scs.begin(), scs.end(), [](const Sc& a, const Sc& b) { return a.chr != b.chr ? a.chr < b.chr : a.offset < b.offset; });
  scs.erase(std::unique(scs.begin(), scs.end(), [](const Sc& a, const Sc& b) { return a.chr == b.chr && a.offset == b.offset; }), scs.end());
  std::string js = "{\"chr_order\":[";
  for (size_t i = 0; i < scaffold_order.size(); ++i) js += (i ? "," : "") + json_escape(scaffold_order[i]);
  js += "],\"chromosomes\":[";
  {
    bool first = true;
    for (const auto& kv : scaffolds) {
      js += std::string(first ? "" : ",") + "{\"index\":" + std::to_string(kv.second.second) + ",\"len\":" + std::to_string(kv.second.first) +
            ",\"name\":" + json_escape(kv.first) + "}";
      first = false;
    }
  }
  js += "],\"supercontigs\":[";
  for (size_t i = 0; i < scs.size(); ++i)
    js += std::string(i ? "," : "") + "{\"chr\":" + json_escape(scs[i].chr) + ",\"len\":" + std::to_string(scs[i].len) + ",\"name\":" +
          json_escape(scs[i].chr + ":" + std::to_string(int(scs[i].offset))) + ",\"offset\":" + std::to_string(scs[i].offset) + "}";
  js += "]}";
  if (!write_file(refdir + "/karyotype.json", js)) { set_error("%s/karyotype.json: %s", refdir_c, strerror(errno)); return BG_EIO; }
  // ---- reference.fasta (write_supercontigs) and, from its text, reference.ref (flat_ref_builder)
  std::stable_sort(data.begin(), data.end(), [](const Datum& a, const Datum& b) { return a.index != b.index ? a.index < b.index : a.start < b.start; });
  std::string fa, flat = "SPR000";
  MpOut scaf_mp, ext_mp;
  uint64_t flat_offset = 0;
  unsigned cur_byte = 0, cur_count = 0;
  for (size_t i = 0; i < data.size(); ++i) {
    const Datum& d = data[i];
    const std::string rec_name = d.name + ":" + std::to_string(long(d.start));
    fa += ">" + rec_name;
    Md5 md5;
    for (size_t j = 0; j < d.seq.size(); ++j) {
      if (j % 80 == 0) fa.push_back('\n');
      fa.push_back(d.seq[j]);
      md5.write(&d.seq[j], 1);
      const char c = d.seq[j];
      cur_byte = (cur_byte << 2) | unsigned(c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3);
      if (++cur_count == 4) { flat.push_back(char(cur_byte)); cur_byte = 0; cur_count = 0; }
    }
    fa.push_back('\n');
    scaf_mp.map(3);
    scaf_mp.uint(1); scaf_mp.raw(rec_name);
    scaf_mp.uint(2); scaf_mp.raw(md5.hex());
    scaf_mp.uint(3); scaf_mp.uint(d.seq.size());
    ext_mp.map(4);
    ext_mp.uint(1); ext_mp.uint(i);
    ext_mp.uint(2); ext_mp.uint(0);
    ext_mp.uint(3); ext_mp.uint(d.seq.size());
    ext_mp.uint(4); ext_mp.uint(flat_offset);
    flat_offset += d.seq.size();
  }
  while (cur_count > 0) {  // pad the last byte with A (flat_ref.cpp:259-261)
    cur_byte <<= 2;
    if (++cur_count == 4) { flat.push_back(char(cur_byte)); cur_byte = 0; cur_count = 0; }
  }
  MpOut idx;
  idx.map(3);
  idx.uint(1); idx.array(data.size()); idx.s += scaf_mp.s;
  idx.uint(2); idx.array(data.size()); idx.s += ext_mp.s;
  idx.uint(3); idx.array(0);
  flat += idx.s;
  for (int i = 3; i >= 0; --i) flat.push_back(char((uint32_t(idx.s.size()) >> (8 * i)) & 0xff));
  if (!write_file(refdir + "/reference.fasta", fa)) { set_error("%s/reference.fasta: %s", refdir_c, strerror(errno)); return BG_EIO; }
  if (!write_file(refdir + "/reference.ref", flat)) { set_error("%s/reference.ref: %s", refdir_c, strerror(errno)); return BG_EIO; }
  return BG_OK;
}
