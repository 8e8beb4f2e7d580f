// refdir.cpp — reader of the reference the discovery walks along: a BioGraph reference
// directory (`karyotype.json` + `reference.ref`, falling back to `reference.fasta`; modules/io/defaults.h:17-50),
// a bare `.ref` file or a FASTA.  Host code only (no device).
//
// reference.ref (bio_base/flat_ref.cpp:140-165 build_ref_from_spec, flat_ref.h:100-172): magic "SPR000", the
// flat genome 2 bits per base (first base in the high bits of a byte, A C G T = 0..3; N-free extents back to
// back), a msgpack index {1: scaffolds [{1: name, 2: md5, 3: size}], 2: extents [{1: scaffold, 2: offset,
// 3: size, 4: flat}], 3: fixups [{1: extent, 2: base, 3: original char}]} (TRANSFER_OBJECT = map keyed by field
// number) and its length as a big-endian u32.  The scaffolds of the index are the FASTA records the file was
// built from — in a reference directory those are the supercontigs "chr:N".
// karyotype.json (bio_base/reference.cpp:27-40, reference_assembly): chromosomes [{index, len, name}],
// supercontigs [{chr, len, name, offset}] = the N-free stretches of every chromosome, flat order = order of
// (chromosome index, offset).
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "bg_internal.h"

namespace {

using bg::set_error;

struct Chromosome { std::string name; uint64_t len; };
struct Supercontig { uint32_t chr; uint64_t offset, len, flat; };

bool read_all(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}
bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool is_dir(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }

// ---- minimal msgpack reader (the types msgpack_transfer.h emits for this index) ----------------------
struct Mp {
  const unsigned char* p;
  const unsigned char* e;
  bool ok = true;
  uint64_t be(int n) {
    if (e - p < n) { ok = false; return 0; }
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | *p++;
    return v;
  }
  bool uint(uint64_t* v) {
    if (p >= e) return ok = false;
    unsigned c = *p++;
    if (c < 0x80) { *v = c; return true; }
    if (c == 0xcc) { *v = be(1); return ok; }
    if (c == 0xcd) { *v = be(2); return ok; }
    if (c == 0xce) { *v = be(4); return ok; }
    if (c == 0xcf) { *v = be(8); return ok; }
    if (c == 0xd0) { *v = uint64_t(int8_t(be(1))); return ok; }
    if (c == 0xd1) { *v = uint64_t(int16_t(be(2))); return ok; }
    if (c == 0xd2) { *v = uint64_t(int32_t(be(4))); return ok; }
    if (c == 0xd3) { *v = be(8); return ok; }
    if (c >= 0xe0) { *v = uint64_t(int8_t(c)); return true; }
    return ok = false;
  }
  bool str(std::string* s) {
    if (p >= e) return ok = false;
    unsigned c = *p++;
    uint64_t n;
    if ((c & 0xe0) == 0xa0) n = c & 0x1f;
    else if (c == 0xd9 || c == 0xc4) n = be(1);
    else if (c == 0xda || c == 0xc5) n = be(2);
    else if (c == 0xdb || c == 0xc6) n = be(4);
    else return ok = false;
    if (!ok || uint64_t(e - p) < n) return ok = false;
    s->assign(reinterpret_cast<const char*>(p), n);
    p += n;
    return true;
  }
  bool container(unsigned fix_mask, unsigned c16, uint64_t* n) {  // fix_mask 0x90 array, 0x80 map
    if (p >= e) return ok = false;
    unsigned c = *p++;
    if ((c & 0xf0) == fix_mask) { *n = c & 0x0f; return true; }
    if (c == c16) { *n = be(2); return ok; }
    if (c == c16 + 1) { *n = be(4); return ok; }
    return ok = false;
  }
  bool array(uint64_t* n) { return container(0x90, 0xdc, n); }
  bool map(uint64_t* n) { return container(0x80, 0xde, n); }
  bool skip() {  // any value
    if (p >= e) return ok = false;
    unsigned c = *p;
    uint64_t n, v;
    std::string s;
    if (c < 0x80 || c >= 0xe0 || (c >= 0xcc && c <= 0xd3)) return uint(&v);
    if ((c & 0xe0) == 0xa0 || c == 0xd9 || c == 0xda || c == 0xdb || c == 0xc4 || c == 0xc5 || c == 0xc6) return str(&s);
    if ((c & 0xf0) == 0x90 || c == 0xdc || c == 0xdd) {
      if (!array(&n)) return false;
      for (uint64_t i = 0; i < n; ++i) if (!skip()) return false;
      return true;
    }
    if ((c & 0xf0) == 0x80 || c == 0xde || c == 0xdf) {
      if (!map(&n)) return false;
      for (uint64_t i = 0; i < 2 * n; ++i) if (!skip()) return false;
      return true;
    }
    if (c == 0xc0 || c == 0xc2 || c == 0xc3) { ++p; return true; }
    if (c == 0xca) { ++p; be(4); return ok; }
    if (c == 0xcb) { ++p; be(8); return ok; }
    return ok = false;
  }
};

// ---- minimal JSON reader (objects, arrays, strings, non-negative integers) ----------------------------
struct Js {
  const char* p;
  const char* e;
  bool ok = true;
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
  bool str(std::string* s) {
    ws();
    if (p >= e || *p != '"') return ok = false;
    ++p;
    s->clear();
    while (p < e && *p != '"') {
      if (*p == '\\' && p + 1 < e) ++p;
      s->push_back(*p++);
    }
    if (p >= e) return ok = false;
    ++p;
    return true;
  }
  bool num(uint64_t* v) {
    ws();
    if (p >= e || *p < '0' || *p > '9') return ok = false;
    *v = 0;
    while (p < e && *p >= '0' && *p <= '9') *v = *v * 10 + uint64_t(*p++ - '0');
    return true;
  }
  bool skip() {
    ws();
    if (p >= e) return ok = false;
    std::string s;
    uint64_t v;
    if (*p == '"') return str(&s);
    if (*p == '{' || *p == '[') {
      const char close = *p == '{' ? '}' : ']';
      const bool obj = *p == '{';
      ++p;
      if (eat(close)) return true;
      do {
        if (obj && (!str(&s) || !eat(':'))) return ok = false;
        if (!skip()) return false;
      } while (eat(','));
      return eat(close) ? true : (ok = false);
    }
    if (*p >= '0' && *p <= '9') return num(&v);
    while (p < e && ((*p >= 'a' && *p <= 'z') || *p == '-' || *p == '.' || (*p >= '0' && *p <= '9'))) ++p;  // true/false/null/floats
    return true;
  }
};

}  // namespace

struct bg_reference {
  std::vector<Chromosome> chromosomes;   // in chr_order
  std::vector<Supercontig> supercontigs; // flat order
  std::string bases;                     // flat genome, one ASCII base per position (ACGT)
  std::string source;                    // "ref" | "fasta"
};

namespace {

// 'A' 'C' 'G' 'T' for those letters in either case, 0 for every other byte
static const struct UpperAcgt {
  unsigned char t[256];
  UpperAcgt() {
    for (int i = 0; i < 256; ++i) t[i] = 0;
    for (const char* p = "ACGT"; *p; ++p) { t[uint8_t(*p)] = uint8_t(*p); t[uint8_t(*p + 32)] = uint8_t(*p); }
  }
  unsigned char operator[](uint8_t b) const { return t[b]; }
} kUpperAcgt;

int load_fasta(const std::string& path, bg_reference* r) {
  std::string txt;
  if (!read_all(path, &txt)) { set_error("Could not open %s: %s", path.c_str(), strerror(errno)); return BG_EIO; }
  r->bases.reserve(r->bases.size() + txt.size());
  size_t i = 0;
  while (i < txt.size()) {
    size_t eol = txt.find('\n', i);
    if (eol == std::string::npos) eol = txt.size();
    if (txt[i] == '>') {
      size_t a = i + 1;
      while (a < eol && (txt[a] == ' ' || txt[a] == '\t')) ++a;
      size_t b = a;
      while (b < eol && txt[b] != ' ' && txt[b] != '\t' && txt[b] != '\r') ++b;
      r->chromosomes.push_back(Chromosome{txt.substr(a, b - a), 0});
    } else if (eol > i) {
      if (r->chromosomes.empty()) r->chromosomes.push_back(Chromosome{"ref", 0});
      Chromosome& c = r->chromosomes.back();
      const uint32_t ci = uint32_t(r->chromosomes.size() - 1);
      // runs of ACGT (either case) go over whole, upper-cased through a table; supercontigs = maximal N-free stretches
      size_t q = i;
      while (q < eol) {
        size_t run = q;
        while (run < eol && kUpperAcgt[uint8_t(txt[run])]) ++run;
        if (run > q) {
          if (r->supercontigs.empty() || r->supercontigs.back().chr != ci ||
              r->supercontigs.back().offset + r->supercontigs.back().len != c.len)
            r->supercontigs.push_back(Supercontig{ci, c.len, 0, uint64_t(r->bases.size())});
          const size_t at = r->bases.size();
          r->bases.resize(at + (run - q));
          for (size_t k = q; k < run; ++k) r->bases[at + (k - q)] = char(kUpperAcgt[uint8_t(txt[k])]);
          r->supercontigs.back().len += run - q;
          c.len += run - q;
          q = run;
        }
        while (q < eol && !kUpperAcgt[uint8_t(txt[q])]) {
          if (txt[q] != '\r') ++c.len;
          ++q;
        }
      }
    }
    i = eol + 1;
  }
  r->source = "fasta";
  return BG_OK;
}

// decoded .ref: names/sizes of its scaffolds, extents, bases
struct SpecRef {
  std::vector<std::pair<std::string, uint64_t>> scaffolds;
  struct Extent { uint64_t scaffold, offset, size, flat; };
  std::vector<Extent> extents;
  std::string bases;
};

int load_spec(const std::string& path, SpecRef* sr) {
  std::string d;
  if (!read_all(path, &d)) { set_error("Could not open %s: %s", path.c_str(), strerror(errno)); return BG_EIO; }
  if (d.size() < 10 || d.compare(0, 6, "SPR000") != 0) {  // flat_ref.cpp:143-149
    set_error("%s is neither a Spec Reference nor a FASTA file", path.c_str());
    return BG_EIO;
  }
  const unsigned char* u = reinterpret_cast<const unsigned char*>(d.data());
  const uint64_t mlen = (uint64_t(u[d.size() - 4]) << 24) | (uint64_t(u[d.size() - 3]) << 16) | (uint64_t(u[d.size() - 2]) << 8) |
                        u[d.size() - 1];
  if (d.size() < 10 + mlen) { set_error("%s: flat_ref reference header size too big", path.c_str()); return BG_EIO; }
  const unsigned char* idx = u + d.size() - 4 - mlen;
  Mp mp{idx, idx + mlen};
  uint64_t nf, key, n;
  if (!mp.map(&nf)) { set_error("%s: bad index", path.c_str()); return BG_EIO; }
  std::vector<std::pair<uint64_t, uint64_t>> fix_pos;
  std::vector<char> fix_chr;
  for (uint64_t f = 0; f < nf && mp.ok; ++f) {
    if (!mp.uint(&key)) break;
    if (key == 1 || key == 2 || key == 3) {
      if (!mp.array(&n)) break;
      for (uint64_t i = 0; i < n && mp.ok; ++i) {
        uint64_t nk, k2, v[5] = {0, 0, 0, 0, 0};
        std::string name, md5;
        if (!mp.map(&nk)) break;
        for (uint64_t j = 0; j < nk && mp.ok; ++j) {
          if (!mp.uint(&k2)) break;
          if (key == 1 && (k2 == 1 || k2 == 2)) mp.str(k2 == 1 ? &name : &md5);
          else if (k2 >= 1 && k2 <= 4) mp.uint(&v[k2]);
          else mp.skip();
        }
        if (key == 1) sr->scaffolds.emplace_back(name, v[3]);
        else if (key == 2) sr->extents.push_back(SpecRef::Extent{v[1], v[2], v[3], v[4]});
        else { fix_pos.emplace_back(v[1], v[2]); fix_chr.push_back(char(v[3])); }
      }
    } else {
      mp.skip();
    }
  }
  if (!mp.ok) { set_error("%s: malformed msgpack index", path.c_str()); return BG_EIO; }
  const uint64_t dna_bytes = uint64_t(idx - u) - 6;
  uint64_t flat_end = 0;
  const uint64_t dna_bases = dna_bytes > (~0ull >> 2) ? ~0ull : dna_bytes * 4;
  for (auto& x : sr->extents) {
    // every extent must lie inside the 2-bit payload (checked without letting flat + size wrap) and name a scaffold
    if (x.flat > dna_bases || x.size > dna_bases - x.flat) { set_error("%s: an extent lies past the end of the DNA buffer", path.c_str()); return BG_EIO; }
    if (x.scaffold >= sr->scaffolds.size()) { set_error("%s: extent of unknown scaffold %llu", path.c_str(), (unsigned long long)x.scaffold); return BG_EIO; }
    const uint64_t slen = sr->scaffolds[x.scaffold].second;
    if (x.offset > slen || x.size > slen - x.offset) { set_error("%s: an extent lies past the end of its scaffold", path.c_str()); return BG_EIO; }
    flat_end = std::max(flat_end, x.flat + x.size);
  }
  sr->bases.resize(flat_end);
  const unsigned char* dna = u + 6;
  for (uint64_t i = 0; i < flat_end; ++i) sr->bases[i] = "ACGT"[(dna[i >> 2] >> (6 - 2 * (i & 3))) & 3];
  // fixups restore characters outside ACGTN in FASTA exports; the walk wants plain bases: ignored here
  (void)fix_pos; (void)fix_chr;
  return BG_OK;
}

int load_karyotype(const std::string& path, bg_reference* r) {
  std::string txt;
  if (!read_all(path, &txt)) { set_error("Could not open %s: %s", path.c_str(), strerror(errno)); return BG_EIO; }
  Js js{txt.data(), txt.data() + txt.size()};
  std::vector<std::string> order;
  std::map<std::string, std::pair<uint64_t, uint64_t>> chroms;  // name -> (index, len)
  struct Sc { std::string chr; uint64_t offset, len; };
  std::vector<Sc> scs;
  if (!js.eat('{')) { set_error("%s: not a JSON object", path.c_str()); return BG_EIO; }
  if (!js.eat('}')) {
    do {
      std::string key, s;
      if (!js.str(&key) || !js.eat(':')) { js.ok = false; break; }
      if (key == "chr_order") {
        if (!js.eat('[')) { js.ok = false; break; }
        if (!js.eat(']')) {
          do { if (!js.str(&s)) break; order.push_back(s); } while (js.eat(','));
          if (!js.eat(']')) js.ok = false;
        }
      } else if (key == "chromosomes" || key == "supercontigs") {
        if (!js.eat('[')) { js.ok = false; break; }
        if (!js.eat(']')) {
          do {
            if (!js.eat('{')) { js.ok = false; break; }
            std::string name, chr;
            uint64_t index = 0, len = 0, offset = 0;
            do {
              std::string k2;
              if (!js.str(&k2) || !js.eat(':')) { js.ok = false; break; }
              if (k2 == "name") js.str(&name);
              else if (k2 == "chr") js.str(&chr);
              else if (k2 == "index") js.num(&index);
              else if (k2 == "len") js.num(&len);
              else if (k2 == "offset") js.num(&offset);
              else js.skip();
            } while (js.ok && js.eat(','));
            if (!js.eat('}')) js.ok = false;
            if (key == "chromosomes") chroms[name] = std::make_pair(index, len);
            else scs.push_back(Sc{chr, offset, len});
          } while (js.ok && js.eat(','));
          if (!js.eat(']')) js.ok = false;
        }
      } else {
        js.skip();
      }
    } while (js.ok && js.eat(','));
  }
  if (!js.ok) { set_error("%s: malformed karyotype", path.c_str()); return BG_EIO; }
  if (order.empty()) {  // chromosomes by index
    std::vector<std::pair<uint64_t, std::string>> v;
    for (auto& kv : chroms) v.emplace_back(kv.second.first, kv.first);
    std::sort(v.begin(), v.end());
    for (auto& x : v) order.push_back(x.second);
  }
  std::map<std::string, uint32_t> pos;
  for (auto& name : order) {
    pos[name] = uint32_t(r->chromosomes.size());
    r->chromosomes.push_back(Chromosome{name, chroms.count(name) ? chroms[name].second : 0});
  }
  for (auto& sc : scs) {
    if (!pos.count(sc.chr)) { set_error("%s: supercontig of unknown chromosome %s", path.c_str(), sc.chr.c_str()); return BG_EIO; }
    r->supercontigs.push_back(Supercontig{pos[sc.chr], sc.offset, sc.len, 0});
  }
  std::sort(r->supercontigs.begin(), r->supercontigs.end(), [](const Supercontig& a, const Supercontig& b) {
    return a.chr != b.chr ? a.chr < b.chr : a.offset < b.offset;
  });
  return BG_OK;
}

}  // namespace

extern "C" int bg_reference_open(const char* path_c, bg_reference** out) {
  if (!path_c || !out) { set_error("bg_reference_open: null argument"); return BG_EINVAL; }
  std::string path(path_c);
  while (path.size() > 1 && path.back() == '/') path.pop_back();
  std::unique_ptr<bg_reference> r(new bg_reference);
  if (is_dir(path)) {
    if (!exists(path + "/karyotype.json")) {
      if (exists(path + "/reference.fasta")) {
        BG_TRY(load_fasta(path + "/reference.fasta", r.get()));
        *out = r.release();
        return BG_OK;
      }
      set_error("%s: no karyotype.json (not a reference directory)", path.c_str());
      return BG_EIO;
    }
    BG_TRY(load_karyotype(path + "/karyotype.json", r.get()));
    // flat positions: the .ref's extents when present, else the supercontigs back to back
    if (exists(path + "/reference.ref")) {
      SpecRef sr;
      BG_TRY(load_spec(path + "/reference.ref", &sr));
      // the .ref of a reference directory is built from reference.fasta, whose records are the supercontigs
      // "chr:N" in flat order (make_ref.cpp); match them to the karyotype by order and length
      if (sr.extents.size() != r->supercontigs.size()) {
        set_error("%s: reference.ref holds %zu extents, karyotype.json %zu supercontigs", path.c_str(), sr.extents.size(),
                  r->supercontigs.size());
        return BG_EIO;
      }
      std::vector<SpecRef::Extent> ex = sr.extents;
      std::sort(ex.begin(), ex.end(), [](const SpecRef::Extent& a, const SpecRef::Extent& b) { return a.flat < b.flat; });
      for (size_t i = 0; i < ex.size(); ++i) {
        if (ex[i].size != r->supercontigs[i].len) {
          set_error("%s: extent %zu of reference.ref is %llu bases, the karyotype says %llu", path.c_str(), i,
                    (unsigned long long)ex[i].size, (unsigned long long)r->supercontigs[i].len);
          return BG_EIO;
        }
        r->supercontigs[i].flat = ex[i].flat;
      }
      r->bases.swap(sr.bases);
      r->source = "ref";
    } else if (exists(path + "/reference.fasta")) {
      bg_reference fa;
      BG_TRY(load_fasta(path + "/reference.fasta", &fa));  // records = supercontigs, in flat order
      uint64_t flat = 0;
      for (auto& sc : r->supercontigs) { sc.flat = flat; flat += sc.len; }
      if (flat != fa.bases.size()) { set_error("%s: reference.fasta and karyotype.json disagree on the size", path.c_str()); return BG_EIO; }
      r->bases.swap(fa.bases);
      r->source = "fasta";
    } else {
      set_error("%s: neither reference.ref nor reference.fasta", path.c_str());
      return BG_EIO;
    }
    *out = r.release();
    return BG_OK;
  }
  // a single file: FASTA (first byte '>') or .ref — flat_ref.cpp:18-24
  std::string head;
  {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { set_error("Could not open %s: %s", path.c_str(), strerror(errno)); return BG_EIO; }
    char c = 0;
    if (fread(&c, 1, 1, f) == 1) head.push_back(c);
    fclose(f);
  }
  if (!head.empty() && head[0] == '>') {
    BG_TRY(load_fasta(path, r.get()));
  } else {
    SpecRef sr;
    BG_TRY(load_spec(path, &sr));
    for (auto& s : sr.scaffolds) r->chromosomes.push_back(Chromosome{s.first, s.second});
    std::vector<SpecRef::Extent> ex = sr.extents;
    std::sort(ex.begin(), ex.end(), [](const SpecRef::Extent& a, const SpecRef::Extent& b) { return a.flat < b.flat; });
    for (auto& x : ex) {
      if (x.scaffold >= r->chromosomes.size()) { set_error("%s: extent of unknown scaffold", path.c_str()); return BG_EIO; }
      r->supercontigs.push_back(Supercontig{uint32_t(x.scaffold), x.offset, x.size, x.flat});
    }
    r->bases.swap(sr.bases);
    r->source = "ref";
  }
  *out = r.release();
  return BG_OK;
}

extern "C" void bg_reference_free(bg_reference* r) { delete r; }
extern "C" uint64_t bg_reference_size(const bg_reference* r) { return r ? r->bases.size() : 0; }
extern "C" uint32_t bg_reference_scaffold_count(const bg_reference* r) { return r ? uint32_t(r->chromosomes.size()) : 0; }
extern "C" const char* bg_reference_scaffold_name(const bg_reference* r, uint32_t i) {
  return r && i < r->chromosomes.size() ? r->chromosomes[i].name.c_str() : "";
}
extern "C" uint64_t bg_reference_scaffold_len(const bg_reference* r, uint32_t i) {
  return r && i < r->chromosomes.size() ? r->chromosomes[i].len : 0;
}
extern "C" uint32_t bg_reference_supercontig_count(const bg_reference* r) { return r ? uint32_t(r->supercontigs.size()) : 0; }
extern "C" int bg_reference_supercontig(const bg_reference* r, uint32_t i, uint32_t* scaffold, uint64_t* offset, uint64_t* len,
                                        uint64_t* flat) {
  if (!r || i >= r->supercontigs.size()) { set_error("bg_reference_supercontig: index out of range"); return BG_EINVAL; }
  const Supercontig& s = r->supercontigs[i];
  if (scaffold) *scaffold = s.chr;
  if (offset) *offset = s.offset;
  if (len) *len = s.len;
  if (flat) *flat = s.flat;
  return BG_OK;
}
extern "C" int bg_reference_bases(const bg_reference* r, uint64_t flat_pos, uint64_t n, char* out) {
  if (!r || (n && !out)) { set_error("bg_reference_bases: null argument"); return BG_EINVAL; }
  if (flat_pos > r->bases.size() || n > r->bases.size() - flat_pos) { set_error("bg_reference_bases: range past the end"); return BG_EINVAL; }
  memcpy(out, r->bases.data() + flat_pos, n);
  return BG_OK;
}

// ---- reference.bwt: bwt_file / bwt_range (bio_base/bwt_file.h:9-66, bwt_file.cpp:20-112) on the host --------------------
// The file (bg_bwt_build writes it, csrc/bwt.hip): magic, offset of the msgpack header, five bit vectors of ca_table[4] bits
// each (previous base A, C, G, T; century mark) in bitcount layout (words, then the format's own rank tables, which are
// not used here: one cumulative count per 64-bit word is made at open), the century table, the header {2: ca_table[6]}.
struct bg_bwt {
  std::string blob;
  uint64_t n = 0, ca[6] = {0, 0, 0, 0, 0, 0};
  const uint64_t* bits[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const uint32_t* century = nullptr;
  std::vector<uint64_t> cum[5];  // set bits before word w
  bool get(int v, uint64_t i) const { return (bits[v][i >> 6] >> (i & 63)) & 1; }
  uint64_t count(int v, uint64_t i) const {  // set bits in [0, i)
    const uint64_t w = i >> 6, r = i & 63;
    return cum[v][w] + (r ? uint64_t(__builtin_popcountll(bits[v][w] & ((1ull << r) - 1))) : 0);
  }
};

extern "C" int bg_bwt_open(const char* path_c, bg_bwt** out) {
  if (!path_c || !out) { set_error("bg_bwt_open: null argument"); return BG_EINVAL; }
  std::string path = path_c;
  if (is_dir(path)) path += "/reference.bwt";
  std::unique_ptr<bg_bwt> b(new bg_bwt);
  if (!read_all(path, &b->blob)) { set_error("bg_bwt_open: cannot read %s", path.c_str()); return BG_EIO; }
  const std::string& f = b->blob;
  uint64_t magic = 0, off = 0;
  if (f.size() >= 16) { memcpy(&magic, f.data(), 8); memcpy(&off, f.data() + 8, 8); }
  if (magic != 0x57426C6172697053ull || off < 16 || off > f.size()) {
    set_error("%s does not appear to be a valid BWT.", path.c_str());  // bwt_file.cpp:13
    return BG_EIO;
  }
  Mp mp{reinterpret_cast<const unsigned char*>(f.data()) + off, reinterpret_cast<const unsigned char*>(f.data()) + f.size()};
  uint64_t nkeys = 0, key = 0, na = 0;
  bool have = false;
  if (mp.map(&nkeys)) {
    for (uint64_t i = 0; i < nkeys && mp.ok; ++i) {
      if (!mp.uint(&key)) break;
      if (key == 2 && mp.array(&na) && na == 6) {
        for (int j = 0; j < 6; ++j) mp.uint(&b->ca[j]);
        have = mp.ok;
      } else {
        mp.skip();
      }
    }
  }
  if (!have) { set_error("%s: malformed BWT header", path.c_str()); return BG_EIO; }
  b->n = b->ca[4];
  const uint64_t nwords = (b->n + 63) / 64, ngroups = (b->n + 511) / 512, naccum = (b->n + 512) / 512;
  const uint64_t bits_size = (nwords + ngroups + naccum) * 8;  // bitcount::compute_size
  if (16 + 5 * bits_size + 4 * b->ca[5] > off) { set_error("%s: BWT body shorter than its header says", path.c_str()); return BG_EIO; }
  for (int v = 0; v < 5; ++v) {
    b->bits[v] = reinterpret_cast<const uint64_t*>(f.data() + 16 + uint64_t(v) * bits_size);
    b->cum[v].resize(nwords + 1);
    uint64_t c = 0;
    for (uint64_t w = 0; w < nwords; ++w) {
      b->cum[v][w] = c;
      c += uint64_t(__builtin_popcountll(b->bits[v][w]));
    }
    b->cum[v][nwords] = c;
  }
  b->century = reinterpret_cast<const uint32_t*>(f.data() + 16 + 5 * bits_size);
  *out = b.release();
  return BG_OK;
}
extern "C" void bg_bwt_free(bg_bwt* b) { delete b; }
extern "C" uint64_t bg_bwt_entries(const bg_bwt* b) { return b ? b->n : 0; }

// bwt_range::find (bwt_file.cpp:64-81): the range of sorted suffixes that start with seq (ACGT, either case); an empty
// range (begin == end) when it does not occur.  A character that is no base is BG_EINVAL.
extern "C" int bg_bwt_find(const bg_bwt* b, const char* seq, uint64_t len, uint64_t* begin, uint64_t* end) {
  if (!b || (len && !seq) || !begin || !end) { set_error("bg_bwt_find: null argument"); return BG_EINVAL; }
  uint64_t lo = 0, hi = b->n;
  for (uint64_t i = len; i-- > 0;) {
    int k;
    switch (seq[i]) {
      case 'A': case 'a': k = 0; break;
      case 'C': case 'c': k = 1; break;
      case 'G': case 'g': k = 2; break;
      case 'T': case 't': k = 3; break;
      default: set_error("bg_bwt_find: '%c' is not a base", seq[i]); return BG_EINVAL;
    }
    lo = b->ca[k] + b->count(k, lo);
    hi = b->ca[k] + b->count(k, hi);
    if (lo == hi) break;
  }
  *begin = lo;
  *end = hi;
  return BG_OK;
}

// bwt_range::get_match (bwt_file.cpp:93-112): the flat reference position of sorted suffix `entry`: steps back through the
// previous bases to the nearest century mark
extern "C" int bg_bwt_match(const bg_bwt* b, uint64_t entry, uint64_t* flat_pos) {
  if (!b || !flat_pos) { set_error("bg_bwt_match: null argument"); return BG_EINVAL; }
  if (entry >= b->n) { set_error("bg_bwt_match: entry %llu of %llu", (unsigned long long)entry, (unsigned long long)b->n); return BG_EINVAL; }
  uint64_t e = entry, dist = 0;
  while (!b->get(4, e)) {
    const int k = (b->get(1, e) ? 1 : 0) + (b->get(2, e) ? 2 : 0) + (b->get(3, e) ? 3 : 0);
    if (k == 0 && !b->get(0, e)) { set_error("bg_bwt_match: suffix %llu starts an extent without a century mark", (unsigned long long)e); return BG_EINTERNAL; }
    e = b->ca[k] + b->count(k, e);
    if (++dist > b->n) { set_error("bg_bwt_match: no century mark reached"); return BG_EINTERNAL; }
  }
  *flat_pos = uint64_t(b->century[b->count(4, e)]) + dist;
  return BG_OK;
}
