// spiral_zip.cpp — writer/reader of the "spiral file" zip container the reference stores a seqset in
// (modules/io/spiral_file_mmap.cpp:350-457 over the forked vendor/minizip): zip64-flavoured local
// headers, method 0 (stored), members created through the mmap path carry CRC 0 (:429), JSON members
// carry their real CRC.  Byte layout of the headers reproduces what the reference's goldens contain
// (golden/e_coli_10000snp.bg/seqset, father_lambda.bg/seqset).
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <fcntl.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <sys/mman.h>

#include <atomic>

#include <fstream>
#include <mutex>
#include <random>
#include <thread>

#include "bg_internal.h"

namespace bg {

// CRC-32 (zip), slicing by 8: ~2 GB/s per host thread instead of ~0.4 byte-at-a-time; the members of a container are
// checksummed on separate threads.
static uint32_t crc32_of(const std::string& s) {
  static uint32_t T[8][256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xff];
  });
  const unsigned char* p = reinterpret_cast<const unsigned char*>(s.data());
  size_t n = s.size();
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t a, b;
    memcpy(&a, p, 4);
    memcpy(&b, p + 4, 4);
    a ^= c;
    c = T[7][a & 0xff] ^ T[6][(a >> 8) & 0xff] ^ T[5][(a >> 16) & 0xff] ^ T[4][a >> 24] ^ T[3][b & 0xff] ^ T[2][(b >> 8) & 0xff] ^
        T[1][(b >> 16) & 0xff] ^ T[0][b >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

static void put16(std::string& o, uint16_t v) { o.push_back(char(v & 0xff)); o.push_back(char(v >> 8)); }
static void put32(std::string& o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back(char((v >> (8 * i)) & 0xff)); }
static void put64(std::string& o, uint64_t v) { for (int i = 0; i < 8; ++i) o.push_back(char((v >> (8 * i)) & 0xff)); }

std::string make_uuid() {
  std::random_device rd;
  std::mt19937_64 g((uint64_t(rd()) << 32) ^ rd() ^ uint64_t(time(nullptr)));
  uint64_t a = g(), b = g();
  char buf[40];
  snprintf(buf, sizeof buf, "%08x-%04x-%04x-%04x-%012llx", uint32_t(a >> 32), uint32_t((a >> 16) & 0xffff),
           uint32_t(a & 0xffff), uint32_t(b >> 48), (unsigned long long)(b & 0xffffffffffffull));
  return buf;
}

int spiral_zip_write(const char* path, const std::vector<ZipPart>& parts) { return spiral_zip_write_layout(path, parts, nullptr); }

// A part with hole_size != 0 gets its headers and hole_size bytes of room, which the caller fills in place
// afterwards (the slices of a sharded seqset, written by every rank at data_offsets[part] + its own offset).
int spiral_zip_write_layout(const char* path, const std::vector<ZipPart>& parts, std::vector<uint64_t>* data_offsets) {
  FILE* f = fopen(path, "wb");
  if (!f) { set_error("Could not open zip for writing: %s: %s", path, strerror(errno)); return BG_EIO; }
  struct Cd { std::string name; uint32_t crc; uint64_t size, offset; };
  std::vector<Cd> cds;
  uint64_t pos = 0;
  if (data_offsets) data_offsets->clear();
  // checksums of the members on the host's threads (largest first), then the headers in one stream with room left for
  // the bodies, which threads fill in place in 8 MiB pieces: a page-cache / tmpfs write is page allocation, and that
  // scales with threads
  std::vector<uint32_t> crcs(parts.size(), 0);
  {
    std::vector<size_t> order;
    for (size_t i = 0; i < parts.size(); ++i)
      if (!parts[i].raw_crc0 && !parts[i].hole_size && parts[i].data.size() >= (1u << 20)) order.push_back(i);
      else if (!parts[i].raw_crc0 && !parts[i].hole_size) crcs[i] = crc32_of(parts[i].data);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return parts[a].data.size() > parts[b].data.size(); });
    std::atomic<size_t> next{0};
    auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < order.size();) crcs[order[k]] = crc32_of(parts[order[k]].data); };
    std::vector<std::thread> pool;
    const size_t nt = std::min<size_t>(order.size(), std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    for (size_t t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
  }
  struct Body { const std::string* data; uint64_t offset; };
  std::vector<Body> bodies;
  size_t part_index = 0;
  for (const ZipPart& p : parts) {
    const bool hole = p.hole_size != 0;
    uint64_t size = hole ? p.hole_size : p.data.size();
    bool big = size >= 0xffffffffull;
    uint32_t crc = crcs[part_index++];
    const bool later = !hole && size >= (4u << 20);  // written in place below
    std::string h;
    put32(h, 0x04034b50);
    put16(h, 45);  // version needed (zip64)
    put16(h, 0);   // flags
    put16(h, 0);   // method: stored
    put16(h, 0);   // time
    put16(h, 0);   // date
    put32(h, crc);
    put32(h, big ? 0xffffffffu : uint32_t(size));
    put32(h, big ? 0xffffffffu : uint32_t(size));
    put16(h, uint16_t(p.name.size()));
    put16(h, 20);  // zip64 extra placeholder
    h += p.name;
    put16(h, 0x0001);
    put16(h, 16);
    put64(h, big ? size : 0);
    put64(h, big ? size : 0);
    if (later) bodies.push_back(Body{&p.data, pos + h.size()});
    if (fwrite(h.data(), 1, h.size(), f) != h.size() ||
        (!hole && !later && size && fwrite(p.data.data(), 1, size, f) != size) ||
        ((hole || later) && fseeko(f, off_t(size), SEEK_CUR) != 0)) {
      set_error("write error on %s: %s", path, strerror(errno));
      fclose(f);
      return BG_EIO;
    }
    if (data_offsets) data_offsets->push_back(pos + h.size());
    cds.push_back(Cd{p.name, crc, size, pos});
    pos += h.size() + size;
  }
  uint64_t cd_start = pos;
  std::string cd;
  for (const Cd& c : cds) {
    bool big_size = c.size >= 0xffffffffull, big_off = c.offset >= 0xffffffffull;
    std::string extra;
    if (big_size || big_off) {
      std::string body;
      if (big_size) { put64(body, c.size); put64(body, c.size); }
      if (big_off) put64(body, c.offset);
      put16(extra, 0x0001);
      put16(extra, uint16_t(body.size()));
      extra += body;
    }
    put32(cd, 0x02014b50);
    put16(cd, (big_size || big_off) ? 45 : 0);   // version made by
    put16(cd, (big_size || big_off) ? 45 : 20);  // version needed
    put16(cd, 0);
    put16(cd, 0);
    put16(cd, 0);
    put16(cd, 0);
    put32(cd, c.crc);
    put32(cd, big_size ? 0xffffffffu : uint32_t(c.size));
    put32(cd, big_size ? 0xffffffffu : uint32_t(c.size));
    put16(cd, uint16_t(c.name.size()));
    put16(cd, uint16_t(extra.size()));
    put16(cd, 0);
    put16(cd, 0);
    put16(cd, 0);
    put32(cd, 0);
    put32(cd, big_off ? 0xffffffffu : uint32_t(c.offset));
    cd += c.name;
    cd += extra;
  }
  uint64_t cd_size = cd.size();
  bool z64 = cd_start >= 0xffffffffull || cds.size() >= 0xffff;
  if (z64) {
    put32(cd, 0x06064b50);
    put64(cd, 44);
    put16(cd, 45);
    put16(cd, 45);
    put32(cd, 0);
    put32(cd, 0);
    put64(cd, cds.size());
    put64(cd, cds.size());
    put64(cd, cd_size);
    put64(cd, cd_start);
    put32(cd, 0x07064b50);
    put32(cd, 0);
    put64(cd, cd_start + cd_size);
    put32(cd, 1);
  }
  put32(cd, 0x06054b50);
  put16(cd, 0);
  put16(cd, 0);
  put16(cd, z64 ? 0xffff : uint16_t(cds.size()));
  put16(cd, z64 ? 0xffff : uint16_t(cds.size()));
  put32(cd, uint32_t(cd_size));
  put32(cd, z64 ? 0xffffffffu : uint32_t(cd_start));
  put16(cd, 0);
  bool ok = fwrite(cd.data(), 1, cd.size(), f) == cd.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok) { set_error("write error on %s: %s", path, strerror(errno)); return BG_EIO; }
  if (!bodies.empty()) {
    const int fd = open(path, O_WRONLY);
    if (fd < 0) { set_error("write error on %s: %s", path, strerror(errno)); return BG_EIO; }
    struct Piece { const char* src; uint64_t off, len; };
    std::vector<Piece> pieces;
    const uint64_t kPiece = 8ull << 20;
    for (const Body& b : bodies)
      for (uint64_t o = 0; o < b.data->size(); o += kPiece)
        pieces.push_back(Piece{b.data->data() + o, b.offset + o, std::min<uint64_t>(kPiece, b.data->size() - o)});
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&] {
      for (size_t k; (k = next.fetch_add(1)) < pieces.size();) {
        const Piece& pc = pieces[k];
        uint64_t done = 0;
        while (done < pc.len) {
          const ssize_t w = pwrite(fd, pc.src + done, pc.len - done, off_t(pc.off + done));
          if (w <= 0) { failed = 1; return; }
          done += uint64_t(w);
        }
      }
    };
    std::vector<std::thread> pool;
    const size_t nt = std::min<size_t>(pieces.size(), std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    for (size_t t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (close(fd) != 0 || failed) { set_error("write error on %s: %s", path, strerror(errno)); return BG_EIO; }
  }
  return BG_OK;
}

static uint16_t get16(const unsigned char* p) { return uint16_t(p[0] | (p[1] << 8)); }
static uint32_t get32(const unsigned char* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
static uint64_t get64(const unsigned char* p) { return uint64_t(get32(p)) | (uint64_t(get32(p + 4)) << 32); }

int spiral_zip_read(const char* path, std::map<std::string, std::string>* parts) { return spiral_zip_read_views(path, parts, nullptr, nullptr); }

// ... with the members of a megabyte and more left where they are: the file is mapped and `big` names their bytes in the mapping,
// which `keep` holds (a 370 MB seqset read into strings first was a tenth of a second of page faults and copies before its upload)
int spiral_zip_read_views(const char* path, std::map<std::string, std::string>* parts,
                          std::map<std::string, std::pair<const char*, uint64_t>>* big, std::shared_ptr<void>* keep) {
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in) { set_error("Could not open %s", path); return BG_EIO; }
  uint64_t fsize = uint64_t(in.tellg());
  const char* base = nullptr;
  if (big && keep && fsize) {
    const int fd = open(path, O_RDONLY);
    if (fd >= 0) {
      void* m = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
      close(fd);
      if (m != MAP_FAILED) {
        base = static_cast<const char*>(m);
        keep->reset(m, [fsize](void* q) { munmap(q, fsize); });
      }
    }
  }
  auto read_at = [&](uint64_t off, uint64_t len, std::string* out) -> bool {
    if (off + len > fsize) return false;
    out->resize(len);
    in.seekg(std::streamoff(off));
    in.read(&(*out)[0], std::streamsize(len));
    return bool(in);
  };
  // end of central directory: scan the tail
  uint64_t tail_len = fsize < 66000 ? fsize : 66000;
  std::string tail;
  if (!read_at(fsize - tail_len, tail_len, &tail)) { set_error("%s: short file", path); return BG_EIO; }
  int64_t eocd = -1;
  for (int64_t i = int64_t(tail.size()) - 22; i >= 0; --i)
    if (get32(reinterpret_cast<const unsigned char*>(tail.data()) + i) == 0x06054b50) { eocd = i; break; }
  if (eocd < 0) { set_error("%s: not a zip file", path); return BG_EIO; }
  const unsigned char* e = reinterpret_cast<const unsigned char*>(tail.data()) + eocd;
  uint64_t n_entries = get16(e + 10), cd_size = get32(e + 12), cd_start = get32(e + 16);
  if (n_entries == 0xffff || cd_start == 0xffffffffu) {
    // zip64 locator sits right before the EOCD
    if (eocd < 20 || get32(e - 20) != 0x07064b50) { set_error("%s: missing zip64 locator", path); return BG_EIO; }
    uint64_t z64_off = get64(e - 20 + 8);
    std::string z;
    if (!read_at(z64_off, 56, &z) || get32(reinterpret_cast<const unsigned char*>(z.data())) != 0x06064b50) {
      set_error("%s: bad zip64 end record", path);
      return BG_EIO;
    }
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(z.data());
    n_entries = get64(zp + 32);
    cd_size = get64(zp + 40);
    cd_start = get64(zp + 48);
  }
  std::string cd;
  if (!read_at(cd_start, cd_size, &cd)) { set_error("%s: bad central directory", path); return BG_EIO; }
  const unsigned char* p = reinterpret_cast<const unsigned char*>(cd.data());
  const unsigned char* pend = p + cd.size();
  for (uint64_t i = 0; i < n_entries; ++i) {
    if (p + 46 > pend || get32(p) != 0x02014b50) { set_error("%s: bad central entry", path); return BG_EIO; }
    uint16_t method = get16(p + 10);
    uint64_t csize = get32(p + 20), usize = get32(p + 24);
    uint16_t nlen = get16(p + 28), elen = get16(p + 30), clen = get16(p + 32);
    uint64_t off = get32(p + 42);
    std::string name(reinterpret_cast<const char*>(p + 46), nlen);
    const unsigned char* x = p + 46 + nlen;
    const unsigned char* xend = x + elen;
    while (x + 4 <= xend) {
      uint16_t id = get16(x), sz = get16(x + 2);
      const unsigned char* b = x + 4;
      if (id == 0x0001) {
        if (usize == 0xffffffffu && b + 8 <= x + 4 + sz) { usize = get64(b); b += 8; }
        if (csize == 0xffffffffu && b + 8 <= x + 4 + sz) { csize = get64(b); b += 8; }
        if (off == 0xffffffffu && b + 8 <= x + 4 + sz) { off = get64(b); b += 8; }
      }
      x += 4 + sz;
    }
    if (method != 0) { set_error("%s: member %s is compressed", path, name.c_str()); return BG_EIO; }
    std::string lh;
    if (!read_at(off, 30, &lh)) { set_error("%s: bad local header", path); return BG_EIO; }
    const unsigned char* l = reinterpret_cast<const unsigned char*>(lh.data());
    if (get32(l) != 0x04034b50) { set_error("%s: bad local header signature", path); return BG_EIO; }
    uint64_t data_off = off + 30 + get16(l + 26) + get16(l + 28);
    if (base && usize >= (uint64_t(1) << 20)) {
      if (data_off + usize > fsize) { set_error("%s: member %s truncated", path, name.c_str()); return BG_EIO; }
      (*big)[name] = std::make_pair(base + data_off, usize);
      p += 46 + nlen + elen + clen;
      continue;
    }
    std::string data;
    if (!read_at(data_off, usize, &data)) { set_error("%s: member %s truncated", path, name.c_str()); return BG_EIO; }
    (*parts)[name] = std::move(data);
    p += 46 + nlen + elen + clen;
  }
  return BG_OK;
}

// SHA-1 with the x86 SHA extensions (sha1rnds4 does four rounds, sha1msg1/2 the message schedule): 2-3 GB/s instead of
// the 0.5 GB/s of the portable rounds below -- a 130 MB readmap's digest was a quarter of a second of `bgbinary create`.
#if defined(__x86_64__)
__attribute__((target("sha,sse4.1,ssse3"))) static void sha1_blocks_ni(uint32_t h[5], const uint8_t* data, size_t nblocks) {
  const __m128i mask = _mm_set_epi64x(0x0001020304050607ll, 0x08090a0b0c0d0e0fll);
  __m128i abcd = _mm_shuffle_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i*>(h)), 0x1B);
  __m128i e0 = _mm_set_epi32(int(h[4]), 0, 0, 0);
  for (; nblocks; --nblocks, data += 64) {
    const __m128i abcd_save = abcd, e_save = e0;
    __m128i w[20];
    for (int g = 0; g < 4; ++g) w[g] = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(data + 16 * g)), mask);
    for (int g = 4; g < 20; ++g)
      w[g] = _mm_sha1msg2_epu32(_mm_xor_si128(_mm_sha1msg1_epu32(w[g - 4], w[g - 3]), w[g - 2]), w[g - 1]);
    __m128i save = abcd, t = _mm_add_epi32(e0, w[0]);
    abcd = _mm_sha1rnds4_epu32(abcd, t, 0);
#define BG_SHA1_STEP(G, F)                    \
    t = _mm_sha1nexte_epu32(save, w[G]);      \
    save = abcd;                              \
    abcd = _mm_sha1rnds4_epu32(abcd, t, F);
    BG_SHA1_STEP(1, 0) BG_SHA1_STEP(2, 0) BG_SHA1_STEP(3, 0) BG_SHA1_STEP(4, 0)
    BG_SHA1_STEP(5, 1) BG_SHA1_STEP(6, 1) BG_SHA1_STEP(7, 1) BG_SHA1_STEP(8, 1) BG_SHA1_STEP(9, 1)
    BG_SHA1_STEP(10, 2) BG_SHA1_STEP(11, 2) BG_SHA1_STEP(12, 2) BG_SHA1_STEP(13, 2) BG_SHA1_STEP(14, 2)
    BG_SHA1_STEP(15, 3) BG_SHA1_STEP(16, 3) BG_SHA1_STEP(17, 3) BG_SHA1_STEP(18, 3) BG_SHA1_STEP(19, 3)
#undef BG_SHA1_STEP
    e0 = _mm_sha1nexte_epu32(save, e_save);
    abcd = _mm_add_epi32(abcd, abcd_save);
  }
  _mm_storeu_si128(reinterpret_cast<__m128i*>(h), _mm_shuffle_epi32(abcd, 0x1B));
  h[4] = uint32_t(_mm_extract_epi32(e0, 3));
}
static bool sha1_have_ni() { return __builtin_cpu_supports("sha") && !getenv("BG_SHA1_PORTABLE"); }
#else
static void sha1_blocks_ni(uint32_t*, const uint8_t*, size_t) {}
static bool sha1_have_ni() { return false; }
#endif

// SHA-1 of a file (FIPS 180-1): the reference names coverage/<sha1>.readmap after the digest of the
// written container (biograph_create.cpp:827-828, sha1_hash over the file).
int sha1_file_hex(const char* path, char* hex41) {
  FILE* f = fopen(path, "rb");
  if (!f) { set_error("%s: %s", path, strerror(errno)); return BG_EIO; }
  uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
  const bool ni = sha1_have_ni();
  auto block = [&](const uint8_t* b) {
    if (ni) { sha1_blocks_ni(h, b, 1); return; }
    uint32_t w[80];
    for (int i = 0; i < 16; ++i)
      w[i] = uint32_t(b[4 * i]) << 24 | uint32_t(b[4 * i + 1]) << 16 | uint32_t(b[4 * i + 2]) << 8 | b[4 * i + 3];
    for (int i = 16; i < 80; ++i) {
      uint32_t x = w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16];
      w[i] = x << 1 | x >> 31;
    }
    uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4];
    for (int i = 0; i < 80; ++i) {
      uint32_t fn, k;
      if (i < 20) { fn = (bb & c) | (~bb & d); k = 0x5A827999u; }
      else if (i < 40) { fn = bb ^ c ^ d; k = 0x6ED9EBA1u; }
      else if (i < 60) { fn = (bb & c) | (bb & d) | (c & d); k = 0x8F1BBCDCu; }
      else { fn = bb ^ c ^ d; k = 0xCA62C1D6u; }
      uint32_t t = (a << 5 | a >> 27) + fn + e + k + w[i];
      e = d; d = c; c = bb << 30 | bb >> 2; bb = a; a = t;
    }
    h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e;
  };
  std::vector<uint8_t> buf(1 << 20);
  uint8_t tail[128];
  size_t ntail = 0;
  uint64_t total = 0;
  for (;;) {
    size_t got = fread(buf.data(), 1, buf.size(), f);
    total += got;
    size_t full = got / 64 * 64;
    if (ni) sha1_blocks_ni(h, buf.data(), full / 64);
    else for (size_t o = 0; o < full; o += 64) block(buf.data() + o);
    if (got < buf.size()) {
      ntail = got - full;
      memcpy(tail, buf.data() + full, ntail);
      break;
    }
  }
  bool bad = ferror(f) != 0;
  fclose(f);
  if (bad) { set_error("%s: read error", path); return BG_EIO; }
  tail[ntail++] = 0x80;
  while (ntail % 64 != 56) tail[ntail++] = 0;
  uint64_t bits = total * 8;
  for (int i = 7; i >= 0; --i) tail[ntail++] = uint8_t(bits >> (8 * i));
  for (size_t o = 0; o < ntail; o += 64) block(tail + o);
  snprintf(hex41, 41, "%08x%08x%08x%08x%08x", h[0], h[1], h[2], h[3], h[4]);
  return BG_OK;
}

}  // namespace bg

// SHA-1 of a file as 40 hex digits + NUL: how a readmap is named inside a .bg directory.  Host only.
extern "C" int bg_sha1_file(const char* path, char* hex41) {
  if (!path || !hex41) { bg::set_error("bg_sha1_file: null argument"); return BG_EINVAL; }
  return bg::sha1_file_hex(path, hex41);
}
