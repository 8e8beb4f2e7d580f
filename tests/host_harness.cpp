// host_harness.cpp — TEST ONLY.  Compiles biograph_amd/csrc/bg_device_funcs.h (the per-thread logic
// of the HIP kernels) with g++ and drives it with plain loops, so the branchy parts (read-correction
// DFS, exact-cover emission, key compare/LCP, push_front_drop) can be checked against the oracle and
// run under sanitizers on the CPU.  Not part of libbgseq.so; never a fallback.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../biograph_amd/csrc/bg_device_funcs.h"

using namespace bg;

static int code_of(char c) {
  uint32_t x = (uint32_t(uint8_t(c)) >> 1) & 3;
  return int(x ^ (x >> 1));
}

struct Packed {
  int W = 1;
  std::vector<uint64_t> words;
  std::vector<uint32_t> nmask;
  std::vector<uint16_t> len;
};

static Packed pack(const char* seq, const uint64_t* offs, uint64_t n) {
  Packed p;
  unsigned mx = 0;
  for (uint64_t i = 0; i < n; ++i) mx = std::max<unsigned>(mx, unsigned(offs[i + 1] - offs[i]));
  p.W = std::max(1u, (mx + 31) / 32);
  p.words.assign(n * p.W, 0);
  p.nmask.assign(n * p.W, 0);
  p.len.resize(n);
  for (uint64_t i = 0; i < n; ++i) {
    int L = int(offs[i + 1] - offs[i]);
    p.len[i] = uint16_t(L);
    for (int j = 0; j < L; ++j) {
      char c = seq[offs[i] + j];
      if (c == 'N') p.nmask[i * p.W + (j >> 5)] |= 1u << (31 - (j & 31));
      else p.words[i * p.W + (j >> 5)] |= uint64_t(code_of(c)) << (62 - 2 * (j & 31));
    }
  }
  return p;
}

static void build_table(const uint64_t* keys, uint64_t n, std::vector<uint64_t>& slots, int& lg,
                        std::vector<uint32_t>* vals = nullptr, const uint32_t* valsrc = nullptr,
                        std::vector<uint32_t>* vals2 = nullptr, const uint32_t* valsrc2 = nullptr) {
  lg = 10;
  while ((1ull << lg) < 2 * n) ++lg;
  slots.assign(1ull << lg, kEmptySlot);
  if (vals) vals->assign(1ull << lg, 0);
  if (vals2) vals2->assign(1ull << lg, 0);
  uint64_t mask = (1ull << lg) - 1;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t s = hash_slot(keys[i], lg);
    while (slots[s] != kEmptySlot) s = (s + 1) & mask;
    slots[s] = keys[i];
    if (vals) (*vals)[s] = valsrc[i];
    if (vals2) (*vals2)[s] = valsrc2[i];
  }
}

extern "C" {

// Read correction through frc_correct_read.  out_seq capacity = offs[n].
int hh_correct_reads(const uint64_t* kmers, uint64_t nk, int k, const char* seq, const uint64_t* offs, uint64_t n,
                     uint32_t max_corrections, uint32_t min_good_run, float trim, char* out_seq, uint64_t* out_offs,
                     uint8_t* keep, uint64_t* ncorr_total, int use_presence) {
  std::vector<uint64_t> slots;
  int lg;
  build_table(kmers, nk, slots, lg);
  KmerTable tab{slots.data(), lg, k};
  Packed p = pack(seq, offs, n);
  uint64_t pos = 0, nc = 0;
  for (uint64_t i = 0; i < n; ++i) {
    out_offs[i] = pos;
    uint64_t ow[kMaxWords];
    int ncorr = 0;
    int L = p.len[i];
    int needed = int(unsigned(double(trim) * double(L)));
    uint64_t present[4] = {0, 0, 0, 0};
    if (use_presence) {  // what k_correct_screen hands to the DFS kernel
      for (int j = 0; j + k <= L; ++j) {
        bool ok = (nwindow32(&p.nmask[i * p.W], p.W, j) >> (32 - k)) == 0 &&
                  tab.has(window32(&p.words[i * p.W], p.W, j) >> (64 - 2 * k));
        if (ok) present[j >> 6] |= 1ull << (j & 63);
      }
    }
    int clen;
    if (use_presence == 2) {
      // the routed search of the sharded build (RoutedTable): re-run with the probes gathered so far until a run
      // misses nothing; a probe's masks are what the owners of its k-mers would answer
      const uint64_t* w = &p.words[i * p.W];
      const uint32_t* nm = &p.nmask[i * p.W];
      std::vector<ProbeDesc> pool;
      uint32_t head = kNoProbe;
      clen = 0;
      for (int round = 0;; ++round) {
        if (round > 4000) return -7;
        RoutedTable rt;
        rt.pool = pool.data();
        rt.head = head;
        rt.k = k;
        rt.n_new = 0;
        rt.missed = false;
        clen = frc_correct_read(rt, w, nm, L, p.W, k, int(max_corrections), int(min_good_run), needed, ow, &ncorr, present);
        if (!rt.missed) break;
        if (rt.n_new == 0) return -8;  // a miss always leaves a new probe
        const int anchor = frc_anchor(present, L, k);
        if (anchor < 0) return -9;
        for (int q = 0; q < rt.n_new; ++q) {
          const int view = int((rt.newmeta[q] >> 16) & 1u), ppos = int(rt.newmeta[q] & 255u);
          ReadView in = frc_view(w, nm, L, p.W, k, present, anchor, view);
          const int run = probe_run(in, ppos, k);
          ProbeDesc d;
          d.q = rt.newq[q];
          d.meta = uint32_t(ppos) | (uint32_t(run) << 8) | (uint32_t(view) << 16);
          d.next = head;
          for (int b = 0; b < 4; ++b) {
            d.mask[b] = 0;
            uint64_t str[2];
            probe_string(in, d.q, ppos, b, run, k, str);
            for (int t = 0; t <= run; ++t) {
              const uint64_t z = probe_window(in, d.q, ppos, b, t, k);
              if ((window32(str, 2, t) >> (64 - 2 * k)) != z) return -10;  // the string's windows are the probe's k-mers
              if (tab.has(z)) d.mask[b] |= 1u << t;
            }
          }
          head = uint32_t(pool.size());
          pool.push_back(d);
        }
      }
    } else {
      clen = frc_correct_read(tab, &p.words[i * p.W], &p.nmask[i * p.W], L, p.W, k, int(max_corrections),
                              int(min_good_run), needed, ow, &ncorr, use_presence ? present : nullptr);
    }
    keep[i] = clen > 0;
    if (clen > 0) nc += ncorr;
    for (int j = 0; j < clen; ++j) out_seq[pos++] = "ACGT"[read_code(ow, j)];
  }
  out_offs[n] = pos;
  *ncorr_total = nc;
  return 0;
}

// Entry set through oriented_words / cover_count / window32 / key_compare / key_lcp.
// Returns number of entries; entries as concatenated ASCII (capacity: caller gives cap bytes).
struct HHEntries {
  std::vector<std::string> e;
  std::vector<uint32_t> shared;
  uint64_t n_records = 0;
  std::vector<uint64_t> bits[4];
  int err = 0;
};

void* hh_build_entries(const char* seq, const uint64_t* offs, uint64_t n, int K) {
  Packed p = pack(seq, offs, n);
  const int W = p.W;
  const uint64_t n_o = 2 * n;
  std::vector<uint64_t> owords(n_o * W);
  std::vector<uint8_t> olen(n_o);
  std::vector<std::pair<uint64_t, uint32_t>> keyed;
  for (uint64_t o = 0; o < n_o; ++o) {
    uint64_t tmp[kMaxWords];
    int L = p.len[o >> 1];
    oriented_words(&p.words[(o >> 1) * W], L, W, (o & 1) != 0, tmp);
    for (int w = 0; w < W; ++w) owords[o * W + w] = L ? tmp[w] : 0;
    olen[o] = uint8_t(L);
    if (L >= K) keyed.emplace_back(tmp[0] >> (64 - 2 * K), uint32_t(o));
  }
  // candidates in full-key order (which implies first-K-mer order): cover_probe searches a K-mer's run by key
  std::stable_sort(keyed.begin(), keyed.end(), [&](auto& a, auto& b) {
    return key_compare(&owords[uint64_t(a.second) * W], p.len[a.second >> 1], &owords[uint64_t(b.second) * W],
                       p.len[b.second >> 1], W) < 0;
  });
  std::vector<uint64_t> skeys, ukeys;
  std::vector<uint32_t> soid, upos, ulen;
  for (size_t i = 0; i < keyed.size(); ++i) {
    skeys.push_back(keyed[i].first);
    soid.push_back(keyed[i].second);
    if (i == 0 || keyed[i].first != keyed[i - 1].first) { ukeys.push_back(keyed[i].first); upos.push_back(uint32_t(i)); ulen.push_back(0); }
    ++ulen.back();
  }
  // the cover table: the k-mer table's buckets, one value word per slot (csrc/bg_device_funcs.h CoverIndex)
  int lg = 10;
  while ((1ull << lg) < 2 * ukeys.size()) ++lg;
  std::vector<uint64_t> slots(1ull << lg, kEmptySlot);
  for (size_t i = 0; i < ukeys.size(); ++i) {
    uint64_t t = hash_slot(ukeys[i], lg);
    while (slots[t] != kEmptySlot) t = (t + 1) & ((1ull << lg) - 1);  // first free slot along the chain of buckets
    const bool single = ulen[i] == 1;
    const uint32_t c = soid[upos[i]];
    // one word per slot: fingerprint | length byte | single flag | candidate or run start (csrc/bg_device_funcs.h cover_slot)
    slots[t] = single ? cover_slot(ukeys[i], p.len[c >> 1], true, c) : cover_slot(ukeys[i], ulen[i] < 255 ? ulen[i] : 255, false, upos[i]);
  }
  // reads shorter than K cannot act as a cover: they are simply absent
  CoverIndex ci{slots.data(), lg, skeys.data(), soid.data(), uint64_t(skeys.size()), K, olen.data()};
  struct Rec { uint64_t w[kMaxWords]; int len; };
  std::vector<Rec> recs;
  for (uint64_t o = 0; o < n_o; ++o) {
    int c = cover_count(ci, owords.data(), o, W);
    int L = p.len[o >> 1];
    for (int j = 0; j < c; ++j) {
      Rec r;
      for (int w = 0; w < W; ++w) r.w[w] = window32(&owords[o * W], W, j + 32 * w);
      r.len = L - j;
      recs.push_back(r);
    }
  }
  auto* out = new HHEntries;
  out->n_records = recs.size();
  std::sort(recs.begin(), recs.end(), [W](const Rec& a, const Rec& b) { return key_compare(a.w, a.len, b.w, b.len, W) < 0; });
  std::vector<Rec> ent;
  std::vector<int> lcp_prev(recs.size(), 0);
  for (size_t i = 1; i < recs.size(); ++i) lcp_prev[i] = key_lcp(recs[i - 1].w, recs[i - 1].len, recs[i].w, recs[i].len, W);
  int run_min = 1 << 30;
  for (size_t i = 0; i < recs.size(); ++i) {
    run_min = std::min(run_min, lcp_prev[i]);
    bool drop = i + 1 < recs.size() && key_lcp(recs[i].w, recs[i].len, recs[i + 1].w, recs[i + 1].len, W) == recs[i].len;
    if (drop) continue;
    ent.push_back(recs[i]);
    out->shared.push_back(ent.size() == 1 ? 0 : uint32_t(run_min));
    run_min = 1 << 30;
  }
  for (auto& r : ent) {
    std::string s;
    for (int j = 0; j < r.len; ++j) s.push_back("ACGT"[read_code(r.w, j)]);
    out->e.push_back(s);
  }
  // prev bits with the same search as k_prev_bits
  size_t ne = ent.size();
  for (int b = 0; b < 4; ++b) out->bits[b].assign((ne + 63) / 64, 0);
  for (size_t i = 0; i < ne; ++i) {
    uint64_t tail[kMaxWords];
    for (int w = 0; w < W; ++w) tail[w] = window32(ent[i].w, W, 1 + 32 * w);
    int tl = ent[i].len - 1;
    size_t lo = 0, hi = ne;
    while (lo < hi) {
      size_t mid = (lo + hi) / 2;
      if (key_compare(ent[mid].w, ent[mid].len, tail, tl, W) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo >= ne || key_lcp(ent[lo].w, ent[lo].len, tail, tl, W) < tl) { out->err = 2; continue; }
    out->bits[ent[i].w[0] >> 62][lo / 64] |= 1ull << (lo & 63);
  }
  return out;
}
uint64_t hh_entries_count(void* h) { return static_cast<HHEntries*>(h)->e.size(); }
uint64_t hh_entries_records(void* h) { return static_cast<HHEntries*>(h)->n_records; }
int hh_entries_err(void* h) { return static_cast<HHEntries*>(h)->err; }
uint64_t hh_entries_total(void* h) { uint64_t t = 0; for (auto& s : static_cast<HHEntries*>(h)->e) t += s.size(); return t; }
void hh_entries_get(void* h, char* seq, uint64_t* offs, uint32_t* shared, uint64_t* bA, uint64_t* bC, uint64_t* bG,
                    uint64_t* bT) {
  auto* en = static_cast<HHEntries*>(h);
  uint64_t pos = 0;
  for (size_t i = 0; i < en->e.size(); ++i) {
    offs[i] = pos;
    memcpy(seq + pos, en->e[i].data(), en->e[i].size());
    pos += en->e[i].size();
    shared[i] = en->shared[i];
  }
  offs[en->e.size()] = pos;
  uint64_t* dst[4] = {bA, bC, bG, bT};
  for (int b = 0; b < 4; ++b) memcpy(dst[b], en->bits[b].data(), en->bits[b].size() * 8);
}
void hh_entries_free(void* h) { delete static_cast<HHEntries*>(h); }

// push_front_drop through SeqsetView (sizes/shared as bytes, rank index supplied by the caller).
void hh_push_front_drop(uint64_t n, const uint8_t* sizes, const uint8_t* shared, const uint64_t* const* bits,
                        const uint64_t* const* subaccum, const uint64_t* const* accum, const uint64_t* fixed,
                        uint64_t nq, const uint64_t* begin, const uint64_t* end, const uint32_t* size,
                        const uint8_t* base, const uint32_t* min_ctx, uint64_t* ob, uint64_t* oe, uint32_t* os) {
  SeqsetView v;
  v.n = n;
  v.sizes = sizes;
  v.shared = shared;
  for (int b = 0; b < 4; ++b) { v.bits[b] = bits[b]; v.subaccum[b] = subaccum[b]; v.accum[b] = accum[b]; }
  for (int b = 0; b < 5; ++b) v.fixed[b] = fixed[b];
  uint64_t n1 = (n + 63) / 64, n2 = (n1 + 63) / 64;
  std::vector<uint8_t> m1(n1 + 1, 255), m2(n2 + 1, 255);
  for (uint64_t i = 0; i < n; ++i) m1[i / 64] = std::min(m1[i / 64], shared[i]);
  for (uint64_t i = 0; i < n1; ++i) m2[i / 64] = std::min(m2[i / 64], m1[i]);
  v.min1 = m1.data();
  v.min2 = m2.data();
  for (uint64_t q = 0; q < nq; ++q)
    v.push_front_drop(begin[q], end[q], size[q], base[q], min_ctx[q], &ob[q], &oe[q], &os[q]);
}

}  // extern "C"
