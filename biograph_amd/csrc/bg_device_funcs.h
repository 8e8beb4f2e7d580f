// bg_device_funcs.h — per-thread logic of the HIP kernels, written as host+device inline functions.
//
// Under hipcc these compile into the gfx950 kernels (build_*.hip, walk.hip).  The same header is
// compiled with g++ by tests/host_harness.cpp ONLY to unit-test the branchy per-thread logic
// against the oracle on the CPU build (sanitizers are not available on the GPU pool).  The product
// library never executes the host instantiation.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BG_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define BG_HD inline
#endif

namespace bg {

constexpr int kMaxWords = 8;         // 8 x 32 = 256 bases >= 255 (biograph_create.cpp:134-139)
constexpr int kMaxCorrections = 8;   // explicit DFS stack depth (fast_read_correct.cpp recursion)
constexpr uint64_t kEmptySlot = ~0ull;
constexpr uint64_t kKmerMask62 = (1ull << 62) - 1;  // kmer_count_table.h:17
constexpr uint64_t kFwdStartsBit = 1ull << 62;      // stored in hash slot: fwd_starts_read
constexpr uint64_t kRevStartsBit = 1ull << 63;      // stored in hash slot: rev_starts_read

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2; MI355X_MICROARCH.md).  Maps
// physical workgroup p of an nb-workgroup grid to a logical chunk so that each XCD walks one contiguous
// eighth of the work.  Only locality depends on it.
// first word of the sort item of an oriented read too short to be a cover (items_build.hip: k_read_items, k_oriented_words)
BG_HD uint64_t cover_item_scatter(uint64_t o) { return (o + 1) * 0x9E3779B97F4A7C15ull; }
BG_HD uint64_t xcd_chunk(uint64_t p, uint64_t nb) {
  uint64_t q = nb >> 3;
  return p < (q << 3) ? (p & 7) * q + (p >> 3) : p;
}

// ---- 2-bit packed reads.  Read i owns words[i*W .. i*W+W): base j sits in word j/32 at bits
// (62-2*(j%32)) (first base in the high bits, dna_sequence.h:97 / kmer.h:30-38), zero (=A) padded.
// nmask[i*W + j/32] bit (31-j%32) is set where the input had 'N' (the base is then stored as A).
BG_HD int read_code(const uint64_t* words, int p) {
  return int((words[p >> 5] >> (62 - 2 * (p & 31))) & 3);
}
BG_HD int read_isn(const uint32_t* nmask, int p) { return int((nmask[p >> 5] >> (31 - (p & 31))) & 1); }

// 32 bases starting at base offset j of a packed word string of nw words (zero padded past end).
BG_HD uint64_t window32(const uint64_t* words, int nw, int j) {
  int q = j >> 5, r = j & 31;
  uint64_t w0 = q < nw ? words[q] : 0;
  if (r == 0) return w0;
  uint64_t w1 = (q + 1) < nw ? words[q + 1] : 0;
  return (w0 << (2 * r)) | (w1 >> (64 - 2 * r));
}
BG_HD uint32_t nwindow32(const uint32_t* nmask, int nw, int j) {
  int q = j >> 5, r = j & 31;
  uint32_t w0 = q < nw ? nmask[q] : 0;
  if (r == 0) return w0;
  uint32_t w1 = (q + 1) < nw ? nmask[q + 1] : 0;
  return (w0 << r) | (w1 >> (32 - r));
}

// Reverse complement of 32 packed bases (A<->T, C<->G = 3-b; dna_base.h:28).
BG_HD uint64_t revcomp64(uint64_t x) {
  x = ~x;
#if defined(__HIP_DEVICE_COMPILE__)
  x = __brevll(x);
#else
  x = ((x >> 32) | (x << 32));
  x = ((x & 0xffff0000ffff0000ull) >> 16) | ((x & 0x0000ffff0000ffffull) << 16);
  x = ((x & 0xff00ff00ff00ff00ull) >> 8) | ((x & 0x00ff00ff00ff00ffull) << 8);
  x = ((x & 0xf0f0f0f0f0f0f0f0ull) >> 4) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
  x = ((x & 0xccccccccccccccccull) >> 2) | ((x & 0x3333333333333333ull) << 2);
  x = ((x & 0xaaaaaaaaaaaaaaaaull) >> 1) | ((x & 0x5555555555555555ull) << 1);
#endif
  // the full bit reversal also reversed the two bits inside every base: swap them back
  return ((x & 0xaaaaaaaaaaaaaaaaull) >> 1) | ((x & 0x5555555555555555ull) << 1);
}
// k-mer (right-aligned, kmer.h) reverse complement: dna_sequence.cpp:330-333.
BG_HD uint64_t kmer_revcomp(uint64_t kmer, int k) { return revcomp64(kmer) >> (64 - 2 * k); }
// dna_sequence.cpp:341-351
BG_HD uint64_t kmer_canon(uint64_t kmer, int k, bool* flipped) {
  uint64_t rc = kmer_revcomp(kmer, k);
  *flipped = rc < kmer;
  return rc < kmer ? rc : kmer;
}
BG_HD uint64_t kmer_shift(uint64_t kmer, int k, int b) {  // fast_read_correct.cpp:5-11
  return ((kmer << 2) | uint64_t(b)) & ((k < 32) ? ~(~0ull << (2 * k)) : ~0ull);
}

// ---- k-mer hash set (open addressing, linear probe; role of kmer_set::find_table_index,
// bio_mapred/kmer_set.h:60-149).  Slot = canonical k-mer in the low 62 bits + two flag bits.
// Probing is linear from a line-aligned start, i.e. a scan of 8-slot (64-byte) buckets: a lookup costs
// one memory round trip unless its bucket is full, and a 64-byte line is what a probe drags from HBM
// anyway.  (The dependent round trips of slot-at-a-time probing, taken by the slowest of a wave's 64
// lanes, were the larger half of the read-correction screen's time.)
constexpr int kBucketSlots = 8;  // (4-slot buckets, two loads per probe: k_cover_count 8.5 -> 9.1 ms, k_correct_dfs 4.5 -> 4.7)
BG_HD uint64_t hash_slot(uint64_t canon, int log2cap) {
  return ((canon * 0x9E3779B97F4A7C15ull) >> (64 - log2cap)) & ~uint64_t(kBucketSlots - 1);
}
// Inserts take the first bucket along the chain that has a free slot, any free slot of it (they start at a
// hashed offset so that concurrent inserts do not all fight for slot 0).
BG_HD int hash_slot_offset(uint64_t canon, int log2cap) {
  return int(((canon * 0x9E3779B97F4A7C15ull) >> (64 - log2cap)) & uint64_t(kBucketSlots - 1));
}
// Slot holding `key` (compared under keymask) or -1.  Tables have at least kBucketSlots slots.
BG_HD int64_t table_find(const uint64_t* slots, int log2cap, uint64_t key, uint64_t keymask) {
  const uint64_t mask = (1ull << log2cap) - 1;
  uint64_t s = hash_slot(key, log2cap);
  for (;;) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(slots + s);
    uint64_t v[kBucketSlots];
#pragma unroll
    for (int u = 0; u < kBucketSlots / 2; ++u) {  // all the 16-byte loads of the bucket in flight together
      const ulonglong2 x = p[u];
      v[2 * u] = x.x;
      v[2 * u + 1] = x.y;
    }
    uint32_t hit = 0, empty = 0;
#pragma unroll
    for (int u = 0; u < kBucketSlots; ++u) {
      hit |= uint32_t((v[u] & keymask) == key && v[u] != kEmptySlot) << u;
      empty |= uint32_t(v[u] == kEmptySlot) << u;
    }
    if (hit) return int64_t(s) + (__ffs(int(hit)) - 1);
    if (empty) return -1;
#else
    bool empty = false;
    for (int u = 0; u < kBucketSlots; ++u) {
      uint64_t v = slots[s + u];
      if (v == kEmptySlot) empty = true;
      else if ((v & keymask) == key) return int64_t(s + u);
    }
    if (empty) return -1;
#endif
    s = (s + kBucketSlots) & mask;
  }
}
struct KmerTable {
  const uint64_t* slots;
  int log2cap;
  int k;
  // Returns slot index or -1.
  BG_HD int64_t find(uint64_t canon) const { return table_find(slots, log2cap, canon, kKmerMask62); }
  BG_HD bool has(uint64_t kmer) const {  // correct_reads.cpp:163-171 kmer_lookup_f
    bool fl;
    return find(kmer_canon(kmer, k, &fl)) >= 0;
  }
  // the search says where a look-up happens (view, virtual index of the k-mer's newest base); a table that
  // holds the whole set does not care
  template <class View>
  BG_HD bool has_at(uint64_t kmer, const View&, int) const { return has(kmer); }
};

// ---- read correction (fast_read_correct.cpp:16-182 + correct_reads.cpp:154-178).
// A corrected read equals the input except at substituted positions, so a DFS result is just
// (length, substitutions).
struct FrcRes {
  int len;
  int ncorr;
  // substitution i: position in bits 8i..8i+7 of spos, base in bits 2i..2i+1 of sbase (packed so that a
  // result stays in registers: arrays indexed by a run-time count would live in scratch memory)
  uint64_t spos;
  uint32_t sbase;
  BG_HD int pos_of(int i) const { return int((spos >> (8 * i)) & 255); }
  BG_HD int base_of(int i) const { return int((sbase >> (2 * i)) & 3); }
};

struct ReadView {
  const uint64_t* words;
  const uint32_t* nmask;
  int origin;  // original-read position of virtual index 0
  int dir;     // +1: left-to-right; -1: reverse complement walking left
  int n;       // virtual length
  int k;
  int W;       // words per read
  // Optional: bit j set <=> the UNMODIFIED read's window starting at base j holds no N and its k-mer is
  // in the set (filled by the screening kernel).  Lets the search skip look-ups of windows no
  // substitution has touched; null = always look up.
  const uint64_t* present;
  BG_HD int get(int t) const {  // 0..3, or 4 for N
    int p = dir > 0 ? origin + t : origin - t;
    if (read_isn(nmask, p)) return 4;
    int c = read_code(words, p);
    return dir > 0 ? c : 3 - c;
  }
  // presence of the original window whose newest base is virtual index t
  BG_HD bool window_present(int t) const {
    int j = dir > 0 ? origin + t - k + 1 : origin - t;  // window start in read coordinates
    return (present[j >> 6] >> (j & 63)) & 1;
  }
  // number of consecutive original windows present from virtual index t on (at most `limit`): what the
  // extension loop would find one look-up at a time, by bit scans
  BG_HD int present_run(int t, int limit) const {
    const int j = dir > 0 ? origin + t - k + 1 : origin - t;
    int run = 0;
    while (run < limit) {
      int ones, avail;
      if (dir > 0) {
        const int jj = j + run;
        avail = 64 - (jj & 63);
        const uint64_t inv = ~(present[jj >> 6] >> (jj & 63));
#if defined(__HIP_DEVICE_COMPILE__)
        ones = inv ? __ffsll((long long)inv) - 1 : 64;
#else
        ones = inv ? __builtin_ctzll(inv) : 64;
#endif
      } else {
        const int jj = j - run;
        if (jj < 0) break;
        avail = (jj & 63) + 1;
        const uint64_t inv = ~(present[jj >> 6] << (63 - (jj & 63)));
#if defined(__HIP_DEVICE_COMPILE__)
        ones = inv ? __clzll((long long)inv) : 64;
#else
        ones = inv ? __builtin_clzll(inv) : 64;
#endif
      }
      if (ones > avail) ones = avail;
      run += ones;
      if (ones < avail) break;
    }
    return run < limit ? run : limit;
  }
  // the t (>= 1) original bases at virtual indices idx0 .. idx0 + t - 1, right-aligned (no N among them)
  BG_HD uint64_t own_bases(int idx0, int t) const {
    if (dir > 0) return window32(words, W, origin + idx0) >> (64 - 2 * t);
    return revcomp64(window32(words, W, origin - idx0 - t + 1) >> (64 - 2 * t)) >> (64 - 2 * t);
  }
  // k-mer (in the view's orientation) of the k original bases ending at virtual index t
  BG_HD uint64_t kmer_ending_at(int t) const {
    if (dir > 0) return window32(words, W, origin + t - k + 1) >> (64 - 2 * k);
    return kmer_revcomp(window32(words, W, origin - t) >> (64 - 2 * k), k);
  }
};

// ---- the k-mer set of a sharded build stays with the ranks that counted it (kmer_counter.h:167-178: a
// k-mer lives in one partition); the search of a read runs where the read is.  What it looks up beyond the
// read's own windows (those come as presence bits) are the windows around a substituted base.  They are
// fetched a *probe* at a time: a probe (q, pos, view) stands for the 4 x (run + 1) k-mers that end at virtual
// indices pos .. pos + run of the view when the base at pos is replaced by b = 0..3, the bases before it are
// the first k - 1 bases of q and the bases after it are the read's own (run < k, cut at an N or the read's
// end): the window the search missed, its three alternatives and the runs they start.  mask[b] bit t = the
// k-mer ending at pos + t is kept.  The search is re-run from scratch with the probes it has (a look-up no
// probe answers is a miss: it is noted as a new probe and taken as "absent"); a run without a miss made
// exactly the look-ups of the search over the whole set, with the same answers.
struct ProbeDesc {
  uint64_t q;
  uint32_t mask[4];
  uint32_t meta;  // pos : 8 | run : 8 | view : 1 (bit 16: the left view)
  uint32_t next;  // older probe of the same read, ~0u = none
};
constexpr uint32_t kNoProbe = ~0u;
constexpr int kMaxNewProbes = 4;  // per read and round

struct RoutedTable {
  const ProbeDesc* pool;
  uint32_t head;
  int k;
  mutable uint64_t newq[kMaxNewProbes];
  mutable uint32_t newmeta[kMaxNewProbes];
  mutable int n_new;
  mutable bool missed;
  BG_HD bool has(uint64_t) const { missed = true; return false; }  // only reached without presence bits: never
  template <class View>
  BG_HD bool has_at(uint64_t kmer, const View& in, int pos) const {
    const uint32_t view = in.dir < 0 ? 1u : 0u;
    const uint64_t kmask = ~(~0ull << (2 * k));
    for (uint32_t d = head; d != kNoProbe; d = pool[d].next) {
      const uint32_t meta = pool[d].meta;
      if (((meta >> 16) & 1u) != view) continue;
      const int t = pos - int(meta & 255u);
      if (t < 0 || t > int((meta >> 8) & 255u)) continue;
      const uint32_t b = uint32_t(kmer >> (2 * t)) & 3u;
      uint64_t z = (pool[d].q & ~3ull) | b;
      if (t) z = ((z << (2 * t)) | in.own_bases(int(meta & 255u) + 1, t)) & kmask;
      if (z == kmer) return (pool[d].mask[b] >> t) & 1u;
    }
    missed = true;
    const uint32_t m = uint32_t(pos) | (view << 16);
    for (int i = 0; i < n_new; ++i)
      if (newmeta[i] == m && (newq[i] >> 2) == (kmer >> 2)) return false;
    if (n_new < kMaxNewProbes) { newq[n_new] = kmer; newmeta[n_new] = m; ++n_new; }
    return false;
  }
};

// correct_internal (fast_read_correct.cpp:16-90) with the recursion unrolled onto an explicit
// stack.  `start`: first virtual index to consume; `kmer`: k-mer of the bases before it.
template <class Table>
BG_HD void frc_solve(const Table& tab, const ReadView& in, int start0, uint64_t kmer0, int k,
                     int min_good_run_top, int min_good_run, int max_corrections, bool require_run_at_end,
                     FrcRes* out) {
  // The frame being worked on lives in locals (registers); the array below is touched only when the
  // search descends into a substitution or returns from one.  (With the current frame addressed as
  // stack[depth] every field access went to scratch memory: 47 of the kernel's 54 GB of HBM traffic.)
  struct Frame {
    int start, pos, b, best_size, best_b;
    int last_sub;  // virtual index of the latest substitution on the path into this frame
    uint64_t kmer;
    FrcRes best;
  };
  Frame stack[kMaxCorrections + 1];
  Frame f;
  int depth = 0;
  f.start = start0;
  f.kmer = kmer0;
  f.last_sub = -(1 << 20);
  f.pos = 0;
  f.b = 0;
  f.best_size = 0;
  f.best_b = 0;
  f.best.len = 0;
  f.best.ncorr = 0;
  FrcRes ret;
  ret.len = 0;
  ret.ncorr = 0;
  bool entering = true;  // true: begin frame f; false: `ret` holds the result of f's child
  for (;;) {
    if (entering) {
      int pos = f.start;
      uint64_t cur = f.kmer;
      while (pos < in.n) {  // :24-38 extend while the next k-mer is present
        if (in.present && pos - f.last_sub >= k) {
          // from here to the end of the run every window is one of the read's own: skip ahead by bit scan
          int run = in.present_run(pos, in.n - pos);
          if (run > 0) {
            pos += run;
            cur = in.kmer_ending_at(pos - 1);
          }
          break;  // the end of the read, or an absent window (or an N): what the loop below would stop at
        }
        int c = in.get(pos);
        if (c == 4) break;
        uint64_t nk = kmer_shift(cur, k, c);
        // a window no substitution reaches (>= k bases past the last one) is the read's own window
        bool has = (in.present && pos - f.last_sub >= k) ? in.window_present(pos) : tab.has_at(nk, in, pos);
        if (!has) break;
        cur = nk;
        ++pos;
      }
      int runlen = pos - f.start;
      int mgr = depth == 0 ? min_good_run_top : min_good_run;
      int maxc = max_corrections - depth;
      if (pos == in.n || runlen < mgr || maxc == 0) {  // :31-33, :41-47
        ret.len = runlen;
        ret.ncorr = 0;
        if (depth == 0) break;
        f = stack[--depth];
        entering = false;
        continue;
      }
      f.pos = pos;
      f.kmer = cur;
      f.b = 0;
      f.best_size = 0;
      f.best_b = 0;
      f.best.len = 0;
      f.best.ncorr = 0;
    } else {
      // child of candidate base f.b returned in `ret` (:67-77)
      if (!(require_run_at_end && ret.len < min_good_run)) {
        if (ret.len >= f.best_size) {
          f.best_size = ret.len + 1;
          f.best_b = f.b;
          f.best = ret;
        }
      }
      f.b++;
    }
    // try loop (:55-78)
    bool descended = false;
    while (f.b < 4) {
      uint64_t tk = kmer_shift(f.kmer, k, f.b);
      if (!tab.has_at(tk, in, f.pos)) { f.b++; continue; }
      if (f.pos + 1 < in.n) {
        stack[depth++] = f;
        int child_start = f.pos + 1, child_last = f.pos;
        f.start = child_start;
        f.kmer = tk;
        f.last_sub = child_last;
        entering = true;
        descended = true;
        break;
      }
      // nothing left after the substituted base: empty child
      if (!(require_run_at_end && 0 < min_good_run)) {
        if (0 >= f.best_size) {
          f.best_size = 1;
          f.best_b = f.b;
          f.best.len = 0;
          f.best.ncorr = 0;
        }
      }
      f.b++;
    }
    if (descended) continue;
    // :80-87 combine
    int runlen = f.pos - f.start;
    if (f.best_size) {
      ret.len = runlen + 1 + f.best.len;
      ret.ncorr = 1 + f.best.ncorr;
      ret.spos = (f.best.ncorr ? f.best.spos << 8 : 0ull) | uint64_t(f.pos & 255);
      ret.sbase = (f.best.ncorr ? f.best.sbase << 2 : 0u) | uint32_t(f.best_b & 3);
    } else {
      ret.len = runlen;
      ret.ncorr = 0;
    }
    if (depth == 0) break;
    f = stack[--depth];
    entering = false;
  }
  *out = ret;
}

// fast_read_correct (fast_read_correct.cpp:94-182) + the keep/drop rule of correct_reads::correct
// (correct_reads.cpp:154-178).  Writes the corrected read (W words, zero padded) and returns its
// length, or 0 if the read is dropped.  needed_good_bases = unsigned(double(float(trim))*len),
// precomputed on the host per length (`needed`).  *ncorr_out gets the number of substitutions.
template <class Table>
BG_HD int frc_correct_read(const Table& tab, const uint64_t* words, const uint32_t* nmask, int len, int W, int k,
                           int max_corrections, int min_good_run, int needed, uint64_t* out_words,
                           int* ncorr_out, const uint64_t* present = nullptr) {
  *ncorr_out = 0;
  if (len < k) return 0;
  // :107-122 find the first position whose k-mer is present (N resets the window)
  int it = 0;
  uint64_t kmer = 0;
  if (present) {  // = the lowest set presence bit
    int nwin = len - k + 1, j = -1;
    for (int w = 0; w * 64 < nwin && j < 0; ++w) {
      uint64_t bits = present[w];
      if (bits) {
#if defined(__HIP_DEVICE_COMPILE__)
        j = w * 64 + (__ffsll((long long)bits) - 1);
#else
        j = w * 64 + __builtin_ctzll(bits);
#endif
      }
    }
    if (j < 0 || j >= nwin) return 0;
    it = j + k;
    kmer = window32(words, W, j) >> (64 - 2 * k);
  } else {
    int left = k;
    for (;;) {
      if (left == 0 && tab.has(kmer)) break;
      if (it == len) return 0;
      if (read_isn(nmask, it)) { ++it; left = k; continue; }
      kmer = kmer_shift(kmer, k, read_code(words, it));
      ++it;
      if (left) --left;
    }
  }
  int kmer_start = it - k;
  FrcRes lres, rres;
  lres.len = 0; lres.ncorr = 0;
  rres.len = 0; rres.ncorr = 0;
  int maxc = max_corrections;
  if (kmer_start > 0) {  // :134-167 correct the left part on the reverse complement
    ReadView lv{words, nmask, kmer_start - 1, -1, kmer_start, k, W, present};
    frc_solve(tab, lv, 0, kmer_revcomp(kmer, k), k, 0, min_good_run, maxc, false, &lres);
    if (lres.len != kmer_start) return 0;
    maxc -= lres.ncorr;
  }
  if (it != len) {  // :169-172
    ReadView rv{words, nmask, it, +1, len - it, k, W, present};
    frc_solve(tab, rv, 0, kmer, k, 0, min_good_run, maxc, true, &rres);
  }
  int clen = kmer_start + k + rres.len;
  if (clen < needed || clen < k) return 0;  // correct_reads.cpp:174-178
  for (int w = 0; w < W; ++w) out_words[w] = words[w];
  for (int i = 0; i < lres.ncorr; ++i) {
    int p = kmer_start - 1 - lres.pos_of(i);
    int b = 3 - lres.base_of(i);
    int sh = 62 - 2 * (p & 31);
    out_words[p >> 5] = (out_words[p >> 5] & ~(3ull << sh)) | (uint64_t(b) << sh);
  }
  for (int i = 0; i < rres.ncorr; ++i) {
    int p = it + rres.pos_of(i);
    int sh = 62 - 2 * (p & 31);
    out_words[p >> 5] = (out_words[p >> 5] & ~(3ull << sh)) | (uint64_t(rres.base_of(i)) << sh);
  }
  // N positions inside the kept part were all substituted (a run never crosses an N), but the
  // stored base under an N is A: nothing else to fix.  Truncate + zero pad.
  for (int w = 0; w < W; ++w) {
    int lo = w * 32;
    if (clen <= lo) out_words[w] = 0;
    else if (clen < lo + 32) out_words[w] &= ~0ull << (2 * (lo + 32 - clen));
  }
  *ncorr_out = lres.ncorr + rres.ncorr;
  return clen;
}

// ---- probes of the routed search (RoutedTable above): the two views of a read and the k-mers of a probe.
// first window whose k-mer is kept (the anchor of fast_read_correct.cpp:107-122), or -1
BG_HD int frc_anchor(const uint64_t* present, int len, int k) {
  const int nwin = len - k + 1;
  for (int w = 0; w * 64 < nwin; ++w) {
    const uint64_t bits = present[w];
    if (bits) {
#if defined(__HIP_DEVICE_COMPILE__)
      const int j = w * 64 + (__ffsll((long long)bits) - 1);
#else
      const int j = w * 64 + __builtin_ctzll(bits);
#endif
      return j < nwin ? j : -1;
    }
  }
  return -1;
}
// view 0: right of the anchor k-mer, left to right; view 1: left of it, on the reverse complement
// (exactly the views frc_correct_read builds)
BG_HD ReadView frc_view(const uint64_t* words, const uint32_t* nmask, int len, int W, int k, const uint64_t* present,
                        int anchor, int view) {
  if (view == 0) return ReadView{words, nmask, anchor + k, +1, len - (anchor + k), k, W, present};
  return ReadView{words, nmask, anchor - 1, -1, anchor, k, W, present};
}
// own bases that follow virtual index pos inside the k - 1 windows a substitution at pos reaches
BG_HD int probe_run(const ReadView& in, int pos, int k) {
  int run = 0;
  for (int t = 1; t < k; ++t) {
    const int idx = pos + t;
    if (idx >= in.n || in.get(idx) == 4) break;
    run = t;
  }
  return run;
}
// the k-mer of probe (q, pos) that ends at pos + t with base b at pos
BG_HD uint64_t probe_window(const ReadView& in, uint64_t q, int pos, int b, int t, int k) {
  uint64_t z = (q & ~3ull) | uint64_t(b);
  if (t) z = ((z << (2 * t)) | in.own_bases(pos + 1, t)) & ~(~0ull << (2 * k));
  return z;
}
// the k + run bases whose windows are probe_window(.., t), t = 0..run, as two left-aligned words
BG_HD void probe_string(const ReadView& in, uint64_t q, int pos, int b, int run, int k, uint64_t out[2]) {
  const uint64_t head = (q & ~3ull) | uint64_t(b);  // k bases, right-aligned
  const int nb = k + run;                            // <= 2k - 1 <= 61
  uint64_t hi = head << (64 - 2 * k), lo = 0;        // left-aligned k bases
  if (run) {
    const uint64_t own = in.own_bases(pos + 1, run);  // run bases, right-aligned; they follow at base offset k
    const int room = 32 - k;                          // bases of the first word still free (>= 1)
    if (run <= room) {
      hi |= own << (2 * (room - run));
    } else {
      hi |= own >> (2 * (run - room));
      lo = own << (64 - 2 * (run - room));
    }
  }
  (void)nb;
  out[0] = hi;
  out[1] = lo;
}

// ---- oriented reads.  Oriented read o = 2*i (forward) or 2*i+1 (reverse complement) of read i.
// Materialises the W words of the oriented sequence (zero padded).
BG_HD void oriented_words(const uint64_t* words, int len, int W, bool rc, uint64_t* out) {
  if (!rc) {
    for (int w = 0; w < W; ++w) out[w] = words[w];
    return;
  }
  // reverse complement of the first `len` bases: word-reverse + per-word revcomp gives the RC of
  // the zero-padded 32*W string; shift left by the pad to re-align.
  int nw = (len + 31) >> 5;
  uint64_t tmp[kMaxWords];
  for (int w = 0; w < nw; ++w) tmp[w] = revcomp64(words[nw - 1 - w]);
  int pad = nw * 32 - len;  // bases of padding now at the FRONT (as T's)
  for (int w = 0; w < W; ++w) out[w] = w < nw ? window32(tmp, nw, w * 32 + pad) : 0;
  // bases past len came from zero-fill in window32: already A.
  int last = len & 31;
  if (last && nw <= W) out[nw - 1] &= ~0ull << (2 * (32 - last));
}

// First differing base between two 32-base words (32 if equal).
BG_HD int lcp_word(uint64_t a, uint64_t b) {
  uint64_t x = a ^ b;
  if (x == 0) return 32;
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x) >> 1;
#else
  return __builtin_clzll(x) >> 1;
#endif
}

#ifndef SUFFIX_WORDS_IN_FLIGHT
#define SUFFIX_WORDS_IN_FLIGHT 5  // k_cover_count on chr20: 1 word at a time 8.51 ms, 2: 8.32, 3: 8.28, 5: 8.16
#endif
// Does oriented sequence `a` (words aw, zero padded), from base offset j, with `rem` bases, match
// the first `rem` bases of sequence b (words bw)?
BG_HD bool suffix_is_prefix(const uint64_t* aw, int anw, int j, int rem, const uint64_t* bw) {
  // several words of b in flight at a time: a match — how every read's search ends — has to see all of them, and
  // fetched one by one behind each other's compare they are as many memory round trips
  const int full = rem >> 5, tail = rem & 31;
  const int nw = full + (tail ? 1 : 0);
  const uint64_t tail_mask = tail ? ~0ull << (2 * (32 - tail)) : ~0ull;
  for (int w = 0; w < nw; w += SUFFIX_WORDS_IN_FLIGHT) {
    uint64_t b[SUFFIX_WORDS_IN_FLIGHT];
#pragma unroll
    for (int u = 0; u < SUFFIX_WORDS_IN_FLIGHT; ++u) b[u] = w + u < nw ? bw[w + u] : 0;
#pragma unroll
    for (int u = 0; u < SUFFIX_WORDS_IN_FLIGHT; ++u) {
      if (w + u >= nw) break;
      const uint64_t m = (w + u == full) ? tail_mask : ~0ull;
      if ((window32(aw, anw, j + 32 * (w + u)) & m) != (b[u] & m)) return false;
    }
  }
  return true;
}

// ---- exact-cover emission: index of the first K bases of every oriented read.
// The look-ups of k_cover_count run into the L1 -> L2 request rate (≈64 G requests/s under load; SQ / TCP counters
// via tools/pmc_kernel.sh).  The table keeps the 64-byte buckets of the k-mer table (the four 16-byte loads of a
// bucket merge into one request; 16-byte pairs were tried and need two dependent requests per miss: 9.4 -> 11.4 ms),
// and what a hit needs next — run start or the one candidate, run length or that candidate's length — sits in one
// 8-byte word per slot instead of three arrays.
struct CoverIndex {
  const uint64_t* slots;  // hash table over the first-K-mers, 8-slot buckets; a slot is ONE word (see cover_slot) that holds what
                          // a hit needs next as well -- the key's word and a value word in a second array made every hit three
                          // dependent random sectors (slot, value, candidate), and these look-ups run at the rate this GPU
                          // serves random sectors
  int log2cap;
  const uint64_t* skeys;  // first-K-mers of the candidate reads, which are sorted by their FULL key
  const uint32_t* soid;   // oriented ids in the same order
  uint64_t n_o;           // number of candidate reads (< 2^31)
  int K;
  const uint8_t* olen;    // length of every oriented read (by oriented id)
};

// A slot: [63:40] a 24-bit fingerprint of the K-mer (never all ones: an empty slot is ~0) | [39:32] a length byte | [31] the
// K-mer's run of candidates is ONE read | [30:0] that read's oriented id, or the start of the run in skeys / soid.  Length
// byte: the one candidate's length, or min(run length, 255).  A fingerprint that matches a different K-mer costs a look at a
// candidate that then fails its check (the candidate's words for a single, skeys for a run): never a wrong answer.
constexpr uint32_t kCoverSingle = 0x80000000u;
BG_HD uint64_t cover_fingerprint(uint64_t key) {
  uint64_t f = (key * 0xD6E8FEB86659FD93ull) >> 40;
  return f == 0xffffffull ? 0xfffffeull : f;
}
BG_HD uint64_t cover_slot(uint64_t key, uint32_t len8, bool single, uint32_t id_or_pos) {
  return (cover_fingerprint(key) << 40) | (uint64_t(len8 & 0xffu) << 32) | (single ? uint64_t(kCoverSingle) : 0ull) | uint64_t(id_or_pos & 0x7fffffffu);
}

// What one slot says about suffix j (rem bases from base j on, first K-mer `key`) of the oriented read s:
// 1 = a candidate of the slot covers it, 0 = the slot holds the suffix's own K-mer and none of its candidates does,
// -1 = the slot belongs to another K-mer with the same fingerprint.
BG_HD int cover_check_slot(const CoverIndex& ci, const uint64_t* owords, const uint64_t* s, int j, int rem, int W, uint64_t key,
                           uint64_t sv) {
  const uint32_t lo32 = uint32_t(sv), len8 = uint32_t(sv >> 32) & 0xffu;
  if (lo32 & kCoverSingle) {
    const uint64_t* cw = owords + uint64_t(lo32 & ~kCoverSingle) * W;
    // (a candidate shorter than the suffix cannot cover it)
    if (int(len8) >= rem && suffix_is_prefix(s, W, j, rem, cw)) return 1;
    return (cw[0] >> (64 - 2 * ci.K)) == key ? 0 : -1;
  }
  uint64_t lo = lo32;
  if (lo >= ci.n_o || ci.skeys[lo] != key) return -1;
  uint64_t hi = lo + len8;
  if (len8 == 255) {  // a long run (a repeat family): its end is the first candidate with another K-mer
    uint64_t a = lo + 255, z = ci.n_o;
    while (a < z) {
      const uint64_t mid = (a + z) >> 1;
      if (ci.skeys[mid] == key) a = mid + 1; else z = mid;
    }
    hi = a;
  }
  const uint64_t end = hi;
  // the candidates of this K-mer are in full-key order: the first one that is not smaller than the suffix is
  // the only one that can have it as a prefix (anything between a string and one of its extensions extends it)
  if (hi - lo <= 4) {  // a handful (mates, duplicates): just try them
    for (uint64_t q = lo; q < hi; ++q) {
      const uint32_t c = ci.soid[q];
      if (int(ci.olen[c]) >= rem && suffix_is_prefix(s, W, j, rem, owords + uint64_t(c) * W)) return 1;
    }
    return 0;
  }
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    const uint32_t c = ci.soid[mid];
    const uint64_t* cw = owords + uint64_t(c) * W;
    int cmp = 0;  // candidate vs suffix in (words, length) order; the suffix's words are zero padded like any key
    for (int w = 0; w < W && cmp == 0; ++w) {
      const uint64_t a = cw[w], bb = window32(s, W, j + 32 * w);
      cmp = a < bb ? -1 : (a > bb ? 1 : 0);
    }
    if (cmp == 0) cmp = int(ci.olen[c]) < rem ? -1 : 0;
    if (cmp < 0) lo = mid + 1; else hi = mid;
  }
  if (lo >= end) return 0;
  const uint32_t c = ci.soid[lo];
  return int(ci.olen[c]) >= rem && suffix_is_prefix(s, W, j, rem, owords + uint64_t(c) * W) ? 1 : 0;
}

// first slot at or after `from` (a slot index; the chain is walked bucket by bucket) whose fingerprint is fp, or -1 at the
// chain's first empty slot.  (The bucket's words die here: the caller reads the one slot again -- kept alive through the
// candidate checks they took a quarter of k_cover_count's waves.)
BG_HD int64_t cover_find(const uint64_t* slots, int log2cap, uint64_t fp, uint64_t from) {
  const uint64_t mask = (1ull << log2cap) - 1;
  uint64_t b = from & ~uint64_t(kBucketSlots - 1);
  uint32_t skip = (1u << (from & (kBucketSlots - 1))) - 1;  // slots of the first bucket before `from`
  for (;;) {
    uint64_t v[kBucketSlots];
#if defined(__HIP_DEVICE_COMPILE__)
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(slots + b);
#pragma unroll
    for (int q = 0; q < kBucketSlots / 2; ++q) {  // all the 16-byte loads of the bucket in flight together
      const ulonglong2 x = p[q];
      v[2 * q] = x.x;
      v[2 * q + 1] = x.y;
    }
#else
    for (int q = 0; q < kBucketSlots; ++q) v[q] = slots[b + q];
#endif
    uint32_t hit = 0, empty = 0;
#pragma unroll
    for (int q = 0; q < kBucketSlots; ++q) {
      hit |= uint32_t((v[q] >> 40) == fp && v[q] != kEmptySlot) << q;
      empty |= uint32_t(v[q] == kEmptySlot) << q;
    }
    hit &= ~skip;
    if (hit) {
#if defined(__HIP_DEVICE_COMPILE__)
      return int64_t(b) + (__ffs(int(hit)) - 1);
#else
      int u = 0;
      while (!((hit >> u) & 1u)) ++u;
      return int64_t(b) + u;
#endif
    }
    if (empty) return -1;  // (an insert takes the first bucket of its chain with a free slot)
    skip = 0;
    b = (b + kBucketSlots) & mask;
  }
}

// is suffix j (>= K bases long) of the oriented read s a prefix of some oriented read?
BG_HD bool cover_probe(const CoverIndex& ci, const uint64_t* owords, const uint64_t* s, int L, int j, int W) {
  const int rem = L - j;
  const uint64_t key = window32(s, W, j) >> (64 - 2 * ci.K);
  const uint64_t fp = cover_fingerprint(key);
  uint64_t from = hash_slot(key, ci.log2cap);
  for (;;) {
    const int64_t h = cover_find(ci.slots, ci.log2cap, fp, from);
    if (h < 0) return false;
    const int r = cover_check_slot(ci, owords, s, j, rem, W, key, ci.slots[h]);
    if (r >= 0) return r != 0;
    // another K-mer behind the same fingerprint (about one look-up in a million): on along the chain
    from = uint64_t(h) + 1;
    if ((from & (kBucketSlots - 1)) == 0) from &= (1ull << ci.log2cap) - 1;
  }
}

// number of suffix records oriented read o emits (0 for a dropped read)
BG_HD int cover_count(const CoverIndex& ci, const uint64_t* owords, uint64_t o, int W) {
  int L = ci.olen[o];
  if (L == 0) return 0;
  const uint64_t* s = owords + o * W;
  int cnt = 1;
  for (int j = 1; j < L; ++j) {
    int rem = L - j;
    if (rem < ci.K) { cnt += rem; break; }  // suffixes shorter than K are never looked up: all emitted
    if (cover_probe(ci, owords, s, L, j, W)) break;
    ++cnt;
  }
  return cnt;
}

// ---- entry keys: W words (zero padded) + length; order = lexicographic with a prefix before its
// extensions == (words, len) tuple order because A=0 pads.
BG_HD int key_compare(const uint64_t* a, int alen, const uint64_t* b, int blen, int W) {
  for (int w = 0; w < W; ++w) {
    if (a[w] < b[w]) return -1;
    if (a[w] > b[w]) return 1;
  }
  return alen < blen ? -1 : (alen > blen ? 1 : 0);
}
// the same with three words of each key in flight at a time: keys that agree for long (records of one locus, a
// prefix and its extension) take fewer memory round trips (k_sort_small_groups on chr20's records: word by word
// 5.02 ms, by two 4.63, by three 4.41, all eight at once 5.22)
BG_HD int key_compare_by_three(const uint64_t* a, int alen, const uint64_t* b, int blen, int W) {
  for (int w = 0; w < W; w += 3) {
    const uint64_t a0 = a[w], b0 = b[w];
    const uint64_t a1 = w + 1 < W ? a[w + 1] : 0, b1 = w + 1 < W ? b[w + 1] : 0;
    const uint64_t a2 = w + 2 < W ? a[w + 2] : 0, b2 = w + 2 < W ? b[w + 2] : 0;
    if (a0 != b0) return a0 < b0 ? -1 : 1;
    if (a1 != b1) return a1 < b1 ? -1 : 1;
    if (a2 != b2) return a2 < b2 ? -1 : 1;
  }
  return alen < blen ? -1 : (alen > blen ? 1 : 0);
}
BG_HD int key_lcp(const uint64_t* a, int alen, const uint64_t* b, int blen, int W) {
  int m = alen < blen ? alen : blen;
  int l = 0;
  for (int w = 0; w < W; ++w) {
    int c = lcp_word(a[w], b[w]);
    l += c;
    if (c < 32) break;
  }
  return l < m ? l : m;
}

// ---- seqset walk (bio_base/seqset.cpp:319-429, io/bitcount.h:218-270, io/int_map_interface.cpp:128-197).
struct SeqsetView {
  uint64_t n;
  const uint8_t* sizes;     // entry_sizes, one byte per entry (max read length 255)
  const uint8_t* shared;
  const uint64_t* bits[4];
  const uint64_t* subaccum[4];
  const uint64_t* accum[4];
  uint64_t fixed[5];
  const uint8_t* min1;      // min of shared[] per 64 entries   (less_than_search k_factor1)
  const uint8_t* min2;      // min of shared[] per 4096 entries (k_factor2)
  // measurement only (bg_walk_counters): [0] += push_front_drop calls, [1] += iterations of its drop loop -- the N_pfd and
  // N_drop of SURVEY.md section 8(d)'s walk bytes (52 N_pfd + 60 N_drop).  Null in every normal run.
  unsigned long long* walk_counters = nullptr;
  // rank blocks (round 4): per base one 64-byte block per 384 entries = [set bits before the block | five 9-bit running
  // counts at its word boundaries | its six words of bits].  A rank look-up through bits / subaccum / accum touches three
  // arrays, i.e. three 64-byte sectors for 24 bytes (measured: the reference map fetches 264 bytes per push_front_drop);
  // through a block it touches one.  Same numbers (seqset_finalize_device builds the blocks from the finalized bitcount);
  // null where a view is built without them.
  const uint64_t* rank64[4] = {nullptr, nullptr, nullptr, nullptr};
  // sel256[b][j] = the entry that holds the (256 j)-th set bit of base b's prev bits, one more slot = n: seqset_range::pop_front
  // is a select (which entry does push_front(b) map onto `begin`?), and a binary search over all n entries is 28 dependent
  // rank look-ups; between two samples lie about a thousand entries = three rank blocks.  Null: the search over everything.
  const uint64_t* sel256[4] = {nullptr, nullptr, nullptr, nullptr};

  // rank4 (round 5): ONE 64-byte block per 96 entries for all four bases -- per base 16 bytes: [set bits before the block,
  // relative to the group of 65536 blocks it lies in | the block's 96 bits as three words].  A frontier asks for the same two
  // positions (begin, end) under four bases: with a block array per base those were eight sectors, here they are two, and one
  // base's look-up is one 16-byte load.  rank4_super[(g >> 16) * 4 + b] = the group's base count.  (rank64 stays for pop_front's
  // select, whose search walks block heads of ONE base.)
  const uint32_t* rank4 = nullptr;
  const uint64_t* rank4_super = nullptr;

  // walk blocks (round 6): ONE 128-byte line per 40 entries with everything push_front_drop reads of them -- words 0..3: per base
  // [set bits before the block, relative to the group of 65536 blocks it lies in : 24 | the block's 40 prev bits : 40]; words
  // 4..8: the 40 sizes; words 9..13: the 40 shared; words 14, 15 spare.  A step of a walk lands on a new entry, checks its size,
  // and the next step asks for the ranks at that entry and (when it drops) for shared around it: with a rank block, sizes and
  // shared those were three lines, here they are one.  wblk_super[(g >> 16) * 4 + b] = the group's base count.  Needs
  // n < 2^35 (the block of an entry by a 32-bit division); null: rank4 / sizes / shared.
  const uint64_t* wblk = nullptr;
  const uint64_t* wblk_super = nullptr;
  static BG_HD uint32_t wblk_of(uint64_t i) { return uint32_t(i >> 3) / 5u; }
  // "are there walk blocks" as a value the scalar unit holds.  Asked as `if (wblk)` inside the walks' divergent loops, the
  // compiler (ROCm 7.2) kept the answer as a lane mask made by a vector compare under ONE iteration's exec and tested it under
  // another's: lanes that were not active when it was made read "there are" and loaded through the null pointer (k_pop_passes
  // with BG_WALK_BLOCKS=0, found by rocgdb: s[2:3] = lane 15's answer, exec = lane 13).  With blocks the wrong turn is harmless
  // -- both ways give the same values -- without them it is a fault; readfirstlane is convergent and stays where it is asked.
  static BG_HD bool uniform_nonnull(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(int(p != nullptr)) != 0;
#else
    return p != nullptr;
#endif
  }
  BG_HD bool has_wblk() const { return uniform_nonnull(wblk); }
  BG_HD unsigned size_at(uint64_t i) const {
    if (has_wblk()) {
      const uint32_t g = wblk_of(i);
      const unsigned r = unsigned(i - uint64_t(g) * 40);
      return reinterpret_cast<const uint8_t*>(wblk + uint64_t(g) * 16 + 4)[r];
    }
    return sizes[i];
  }
  BG_HD unsigned shared_at(uint64_t i) const {
    if (has_wblk()) {
      const uint32_t g = wblk_of(i);
      const unsigned r = unsigned(i - uint64_t(g) * 40);
      return reinterpret_cast<const uint8_t*>(wblk + uint64_t(g) * 16 + 9)[r];
    }
    return shared[i];
  }
  // the eight shared values of entries [8 wi, 8 wi + 8), entry 8 wi in the low byte
  BG_HD uint64_t shared_word(uint64_t wi) const {
    if (has_wblk()) {
      const uint32_t g = uint32_t(wi) / 5u;
      return wblk[uint64_t(g) * 16 + 9 + (uint32_t(wi) - g * 5u)];
    }
    return reinterpret_cast<const uint64_t*>(shared)[wi];
  }

  BG_HD uint64_t count(int b, uint64_t i) const {  // bitcount.h:218-270
    if (has_wblk()) {
      const uint32_t g = wblk_of(i);
      const unsigned r = unsigned(i - uint64_t(g) * 40);
      const uint64_t w = wblk[uint64_t(g) * 16 + unsigned(b)];
#if defined(__HIP_DEVICE_COMPILE__)
      const uint64_t c = __popcll(w & ((1ull << r) - 1));
#else
      const uint64_t c = __builtin_popcountll(w & ((1ull << r) - 1));
#endif
      return wblk_super[uint64_t(g >> 16) * 4 + unsigned(b)] + (w >> 40) + c;
    }
    if (uniform_nonnull(rank4)) {
      const uint64_t g = i / 96;
      const unsigned r = unsigned(i - g * 96), w = r >> 5;
#if defined(__HIP_DEVICE_COMPILE__)
      const uint4 q = *reinterpret_cast<const uint4*>(rank4 + g * 16 + unsigned(b) * 4);
      const uint32_t w0 = q.y, w1 = q.z, w2 = q.w, rel = q.x;
      uint32_t c = __popc((w == 0 ? w0 : (w == 1 ? w1 : w2)) & ((1u << (r & 31)) - 1));
      if (w > 0) c += __popc(w0);
      if (w > 1) c += __popc(w1);
#else
      const uint32_t* q = rank4 + g * 16 + unsigned(b) * 4;
      const uint32_t rel = q[0];
      uint32_t c = uint32_t(__builtin_popcount(q[1 + w] & ((1u << (r & 31)) - 1)));
      if (w > 0) c += uint32_t(__builtin_popcount(q[1]));
      if (w > 1) c += uint32_t(__builtin_popcount(q[2]));
#endif
      return rank4_super[(g >> 16) * 4 + unsigned(b)] + rel + c;
    }
    if (uniform_nonnull(rank64[0])) {  // (the four are there or not together)
      const uint64_t g = i / 384;
      const unsigned r = unsigned(i - g * 384), w = r >> 6;
      const uint64_t* blk = rank64[b] + g * 8;
      uint64_t x = blk[2 + w] & ((1ull << (r & 63)) - 1);
#if defined(__HIP_DEVICE_COMPILE__)
      const uint64_t bc = __popcll(x);
#else
      const uint64_t bc = __builtin_popcountll(x);
#endif
      const uint64_t sub = w ? (blk[1] >> (9 * (w - 1))) & 0x1ff : 0;
      return blk[0] + sub + bc;
    }
    uint64_t nw = (n + 63) >> 6;
    uint64_t x = (i >> 6) < nw ? bits[b][i >> 6] : 0;
    x &= (1ull << (i & 63)) - 1;
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t bc = __popcll(x);
#else
    uint64_t bc = __builtin_popcountll(x);
#endif
    uint64_t group = i >> 9, sub = (i >> 6) & 7;
    uint64_t ngroups = (n + 511) >> 9;
    uint64_t sa = group < ngroups ? subaccum[b][group] : 0;
    sa >>= 56 - 8 * sub;
    sa &= 0xffffffffffffff00ull;
    sa = (sa >> 32) + (sa & 0xffffffffull);
    sa = (sa >> 16) + (sa & 0xffff);
    sa = (sa >> 8) + (sa & 0xff);
    return accum[b][group] + sa + bc;
  }
  // bit 8 i of the result = byte i of x is below v (0 <= v <= 255; exact in every byte: two passes over 16-bit lanes, so that no
  // borrow crosses a byte)
  static BG_HD uint64_t bytes_below(uint64_t x, unsigned v) {
    const uint64_t K = 0x0001000100010001ull, B = 0x0100010001000100ull, M = 0x00ff00ff00ff00ffull;
    const uint64_t te = ((x & M) | B) - uint64_t(v) * K, to = (((x >> 8) & M) | B) - uint64_t(v) * K;
    return ((~te >> 8) & K) | (((~to >> 8) & K) << 8);
  }
#if (defined(__HIP_DEVICE_COMPILE__) && !defined(BG_BYTE_SCAN)) || defined(BG_WORD_SCAN_HOST)  // (BG_BYTE_SCAN: tools/build_variant.sh, A/B)
  // The two searches over `shared` that the drop loop of push_front_drop runs, eight entries a load: a drop widens a range by
  // the entries that share the shorter context -- tens of them under 30x coverage -- and one byte a DEPENDENT load was what a
  // walk step spent most of its time in (branch trace: 0.7 drop iterations a call).  `shared` is 8-byte aligned and its
  // allocation a multiple of 8 bytes (the arena rounds to 512), so the word holding a valid entry is readable; bytes past n
  // in it never win (forward: a hit at or past n means none before it; backward: they lie above the start).
  // first pos >= start with shared[pos] < v, or n
  BG_HD uint64_t next_forward_lt(uint64_t pos, unsigned v) const {
    uint64_t p = pos;
    while (p < n) {
      const uint64_t f = bytes_below(shared_word(p >> 3), v) >> (8 * unsigned(p & 7));
      if (f) {
        const uint64_t r = p + uint64_t(__builtin_ctzll(f) >> 3);
        return r < n ? r : n;
      }
      p = (p | 7) + 1;
      if ((p & 63) == 0) {  // whole blocks of 64 (4096) entries without a value below v are skipped through the min index
        uint64_t f1 = p >> 6;
        while (p < n) {
          if ((f1 & 63) == 0 && min2[f1 >> 6] >= v) { f1 += 64; p = f1 << 6; continue; }
          if (min1[f1] >= v) { ++f1; p = f1 << 6; continue; }
          break;
        }
      }
    }
    return n;
  }
  // last pos <= start with shared[pos] < v, or 0
  BG_HD uint64_t next_backward_lt(uint64_t pos, unsigned v) const {
    uint64_t p = pos;
    for (;;) {
      const unsigned k = unsigned(p & 7);
      const uint64_t f = bytes_below(shared_word(p >> 3), v) << (8 * (7 - k));  // byte k on top, the bytes above it gone
      if (f) return p - uint64_t(__builtin_clzll(f) >> 3);
      if (p < 8) return 0;
      p = (p & ~7ull) - 1;
      if ((p & 63) == 63) {
        uint64_t f1 = p >> 6;
        for (;;) {
          if ((f1 & 63) == 63 && min2[f1 >> 6] >= v) {
            if (f1 < 64) return 0;
            f1 -= 64;
            continue;
          }
          if (min1[f1] >= v) {
            if (f1 == 0) return 0;
            --f1;
            continue;
          }
          break;
        }
        p = (f1 << 6) + 63;
      }
    }
  }
#else
  // first pos >= start with shared[pos] < v, or n
  BG_HD uint64_t next_forward_lt(uint64_t pos, unsigned v) const {
    if (shared_at(pos) < v) return pos;
    while (pos < n) {
      ++pos;
      if (pos == n) return pos;
      if ((pos & 63) == 0) {
        uint64_t f1 = pos >> 6;
        if ((f1 & 63) == 0 && min2[f1 >> 6] >= v) {
          pos = ((f1 >> 6) + 1) * 4096 - 1;
          if (pos >= n) return n;
          continue;
        }
        if (min1[f1] >= v) {
          pos = (f1 + 1) * 64 - 1;
          if (pos >= n) return n;
          continue;
        }
      }
      if (shared_at(pos) < v) return pos;
    }
    return pos;
  }
  // last pos <= start with shared[pos] < v, or 0
  BG_HD uint64_t next_backward_lt(uint64_t pos, unsigned v) const {
    if (shared_at(pos) < v) return pos;
    while (pos > 0) {
      --pos;
      if ((pos & 63) == 63) {
        uint64_t f1 = pos >> 6;
        if ((f1 & 63) == 63 && min2[f1 >> 6] >= v) { pos = (f1 >> 6) * 4096; continue; }
        if (min1[f1] >= v) { pos = f1 * 64; continue; }
      }
      if (shared_at(pos) < v) return pos;
    }
    return pos;
  }
#endif
  BG_HD void push_front(uint64_t begin, uint64_t end, uint32_t size, int b, uint64_t* ob, uint64_t* oe,
                        uint32_t* os) const {  // seqset.cpp:319-338
    uint64_t nb = fixed[b] + count(b, begin), ne = fixed[b] + count(b, end);
    if (nb < ne && size_at(nb) < size + 1) nb++;
    *ob = nb; *oe = ne; *os = size + 1;
  }
  BG_HD void push_front_drop(uint64_t begin, uint64_t end, uint32_t size, int b, uint32_t min_ctx,
                             uint64_t* ob, uint64_t* oe, uint32_t* os) const {  // seqset.cpp:350-429
    uint64_t fx = fixed[b];
    uint64_t o_begin = begin, o_end = end;
    uint32_t octx = size;
    uint64_t sb = count(b, o_begin), se = count(b, o_end);
    *ob = 0; *oe = 0; *os = 0;
#ifdef __HIP_DEVICE_COMPILE__
    if (uniform_nonnull(walk_counters)) atomicAdd(&walk_counters[0], 1ull);
#endif
    if (octx < min_ctx) return;
    // (Measured and put away: deciding a failing base up front from the widest range the loop may reach -- the entries that
    // share min_ctx bases -- instead of giving the context up level by level.  A failing base nearly always fails in the loop's
    // FIRST iteration, its two neighbours sharing fewer than min_ctx bases, so the extra searches and rank look-ups only cost:
    // branch trace 57.6 -> 63.4 ms.)
    while (sb == se || (sb + 1 == se && size_at(fx + sb) < octx + 1)) {
#ifdef __HIP_DEVICE_COMPILE__
      if (uniform_nonnull(walk_counters)) atomicAdd(&walk_counters[1], 1ull);
#endif
      // No entry starts with base b at all (a seqset of a two-letter genome, a toy input): with min_ctx == 0 the
      // reference's loop reaches the whole range and spins there for ever (seqset.cpp:368-421 has no exit for
      // it); the extension simply does not exist.  Found by tools/fuzz_parity.py (FUZZ_WALK).
      if (o_begin == 0 && o_end == n && sb == se) return;
      uint32_t d0 = shared_at(o_begin);
      uint32_t d1 = o_end == n ? 0u : uint32_t(shared_at(o_end));
      uint32_t drop = d0 > d1 ? d0 : d1;
      if (sb != se) {
        uint32_t d2 = uint32_t(size_at(fx + sb)) - 1;
        if (d2 > drop) drop = d2;
      }
      if (drop < min_ctx) return;
      bool ub = false, ue = false;
      if (o_begin > 0 && d0 >= drop) { o_begin = next_backward_lt(o_begin, drop); ub = true; }
      if (o_end < n && d1 >= drop) { o_end = next_forward_lt(o_end, drop); ue = true; }
      if (ub) sb = count(b, o_begin);
      if (ue) se = count(b, o_end);
      octx = drop;
    }
    uint64_t nb = fx + sb, ne = fx + se;
    if (nb < ne && size_at(nb) < octx + 1) nb++;
    *ob = nb; *oe = ne; *os = octx + 1;
  }
};

}  // namespace bg
