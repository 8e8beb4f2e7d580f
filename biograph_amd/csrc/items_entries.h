// items_entries.h -- the entries of a seqset built from items, searched without a key table (items_build.hip makes them; the
// prev bits' tail search and the readmap's look-ups read them).  An entry is its first 32 bases and the record it came from --
// oriented read << 24 | offset << 16 | length << 8 | flags -- and the rest of its string is read out of the oriented reads.
#pragma once
#include "bg_device_funcs.h"

namespace bg {

struct __attribute__((aligned(16))) Item { uint64_t fw, id; };

// id bits 55..63 are free in a sorted item: bit 63 = "the 8 bits below are this item's LCP with the item after it" (items_build.hip)
constexpr uint64_t kItLcpValid = 1ull << 63;
constexpr int kItLcpShift = 55;
constexpr uint64_t kItIdMask = (1ull << kItLcpShift) - 1;

__device__ __forceinline__ uint64_t it_suffix_word(const uint64_t* __restrict__ s, int W, int j, int w) {  // word w of suffix j of a read
  const int sq = (j >> 5) + w, sr = j & 31;
  const uint64_t cur = sq < W ? s[sq] : 0, next = sq + 1 < W ? s[sq + 1] : 0;
  return sr ? (cur << (2 * sr)) | (next >> (64 - 2 * sr)) : cur;
}

struct ItEntries {
  const Item* ent;        // every entry's first word and its record
  const uint8_t* sizes;
  const uint64_t* owords;
  const uint32_t* dir;
  uint64_t n;
  int W, bases;
};
// entry e against (key, kl) beyond the first word (equal): rest of the string through the read, then the length
__device__ __forceinline__ int it_cmp_rest(const ItEntries& t, uint64_t e, const uint64_t* key, int kl) {
  const uint64_t id = t.ent[e].id;
  const uint64_t* s = t.owords + (id >> 24) * uint64_t(t.W);
  const int j = int(id >> 16) & 255;
  for (int w = 1; w < t.W; ++w) {
    const uint64_t x = it_suffix_word(s, t.W, j, w);
    if (x != key[w]) return x < key[w] ? -1 : 1;
  }
  const int el = t.sizes[e];
  return el < kl ? -1 : (el > kl ? 1 : 0);
}
// first entry >= (key, kl) in (string, length) order.  *alone = the key's first word occurs in exactly one entry (which is then
// returned without a look at the reads: when the key is at least a word long and the entries are closed under pop_front, that
// entry is the one that extends it -- k_prev_bits_pre2's shortcut on one word instead of two)
__device__ __forceinline__ uint64_t it_lower_bound(const ItEntries& t, const uint64_t* key, int kl, bool* alone) {
  const uint32_t p = uint32_t(key[0] >> (64 - 2 * t.bases));
  uint64_t a = t.dir[p], z = t.dir[p + 1];
  const uint64_t hi = z;
  while (a < z) {  // first entry whose first word is >= the key's
    const uint64_t mid = (a + z) >> 1;
    if (t.ent[mid].fw < key[0]) a = mid + 1; else z = mid;
  }
  uint64_t lo1 = a;
  *alone = false;
  if (lo1 >= hi || t.ent[lo1].fw != key[0]) return lo1;
  uint64_t hi1 = lo1 + 1;  // the entries that share the first word lie side by side: a few (a repeat family: thousands)
  while (hi1 < hi && hi1 < lo1 + 8 && t.ent[hi1].fw == key[0]) ++hi1;
  if (hi1 == lo1 + 8 && hi1 < hi && t.ent[hi1].fw == key[0]) {
    uint64_t a2 = hi1 + 1, z2 = hi;
    while (a2 < z2) {  // first entry whose first word is greater
      const uint64_t mid = (a2 + z2) >> 1;
      if (t.ent[mid].fw <= key[0]) a2 = mid + 1; else z2 = mid;
    }
    hi1 = a2;
  }
  if (hi1 == lo1 + 1 && kl >= 32) { *alone = true; return lo1; }
  while (lo1 < hi1) {
    const uint64_t mid = (lo1 + hi1) >> 1;
    if (it_cmp_rest(t, mid, key, kl) < 0) lo1 = mid + 1; else hi1 = mid;
  }
  return lo1;
}
// LCP of entry e with (key, kl), in bases
__device__ __forceinline__ int it_entry_lcp(const ItEntries& t, uint64_t e, const uint64_t* key, int kl) {
  const Item it = t.ent[e];
  int l = lcp_word(it.fw, key[0]);
  if (l == 32) {
    const uint64_t id = it.id;
    const uint64_t* s = t.owords + (id >> 24) * uint64_t(t.W);
    const int j = int(id >> 16) & 255;
    for (int w = 1; w < t.W; ++w) {
      const int c = lcp_word(it_suffix_word(s, t.W, j, w), key[w]);
      l += c;
      if (c < 32) break;
    }
  }
  const int el = t.sizes[e], m = el < kl ? el : kl;
  return l < m ? l : m;
}

}  // namespace bg
