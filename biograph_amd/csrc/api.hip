// api.hip — context, seqset object (parts, container I/O) and the whole-create entry point of the
// C ABI declared in include/bgseq.h.
#include <string.h>
#include <time.h>

#include "bg_device_funcs.h"
#include <errno.h>
#include <string.h>

#include "bg_internal.h"

namespace bg {
const char* last_error_cstr();

// packed_varbit_vector layout (io/packed_varbit_vector.cpp:80-147): value i occupies bits
// [i*bpv, (i+1)*bpv) of a little-endian bit string stored as uint64 words.  One thread per word.
__global__ void k_pack_varbit(const uint8_t* __restrict__ vals, uint64_t n, int bpv, uint64_t nwords,
                              uint64_t* __restrict__ out) {
  uint64_t w = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint64_t bit_lo = w * 64, bit_hi = bit_lo + 64;
  uint64_t first = bit_lo / bpv;
  uint64_t word = 0;
  for (uint64_t i = first; i < n; ++i) {
    uint64_t b = i * bpv;
    if (b >= bit_hi) break;
    uint64_t v = vals[i];
    if (b >= bit_lo) word |= v << (b - bit_lo);
    else word |= v >> (bit_lo - b);
  }
  out[w] = word;
}

__global__ void k_unpack_varbit(const uint64_t* __restrict__ words, uint64_t n, int bpv, uint8_t* __restrict__ vals) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (bpv == 0) { vals[i] = 0; return; }
  uint64_t b = i * bpv;
  uint64_t v = words[b >> 6] >> (b & 63);
  if ((b & 63) + bpv > 64) v |= words[(b >> 6) + 1] << (64 - (b & 63));
  vals[i] = uint8_t(v & ((1ull << bpv) - 1));
}

__global__ void k_u8_to_u32(const uint8_t* __restrict__ a, uint64_t n, uint32_t* __restrict__ out) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i];
}
__global__ void k_u32_to_u8(const uint32_t* __restrict__ a, uint64_t n, uint8_t* __restrict__ out) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = uint8_t(a[i]);
}

// entry sequences by FM navigation is the reference's way (seqset.cpp:609-721); after a build the
// keys are resident, so this simply unpacks them.
__global__ void k_entries_ascii(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ sizes, uint64_t first,
                                uint64_t count, int W, const uint64_t* __restrict__ offs, uint8_t* __restrict__ seq) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint64_t e = first + i;
  int L = sizes[e];
  uint8_t* p = seq + offs[i];
  for (int j = 0; j < L; ++j) p[j] = "ACGT"[read_code(keys + e * W, j)];
}

// Without resident keys (seqset opened from a file): walk pop_front by select on prev_b.  One thread
// per entry; select = binary search over accum + word scan.  Used by tests / SDK only.
__global__ void k_entries_ascii_fm(SeqsetView v, uint64_t first, uint64_t count, const uint64_t* __restrict__ offs,
                                   uint8_t* __restrict__ seq) {
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint64_t cur = first + i;
  int L = v.sizes[cur];
  uint8_t* p = seq + offs[i];
  for (int j = 0; j < L; ++j) {
    int b = 0;
    while (b < 3 && cur >= v.fixed[b + 1]) ++b;
    p[j] = "ACGT"[b];
    uint64_t rank = cur - v.fixed[b];  // the (rank)-th set bit of prev_b is the pop_front target
    uint64_t lo = 0, hi = v.n;         // smallest t with count(b, t+1) > rank
    while (lo < hi) {
      uint64_t mid = (lo + hi) >> 1;
      if (v.count(b, mid + 1) > rank) hi = mid; else lo = mid + 1;
    }
    cur = lo;
  }
}

static unsigned bits_for_value(uint64_t v) { unsigned b = 0; while (v) { ++b; v >>= 1; } return b; }

static SeqsetView view_of(const bg_seqset* ss) {
  SeqsetView v;
  v.n = ss->n;
  v.sizes = ss->sizes.p;
  v.shared = ss->shared.p;
  for (int b = 0; b < 4; ++b) { v.bits[b] = ss->bits[b].p; v.subaccum[b] = ss->subaccum[b].p; v.accum[b] = ss->accum[b].p; }
  for (int b = 0; b < 5; ++b) v.fixed[b] = ss->fixed[b];
  v.min1 = ss->min1.p;
  v.min2 = ss->min2.p;
  return v;
}

static int download(bg_ctx* ctx, void* h, const void* d, uint64_t bytes) {
  if (!bytes) return BG_OK;
  if (bytes >= (8u << 20)) return staged_download(ctx, h, d, bytes, 0);  // a pageable copy of this size runs at 2-3 GB/s
  BG_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

// bytes of one payload part, to host
static int part_bytes(bg_ctx* ctx, const bg_seqset* ss, const std::string& part, std::string* out) {
  const uint64_t n = ss->n;
  if (part == "fixed") { out->assign(reinterpret_cast<const char*>(ss->fixed), 40); return BG_OK; }
  if (part == "entry_sizes/elements" || part == "shared/elements") {
    bool is_sizes = part[0] == 'e';
    uint64_t maxv = is_sizes ? ss->max_len : (ss->max_len ? ss->max_len - 1 : 0);
    int bpv = int(bits_for_value(maxv));
    uint64_t nwords = (n * bpv + 63) / 64;
    out->resize(nwords * 8);
    if (!nwords) return BG_OK;
    DevBuf<uint64_t> packed;
    BG_TRY(packed.alloc(ctx, nwords));
    k_pack_varbit<<<cdiv(nwords, 256), 256, 0, ctx->stream>>>(is_sizes ? ss->sizes.p : ss->shared.p, n, bpv, nwords,
                                                               packed.p);
    BG_LAUNCH_CHECK();
    return download(ctx, &(*out)[0], packed.p, nwords * 8);
  }
  if (part.size() > 7 && part.compare(0, 5, "prev_") == 0) {
    int b = part[5] == 'A' ? 0 : part[5] == 'C' ? 1 : part[5] == 'G' ? 2 : part[5] == 'T' ? 3 : -1;
    std::string what = part.substr(7);
    if (b >= 0) {
      const DevBuf<uint64_t>* src = what == "bits" ? &ss->bits[b] : what == "subaccum" ? &ss->subaccum[b]
                                    : what == "accum" ? &ss->accum[b] : nullptr;
      if (src) {
        uint64_t words = what == "bits" ? (n + 63) / 64 : what == "subaccum" ? (n + 511) / 512 : (n + 512) / 512;
        out->resize(words * 8);
        return download(ctx, &(*out)[0], src->p, words * 8);
      }
    }
  }
  set_error("bg_seqset_part: unknown part '%s'", part.c_str());
  return BG_EINVAL;
}

static std::string json_escape(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(c); }
    else if (c == '\n') o += "\\n";
    else o.push_back(c);
  }
  return o;
}

static std::string version_json(const char* type, int maj, int min, int pat) {
  char buf[256];
  snprintf(buf, sizeof buf,
           "{\"part_type\":\"%s\",\"version\":{\"build\":\"\",\"major\":%d,\"minor\":%d,\"patch\":%d,\"pre\":\"\"}}", type,
           maj, min, pat);
  return buf;
}

}  // namespace bg

using namespace bg;

extern "C" int bg_abi_version(void) { return BG_ABI_VERSION; }
extern "C" int bg_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
extern "C" const char* bg_last_error(void) { return bg::last_error_cstr(); }

extern "C" int bg_ctx_create(int device, void* stream, bg_ctx** out) {
  if (!out) { set_error("bg_ctx_create: null argument"); return BG_EINVAL; }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    set_error("bg_ctx_create: no HIP device available (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    (void)hipGetLastError();
    return BG_ENODEV;
  }
  if (device < 0 || device >= count) { set_error("bg_ctx_create: device %d out of range (%d)", device, count); return BG_EINVAL; }
  BG_HIP(hipSetDevice(device));
  bg_ctx* c = new bg_ctx;
  c->device = device;
  c->stream = static_cast<hipStream_t>(stream);
  {
    // what the arena may keep from the driver without using it (BG_POOL_CAP_GB, default 35 % of the device): the
    // caller's own allocator (torch in the multi-GPU driver) lives beside it and cannot ask it to trim
    size_t cap = ~size_t(0);
    if (const char* e = getenv("BG_POOL_CAP_GB")) {
      cap = size_t(atof(e) * 1e9);
    } else {
      size_t fr = 0, total = 0;
      if (hipMemGetInfo(&fr, &total) == hipSuccess) cap = size_t(double(total) * 0.35);
      else (void)hipGetLastError();
    }
    if (bg_ctx::DeviceArena* sh = shared_arena(device)) c->arena = sh;  // one arena with torch: nothing to cap
    else c->own_arena.set_cap(cap);
  }
  *out = c;
  return BG_OK;
}

// A context with a stream of its own (non-blocking: it does not meet other contexts on the null stream), for callers
// that run several contexts side by side from several host threads.  The stream goes with the context.
extern "C" int bg_ctx_create_own_stream(int device, bg_ctx** out) {
  BG_TRY(bg_ctx_create(device, nullptr, out));
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_error("bg_ctx_create_own_stream: %s", hipGetErrorString(e));
    bg_ctx_destroy(*out);
    *out = nullptr;
    return BG_ENODEV;
  }
  (*out)->stream = s;
  (*out)->owns_stream = true;
  // The shared arena hands a released chunk out again at once, which is only safe among users ordered on ONE stream
  // (torch's current stream = the stream of the contexts made with bg_ctx_create).  A context on a stream of its own
  // therefore always keeps its private arena: chunks never move between streams.
  if ((*out)->arena != &(*out)->own_arena) {
    (*out)->arena = &(*out)->own_arena;
    size_t fr = 0, total = 0;
    if (hipMemGetInfo(&fr, &total) == hipSuccess) (*out)->own_arena.set_cap(size_t(double(total) * 0.35));
    else (void)hipGetLastError();
  }
  return BG_OK;
}

extern "C" void bg_ctx_destroy(bg_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->owns_stream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
  ctx->trim();
  ctx->clear_timings();
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (void* p : ctx->pinned) if (p) (void)hipHostFree(p);
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int bg_ctx_trim(bg_ctx* ctx) { return ctx ? ctx->trim() : BG_EINVAL; }

// Pre-faults device memory for what is about to run: a one-shot command pays the driver ~30 us per MiB the first time it
// touches memory (0.85 s of a 1.9 s `create` of a chr20-sized sample), and that can happen on a side thread while the main
// one is still reading its input.  The driver call runs outside the arena's lock, so the context stays usable meanwhile;
// the segment then joins the arena as free memory.  Safe to call from another thread than the one using the context.
extern "C" int bg_ctx_reserve(bg_ctx* ctx, uint64_t bytes) {
  if (!ctx) { set_error("bg_ctx_reserve: null argument"); return BG_EINVAL; }
  if (!bytes) return BG_OK;
  BG_HIP(hipSetDevice(ctx->device));
  // BG_RESERVE_MODE (measurements, tools/map_rate.hip): "malloc" = one hipMalloc of the whole size, "vmm1g" = the address
  // range below in 1 GiB pieces instead of 256 MiB ones
  const char* mode = getenv("BG_RESERVE_MODE");
  if (mode && !strcmp(mode, "malloc")) {
    void* base = nullptr;
    const size_t seg = (size_t(bytes) + (size_t(1) << 21) - 1) & ~((size_t(1) << 21) - 1);
    if (hipMalloc(&base, seg) != hipSuccess) {
      (void)hipGetLastError();
      set_error("bg_ctx_reserve: the driver refused %zu MiB", seg >> 20);
      return BG_ENOMEM;
    }
    std::lock_guard<std::mutex> lock(ctx->arena_mutex());
    ctx->arena->adopt(base, seg);
    return BG_OK;
  }
  // one address range, a gigabyte of physical memory mapped into it at a time (between two pieces everybody else gets
  // through the driver; 256 MiB pieces, the first version, turned out to be the slow way: 60 us per MiB against 1.6)
  const size_t piece = (mode && !strcmp(mode, "vmm256")) ? size_t(256) << 20 : size_t(1) << 30;
  const size_t seg = (size_t(bytes) + piece - 1) & ~(piece - 1);
  void* base = vmm_segment(ctx->device, seg, piece);
  if (!base) { set_error("bg_ctx_reserve: the driver refused %zu MiB", seg >> 20); return BG_ENOMEM; }
  const size_t done = seg;
  std::lock_guard<std::mutex> lock(ctx->arena_mutex());
  ctx->arena->adopt(base, done);
  return BG_OK;
}

extern "C" int bg_ctx_memory(bg_ctx* ctx, uint64_t* live, uint64_t* cached, uint64_t* peak, int reset_peak) {
  if (!ctx) return BG_EINVAL;
  std::lock_guard<std::mutex> lock(ctx->arena_mutex());
  if (live) *live = ctx->arena->live();
  if (cached) *cached = ctx->arena->cached();
  if (peak) *peak = ctx->arena->peak();
  if (reset_peak) ctx->arena->reset_peak();
  return BG_OK;
}

extern "C" int bg_ctx_set_profiling(bg_ctx* ctx, int on) {
  if (!ctx) return BG_EINVAL;
  ctx->profiling = on != 0;
  ctx->clear_timings();
  return BG_OK;
}

extern "C" int bg_ctx_timings(bg_ctx* ctx, bg_timing* out, int cap) {
  if (!ctx) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  std::vector<std::string> order;
  std::map<std::string, bg_timing> agg;
  for (auto& r : ctx->recs) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) { (void)hipGetLastError(); continue; }
    auto it = agg.find(r.name);
    if (it == agg.end()) {
      bg_timing t;
      memset(&t, 0, sizeof t);
      snprintf(t.name, sizeof t.name, "%s", r.name.c_str());
      it = agg.emplace(r.name, t).first;
      order.push_back(r.name);
    }
    it->second.ms += ms;
    it->second.launches += 1;
    it->second.bytes += r.bytes;
  }
  int k = 0;
  for (auto& name : order) {
    if (out && k < cap) out[k] = agg[name];
    ++k;
  }
  return k;
}

extern "C" int bg_dev_alloc(bg_ctx* ctx, uint64_t bytes, void** d_out) {
  if (!ctx || !d_out) return BG_EINVAL;
  BG_HIP(hipSetDevice(ctx->device));
  return ctx->alloc(bytes ? bytes : 1, d_out);  // from the context's arena, like everything else
}
extern "C" int bg_dev_free(bg_ctx* ctx, void* p) {
  if (!ctx) return BG_EINVAL;
  if (!p) return BG_OK;
  BG_HIP(hipSetDevice(ctx->device));
  bool ok;
  {
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    ok = ctx->arena->release(p);
  }
  if (!ok) { set_error("bg_dev_free: %p was not allocated by this context", p); return BG_EINVAL; }
  return BG_OK;
}
extern "C" int bg_dev_upload(bg_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes) {
  if (!ctx) return BG_EINVAL;
  BG_HIP(hipSetDevice(ctx->device));
  if (!bytes) return BG_OK;
  BG_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}
extern "C" int bg_dev_download(bg_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes) {
  if (!ctx) return BG_EINVAL;
  BG_HIP(hipSetDevice(ctx->device));
  return download(ctx, h_dst, d_src, bytes);
}

// ---- seqset object

extern "C" uint64_t bg_seqset_size(const bg_seqset* ss) { return ss ? ss->n : 0; }
extern "C" uint32_t bg_seqset_max_len(const bg_seqset* ss) { return ss ? ss->max_len : 0; }
extern "C" int bg_seqset_uuid(const bg_seqset* ss, char* h_out, uint64_t cap) {
  if (!ss || !h_out || cap < ss->uuid.size() + 1) { set_error("bg_seqset_uuid: buffer too small"); return BG_EINVAL; }
  memcpy(h_out, ss->uuid.c_str(), ss->uuid.size() + 1);
  return BG_OK;
}
extern "C" void bg_seqset_free(bg_seqset* ss) { delete ss; }

extern "C" int bg_seqset_part(bg_ctx* ctx, const bg_seqset* ss, const char* part, void* h_out, uint64_t* size) {
  if (!ctx || !ss || !part || !size) { set_error("bg_seqset_part: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::string data;
  BG_TRY(part_bytes(ctx, ss, part, &data));
  if (h_out) {
    if (*size < data.size()) { set_error("bg_seqset_part: buffer too small"); *size = data.size(); return BG_EINVAL; }
    memcpy(h_out, data.data(), data.size());
  }
  *size = data.size();
  return BG_OK;
}

extern "C" int bg_seqset_arrays(bg_ctx* ctx, const bg_seqset* ss, uint32_t* h_sizes, uint32_t* h_shared) {
  if (!ctx || !ss) { set_error("bg_seqset_arrays: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  uint64_t n = ss->n;
  if (!n) return BG_OK;
  DevBuf<uint32_t> tmp;
  BG_TRY(tmp.alloc(ctx, n));
  for (int which = 0; which < 2; ++which) {
    uint32_t* dst = which ? h_shared : h_sizes;
    if (!dst) continue;
    k_u8_to_u32<<<cdiv(n, 256), 256, 0, ctx->stream>>>(which ? ss->shared.p : ss->sizes.p, n, tmp.p);
    BG_LAUNCH_CHECK();
    BG_TRY(download(ctx, dst, tmp.p, n * 4));
  }
  return BG_OK;
}

extern "C" int bg_seqset_entries(bg_ctx* ctx, const bg_seqset* ss, uint64_t first, uint64_t count, uint8_t* h_seq,
                                 uint64_t* h_offs) {
  if (!ctx || !ss || !h_offs) { set_error("bg_seqset_entries: null argument"); return BG_EINVAL; }
  if (first + count > ss->n) { set_error("bg_seqset_entries: range past the end"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::vector<uint8_t> sz(count);
  BG_TRY(download(ctx, sz.data(), ss->sizes.p + first, count));
  uint64_t pos = 0;
  for (uint64_t i = 0; i < count; ++i) { h_offs[i] = pos; pos += sz[i]; }
  h_offs[count] = pos;
  if (!count || !h_seq) return BG_OK;
  DevBuf<uint64_t> d_offs;
  DevBuf<uint8_t> d_seq;
  BG_TRY(d_offs.alloc(ctx, count + 1));
  BG_TRY(d_seq.alloc(ctx, pos));
  BG_HIP(hipMemcpyAsync(d_offs.p, h_offs, (count + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BG_TRY(seqset_ensure_keys(ctx, const_cast<bg_seqset*>(ss)));  // (a build from items makes its keys when asked)
  if (ss->keys.p && ss->W) {
    k_entries_ascii<<<cdiv(count, 256), 256, 0, ctx->stream>>>(ss->keys.p, ss->sizes.p, first, count, ss->W, d_offs.p,
                                                              d_seq.p);
  } else {
    k_entries_ascii_fm<<<cdiv(count, 64), 64, 0, ctx->stream>>>(view_of(ss), first, count, d_offs.p, d_seq.p);
  }
  BG_LAUNCH_CHECK();
  return download(ctx, h_seq, d_seq.p, pos);
}

extern "C" int bg_seqset_from_arrays(bg_ctx* ctx, uint64_t n, const uint32_t* h_sizes, const uint32_t* h_shared,
                                     const uint64_t* const h_bits[4], bg_seqset** out) {
  if (!ctx || !out || (n && (!h_sizes || !h_shared || !h_bits))) { set_error("bg_seqset_from_arrays: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_seqset> ss(new bg_seqset);
  ss->ctx = ctx;
  ss->n = n;
  ss->uuid = make_uuid();
  BG_TRY(ss->sizes.alloc(ctx, n));
  BG_TRY(ss->shared.alloc(ctx, n));
  uint32_t mx = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (h_sizes[i] > 255 || h_shared[i] > 255) { set_error("bg_seqset_from_arrays: entry longer than 255"); return BG_EINVAL; }
    mx = h_sizes[i] > mx ? h_sizes[i] : mx;
  }
  ss->max_len = mx;
  if (n) {
    DevBuf<uint32_t> tmp;
    BG_TRY(tmp.alloc(ctx, n));
    BG_HIP(hipMemcpyAsync(tmp.p, h_sizes, n * 4, hipMemcpyHostToDevice, ctx->stream));
    k_u32_to_u8<<<cdiv(n, 256), 256, 0, ctx->stream>>>(tmp.p, n, ss->sizes.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipStreamSynchronize(ctx->stream));
    BG_HIP(hipMemcpyAsync(tmp.p, h_shared, n * 4, hipMemcpyHostToDevice, ctx->stream));
    k_u32_to_u8<<<cdiv(n, 256), 256, 0, ctx->stream>>>(tmp.p, n, ss->shared.p);
    BG_LAUNCH_CHECK();
    BG_HIP(hipStreamSynchronize(ctx->stream));
  }
  uint64_t nwords = (n + 63) / 64;
  for (int b = 0; b < 4; ++b) {
    BG_TRY(ss->bits[b].alloc(ctx, nwords));
    if (nwords) BG_HIP(hipMemcpyAsync(ss->bits[b].p, h_bits[b], nwords * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  BG_TRY(seqset_finalize_device(ctx, ss.get()));
  *out = ss.release();
  return BG_OK;
}

// file_info.json (io/spiral_file.cpp:194-222): uuid, timestamps, build stamp, argv — never comparable
static std::string file_info_json(const std::string& uuid, int argc, const char* const* argv) {
  time_t now = time(nullptr);
  char tbuf[64];
  ctime_r(&now, tbuf);
  std::string ttext(tbuf);
  if (!ttext.empty() && ttext.back() == '\n') ttext.pop_back();
  std::string cmd = "[";
  for (int i = 0; i < argc; ++i) {
    if (i) cmd += ",";
    cmd += "\"" + json_escape(argv[i] ? argv[i] : "") + "\"";
  }
  cmd += "]";
  char num[32];
  snprintf(num, sizeof num, "%lld", (long long)now);
  return std::string("{\"build_host\":\"\",\"build_is_clean\":true,\"build_revision\":\"biograph_amd\",\"build_timestamp\":0,"
                     "\"build_timestamp_text\":\"\",\"build_user\":\"\",\"command_line\":") + cmd +
         ",\"create_timestamp\":" + num + ",\"create_timestamp_text\":\"" + ttext + "\",\"uuid\":\"" + json_escape(uuid) + "\"}";
}

constexpr int kPayloadParts = 15;  // fixed, entry_sizes/elements, shared/elements, prev_{A,C,G,T}/{bits,subaccum,accum}

// The members of a seqset container in creation order (spiral_file_mmap.cpp:350-457).  payload(name, &data) fills a
// payload member; with hole_sizes (kPayloadParts entries, payload order) the payload members only get room.
template <class F>
static int seqset_members(uint64_t n, uint32_t max_len, const std::string& uuid, int argc, const char* const* argv, F payload,
                          const uint64_t* hole_sizes, std::vector<ZipPart>* parts) {
  int pi = 0;
  auto add_payload = [&](const std::string& name) -> int {
    ZipPart p{name, std::string(), true};
    if (hole_sizes) {
      p.hole_size = hole_sizes[pi];
      if (!p.hole_size) p.data.clear();
    } else {
      BG_TRY(payload(name, &p.data));
    }
    ++pi;
    parts->push_back(std::move(p));
    return BG_OK;
  };
  parts->push_back(ZipPart{"file_info.json", file_info_json(uuid, argc, argv), false});
  parts->push_back(ZipPart{"part_info.json", version_json("seqset", 1, 1, 0), false});  // seqset.cpp:12
  char sj[64];
  snprintf(sj, sizeof sj, "{\"num_entries\":%llu}", (unsigned long long)n);
  parts->push_back(ZipPart{"seqset.json", sj, false});
  BG_TRY(add_payload("fixed"));
  for (int which = 0; which < 2; ++which) {
    std::string dir = which ? "shared" : "entry_sizes";
    uint64_t maxv = which ? (max_len ? max_len - 1 : 0) : max_len;
    parts->push_back(ZipPart{dir + "/part_info.json", version_json("packed_varbit_vector", 1, 0, 0), false});
    char mj[128];
    snprintf(mj, sizeof mj, "{\"bits_per_value\":%u,\"element_count\":%llu,\"max_value\":%llu}", bits_for_value(maxv),
             (unsigned long long)n, (unsigned long long)maxv);
    parts->push_back(ZipPart{dir + "/packed_varbit_vector.json", mj, false});
    BG_TRY(add_payload(dir + "/elements"));
  }
  for (int b = 0; b < 4; ++b) {
    std::string dir = std::string("prev_") + "ACGT"[b];
    parts->push_back(ZipPart{dir + "/part_info.json", version_json("bitcount", 1, 0, 0), false});
    char bj[64];
    snprintf(bj, sizeof bj, "{\"nbits\":%llu}", (unsigned long long)n);
    parts->push_back(ZipPart{dir + "/bitcount.json", bj, false});
    for (const char* w : {"bits", "subaccum", "accum"}) BG_TRY(add_payload(dir + "/" + w));
  }
  return BG_OK;
}

extern "C" int bg_seqset_write(bg_ctx* ctx, const bg_seqset* ss, const char* path, int argc, const char* const* argv) {
  if (!ctx || !ss || !path) { set_error("bg_seqset_write: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::vector<ZipPart> parts;
  BG_TRY(seqset_members(ss->n, ss->max_len, ss->uuid, argc, argv,
                        [&](const std::string& name, std::string* data) { return part_bytes(ctx, ss, name, data); }, nullptr,
                        &parts));
  return spiral_zip_write(path, parts);
}

// ---- a sharded seqset leaves as slices: every rank holds a 512-aligned stretch of the entries and writes its
// bytes of every member in place (no gather; biograph_amd/sharded.py).  An aligned slice is a seqset of its own
// whose packed members are byte for byte the global members' stretch; only accum needs the popcounts of the
// ranks before it added, and `fixed` is global.

namespace bg {
__global__ void k_bits_to_masks(const uint64_t* __restrict__ b0, const uint64_t* __restrict__ b1, const uint64_t* __restrict__ b2,
                                const uint64_t* __restrict__ b3, uint64_t n, uint8_t* __restrict__ mask) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t w = i >> 6;
  const int s = int(i & 63);
  mask[i] = uint8_t(((b0[w] >> s) & 1) | (((b1[w] >> s) & 1) << 1) | (((b2[w] >> s) & 1) << 2) | (((b3[w] >> s) & 1) << 3));
}
__global__ __launch_bounds__(256) void k_masks_to_bits(const uint8_t* __restrict__ mask, uint64_t n, uint64_t* __restrict__ b0,
                                                       uint64_t* __restrict__ b1, uint64_t* __restrict__ b2,
                                                       uint64_t* __restrict__ b3) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one wave = one word of each vector
  const uint32_t m = i < n ? mask[i] : 0u;
  const uint64_t w0 = __ballot(m & 1u), w1 = __ballot(m & 2u), w2 = __ballot(m & 4u), w3 = __ballot(m & 8u);
  if ((threadIdx.x & 63) == 0 && i < n) {
    b0[i >> 6] = w0;
    b1[i >> 6] = w1;
    b2[i >> 6] = w2;
    b3[i >> 6] = w3;
  }
}
__global__ void k_add_u64(uint64_t* __restrict__ a, uint64_t n, uint64_t v) {
  const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) a[i] += v;
}
}  // namespace bg

extern "C" int bg_seqset_slice_masks(bg_ctx* ctx, const bg_seqset* ss, uint8_t* d_sizes, uint8_t* d_shared, uint8_t* d_mask) {
  if (!ctx || !ss || (ss->n && (!d_sizes || !d_shared || !d_mask))) { set_error("bg_seqset_slice_masks: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  const uint64_t n = ss->n;
  if (n) {
    if (!ss->bits[0].p) { set_error("bg_seqset_slice_masks: prev bits not built"); return BG_EINVAL; }
    BG_HIP(hipMemcpyAsync(d_sizes, ss->sizes.p, n, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(d_shared, ss->shared.p, n, hipMemcpyDeviceToDevice, ctx->stream));
    k_bits_to_masks<<<cdiv(n, 256), 256, 0, ctx->stream>>>(ss->bits[0].p, ss->bits[1].p, ss->bits[2].p, ss->bits[3].p, n, d_mask);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return BG_OK;
}

extern "C" int bg_seqset_slice_build(bg_ctx* ctx, uint64_t m, const uint8_t* d_sizes, const uint8_t* d_shared,
                                     const uint8_t* d_mask, uint32_t max_len, bg_seqset** out) {
  if (!ctx || !out || (m && (!d_sizes || !d_shared || !d_mask))) { set_error("bg_seqset_slice_build: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<bg_seqset> ss(new bg_seqset);
  ss->ctx = ctx;
  ss->n = m;
  ss->W = 0;
  ss->max_len = max_len;  // of the WHOLE seqset: it fixes the packed widths
  ss->uuid = make_uuid();
  BG_TRY(ss->sizes.alloc(ctx, m));
  BG_TRY(ss->shared.alloc(ctx, m));
  BG_TRY(alloc_prev_bits(ctx, ss.get()));
  if (m) {
    BG_HIP(hipMemcpyAsync(ss->sizes.p, d_sizes, m, hipMemcpyDeviceToDevice, ctx->stream));
    BG_HIP(hipMemcpyAsync(ss->shared.p, d_shared, m, hipMemcpyDeviceToDevice, ctx->stream));
    k_masks_to_bits<<<cdiv(m, 256), 256, 0, ctx->stream>>>(d_mask, m, ss->bits[0].p, ss->bits[1].p, ss->bits[2].p, ss->bits[3].p);
    BG_LAUNCH_CHECK();
  }
  ctx->tic("seqset_finalize");
  BG_TRY(seqset_finalize_device(ctx, ss.get(), true));
  ctx->toc(m * 3);
  *out = ss.release();
  return BG_OK;
}

// set prev bits per base inside the slice (what the ranks after it add to their accum)
extern "C" int bg_seqset_slice_popcounts(const bg_seqset* ss, uint64_t* h_out4) {
  if (!ss || !h_out4) { set_error("bg_seqset_slice_popcounts: null argument"); return BG_EINVAL; }
  for (int b = 0; b < 4; ++b) h_out4[b] = ss->fixed[b + 1] - ss->fixed[b];
  return BG_OK;
}

static void payload_sizes(uint64_t n, uint32_t max_len, uint64_t* sz) {
  const int bs = int(bits_for_value(max_len)), bh = int(bits_for_value(max_len ? max_len - 1 : 0));
  sz[0] = 40;
  sz[1] = (n * bs + 63) / 64 * 8;
  sz[2] = (n * bh + 63) / 64 * 8;
  for (int b = 0; b < 4; ++b) {
    sz[3 + 3 * b] = (n + 63) / 64 * 8;
    sz[4 + 3 * b] = (n + 511) / 512 * 8;
    sz[5 + 3 * b] = (n + 512) / 512 * 8;
  }
}

// Rank 0: the container with room for every payload member; h_offsets[15] = file offset of each member's data
// (payload order: fixed, entry_sizes/elements, shared/elements, prev_A/bits, prev_A/subaccum, prev_A/accum, ...).
// `fixed` is written here.
extern "C" int bg_seqset_sharded_create_file(const char* path, uint64_t n_total, uint32_t max_len, const uint64_t* h_fixed5,
                                             int argc, const char* const* argv, uint64_t* h_offsets, char* h_uuid37) {
  if (!path || !h_fixed5 || !h_offsets) { set_error("bg_seqset_sharded_create_file: null argument"); return BG_EINVAL; }
  const std::string uuid = make_uuid();
  if (h_uuid37) snprintf(h_uuid37, 37, "%s", uuid.c_str());
  uint64_t sz[kPayloadParts];
  payload_sizes(n_total, n_total ? max_len : 0, sz);
  std::vector<ZipPart> parts;
  BG_TRY(seqset_members(n_total, n_total ? max_len : 0, uuid, argc, argv,
                        [](const std::string&, std::string*) { return BG_OK; }, sz, &parts));
  for (ZipPart& p : parts)
    if (p.name == "fixed") { p.hole_size = 0; p.data.assign(reinterpret_cast<const char*>(h_fixed5), 40); }
  std::vector<uint64_t> offs;
  BG_TRY(spiral_zip_write_layout(path, parts, &offs));
  int pi = 0;
  for (size_t i = 0; i < parts.size(); ++i)
    if (parts[i].raw_crc0) h_offsets[pi++] = offs[i];
  return BG_OK;
}

// Every rank: the slice's bytes of every member, in place.  first_entry = global index of the slice's first
// entry (a multiple of 512); is_last: this rank also owns the closing accum slot; h_base_pop[4] = set prev bits
// per base in all entries before first_entry.
extern "C" int bg_seqset_slice_write(bg_ctx* ctx, const bg_seqset* slice, const char* path, const uint64_t* h_offsets,
                                     uint64_t first_entry, int is_last, const uint64_t* h_base_pop4) {
  if (!ctx || !slice || !path || !h_offsets || !h_base_pop4) { set_error("bg_seqset_slice_write: null argument"); return BG_EINVAL; }
  if (first_entry % 512) { set_error("bg_seqset_slice_write: slice starts at entry %llu, not a multiple of 512", (unsigned long long)first_entry); return BG_EINVAL; }
  if (!is_last && slice->n % 512) { set_error("bg_seqset_slice_write: an inner slice of %llu entries is not 512-aligned", (unsigned long long)slice->n); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  FILE* f = fopen(path, "r+b");
  if (!f) { set_error("Could not open %s for writing: %s", path, strerror(errno)); return BG_EIO; }
  const uint64_t m = slice->n;
  const int bs = int(bits_for_value(slice->max_len)), bh = int(bits_for_value(slice->max_len ? slice->max_len - 1 : 0));
  auto put = [&](int member, uint64_t byte_off, const std::string& data) -> int {
    if (data.empty()) return BG_OK;
    if (fseeko(f, off_t(h_offsets[member] + byte_off), SEEK_SET) != 0 || fwrite(data.data(), 1, data.size(), f) != data.size()) {
      set_error("write error on %s: %s", path, strerror(errno));
      return BG_EIO;
    }
    return BG_OK;
  };
  int rc = BG_OK;
  std::string data;
  if (rc == BG_OK) rc = part_bytes(ctx, slice, "entry_sizes/elements", &data);
  if (rc == BG_OK) rc = put(1, first_entry * bs / 8, data);
  if (rc == BG_OK) rc = part_bytes(ctx, slice, "shared/elements", &data);
  if (rc == BG_OK) rc = put(2, first_entry * bh / 8, data);
  for (int b = 0; b < 4 && rc == BG_OK; ++b) {
    const std::string dir = std::string("prev_") + "ACGT"[b];
    rc = part_bytes(ctx, slice, dir + "/bits", &data);
    if (rc == BG_OK) rc = put(3 + 3 * b, first_entry / 8, data);
    if (rc == BG_OK) rc = part_bytes(ctx, slice, dir + "/subaccum", &data);
    if (rc == BG_OK) rc = put(4 + 3 * b, first_entry / 512 * 8, data);
    if (rc == BG_OK) rc = part_bytes(ctx, slice, dir + "/accum", &data);
    if (rc == BG_OK) {
      uint64_t* a = reinterpret_cast<uint64_t*>(&data[0]);
      const uint64_t na = data.size() / 8, keep = is_last ? na : m / 512;
      for (uint64_t i = 0; i < keep; ++i) a[i] += h_base_pop4[b];
      data.resize(keep * 8);
      rc = put(5 + 3 * b, first_entry / 512 * 8, data);
    }
  }
  if (fclose(f) != 0 && rc == BG_OK) { set_error("write error on %s: %s", path, strerror(errno)); rc = BG_EIO; }
  return rc;
}

static bool json_u64(const std::string& js, const char* key, uint64_t* out) {
  std::string pat = std::string("\"") + key + "\":";
  size_t p = js.find(pat);
  if (p == std::string::npos) return false;
  *out = strtoull(js.c_str() + p + pat.size(), nullptr, 10);
  return true;
}

namespace bg {
__global__ void k_max_u8(const uint8_t* __restrict__ a, uint64_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) m = a[i] > m ? a[i] : m;
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = uint32_t(__shfl_xor(int(m), d, 64));
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
}  // namespace bg

extern "C" int bg_seqset_open(bg_ctx* ctx, const char* path, bg_seqset** out) {
  if (!ctx || !path || !out) { set_error("bg_seqset_open: null argument"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  std::map<std::string, std::string> parts;
  // (the large members stay in the mapped file: they go up from there)
  std::map<std::string, std::pair<const char*, uint64_t>> big;
  std::shared_ptr<void> mapping;
  BG_TRY(spiral_zip_read_views(path, &parts, &big, &mapping));
  auto has = [&](const std::string& name) { return big.count(name) || parts.count(name); };
  auto bytes_of = [&](const std::string& name) -> std::pair<const char*, uint64_t> {
    auto b = big.find(name);
    if (b != big.end()) return b->second;
    auto s = parts.find(name);
    if (s != parts.end()) return std::make_pair(s->second.data(), uint64_t(s->second.size()));
    return std::make_pair(static_cast<const char*>(nullptr), uint64_t(0));
  };
  // (measured: the large members through the pinned double buffer instead of the driver's staging of pageable memory -- the same
  // 0.14-0.18 s for a chr20-sized seqset; the uploads are not what the open waits for)
  auto up = [&](void* d_dst, const char* src, uint64_t bytes) -> int {
    if (bytes) BG_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return BG_OK;
  };
  uint64_t n = 0;
  if (!parts.count("seqset.json") || !json_u64(parts["seqset.json"], "num_entries", &n)) {
    set_error("%s: no seqset.json", path);
    return BG_EIO;
  }
  std::unique_ptr<bg_seqset> ss(new bg_seqset);
  ss->ctx = ctx;
  ss->n = n;
  {
    const std::string& fi = parts["file_info.json"];
    size_t p = fi.find("\"uuid\":\"");
    if (p != std::string::npos) ss->uuid = fi.substr(p + 8, fi.find('"', p + 8) - (p + 8));
  }
  BG_TRY(ss->sizes.alloc(ctx, n));
  BG_TRY(ss->shared.alloc(ctx, n));
  for (int which = 0; which < 2; ++which) {
    std::string dir = which ? "shared" : "entry_sizes";
    uint8_t* dst = which ? ss->shared.p : ss->sizes.p;
    if (has(dir)) {  // old layout: raw uint8 (int_map_interface.cpp detect_subpart_or_uint8_membuf)
      const auto raw = bytes_of(dir);
      if (raw.second != n) { set_error("%s: bad %s size", path, dir.c_str()); return BG_EIO; }
      BG_TRY(up(dst, raw.first, n));
    } else {
      const std::string& md = parts[dir + "/packed_varbit_vector.json"];
      const auto el = bytes_of(dir + "/elements");
      uint64_t bpv = 0, cnt = 0;
      if (!json_u64(md, "bits_per_value", &bpv) || !json_u64(md, "element_count", &cnt) || cnt != n || bpv > 8 ||
          el.second != ((n * bpv + 63) / 64) * 8) {
        set_error("%s: unsupported %s encoding", path, dir.c_str());
        return BG_EIO;
      }
      DevBuf<uint64_t> packed;
      BG_TRY(packed.alloc(ctx, el.second / 8 + 1));
      BG_TRY(up(packed.p, el.first, el.second));
      if (n) {
        k_unpack_varbit<<<cdiv(n, 256), 256, 0, ctx->stream>>>(packed.p, n, int(bpv), dst);
        BG_LAUNCH_CHECK();
      }
      BG_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  uint64_t nwords = (n + 63) / 64;
  for (int b = 0; b < 4; ++b) {
    std::string name = std::string("prev_") + "ACGT"[b] + "/bits";
    const auto bits = bytes_of(name);
    if (!has(name) || bits.second != nwords * 8) { set_error("%s: bad %s", path, name.c_str()); return BG_EIO; }
    BG_TRY(ss->bits[b].alloc(ctx, nwords));
    BG_TRY(up(ss->bits[b].p, bits.first, nwords * 8));
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  // max_len from sizes (on the device: the bytes do not come back for it)
  {
    DevBuf<uint32_t> d_mx;
    BG_TRY(d_mx.alloc(ctx, 1));
    BG_HIP(hipMemsetAsync(d_mx.p, 0, 4, ctx->stream));
    if (n) {
      k_max_u8<<<unsigned(std::min<uint64_t>(cdiv(n, 256 * 16), 65535)), 256, 0, ctx->stream>>>(ss->sizes.p, n, d_mx.p);
      BG_LAUNCH_CHECK();
    }
    uint32_t mx = 0;
    BG_HIP(copy_d2h(ctx, &mx, d_mx.p, 4));
    ss->max_len = mx;
  }
  BG_TRY(seqset_finalize_device(ctx, ss.get()));
  // cross-check the stored rank index against the recomputed one
  for (int b = 0; b < 4; ++b) {
    std::string name = std::string("prev_") + "ACGT"[b] + "/accum";
    std::string mine;
    BG_TRY(part_bytes(ctx, ss.get(), name, &mine));
    const auto stored = bytes_of(name);
    if (has(name) && (stored.second != mine.size() || memcmp(stored.first, mine.data(), mine.size()) != 0)) {
      set_error("%s: %s does not match its bits", path, name.c_str());
      return BG_EIO;
    }
  }
  *out = ss.release();
  return BG_OK;
}

// ---- whole create path

extern "C" void bg_create_params_default(bg_create_params* p) {
  if (!p) return;
  p->kmer_size = 30;                      // biograph_create.cpp:272
  p->min_kmer_count = 5;                  // :275
  p->correct.max_corrections = 8;         // :288
  p->correct.min_good_run = 2;            // :291
  p->correct.trim_after_portion = 0.7f;   // :281
}

static int create_impl(bg_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                       const bg_create_params* p, int is_paired, bg_seqset** out, bg_readmap** out_rm,
                       bg_build_stats* stats) {
  if (!ctx || !p || !out) { set_error("bg_create: null argument"); return BG_EINVAL; }
  bg_build_stats st;
  memset(&st, 0, sizeof st);
  bg_reads* reads = nullptr;
  bg_kmerset* ks = nullptr;
  bg_reads* corrected = nullptr;
  int rc = bg_pack_reads(ctx, d_seq, d_offs, n_reads, &reads);
  if (rc == BG_OK) {
    st.n_reads = n_reads;
    st.n_bases = reads->total_bases;
    rc = kmer_count_impl(ctx, reads, p->kmer_size, p->min_kmer_count, false, &ks);  // nobody reads the start flags here
  }
  bg_correct_stats cs;
  memset(&cs, 0, sizeof cs);
  if (rc == BG_OK) {
    st.n_kmers = ks->instances;
    st.n_distinct_kmers = ks->distinct;
    st.n_kept_kmers = ks->size;
    rc = bg_correct_reads(ctx, reads, ks, &p->correct, &corrected, &cs);
  }
  bg_reads_free(reads);
  bg_kmerset_free(ks);
  if (rc == BG_OK) {
    st.n_corrected_reads = cs.corrected_reads;
    st.n_corrected_bases = cs.corrected_bases;
    bg_build_stats s2;
    memset(&s2, 0, sizeof s2);
    rc = bg_seqset_build(ctx, corrected, p->kmer_size, out, &s2);
    st.n_records = s2.n_records;
    st.n_tied_records = s2.n_tied_records;
    st.n_entries = s2.n_entries;
    st.max_len = s2.max_len;
  }
  if (rc == BG_OK && out_rm) {
    rc = bg_readmap_build(ctx, *out, corrected, is_paired, out_rm);
    if (rc != BG_OK) { bg_seqset_free(*out); *out = nullptr; }
  }
  bg_reads_free(corrected);
  if (stats) *stats = st;
  return rc;
}

extern "C" int bg_create(bg_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                         const bg_create_params* p, bg_seqset** out, bg_build_stats* stats) {
  return create_impl(ctx, d_seq, d_offs, n_reads, p, 0, out, nullptr, stats);
}

extern "C" int bg_create_sample(bg_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offs, uint64_t n_reads,
                                const bg_create_params* p, int is_paired, bg_seqset** out, bg_readmap** out_rm,
                                bg_build_stats* stats) {
  if (!out_rm) { set_error("bg_create_sample: null argument"); return BG_EINVAL; }
  return create_impl(ctx, d_seq, d_offs, n_reads, p, is_paired, out, out_rm, stats);
}

// ---- `bgbinary discovery --ref-map FILE` (biograph_variants.cpp:171,643-666): the reference map kept between runs.  The
// reference's file is a spiral container holding one packed_vector<unsigned, 8> (ref_map.cpp:24-38: one byte per seqset entry =
// [fwd flag | rev flag | count:6], ref_map.h:49-53): file_info.json, part_info.json ("packed_vector" 1.0.0), packed_data (the
// bytes, in 64-bit words), packed_vector.json.
extern "C" int bg_ref_map_write(const char* path, const uint8_t* h_map, uint64_t n, int argc, const char* const* argv) {
  if (!path || (n && !h_map)) { set_error("bg_ref_map_write: null argument"); return BG_EINVAL; }
  std::vector<ZipPart> parts;
  parts.push_back(ZipPart{"file_info.json", file_info_json(make_uuid(), argc, argv), false});
  parts.push_back(ZipPart{"part_info.json", version_json("packed_vector", 1, 0, 0), false});
  std::string data(size_t((n + 7) / 8 * 8), '\0');
  if (n) memcpy(&data[0], h_map, n);
  parts.push_back(ZipPart{"packed_data", data, true});
  char js[96];
  snprintf(js, sizeof js, "{\"value_count\":%llu,\"value_width_bits\":8}", (unsigned long long)n);
  parts.push_back(ZipPart{"packed_vector.json", js, false});
  return spiral_zip_write(path, parts);
}

extern "C" int bg_ref_map_read(const char* path, uint8_t* h_map_out, uint64_t n) {
  if (!path || (n && !h_map_out)) { set_error("bg_ref_map_read: null argument"); return BG_EINVAL; }
  std::map<std::string, std::string> parts;
  BG_TRY(spiral_zip_read(path, &parts));
  auto js = parts.find("packed_vector.json");
  auto data = parts.find("packed_data");
  if (js == parts.end() || data == parts.end()) { set_error("%s: not a reference map (no packed_vector at its root)", path); return BG_EINVAL; }
  unsigned long long count = 0, width = 0;
  const size_t pc = js->second.find("\"value_count\":"), pw = js->second.find("\"value_width_bits\":");
  if (pc != std::string::npos) count = strtoull(js->second.c_str() + pc + 14, nullptr, 10);
  if (pw != std::string::npos) width = strtoull(js->second.c_str() + pw + 19, nullptr, 10);
  if (width != 8) { set_error("%s: values of %llu bits, a reference map holds 8", path, width); return BG_EINVAL; }
  if (count != n || data->second.size() < n) {  // ref_map.cpp:37 CHECK_EQ(m_ref_map->size(), m_seqset->size())
    set_error("%s: a reference map of %llu entries, the seqset has %llu", path, count, (unsigned long long)n);
    return BG_EINVAL;
  }
  if (n) memcpy(h_map_out, data->second.data(), n);
  return BG_OK;
}
