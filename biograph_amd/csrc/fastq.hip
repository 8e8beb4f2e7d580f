// fastq.hip — FASTQ import on the GPU (fastq_reader + read_importer_base::read_fastq,
// modules/build_seqset/read_importer.cpp:604, read_importer.h:65-141; --cut-reads read_importer.cpp:157-169).
//
// The reference parses FASTQ with one reader thread per file.  Here the host only moves bytes: the text of
// every file goes to HBM as it is (bg_dev_upload_pinned: parallel copies into two pinned staging buffers,
// DMA of one overlapping the fill of the other) and the parsing is three streaming passes at HBM speed:
// newline positions (count / scan / emit), record spans with the checks the reader makes ('@' and '+' lines,
// four lines per record), and a gather of the sequence lines into the concatenated layout bg_pack_reads
// takes.  Qualities are dropped as in the reference (k_keep_quality_scores = false, read_importer.cpp:22).
// Pairing follows the reference: two files read in lock step (mates = records of equal index) or one
// interleaved file (mates = records 2i, 2i + 1); mates end up adjacent (reads 2i, 2i + 1).
#include <string.h>
#include <unistd.h>

#include <thread>
#include <vector>

#include "bg_internal.h"

namespace bg {

constexpr int FQ_THREADS = 256;
constexpr int FQ_ITEMS = 16;                       // bytes per thread (one 16-byte load)
constexpr int FQ_TILE = FQ_THREADS * FQ_ITEMS;     // 4096 bytes per workgroup

__device__ __forceinline__ uint32_t fq_nl_mask(const uint8_t* __restrict__ text, uint64_t n, uint64_t base) {
  uint32_t m = 0;
  if (base + FQ_ITEMS <= n && ((reinterpret_cast<uintptr_t>(text) + base) & 15) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + base);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (((w[q] >> (8 * b)) & 0xff) == '\n') m |= 1u << (4 * q + b);
  } else {
    for (int i = 0; i < FQ_ITEMS; ++i)
      if (base + i < n && text[base + i] == '\n') m |= 1u << i;
  }
  return m;
}

__device__ __forceinline__ uint32_t fq_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__global__ __launch_bounds__(FQ_THREADS) void fq_count_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                              uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t sm[4];
  const uint64_t base = uint64_t(blockIdx.x) * FQ_TILE + uint64_t(threadIdx.x) * FQ_ITEMS;
  const uint32_t c = __popc(fq_nl_mask(text, n, base));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = fq_wave_incl_scan(c, lane);
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(FQ_THREADS) void fq_emit_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                             const uint64_t* __restrict__ tile_offs,
                                                             uint64_t* __restrict__ nl) {
  __shared__ uint32_t sm[4];
  const uint64_t base = uint64_t(blockIdx.x) * FQ_TILE + uint64_t(threadIdx.x) * FQ_ITEMS;
  uint32_t m = fq_nl_mask(text, n, base);
  const uint32_t c = __popc(m);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = fq_wave_incl_scan(c, lane);
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  uint32_t wbase = 0;
  for (int i = 0; i < w; ++i) wbase += sm[i];
  uint64_t o = tile_offs[blockIdx.x] + wbase + inc - c;
  while (m) {
    const int i = __ffs(int(m)) - 1;
    m &= m - 1;
    nl[o++] = base + uint64_t(i);
  }
}

// Record r of one source: lines 4r .. 4r+3.  Line L spans [L ? nl[L-1] + 1 : 0, L < n_nl ? nl[L] : n).
// err[0] = lowest malformed record + 1 (0 = none), err[1] = lowest record whose cut region is empty + 1.
__global__ void fq_spans_kernel(const uint8_t* __restrict__ text, uint64_t n, const uint64_t* __restrict__ nl, uint64_t n_nl,
                                uint64_t n_rec, uint64_t id_base, int id_stride, int id_phase, uint32_t cut_start,
                                uint32_t cut_end, uint64_t* __restrict__ addr, uint32_t* __restrict__ lens,
                                unsigned long long* __restrict__ err) {
  const uint64_t r = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n_rec) return;
  auto line_start = [&](uint64_t L) { return L ? nl[L - 1] + 1 : 0ull; };
  auto line_end = [&](uint64_t L) { return L < n_nl ? nl[L] : n; };
  const uint64_t h0 = line_start(4 * r), s0 = line_start(4 * r + 1), p0 = line_start(4 * r + 2);
  uint64_t s1 = line_end(4 * r + 1);
  if (h0 >= n || text[h0] != '@' || p0 >= n || text[p0] != '+') atomicMin(&err[0], (unsigned long long)(r + 1));
  if (s1 > s0 && text[s1 - 1] == '\r') --s1;
  {  // fastq.cpp:121-124: "Quality line not same length as sequence" (a truncated record must not pass)
    const uint64_t q0 = line_start(4 * r + 3);
    uint64_t q1 = line_end(4 * r + 3);
    if (q1 > q0 && q0 < n && text[q1 - 1] == '\r') --q1;
    if (q0 > n || q1 - (q0 < q1 ? q0 : q1) != s1 - s0) atomicMin(&err[0], (unsigned long long)(r + 1));
  }
  uint64_t start = s0, len = s1 - s0;
  if (cut_end) {  // read_batch::cut_reads, read_importer.cpp:157-169
    const uint64_t this_end = len < cut_end ? len : cut_end;
    if (this_end <= cut_start) { atomicMin(&err[1], (unsigned long long)(r + 1)); len = 0; }
    else { start += cut_start; len = this_end - cut_start; }
  }
  const uint64_t id = id_base + r * uint64_t(id_stride) + uint64_t(id_phase);
  addr[id] = reinterpret_cast<uint64_t>(text) + start;
  lens[id] = uint32_t(len);
}

// one wave per read: the bases of read i go to seq[offs[i] ..)
__global__ __launch_bounds__(256) void fq_gather_kernel(const uint64_t* __restrict__ addr, const uint32_t* __restrict__ lens,
                                                        const uint64_t* __restrict__ offs, uint64_t n_reads,
                                                        uint8_t* __restrict__ seq) {
  const uint64_t i = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (i >= n_reads) return;
  const int lane = threadIdx.x & 63;
  const uint8_t* src = reinterpret_cast<const uint8_t*>(addr[i]);
  uint8_t* dst = seq + offs[i];
  const uint32_t len = lens[i];
  for (uint32_t j = lane; j < len; j += 64) dst[j] = src[j];
}

__global__ void fq_set_last_offset(const uint64_t* __restrict__ offs, const uint32_t* __restrict__ lens, uint64_t n,
                                   uint64_t* __restrict__ out_last) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out_last = n ? offs[n - 1] + lens[n - 1] : 0;
}

}  // namespace bg

using namespace bg;

// Host -> device through two pinned staging buffers: `threads` host threads fill one (fill(dst, offset, bytes))
// while the DMA engine drains the other (a pageable hipMemcpy stages through one small internal buffer,
// single-threaded).
int bg_ctx::pinned_buffers() {
  for (int b = 0; b < 2; ++b)
    if (!pinned[b] && hipHostMalloc(&pinned[b], kPinnedChunk, hipHostMallocDefault) != hipSuccess) {
      pinned[b] = nullptr;
      bg::set_error("pinned staging buffer: out of memory");
      return BG_ENOMEM;
    }
  return BG_OK;
}

template <class Fill>
static int staged_upload(bg_ctx* ctx, void* d_dst, uint64_t bytes, int threads, Fill fill, const char* who) {
  BG_HIP(hipSetDevice(ctx->device));
  if (!bytes) return BG_OK;
  if (threads < 1) threads = int(std::thread::hardware_concurrency());
  if (threads < 1) threads = 1;
  if (threads > 12) threads = 12;  // 8 threads fill a 64 MiB piece as fast as the bus drains it (tools/upload_rate.py); more only add start-up
  const uint64_t chunk = bg_ctx::kPinnedChunk;
  BG_TRY(ctx->pinned_buffers());
  void* const* pin = ctx->pinned;
  hipEvent_t ev[2];
  BG_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
  BG_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  int rc = BG_OK;
  uint64_t done = 0;
  for (int c = 0; done < bytes && rc == BG_OK; ++c) {
    const int b = c & 1;
    const uint64_t sz = bytes - done < chunk ? bytes - done : chunk;
    if (c >= 2 && hipEventSynchronize(ev[b]) != hipSuccess) { rc = BG_ENODEV; break; }
    char* dst = static_cast<char*>(pin[b]);
    const int nt = sz < (8u << 20) ? 1 : threads;
    std::vector<int> ok(nt, 1);
    if (nt == 1) {
      ok[0] = fill(dst, done, sz) ? 1 : 0;
    } else {
      std::vector<std::thread> pool;
      const uint64_t per = (sz + nt - 1) / nt;
      for (int t = 0; t < nt; ++t) {
        const uint64_t lo = uint64_t(t) * per, hi = lo + per < sz ? lo + per : sz;
        if (lo < hi) pool.emplace_back([&, t, lo, hi] { ok[t] = fill(dst + lo, done + lo, hi - lo) ? 1 : 0; });
      }
      for (auto& th : pool) th.join();
    }
    for (int v : ok) if (!v) rc = BG_EIO;
    if (rc != BG_OK) break;
    if (hipMemcpyAsync(static_cast<char*>(d_dst) + done, pin[b], sz, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipEventRecord(ev[b], ctx->stream) != hipSuccess) { rc = BG_ENODEV; break; }
    done += sz;
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == BG_OK) rc = BG_ENODEV;
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
  if (rc == BG_EIO) set_error("%s: read error", who);
  else if (rc != BG_OK) set_error("%s: HIP error during the staged copy", who);
  return rc;
}

// device -> host the same way round: piece c + 1 crosses the bus into one pinned buffer while the host's threads copy
// piece c out of the other
int bg::staged_download(bg_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes, int threads) {
  BG_HIP(hipSetDevice(ctx->device));
  if (!bytes) return BG_OK;
  if (threads < 1) threads = int(std::thread::hardware_concurrency());
  if (threads < 1) threads = 1;
  if (threads > 32) threads = 32;
  const uint64_t chunk = bg_ctx::kPinnedChunk;
  BG_TRY(ctx->pinned_buffers());
  void* const* pin = ctx->pinned;
  auto drain = [&](int b, uint64_t off, uint64_t sz) {
    char* dst = static_cast<char*>(h_dst) + off;
    const char* src = static_cast<const char*>(pin[b]);
    const int nt = sz < (8u << 20) ? 1 : threads;
    if (nt == 1) { memcpy(dst, src, sz); return; }
    std::vector<std::thread> pool;
    const uint64_t per = (sz + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
      const uint64_t lo = uint64_t(t) * per, hi = lo + per < sz ? lo + per : sz;
      if (lo < hi) pool.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& th : pool) th.join();
  };
  uint64_t issued = 0, prev_off = 0, prev_sz = 0;
  for (int c = 0; issued < bytes; ++c) {
    const int b = c & 1;
    const uint64_t sz = bytes - issued < chunk ? bytes - issued : chunk;
    BG_HIP(hipMemcpyAsync(pin[b], static_cast<const char*>(d_src) + issued, sz, hipMemcpyDeviceToHost, ctx->stream));
    if (c) drain(b ^ 1, prev_off, prev_sz);  // the previous piece is complete (synchronised below, one iteration ago)
    BG_HIP(hipStreamSynchronize(ctx->stream));
    prev_off = issued;
    prev_sz = sz;
    issued += sz;
  }
  drain(int(((issued + chunk - 1) / chunk - 1) & 1), prev_off, prev_sz);
  return BG_OK;
}

extern "C" int bg_dev_upload_pinned(bg_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes, int threads) {
  if (!ctx || (bytes && (!d_dst || !h_src))) { set_error("bg_dev_upload_pinned: null argument"); return BG_EINVAL; }
  const char* src = static_cast<const char*>(h_src);
  return staged_upload(ctx, d_dst, bytes, threads,
                       [src](char* dst, uint64_t off, uint64_t n) { memcpy(dst, src + off, n); return true; }, "bg_dev_upload_pinned");
}

// The same from an open file: every thread pread()s its slice of a chunk straight into the pinned buffer (no mapping
// of the file, so no page-table faults on 4 KiB pages of a multi-gigabyte FASTQ).
extern "C" int bg_dev_upload_file(bg_ctx* ctx, void* d_dst, int fd, uint64_t file_offset, uint64_t bytes, int threads) {
  if (!ctx || fd < 0 || (bytes && !d_dst)) { set_error("bg_dev_upload_file: bad argument"); return BG_EINVAL; }
  return staged_upload(ctx, d_dst, bytes, threads,
                       [fd, file_offset](char* dst, uint64_t off, uint64_t n) {
                         while (n) {
                           const ssize_t got = pread(fd, dst, n, off_t(file_offset + off));
                           if (got <= 0) return false;
                           dst += got;
                           off += uint64_t(got);
                           n -= uint64_t(got);
                         }
                         return true;
                       }, "bg_dev_upload_file");
}

extern "C" int bg_fastq_import(bg_ctx* ctx, uint32_t n_src, const uint8_t* const* d_text, const uint64_t* h_bytes, int paired_files,
                               int interleaved, uint32_t cut_start, uint32_t cut_end, uint8_t** d_seq_out, uint64_t** d_offs_out,
                               uint64_t* n_reads_out, uint64_t* n_bases_out) {
  if (!ctx || !d_text || !h_bytes || !d_seq_out || !d_offs_out || !n_reads_out) { set_error("bg_fastq_import: null argument"); return BG_EINVAL; }
  if (paired_files && interleaved) { set_error("--interleaved and --pair exclude each other"); return BG_EINVAL; }
  if (paired_files && (n_src & 1)) { set_error("bg_fastq_import: mate files come in twos"); return BG_EINVAL; }
  if (cut_end && cut_end <= cut_start) { set_error("--cut-reads: the end of the range must lie past its start"); return BG_EINVAL; }
  BG_HIP(hipSetDevice(ctx->device));
  struct Src { DevBuf<uint64_t> nl; uint64_t n_nl = 0, n_rec = 0; };
  std::vector<Src> src(n_src);
  uint64_t n_reads = 0;
  for (uint32_t s = 0; s < n_src; ++s) {
    const uint64_t n = h_bytes[s];
    if (n && !d_text[s]) { set_error("bg_fastq_import: null text"); return BG_EINVAL; }
    const uint64_t nt = (n + FQ_TILE - 1) / FQ_TILE;
    if (nt >= (1ull << 31)) { set_error("bg_fastq_import: one source holds %llu bytes; split it", (unsigned long long)n); return BG_EINVAL; }
    uint64_t n_nl = 0;
    if (n) {
      DevBuf<uint32_t> counts;
      DevBuf<uint64_t> offs, total;
      BG_TRY(counts.alloc(ctx, nt));
      BG_TRY(offs.alloc(ctx, nt));
      BG_TRY(total.alloc(ctx, 1));
      ctx->tic("fastq_newlines");
      fq_count_kernel<<<unsigned(nt), FQ_THREADS, 0, ctx->stream>>>(d_text[s], n, counts.p);
      BG_LAUNCH_CHECK();
      BG_TRY(exclusive_scan_u32_to_u64(ctx, counts.p, offs.p, nt, total.p));
      BG_HIP(hipMemcpyAsync(&n_nl, total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
      BG_HIP(hipStreamSynchronize(ctx->stream));
      BG_TRY(src[s].nl.alloc(ctx, n_nl));
      if (n_nl) {
        fq_emit_kernel<<<unsigned(nt), FQ_THREADS, 0, ctx->stream>>>(d_text[s], n, offs.p, src[s].nl.p);
        BG_LAUNCH_CHECK();
      }
      ctx->toc(2 * n + n_nl * 8);
    }
    src[s].n_nl = n_nl;
    // lines = newlines (+ 1 for an unterminated last line); blank lines at the very end are not records
    uint64_t lines = n_nl;
    if (n) {
      char tail[64];
      const uint64_t tl = n < sizeof tail ? n : sizeof tail;
      BG_HIP(hipMemcpyAsync(tail, d_text[s] + (n - tl), tl, hipMemcpyDeviceToHost, ctx->stream));
      BG_HIP(hipStreamSynchronize(ctx->stream));
      if (tail[tl - 1] != '\n') {
        ++lines;
      } else {
        // a blank line is "\n" or "\r\n" (a CRLF file may end in "\r\n\r\n")
        uint64_t i = tl - 1;  // tail[i] is the newline that ends the last line
        while (lines > 0) {
          uint64_t j = i;
          if (j > 0 && tail[j - 1] == '\r') --j;
          if (j == 0 || tail[j - 1] != '\n') break;
          --lines;
          i = j - 1;
        }
      }
    }
    if (lines % 4) { set_error("source %u: malformed FASTQ (%llu lines are not a whole number of records)", s, (unsigned long long)lines); return BG_EINVAL; }
    src[s].n_rec = lines / 4;
    if (interleaved && (src[s].n_rec & 1)) { set_error("source %u: an interleaved file must hold an even number of records", s); return BG_EINVAL; }
    if (paired_files && (s & 1) && src[s].n_rec != src[s - 1].n_rec) { set_error("Pair files hold different numbers of reads"); return BG_EINVAL; }
    n_reads += src[s].n_rec;
  }
  DevBuf<uint64_t> addr, offs;
  DevBuf<uint32_t> lens;
  DevBuf<unsigned long long> err;
  BG_TRY(addr.alloc(ctx, n_reads));
  BG_TRY(lens.alloc(ctx, n_reads));
  BG_TRY(err.alloc(ctx, 2));
  BG_HIP(hipMemsetAsync(err.p, 0xff, 16, ctx->stream));
  uint64_t id_base = 0;
  ctx->tic("fastq_spans");
  for (uint32_t s = 0; s < n_src; ++s) {
    const uint64_t nr = src[s].n_rec;
    const int stride = paired_files ? 2 : 1, phase = paired_files ? int(s & 1) : 0;
    if (nr) {
      fq_spans_kernel<<<cdiv(nr, 256), 256, 0, ctx->stream>>>(d_text[s], h_bytes[s], src[s].nl.p, src[s].n_nl, nr, id_base, stride,
                                                             phase, cut_start, cut_end, addr.p, lens.p, err.p);
      BG_LAUNCH_CHECK();
    }
    if (!paired_files || (s & 1)) id_base += paired_files ? 2 * nr : nr;
  }
  ctx->toc(n_reads * 60);
  unsigned long long h_err[2] = {0, 0};
  BG_HIP(hipMemcpyAsync(h_err, err.p, 16, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (h_err[0] != ~0ull) { set_error("malformed FASTQ record %llu ('@' or '+' line missing, or quality line not the sequence's length)", h_err[0]); return BG_EINVAL; }
  if (h_err[1] != ~0ull) { set_error("--cut-reads: record %llu is shorter than the start of the range", h_err[1]); return BG_EINVAL; }
  uint64_t* d_offs = nullptr;
  uint8_t* d_seq = nullptr;
  void* q = nullptr;
  BG_TRY(ctx->alloc((n_reads + 1) * 8, &q));
  d_offs = static_cast<uint64_t*>(q);
  DevBuf<uint64_t> total;
  BG_TRY(total.alloc(ctx, 1));
  uint64_t n_bases = 0;
  int rc = exclusive_scan_u32_to_u64(ctx, lens.p, d_offs, n_reads, total.p);
  if (rc == BG_OK && hipMemcpyAsync(&n_bases, total.p, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = BG_ENODEV;
  if (rc == BG_OK && hipMemcpyAsync(d_offs + n_reads, total.p, 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = BG_ENODEV;
  if (rc == BG_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = BG_ENODEV;
  if (rc == BG_OK) rc = ctx->alloc(n_bases ? n_bases : 1, &q);
  if (rc != BG_OK) { ctx->release(d_offs); if (rc == BG_ENODEV) set_error("bg_fastq_import: HIP error"); return rc; }
  d_seq = static_cast<uint8_t*>(q);
  if (n_reads) {
    ctx->tic("fastq_gather");
    fq_gather_kernel<<<cdiv(n_reads, 4), 256, 0, ctx->stream>>>(addr.p, lens.p, d_offs, n_reads, d_seq);
    ctx->toc(n_bases * 2 + n_reads * 20);
  }
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
    ctx->release(d_offs);
    ctx->release(d_seq);
    set_error("bg_fastq_import: HIP error in the gather");
    return BG_ENODEV;
  }
  *d_seq_out = d_seq;
  *d_offs_out = d_offs;
  *n_reads_out = n_reads;
  if (n_bases_out) *n_bases_out = n_bases;
  return BG_OK;
}
