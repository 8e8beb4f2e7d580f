// compact.h — stream compaction of flagged positions (flags: one byte per element).
// count per 2048-element tile -> device-wide scan -> emit ascending positions.
#pragma once
#include "bg_internal.h"

namespace bg {

constexpr int CP_THREADS = 256;
constexpr int CP_ITEMS = 8;
constexpr int CP_TILE = CP_THREADS * CP_ITEMS;

__device__ __forceinline__ uint32_t cp_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__global__ __launch_bounds__(CP_THREADS) static void cp_count_kernel(const uint8_t* __restrict__ flags, uint64_t n,
                                                                     uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t sm[4];
  uint64_t base = uint64_t(blockIdx.x) * CP_TILE + uint64_t(threadIdx.x) * CP_ITEMS;
  uint32_t c = 0;
  if (base + CP_ITEMS <= n && (reinterpret_cast<uintptr_t>(flags) & 7) == 0) {
    // the thread's eight flags in one load (a byte a lane and load moves 64 bytes per instruction)
    uint64_t f8 = *reinterpret_cast<const uint64_t*>(flags + base);
    f8 |= f8 >> 4;
    f8 |= f8 >> 2;
    f8 |= f8 >> 1;  // bit 0 of every byte = that byte is not zero
    c = uint32_t(__popcll(f8 & 0x0101010101010101ull));
  } else {
#pragma unroll
    for (int i = 0; i < CP_ITEMS; ++i)
      if (base + i < n && flags[base + i]) ++c;
  }
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = cp_wave_incl_scan(c, lane);
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

template <class IdxT>
__global__ __launch_bounds__(CP_THREADS) static void cp_emit_kernel(const uint8_t* __restrict__ flags, uint64_t n,
                                                                    const uint64_t* __restrict__ tile_offs,
                                                                    IdxT* __restrict__ out) {
  __shared__ uint32_t sm[4];
  uint64_t base = uint64_t(blockIdx.x) * CP_TILE + uint64_t(threadIdx.x) * CP_ITEMS;
  uint8_t f[CP_ITEMS];
  uint32_t c = 0;
  if (base + CP_ITEMS <= n && (reinterpret_cast<uintptr_t>(flags) & 7) == 0) {
    const uint64_t f8 = *reinterpret_cast<const uint64_t*>(flags + base);
#pragma unroll
    for (int i = 0; i < CP_ITEMS; ++i) {
      f[i] = uint8_t(f8 >> (8 * i));
      if (f[i]) ++c;
    }
  } else {
#pragma unroll
    for (int i = 0; i < CP_ITEMS; ++i) {
      f[i] = (base + i < n) ? flags[base + i] : 0;
      if (f[i]) ++c;
    }
  }
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = cp_wave_incl_scan(c, lane);
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  uint32_t wbase = 0;
  for (int i = 0; i < w; ++i) wbase += sm[i];
  uint64_t o = tile_offs[blockIdx.x] + wbase + inc - c;
#pragma unroll
  for (int i = 0; i < CP_ITEMS; ++i)
    if (f[i]) out[o++] = IdxT(base + i);
}

// Writes the ascending positions of non-zero flags into `out` (allocated here) and their number
// into *h_count (host; synchronises the stream).
template <class IdxT>
static int compact_flagged(bg_ctx* ctx, const uint8_t* d_flags, uint64_t n, DevBuf<IdxT>* out, uint64_t* h_count) {
  *h_count = 0;
  if (n == 0) return out->alloc(ctx, 1);
  uint64_t nt = (n + CP_TILE - 1) / CP_TILE;
  DevBuf<uint32_t> counts;
  DevBuf<uint64_t> offs, total;
  BG_TRY(counts.alloc(ctx, nt));
  BG_TRY(offs.alloc(ctx, nt));
  BG_TRY(total.alloc(ctx, 1));
  cp_count_kernel<<<unsigned(nt), CP_THREADS, 0, ctx->stream>>>(d_flags, n, counts.p);
  BG_LAUNCH_CHECK();
  BG_TRY(exclusive_scan_u32_to_u64(ctx, counts.p, offs.p, nt, total.p));
  // The positions go out before the host knows how many there are, into room for all n (the arena takes the unused tail
  // back in place): the wait for the count runs beside the kernel instead of in front of it (a 40 us hole a call, eight
  // calls a build step).  Inputs too large for that wait first, as before.
  const bool ahead = n * sizeof(IdxT) <= (size_t(2) << 30) && out->alloc(ctx, n) == BG_OK;
  if (ahead) {
    cp_emit_kernel<IdxT><<<unsigned(nt), CP_THREADS, 0, ctx->stream>>>(d_flags, n, offs.p, out->p);
    BG_LAUNCH_CHECK();
  }
  BG_HIP(hipMemcpyAsync(h_count, total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  if (ahead) {
    out->shrink(*h_count ? *h_count : 1);
    return BG_OK;
  }
  BG_TRY(out->alloc(ctx, *h_count ? *h_count : 1));
  if (*h_count) {
    cp_emit_kernel<IdxT><<<unsigned(nt), CP_THREADS, 0, ctx->stream>>>(d_flags, n, offs.p, out->p);
    BG_LAUNCH_CHECK();
  }
  return BG_OK;
}

}  // namespace bg
