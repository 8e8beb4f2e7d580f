// probe_rate.hip — microbenchmark: rate of independent random 64-byte bucket probes (what the k-mer
// table lookups are) as a function of table size and of probes in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_rate.hip -o /tmp/probe_rate && /tmp/probe_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MLP, int BYTES>
__global__ __launch_bounds__(256) void k_probe(const uint64_t* __restrict__ tab, uint64_t mask, uint64_t n_per_thread,
                                               uint64_t* __restrict__ out) {
  uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  uint64_t x = t * 0x9E3779B97F4A7C15ull + 12345;
  uint64_t acc = 0;
  for (uint64_t i = 0; i < n_per_thread; i += MLP) {
    uint64_t a[MLP];
#pragma unroll
    for (int m = 0; m < MLP; ++m) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      a[m] = ((x >> 20) & mask) & ~uint64_t(7);
    }
#pragma unroll
    for (int m = 0; m < MLP; ++m) {
      if (BYTES == 64) {
        const ulonglong2* p = reinterpret_cast<const ulonglong2*>(tab + a[m]);
        ulonglong2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        acc += v0.x ^ v0.y ^ v1.x ^ v1.y ^ v2.x ^ v2.y ^ v3.x ^ v3.y;
      } else {
        acc += tab[a[m]];
      }
    }
  }
  if (acc == 0x1234567) out[0] = acc;
}

template <int MLP, int BYTES>
static void run(const uint64_t* tab, uint64_t slots, uint64_t* out) {
  const uint64_t threads = 256ull * 256 * 8, per = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_probe<MLP, BYTES><<<threads / 256, 256>>>(tab, slots - 1, 64, out);
  hipEventRecord(e0);
  k_probe<MLP, BYTES><<<threads / 256, 256>>>(tab, slots - 1, per, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("table %6.0f MiB  bytes/probe %2d  in flight/lane %d : %7.1f G probes/s  (%.2f TB/s of lines)\n",
         slots * 8.0 / (1 << 20), BYTES, MLP, threads * per / (ms * 1e-3) / 1e9, threads * per * 64.0 / (ms * 1e-3) / 1e12);
}

int main() {
  uint64_t max_slots = 1ull << 30;  // 8 GiB
  uint64_t *tab, *out;
  if (getenv("PROBE_VMM")) {  // the table through the virtual-memory API in 1 GiB pieces at a 1 GiB-aligned address
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t piece = size_t(1) << 30, total = max_slots * 8;
    void* va = nullptr;
    hipMemAddressReserve(&va, total, piece, nullptr, 0);
    for (size_t off = 0; off < total; off += piece) {
      hipMemGenericAllocationHandle_t h;
      hipMemCreate(&h, piece, &prop, 0);
      hipMemMap((char*)va + off, piece, 0, h, 0);
      hipMemRelease(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemSetAccess(va, total, &acc, 1);
    tab = (uint64_t*)va;
    printf("table mapped through the VMM API in 1 GiB pieces\n");
  } else
  hipMalloc(&tab, max_slots * 8);
  hipMalloc(&out, 8);
  hipMemset(tab, 0x5a, max_slots * 8);
  for (uint64_t slots = 1ull << 26; slots <= max_slots; slots <<= 1) {
    run<1, 64>(tab, slots, out);
    run<4, 64>(tab, slots, out);
    run<4, 8>(tab, slots, out);
  }
  return 0;
}
