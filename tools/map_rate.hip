// map_rate.hip — how fast does a cold process get device memory?  hipMalloc alone / from several threads at once / through
// the virtual-memory API (hipMemCreate + hipMemMap) / hipMallocAsync, each followed by a first touch (hipMemsetAsync).
//   hipcc --offload-arch=gfx950 -O2 tools/map_rate.hip -o tools/map_rate.bin -lpthread && tools/map_rate.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

static void plain(size_t gib, int threads) {
  const size_t per = (gib << 30) / threads;
  std::vector<void*> p(threads, nullptr);
  std::vector<std::thread> th;
  const double t0 = now();
  for (int t = 0; t < threads; ++t) th.emplace_back([&, t] { (void)hipSetDevice(0); if (hipMalloc(&p[t], per) != hipSuccess) p[t] = nullptr; });
  for (auto& x : th) x.join();
  const double t1 = now();
  for (int t = 0; t < threads; ++t) if (p[t]) (void)hipMemsetAsync(p[t], 1, per, 0);
  (void)hipDeviceSynchronize();
  const double t2 = now();
  for (int t = 0; t < threads; ++t) if (p[t]) (void)hipMemsetAsync(p[t], 2, per, 0);
  (void)hipDeviceSynchronize();
  const double t3 = now();
  printf("hipMalloc %2zu GiB in %d thread(s): alloc %.3f s (%.1f us/MiB)  first touch %.3f s  second touch %.3f s\n", gib, threads, t1 - t0,
         (t1 - t0) * 1e6 / double(gib << 10), t2 - t1, t3 - t2);
  const double t4 = now();
  for (int t = 0; t < threads; ++t) if (p[t]) (void)hipFree(p[t]);
  printf("   free %.3f s\n", now() - t4);
}

static void vmm(size_t gib, size_t chunk_mib) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t total = gib << 30, chunk = ((chunk_mib << 20) + gran - 1) / gran * gran;
  void* va = nullptr;
  const double t0 = now();
  CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> hs;
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
    hs.push_back(h);
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &acc, 1));
  const double t1 = now();
  (void)hipMemsetAsync(va, 1, total, 0);
  (void)hipDeviceSynchronize();
  const double t2 = now();
  printf("VMM %2zu GiB in %zu MiB chunks (granularity %zu KiB): reserve+create+map+access %.3f s (%.1f us/MiB)  first touch %.3f s\n", gib,
         chunk >> 20, gran >> 10, t1 - t0, (t1 - t0) * 1e6 / double(gib << 10), t2 - t1);
  (void)hipMemUnmap(va, total);
  for (auto h : hs) (void)hipMemRelease(h);
  (void)hipMemAddressFree(va, total);
}

static void async_pool(size_t gib) {
  void* p = nullptr;
  const double t0 = now();
  CK(hipMallocAsync(&p, gib << 30, 0));
  CK(hipStreamSynchronize(0));
  const double t1 = now();
  (void)hipMemsetAsync(p, 1, gib << 30, 0);
  (void)hipDeviceSynchronize();
  const double t2 = now();
  printf("hipMallocAsync %2zu GiB: alloc %.3f s (%.1f us/MiB)  first touch %.3f s\n", gib, t1 - t0, (t1 - t0) * 1e6 / double(gib << 10), t2 - t1);
  (void)hipFreeAsync(p, 0);
  (void)hipDeviceSynchronize();
}

int main(int argc, char** argv) {
  // one measurement per process (the driver keeps what a process has mapped: a second measurement in the same process sees
  // memory the first one left behind):  map_rate.bin plain GIB THREADS | vmm GIB CHUNK_MIB | async GIB | upload GIB
  const double t0 = now();
  (void)hipSetDevice(0);
  (void)hipFree(nullptr);
  printf("runtime up in %.3f s\n", now() - t0);
  const char* mode = argc > 1 ? argv[1] : "plain";
  const size_t a = argc > 2 ? size_t(atoll(argv[2])) : 32, b = argc > 3 ? size_t(atoll(argv[3])) : 1;
  if (!strcmp(mode, "plain")) plain(a, int(b));
  else if (!strcmp(mode, "vmm")) vmm(a, b);
  else if (!strcmp(mode, "async")) async_pool(a);
  else if (!strcmp(mode, "upload")) {
    void* h = nullptr;
    void* d = nullptr;
    (void)hipHostMalloc(&h, size_t(1) << 30, 0);
    (void)hipMalloc(&d, size_t(1) << 30);
    std::thread up([&] { (void)hipSetDevice(0); hipStream_t s; (void)hipStreamCreate(&s); const double x = now(); for (int i = 0; i < 8; ++i) (void)hipMemcpyAsync(d, h, size_t(1) << 30, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); printf("   8 GiB uploaded beside it in %.3f s\n", now() - x); });
    if (b) vmm(a, b); else plain(a, 1);
    up.join();
  }
  return 0;
}
