// comm.hip — the communicator of the N-rank build (include/bgseq.h "Communicator"; comm.h for the library's own callers).
//
// One rank per process and GPU, as the reference runs one `parallel_for` worker per partition over shared disk files
// (modules/build_seqset/part_repo.cpp:32-45, modules/biograph/biograph_create.cpp:431-811): here the partitions live in the
// ranks' HBM and what the reference moves through files moves through
//   * RCCL over xGMI (backend "rccl"): grouped ncclSend / ncclRecv on device buffers where they lie, every piece at most
//     1 GiB - 4 KiB (round 1 measured corrupt data at pieces of exactly 2^31 / 2^32 / 2^33 bytes on this stack), larger
//     exchanges in several groups; librccl is opened with dlopen when the first communicator is made;
//   * POSIX shared memory (backend "shm"): every rank stages what it sends in a segment of its own, the others copy their
//     pieces out of it — for several ranks that share ONE GPU (tests, a box without xGMI); RCCL refuses two ranks on a device.
// A rank alone short-cuts every collective to a copy unless BG_COMM_NO_SHORTCUT=1 (then its pieces go through the backend
// to itself: what lets a one-GPU box exercise the RCCL path).
#include "comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

namespace bg {
namespace {

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.lib) return BG_OK;
  // a process that already holds an RCCL (torch ships its own) keeps using that one
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names)
    if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
  if (!lib)
    for (const char* n : names)
      if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!lib) {
    set_error("bg_comm: cannot load librccl (%s)", dlerror());
    return BG_ENODEV;
  }
#define BG_SYM(field, name)                                              \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name)); \
  if (!g_rccl.field) {                                                   \
    set_error("bg_comm: librccl lacks %s", name);                        \
    return BG_ENODEV;                                                    \
  }
  BG_SYM(GetUniqueId, "ncclGetUniqueId")
  BG_SYM(CommInitRank, "ncclCommInitRank")
  BG_SYM(CommDestroy, "ncclCommDestroy")
  BG_SYM(CommAbort, "ncclCommAbort")
  BG_SYM(Send, "ncclSend")
  BG_SYM(Recv, "ncclRecv")
  BG_SYM(GroupStart, "ncclGroupStart")
  BG_SYM(GroupEnd, "ncclGroupEnd")
  BG_SYM(AllGather, "ncclAllGather")
  BG_SYM(AllReduce, "ncclAllReduce")
  BG_SYM(GetErrorString, "ncclGetErrorString")
  BG_SYM(GetVersion, "ncclGetVersion")
#undef BG_SYM
  g_rccl.lib = lib;
  return BG_OK;
}

int nccl_fail(ncclResult_t r, const char* what, int line) {
  set_error("RCCL error '%s' in %s (comm.hip:%d)", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what, line);
  return BG_EINTERNAL;
}
#define BG_NCCL(expr)                                               \
  do {                                                              \
    ncclResult_t r_ = (expr);                                       \
    if (r_ != ncclSuccess) return ::bg::nccl_fail(r_, #expr, __LINE__);   \
  } while (0)

// ---- the shared-memory fabric: a control block every rank maps, and one data segment per rank
constexpr size_t kSmall = size_t(256) << 10;  // per-rank slot of the host all-gather
struct FabricHdr {
  std::atomic<uint32_t> ready;    // 0x42474331 once rank 0 has laid the block out
  std::atomic<uint32_t> arrived, gen, failed;
  uint32_t world;
  uint32_t pad[11];
};
struct Fabric {
  FabricHdr* hdr = nullptr;
  uint64_t* seg_bytes = nullptr;  // [world]  published size of every rank's data segment
  uint64_t* offs = nullptr;       // [world][world]  byte offset, in the sender's segment, of its piece for each destination
  char* small = nullptr;          // [world][kSmall]
  size_t bytes = 0;
  static size_t size_for(int world) { return sizeof(FabricHdr) + size_t(world) * 8 + size_t(world) * world * 8 + size_t(world) * kSmall; }
  void lay(void* p, int world) {
    char* c = static_cast<char*>(p);
    hdr = reinterpret_cast<FabricHdr*>(c);
    seg_bytes = reinterpret_cast<uint64_t*>(c + sizeof(FabricHdr));
    offs = seg_bytes + world;
    small = reinterpret_cast<char*>(offs + size_t(world) * world);
    bytes = size_for(world);
  }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace
}  // namespace bg

struct bg_comm {
  bg_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  int backend = 0;  // 0 rccl, 1 shm
  bool shortcut = false;
  size_t max_piece = (size_t(1) << 30) - 4096;
  // rccl
  ncclComm_t nccl = nullptr;
  hipStream_t xstream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  bool flying = false;
  void* d_small = nullptr;  // (world + 1) * kSmall: [mine][everybody's]
  void* h_small = nullptr;  // pinned, same layout
  // rccl watchdog: an event behind every operation handed to RCCL; one that has not completed timeout_s after it was queued
  // aborts the communicator (ncclCommAbort: the kernels that wait for a peer leave, the streams drain) and every later call
  // on it fails -- a collective that a peer never enters would otherwise hang the build, and whatever runs it, for ever.
  std::thread watchdog;
  std::mutex wd_mu;
  std::condition_variable wd_cv;
  bool wd_stop = false;
  std::deque<std::pair<hipEvent_t, double>> wd_pending;
  std::vector<hipEvent_t> wd_free;
  std::atomic<bool> aborted{false};
  // shm
  std::string name;
  bg::Fabric fab;
  std::vector<char*> seg;        // mapping of every rank's data segment
  std::vector<size_t> seg_mapped;
  int lock_fd = -1;              // rank 0: the fabric's descriptor, flock'ed for as long as the communicator lives
  double timeout_s = 600.0;
  // statistics
  uint64_t n_collectives = 0, bytes_backend = 0, bytes_local = 0;
  double seconds = 0.0;
};

namespace bg {
namespace {

std::string seg_name(const bg_comm* c, int r) { return "/" + c->name + "-r" + std::to_string(r); }

int shm_map_segment(bg_comm* c, int r, size_t want) {
  if (c->seg_mapped[r] >= want && c->seg[r]) return BG_OK;
  if (c->seg[r]) munmap(c->seg[r], c->seg_mapped[r]);
  c->seg[r] = nullptr;
  c->seg_mapped[r] = 0;
  int fd = shm_open(seg_name(c, r).c_str(), O_RDWR, 0600);
  if (fd < 0) {
    set_error("bg_comm(shm): cannot open the segment of rank %d: %s", r, strerror(errno));
    return BG_EIO;
  }
  struct stat sb;
  if (fstat(fd, &sb) != 0 || size_t(sb.st_size) < want) {
    close(fd);
    set_error("bg_comm(shm): the segment of rank %d holds %lld bytes, %zu were announced", r, (long long)sb.st_size, want);
    return BG_EIO;
  }
  void* p = mmap(nullptr, size_t(sb.st_size), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    set_error("bg_comm(shm): mmap of rank %d's segment (%lld bytes): %s", r, (long long)sb.st_size, strerror(errno));
    return BG_ENOMEM;
  }
  c->seg[r] = static_cast<char*>(p);
  c->seg_mapped[r] = size_t(sb.st_size);
  return BG_OK;
}

// my own segment holds at least `want` bytes (grown in steps: a build asks for a few sizes over and over)
int shm_own_segment(bg_comm* c, size_t want) {
  const int r = c->rank;
  if (c->fab.seg_bytes[r] >= want && c->seg[r]) return BG_OK;
  size_t sz = want + want / 4 + (size_t(1) << 20);
  sz = (sz + 4095) & ~size_t(4095);
  int fd = shm_open(seg_name(c, r).c_str(), O_RDWR | O_CREAT, 0600);
  if (fd < 0) {
    set_error("bg_comm(shm): cannot create a segment: %s", strerror(errno));
    return BG_EIO;
  }
  if (ftruncate(fd, off_t(sz)) != 0) {
    close(fd);
    set_error("bg_comm(shm): cannot size a segment to %zu bytes: %s", sz, strerror(errno));
    return BG_ENOMEM;
  }
  close(fd);
  c->fab.seg_bytes[r] = sz;
  return shm_map_segment(c, r, sz);
}

int shm_barrier(bg_comm* c) {
  FabricHdr* h = c->fab.hdr;
  const uint32_t g = h->gen.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == uint32_t(c->world)) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->gen.store(g + 1, std::memory_order_release);
    return BG_OK;
  }
  const double t0 = now_s();
  unsigned spins = 0;
  while (h->gen.load(std::memory_order_acquire) == g) {
    if (h->failed.load(std::memory_order_relaxed)) {
      set_error("bg_comm(shm): another rank failed");
      return BG_EIO;
    }
    if (++spins > 2000) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((spins & 1023) == 0 && now_s() - t0 > c->timeout_s) {
        h->failed.store(1);
        set_error("bg_comm(shm): rank %d waited %.0f s at a barrier (BG_COMM_TIMEOUT_S)", c->rank, c->timeout_s);
        return BG_EIO;
      }
    } else {
      std::this_thread::yield();
    }
  }
  return BG_OK;
}

// One staged transfer: my `send_total` bytes go into my segment, the piece for rank d starts at send_off[d]; I take
// recv_bytes[s] bytes from rank s's segment (at the offset it published for me), back to back in rank order.
int shm_transfer(bg_comm* c, const void* d_send, size_t send_total, const uint64_t* send_off, const uint64_t* recv_bytes, void* d_recv) {
  bg_ctx* ctx = c->ctx;
  BG_TRY(shm_own_segment(c, send_total ? send_total : 1));
  if (send_total) BG_HIP(hipMemcpyAsync(c->seg[c->rank], d_send, send_total, hipMemcpyDeviceToHost, ctx->stream));
  BG_HIP(hipStreamSynchronize(ctx->stream));
  for (int d = 0; d < c->world; ++d) c->fab.offs[size_t(c->rank) * c->world + d] = send_off[d];
  BG_TRY(shm_barrier(c));
  size_t at = 0;
  for (int s = 0; s < c->world; ++s) {
    const size_t nb = recv_bytes[s];
    if (!nb) continue;
    BG_TRY(shm_map_segment(c, s, c->fab.seg_bytes[s]));
    const uint64_t off = c->fab.offs[size_t(s) * c->world + c->rank];
    BG_HIP(hipMemcpyAsync(static_cast<char*>(d_recv) + at, c->seg[s] + off, nb, hipMemcpyHostToDevice, ctx->stream));
    at += nb;
    if (s == c->rank) c->bytes_local += nb; else c->bytes_backend += nb;
  }
  BG_HIP(hipStreamSynchronize(ctx->stream));
  return shm_barrier(c);  // nobody overwrites its segment while it is being read
}

// ---- the RCCL watchdog (bg_comm::watchdog)
int rccl_alive(bg_comm* c) {
  if (c->aborted.load() || (c->backend == 0 && !c->shortcut && !c->nccl)) {
    set_error("bg_comm(rccl): rank %d: a collective did not complete within %.0f s (BG_COMM_TIMEOUT_S) and the communicator was aborted",
              c->rank, c->timeout_s);
    return BG_EIO;
  }
  return BG_OK;
}
int rccl_handle(bg_comm* c, ncclComm_t* out) {  // the communicator, unless the watchdog has taken it away
  std::lock_guard<std::mutex> lk(c->wd_mu);
  *out = c->nccl;
  if (c->aborted.load() || !c->nccl) return rccl_alive(c);
  return BG_OK;
}
int rccl_note(bg_comm* c, hipStream_t stream) {  // behind an operation just handed to RCCL on `stream`
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lk(c->wd_mu);
    if (!c->wd_free.empty()) { ev = c->wd_free.back(); c->wd_free.pop_back(); }
  }
  if (!ev) BG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  BG_HIP(hipEventRecord(ev, stream));
  std::lock_guard<std::mutex> lk(c->wd_mu);
  c->wd_pending.emplace_back(ev, now_s());
  return BG_OK;
}
void rccl_watch(bg_comm* c) {
  (void)hipSetDevice(c->ctx->device);
  std::unique_lock<std::mutex> lk(c->wd_mu);
  while (!c->wd_stop) {
    c->wd_cv.wait_for(lk, std::chrono::milliseconds(c->timeout_s < 2.0 ? 20 : 500));
    while (!c->wd_pending.empty() && hipEventQuery(c->wd_pending.front().first) == hipSuccess) {
      c->wd_free.push_back(c->wd_pending.front().first);
      c->wd_pending.pop_front();
    }
    (void)hipGetLastError();  // (hipErrorNotReady of the query)
    if (!c->wd_pending.empty() && !c->aborted.load() && now_s() - c->wd_pending.front().second > c->timeout_s) {
      c->aborted.store(true);
      ncclComm_t comm = c->nccl;
      c->nccl = nullptr;
      lk.unlock();
      fprintf(stderr, "bg_comm(rccl): rank %d: a collective has been waiting for %.0f s: aborting the communicator\n", c->rank, c->timeout_s);
      if (comm) (void)g_rccl.CommAbort(comm);
      lk.lock();
    }
  }
}

// Grouped send / recv over RCCL: piece (me -> d) = [send_off[d], +send_bytes[d]), piece (s -> me) lands at recv_off[s];
// every piece in chunks of at most max_piece, chunk t of every pair in the t-th group.
int rccl_transfer(bg_comm* c, const void* d_send, const uint64_t* send_off, const uint64_t* send_bytes, void* d_recv,
                  const uint64_t* recv_off, const uint64_t* recv_bytes, hipStream_t stream) {
  const int P = c->world;
  size_t rounds = 0;
  for (int p = 0; p < P; ++p) {
    rounds = std::max(rounds, size_t((send_bytes[p] + c->max_piece - 1) / c->max_piece));
    rounds = std::max(rounds, size_t((recv_bytes[p] + c->max_piece - 1) / c->max_piece));
  }
  const bool self_through_backend = !c->shortcut && P == 1;  // BG_COMM_NO_SHORTCUT at world 1
  for (size_t t = 0; t < rounds; ++t) {
    const uint64_t lo = uint64_t(t) * c->max_piece;
    ncclComm_t comm = nullptr;
    BG_TRY(rccl_handle(c, &comm));
    BG_NCCL(g_rccl.GroupStart());
    for (int p = 0; p < P; ++p) {
      if (p == c->rank && !self_through_backend) continue;
      if (send_bytes[p] > lo) {
        const size_t nb = size_t(std::min<uint64_t>(send_bytes[p] - lo, c->max_piece));
        BG_NCCL(g_rccl.Send(static_cast<const char*>(d_send) + send_off[p] + lo, nb, ncclUint8, p, comm, stream));
        c->bytes_backend += nb;
      }
      if (recv_bytes[p] > lo) {
        const size_t nb = size_t(std::min<uint64_t>(recv_bytes[p] - lo, c->max_piece));
        BG_NCCL(g_rccl.Recv(static_cast<char*>(d_recv) + recv_off[p] + lo, nb, ncclUint8, p, comm, stream));
      }
    }
    BG_NCCL(g_rccl.GroupEnd());
    BG_TRY(rccl_note(c, stream));
  }
  if (!self_through_backend && send_bytes[c->rank]) {  // the piece a rank keeps is a device copy
    BG_HIP(hipMemcpyAsync(static_cast<char*>(d_recv) + recv_off[c->rank], static_cast<const char*>(d_send) + send_off[c->rank],
                          send_bytes[c->rank], hipMemcpyDeviceToDevice, stream));
    c->bytes_local += send_bytes[c->rank];
  }
  return BG_OK;
}

}  // namespace

// A rank that leaves the build on an error: what it has in flight lands first (the buffers go back to the arena behind it),
// and the ranks of a shared-memory communicator are told (they leave their next wait at once instead of at the timeout).
void comm_fail(bg_comm* c) {
  if (!c) return;
  if (c->flying) { (void)hipStreamSynchronize(c->xstream); c->flying = false; }
  if (c->ctx) (void)hipStreamSynchronize(c->ctx->stream);
  if (c->backend == 1 && c->fab.hdr) c->fab.hdr->failed.store(1);
}
bool comm_is_shortcut(const bg_comm* c) { return c->shortcut; }
int comm_rank(const bg_comm* c) { return c->rank; }
int comm_world(const bg_comm* c) { return c->world; }

int comm_exchange_wait(bg_comm* c) {
  if (!c->flying) return BG_OK;
  c->flying = false;
  BG_HIP(hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));
  return BG_OK;
}

int comm_exchange(bg_comm* c, const void* d_send, const uint64_t* send_rows, const uint64_t* recv_rows, size_t row_bytes, void* d_recv,
                  bool async) {
  const int P = c->world;
  const double t0 = now_s();
  ++c->n_collectives;
  std::vector<uint64_t> soff(P), sby(P), roff(P), rby(P);
  uint64_t s_at = 0, r_at = 0;
  for (int p = 0; p < P; ++p) {
    soff[p] = s_at; sby[p] = send_rows[p] * row_bytes; s_at += sby[p];
    roff[p] = r_at; rby[p] = recv_rows[p] * row_bytes; r_at += rby[p];
  }
  if (c->shortcut) {
    if (sby[0] != rby[0]) { set_error("bg_comm: a rank alone receives what it sends"); return BG_EINVAL; }
    if (sby[0] && d_send != d_recv) BG_HIP(hipMemcpyAsync(d_recv, d_send, sby[0], hipMemcpyDeviceToDevice, c->ctx->stream));
    c->bytes_local += sby[0];
    return BG_OK;
  }
  int rc;
  if (c->backend == 1) {
    rc = shm_transfer(c, d_send, size_t(s_at), soff.data(), rby.data(), d_recv);
  } else {
    BG_TRY(comm_exchange_wait(c));  // one exchange in flight at a time
    hipStream_t st = c->ctx->stream;
    if (async) {
      BG_HIP(hipEventRecord(c->ev_ready, c->ctx->stream));
      BG_HIP(hipStreamWaitEvent(c->xstream, c->ev_ready, 0));
      st = c->xstream;
    }
    rc = rccl_transfer(c, d_send, soff.data(), sby.data(), d_recv, roff.data(), rby.data(), st);
    if (rc == BG_OK && async) {
      BG_HIP(hipEventRecord(c->ev_done, c->xstream));
      c->flying = true;
    }
  }
  c->seconds += now_s() - t0;
  return rc;
}

int comm_all_gather_dev(bg_comm* c, const void* d_send, const uint64_t* bytes, void* d_recv) {
  const int P = c->world;
  const double t0 = now_s();
  ++c->n_collectives;
  std::vector<uint64_t> soff(P, 0), sby(P, bytes[c->rank]), roff(P), rby(P);
  uint64_t at = 0;
  for (int p = 0; p < P; ++p) { roff[p] = at; rby[p] = bytes[p]; at += bytes[p]; }
  int rc = BG_OK;
  if (c->shortcut) {
    if (bytes[0] && d_send != d_recv) BG_HIP(hipMemcpyAsync(d_recv, d_send, bytes[0], hipMemcpyDeviceToDevice, c->ctx->stream));
    c->bytes_local += bytes[0];
  } else if (c->backend == 1) {
    rc = shm_transfer(c, d_send, size_t(bytes[c->rank]), soff.data(), rby.data(), d_recv);
  } else {
    BG_TRY(comm_exchange_wait(c));
    rc = rccl_transfer(c, d_send, soff.data(), sby.data(), d_recv, roff.data(), rby.data(), c->ctx->stream);
  }
  c->seconds += now_s() - t0;
  return rc;
}

int comm_all_gather_host(bg_comm* c, const void* mine, size_t bytes, void* all) {
  const int P = c->world;
  if (c->shortcut) {
    memcpy(all, mine, bytes);
    return BG_OK;
  }
  if (!bytes) return BG_OK;
  const double t0 = now_s();
  ++c->n_collectives;
  for (size_t lo = 0; lo < bytes; lo += kSmall) {
    const size_t nb = std::min(kSmall, bytes - lo);
    if (c->backend == 1) {
      memcpy(c->fab.small + size_t(c->rank) * kSmall, static_cast<const char*>(mine) + lo, nb);
      BG_TRY(shm_barrier(c));
      for (int r = 0; r < P; ++r) memcpy(static_cast<char*>(all) + size_t(r) * bytes + lo, c->fab.small + size_t(r) * kSmall, nb);
      BG_TRY(shm_barrier(c));
    } else if (nb) {
      BG_TRY(comm_exchange_wait(c));
      const size_t padded = (nb + 15) & ~size_t(15);
      char* h = static_cast<char*>(c->h_small);
      char* d = static_cast<char*>(c->d_small);
      memcpy(h, static_cast<const char*>(mine) + lo, nb);
      BG_HIP(hipMemcpyAsync(d, h, padded, hipMemcpyHostToDevice, c->ctx->stream));
      ncclComm_t comm = nullptr;
      BG_TRY(rccl_handle(c, &comm));
      BG_NCCL(g_rccl.AllGather(d, d + kSmall, padded, ncclUint8, comm, c->ctx->stream));
      BG_TRY(rccl_note(c, c->ctx->stream));
      BG_HIP(hipMemcpyAsync(h + kSmall, d + kSmall, padded * P, hipMemcpyDeviceToHost, c->ctx->stream));
      BG_HIP(hipStreamSynchronize(c->ctx->stream));
      for (int r = 0; r < P; ++r) memcpy(static_cast<char*>(all) + size_t(r) * bytes + lo, h + kSmall + size_t(r) * padded, nb);
      c->bytes_backend += padded * size_t(P > 1 ? P - 1 : 1);
    }
  }
  c->seconds += now_s() - t0;
  return BG_OK;
}

int comm_barrier(bg_comm* c) {
  if (c->shortcut) return BG_OK;
  if (c->backend == 1) return shm_barrier(c);
  uint64_t one = 1;
  std::vector<uint64_t> all(c->world);
  return comm_all_gather_host(c, &one, sizeof one, all.data());
}

int comm_all_reduce_sum_u64(bg_comm* c, uint64_t* d_buf, size_t n) {
  if (c->shortcut || !n) return BG_OK;
  const double t0 = now_s();
  ++c->n_collectives;
  int rc = BG_OK;
  if (c->backend == 1) {
    const int P = c->world;
    const size_t nb = n * sizeof(uint64_t);
    BG_TRY(shm_own_segment(c, nb));
    BG_HIP(hipMemcpyAsync(c->seg[c->rank], d_buf, nb, hipMemcpyDeviceToHost, c->ctx->stream));
    BG_HIP(hipStreamSynchronize(c->ctx->stream));
    BG_TRY(shm_barrier(c));
    std::vector<uint64_t> sum(n, 0);
    for (int s = 0; s < P; ++s) {
      BG_TRY(shm_map_segment(c, s, c->fab.seg_bytes[s]));
      const uint64_t* v = reinterpret_cast<const uint64_t*>(c->seg[s]);
      for (size_t i = 0; i < n; ++i) sum[i] += v[i];
      if (s != c->rank) c->bytes_backend += nb;
    }
    BG_HIP(hipMemcpyAsync(d_buf, sum.data(), nb, hipMemcpyHostToDevice, c->ctx->stream));
    BG_HIP(hipStreamSynchronize(c->ctx->stream));
    rc = shm_barrier(c);
  } else {
    BG_TRY(comm_exchange_wait(c));
    ncclComm_t comm = nullptr;
    BG_TRY(rccl_handle(c, &comm));
    BG_NCCL(g_rccl.AllReduce(d_buf, d_buf, n, ncclUint64, ncclSum, comm, c->ctx->stream));
    BG_TRY(rccl_note(c, c->ctx->stream));
    c->bytes_backend += n * sizeof(uint64_t);
  }
  c->seconds += now_s() - t0;
  return rc;
}

}  // namespace bg

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

int bg_comm_unique_id(char* id_out) {
  if (!id_out) return BG_EINVAL;
  BG_TRY(bg::rccl_load());
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == BG_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  BG_NCCL(bg::g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return BG_OK;
}

void bg_comm_free(bg_comm* c) {
  if (!c) return;
  if (c->ctx) (void)hipSetDevice(c->ctx->device);
  if (c->watchdog.joinable()) {
    { std::lock_guard<std::mutex> lk(c->wd_mu); c->wd_stop = true; }
    c->wd_cv.notify_all();
    c->watchdog.join();
  }
  if (c->flying) (void)hipStreamSynchronize(c->xstream);
  if (c->nccl) (void)bg::g_rccl.CommDestroy(c->nccl);
  for (auto& pe : c->wd_pending) (void)hipEventDestroy(pe.first);
  for (hipEvent_t e : c->wd_free) (void)hipEventDestroy(e);
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->d_small) c->ctx->release(c->d_small);
  if (c->h_small) (void)hipHostFree(c->h_small);
  for (size_t r = 0; r < c->seg.size(); ++r)
    if (c->seg[r]) munmap(c->seg[r], c->seg_mapped[r]);
  if (c->backend == 1 && !c->name.empty()) {
    shm_unlink(bg::seg_name(c, c->rank).c_str());
    if (c->fab.hdr) munmap(c->fab.hdr, c->fab.bytes);
    if (c->rank == 0) shm_unlink(("/" + c->name).c_str());
    if (c->lock_fd >= 0) close(c->lock_fd);  // (drops the lock)
  }
  delete c;
}

int bg_comm_init(bg_ctx* ctx, int rank, int world, const char* id, const char* backend, bg_comm** out) {
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world || world > 127) {
    bg::set_error("bg_comm_init: bad arguments (rank %d of %d; at most 127 ranks: a row names its destination in 7 bits)", rank, world);
    return BG_EINVAL;
  }
  const std::string be = backend && *backend ? backend : "rccl";
  if (be != "rccl" && be != "shm") {
    bg::set_error("bg_comm_init: unknown backend '%s' (rccl, shm)", be.c_str());
    return BG_EINVAL;
  }
  std::unique_ptr<bg_comm, void (*)(bg_comm*)> c(new bg_comm, bg_comm_free);
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->backend = be == "shm" ? 1 : 0;
  const char* ns = getenv("BG_COMM_NO_SHORTCUT");
  c->shortcut = world == 1 && !(ns && *ns && *ns != '0');
  if (const char* mp = getenv("BG_COMM_MAX_PIECE")) c->max_piece = std::max<size_t>(16, size_t(strtoull(mp, nullptr, 10)));
  if (const char* to = getenv("BG_COMM_TIMEOUT_S")) c->timeout_s = atof(to);
  BG_HIP(hipSetDevice(ctx->device));
  if (c->shortcut) {
    *out = c.release();
    return BG_OK;
  }
  if (c->backend == 0) {
    if (!id) { bg::set_error("bg_comm_init(rccl): the unique id of bg_comm_unique_id is needed"); return BG_EINVAL; }
    BG_TRY(bg::rccl_load());
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    BG_NCCL(bg::g_rccl.CommInitRank(&c->nccl, world, uid, rank));
    BG_HIP(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    BG_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    BG_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    BG_TRY(ctx->alloc(size_t(world + 1) * bg::kSmall, &c->d_small));
    BG_HIP(hipHostMalloc(&c->h_small, size_t(world + 1) * bg::kSmall, hipHostMallocDefault));
    if (!getenv("BG_COMM_TIMEOUT_S")) c->timeout_s = 180.0;  // (a rank's collectives wait for its peers' compute: seconds; never minutes)
    bg_comm* raw = c.get();
    c->watchdog = std::thread([raw] { bg::rccl_watch(raw); });
  } else {
    if (!id || !*id || strlen(id) > 100 || strchr(id, '/')) { bg::set_error("bg_comm_init(shm): the id names the shared segment (a short string without '/')"); return BG_EINVAL; }
    c->name = std::string("bgcomm-") + id;
    c->seg.assign(world, nullptr);
    c->seg_mapped.assign(world, 0);
    const size_t fb = bg::Fabric::size_for(world);
    const std::string fname = "/" + c->name;
    int fd = -1;
    // A segment a crashed run left under this name must not be taken for the new one (it is complete: the right size,
    // `ready` set): rank 0 holds an exclusive flock on the fabric for as long as its communicator lives, makes it under a
    // name of its own and only then gives it the shared name, so the shared name never shows an unlocked live fabric —
    // and the other ranks skip whatever they can lock (nobody holds it: a leftover) until rank 0's appears.
    if (rank == 0) {
      shm_unlink(fname.c_str());
      const std::string tmp = fname + "-new" + std::to_string(long(getpid()));
      shm_unlink(tmp.c_str());
      fd = shm_open(tmp.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
      if (fd < 0 || flock(fd, LOCK_EX | LOCK_NB) != 0 || ftruncate(fd, off_t(fb)) != 0 ||
          rename(("/dev/shm" + tmp).c_str(), ("/dev/shm" + fname).c_str()) != 0) {
        bg::set_error("bg_comm_init(shm): cannot create %s: %s", fname.c_str(), strerror(errno));
        if (fd >= 0) close(fd);
        shm_unlink(tmp.c_str());
        return BG_EIO;
      }
      c->lock_fd = dup(fd);
    } else {
      const double t0 = bg::now_s();
      for (;;) {
        fd = shm_open(fname.c_str(), O_RDWR, 0600);
        struct stat sb;
        if (fd >= 0 && flock(fd, LOCK_SH | LOCK_NB) == 0) {  // nobody holds it: left behind by a run that died
          flock(fd, LOCK_UN);
          close(fd);
          fd = -1;
        }
        if (fd >= 0 && fstat(fd, &sb) == 0 && size_t(sb.st_size) >= fb) break;
        if (fd >= 0) close(fd);
        if (bg::now_s() - t0 > c->timeout_s) { bg::set_error("bg_comm_init(shm): rank 0 never created %s", fname.c_str()); return BG_EIO; }
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    }
    void* p = mmap(nullptr, fb, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { bg::set_error("bg_comm_init(shm): mmap: %s", strerror(errno)); return BG_ENOMEM; }
    c->fab.lay(p, world);
    if (rank == 0) {
      c->fab.hdr->world = uint32_t(world);
      c->fab.hdr->arrived.store(0);
      c->fab.hdr->gen.store(0);
      c->fab.hdr->failed.store(0);
      for (int r = 0; r < world; ++r) c->fab.seg_bytes[r] = 0;
      c->fab.hdr->ready.store(0x42474331u, std::memory_order_release);
    } else {
      const double t0 = bg::now_s();
      while (c->fab.hdr->ready.load(std::memory_order_acquire) != 0x42474331u) {
        if (bg::now_s() - t0 > c->timeout_s) { bg::set_error("bg_comm_init(shm): rank 0 never finished %s", fname.c_str()); return BG_EIO; }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      if (c->fab.hdr->world != uint32_t(world)) { bg::set_error("bg_comm_init(shm): %s was made for %u ranks", fname.c_str(), c->fab.hdr->world); return BG_EINVAL; }
    }
    shm_unlink(bg::seg_name(c.get(), rank).c_str());
    BG_TRY(bg::shm_own_segment(c.get(), size_t(1) << 20));
    BG_TRY(bg::shm_barrier(c.get()));
  }
  *out = c.release();
  return BG_OK;
}

int bg_comm_rank(const bg_comm* c) { return c ? c->rank : -1; }
int bg_comm_world(const bg_comm* c) { return c ? c->world : 0; }

int bg_comm_get_stats(const bg_comm* c, bg_comm_stats* out) {
  if (!c || !out) return BG_EINVAL;
  out->collectives = c->n_collectives;
  out->bytes_through_backend = c->bytes_backend;
  out->bytes_kept_local = c->bytes_local;
  out->seconds_in_calls = c->seconds;
  out->backend = c->shortcut ? 2 : c->backend;
  out->rccl_version = 0;
  if (c->backend == 0 && bg::g_rccl.GetVersion) { int v = 0; if (bg::g_rccl.GetVersion(&v) == ncclSuccess) out->rccl_version = v; }
  return BG_OK;
}

int bg_comm_all_to_all_v(bg_comm* c, const void* d_send, const uint64_t* h_send_rows, uint64_t row_bytes, void* d_recv, uint64_t recv_cap_rows,
                         uint64_t* h_recv_rows) {
  if (!c || !h_send_rows || !h_recv_rows || !row_bytes) return BG_EINVAL;
  const int P = c->world;
  std::vector<uint64_t> cm(size_t(P) * P);
  BG_TRY(bg::comm_all_gather_host(c, h_send_rows, size_t(P) * 8, cm.data()));
  uint64_t tot = 0;
  for (int s = 0; s < P; ++s) { h_recv_rows[s] = cm[size_t(s) * P + c->rank]; tot += h_recv_rows[s]; }
  if (tot > recv_cap_rows) { bg::set_error("bg_comm_all_to_all_v: %llu rows arrive, room for %llu", (unsigned long long)tot, (unsigned long long)recv_cap_rows); return BG_ENOMEM; }
  return bg::comm_exchange(c, d_send, h_send_rows, h_recv_rows, size_t(row_bytes), d_recv);
}

int bg_comm_exchange(bg_comm* c, const void* d_send, const uint64_t* h_send_rows, const uint64_t* h_recv_rows, uint64_t row_bytes, void* d_recv) {
  if (!c || !h_send_rows || !h_recv_rows || !row_bytes) return BG_EINVAL;
  return bg::comm_exchange(c, d_send, h_send_rows, h_recv_rows, size_t(row_bytes), d_recv);
}

int bg_comm_all_gather_host(bg_comm* c, const void* h_mine, uint64_t bytes, void* h_all) {
  if (!c) return BG_EINVAL;
  return bg::comm_all_gather_host(c, h_mine, size_t(bytes), h_all);
}

int bg_comm_all_gather(bg_comm* c, const void* d_send, const uint64_t* h_bytes, void* d_recv) {
  if (!c || !h_bytes) return BG_EINVAL;
  return bg::comm_all_gather_dev(c, d_send, h_bytes, d_recv);
}

int bg_comm_all_reduce_sum_u64(bg_comm* c, uint64_t* d_buf, uint64_t n) {
  if (!c) return BG_EINVAL;
  return bg::comm_all_reduce_sum_u64(c, d_buf, size_t(n));
}

int bg_comm_barrier(bg_comm* c) { return c ? bg::comm_barrier(c) : BG_EINVAL; }

}  // extern "C"
