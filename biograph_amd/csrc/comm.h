// comm.h — the communicator behind the N-rank build (csrc/comm.hip): one rank per process and GPU.
// Backends: RCCL over xGMI (rccl.h through dlopen: a process that never shards never loads it) and "shm" — host-staged
// exchanges through POSIX shared memory, for several ranks on ONE GPU (tests; RCCL refuses two ranks on a device).
#pragma once
#include <vector>

#include "bg_internal.h"

struct bg_comm;

namespace bg {

// Rows of row_bytes bytes travel as send_rows[d] rows to rank d (pieces back to back in d_send, in rank order) and arrive as
// recv_rows[s] rows from rank s (back to back in d_recv, in rank order).  Both sides know their counts.  Queued on the
// context's stream; `async` queues it on the communicator's side stream instead (behind what the context's stream holds
// now) and comm_exchange_wait makes the context's stream wait for it.
int comm_exchange(bg_comm* c, const void* d_send, const uint64_t* send_rows, const uint64_t* recv_rows, size_t row_bytes, void* d_recv,
                  bool async = false);
int comm_exchange_wait(bg_comm* c);
// Every rank's `bytes` host bytes on every rank, in rank order (counts, boundary rows: small things only).
int comm_all_gather_host(bg_comm* c, const void* mine, size_t bytes, void* all);
// Every rank's device buffer (bytes[r] bytes from rank r; every rank passes the same table) back to back on every rank.
int comm_all_gather_dev(bg_comm* c, const void* d_send, const uint64_t* bytes, void* d_recv);
int comm_all_reduce_sum_u64(bg_comm* c, uint64_t* d_buf, size_t n);
int comm_barrier(bg_comm* c);
// a rank leaving the build on an error: lands what it has in flight, tells the ranks of a shared-memory communicator
void comm_fail(bg_comm* c);
// true: a rank alone whose collectives are identities (world 1 without BG_COMM_NO_SHORTCUT=1)
bool comm_is_shortcut(const bg_comm* c);
int comm_rank(const bg_comm* c);
int comm_world(const bg_comm* c);

}  // namespace bg
