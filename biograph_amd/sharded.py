"""Sharded (multi-GPU) seqset build: one process per GPU, buffers exchanged with torch.distributed
(backend "nccl" = RCCL over xGMI).  SURVEY.md §8e / DESIGN.md §7.

Reads are sharded arbitrarily across ranks.  Exchange steps (everything else is rank-local HIP work
behind the C ABI):

  1. k-mer instances   partition by hash(kmer) -> all_to_all  -> per-rank exact counts + filter
  2. kept k-mers       all_gather                              -> replicated hash set
  3. corrected reads   all_gather (packed 2-bit)               -> replicated cover index
  4. suffix records    partition by leading bases (balanced splitters from an all-reduced
                       histogram) -> all_to_all                -> per-rank sort + prefix-dedup
  5. boundary entries  all_gather of first/last entries        -> cross-shard prefix drop, shared[0]
  6. tails s[1:]       partition by owner of lower_bound -> all_to_all -> prev_{ACGT} bits
  7. slices            gather -> concatenation (shards are lexicographic ranges) + finalize

`Comm` abstracts "the ranks this process drives": `TorchDistComm` (one rank, real collectives) and
`LocalSimComm` (all ranks in one process on one GPU — used by the 1-GPU parity tests, where RCCL
cannot run two ranks on one device).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi
from .capi import BuildStats

PREFIX_BASES = 8  # splitter resolution: 4^8 bins of leading bases
# the splitter histogram counts every PREFIX_SAMPLE-th record of every rank (balance needs no more; the same step
# on all ranks, so that the all-reduced bins weigh every rank alike)
PREFIX_SAMPLE = max(1, int(os.environ.get("BG_PREFIX_SAMPLE", "8")))


# ------------------------------------------------------------------------------------------------
# communicators

class LocalSimComm:
    """All `world` ranks live in this process; collectives are plain tensor shuffles."""

    def __init__(self, world):
        self.world = world
        self.local_ranks = list(range(world))
        self.off_rank_bytes = [0] * world  # what each rank would put on its links (all_to_all pieces not kept + gathers)

    def all_to_all_v(self, send):
        """send[r] = (rows [n, width], counts list[world]) -> (recv, cm): recv[r] = rows from all sources in rank
        order, cm[s][d] = rows source s sent to destination d."""
        pieces = {r: [] for r in self.local_ranks}
        cm = [[0] * self.world for _ in range(self.world)]
        for src in self.local_ranks:
            rows, counts = send[src]
            off = 0
            row_bytes = rows.element_size() * (rows.numel() // max(rows.shape[0], 1)) if rows.shape[0] else 0
            for dst in range(self.world):
                pieces[dst].append(rows[off:off + counts[dst]])
                cm[src][dst] = int(counts[dst])
                if dst != src:
                    self.off_rank_bytes[src] += int(counts[dst]) * row_bytes
                off += counts[dst]
        return {r: torch.cat(pieces[r], dim=0) for r in self.local_ranks}, cm

    def all_to_all_v_start(self, send):
        """all_to_all_v in two halves, so that the caller can compute while the rows travel: -> a handle for
        all_to_all_v_finish.  (Simulated ranks share one device: nothing to overlap, the exchange happens here.)"""
        return self.all_to_all_v(send)

    def all_to_all_v_finish(self, handle):
        return handle

    def all_to_all_reply(self, send, cm):
        """The way back of an exchange whose counts matrix was cm: send[r] = one row per row r received (in received
        order); every rank gets one row per row it sent, in sent order."""
        return self.all_to_all_v({r: (send[r], [cm[s][r] for s in range(self.world)]) for r in self.local_ranks})[0]

    def all_to_all_again(self, send, cm):
        """One more exchange in the direction and with the counts of the one that gave cm (send[r] = one row per
        row sent then, in the same order)."""
        return self.all_to_all_v({r: (send[r], cm[r]) for r in self.local_ranks})[0]

    def gather_ints(self, vals):
        """vals[r] = list of ints of local rank r -> [list of rank 0, list of rank 1, ...] on every rank."""
        return [list(vals[r]) for r in range(self.world)]

    def all_gather_v(self, send):
        """send[r] = tensor -> every rank gets the list [tensor of rank 0, ...]."""
        for r in range(self.world):
            self.off_rank_bytes[r] += send[r].numel() * send[r].element_size() * (self.world - 1)
        lst = [send[r] for r in range(self.world)]
        return {r: lst for r in self.local_ranks}

    def all_gather_obj(self, send):
        lst = [send[r] for r in range(self.world)]
        return {r: lst for r in self.local_ranks}

    def all_reduce_sum(self, send):
        tot = None
        for r in self.local_ranks:
            tot = send[r].clone() if tot is None else tot + send[r]
        return {r: tot for r in self.local_ranks}

    def barrier(self):
        pass


class TorchDistComm:
    """One rank per process over an initialised torch.distributed process group."""

    def __init__(self, device, stage_through_host=False):
        """stage_through_host: run every collective on CPU copies (gloo backend; lets several ranks share
        one GPU in tests — RCCL refuses two ranks on one device)."""
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.local_ranks = [self.rank]
        self.out_device = device
        self.device = torch.device("cpu") if stage_through_host else device
        if os.environ.get("BG_COMM_MAX_PIECE"):  # tests: force the multi-round exchange
            self.MAX_PIECE_BYTES = int(os.environ["BG_COMM_MAX_PIECE"])

    def _in(self, t):
        return t.to(self.device) if t.device != self.device else t

    def _out(self, t):
        return t.to(self.out_device) if t.device != self.out_device else t

    # RCCL 2.26 (ROCm 7.0 build shipped with torch 2.10) returned corrupt data for all_to_all pieces of
    # exactly 2^31 / 2^32 / 2^33 bytes on MI355X (tools/debug_nccl_big.py); every piece is therefore
    # kept <= 1 GiB - 4 KiB and large exchanges run as several rounds.
    MAX_PIECE_BYTES = (1 << 30) - 4096

    def gather_ints(self, vals):
        """vals[rank] = list of ints -> the lists of all ranks (one small tensor collective, no pickling)."""
        dist = self.dist
        if self.world == 1:
            return [[int(v) for v in vals[self.rank]]]
        mine = torch.tensor([int(v) for v in vals[self.rank]], dtype=torch.int64, device=self.device)
        out = torch.empty((self.world, mine.numel()), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(out, mine.view(1, -1)) if mine.numel() else None
        return [[int(x) for x in row] for row in out.tolist()] if mine.numel() else [[] for _ in range(self.world)]

    def all_to_all_v(self, send):
        """-> (recv, cm) as LocalSimComm.all_to_all_v; cm comes from ONE all_gather of the counts."""
        rows, counts = send[self.rank]
        cm = self.gather_ints({self.rank: [int(c) for c in counts]})
        return {self.rank: self._exchange(rows, [int(c) for c in counts], cm)}, cm

    def all_to_all_v_start(self, send):
        """all_to_all_v in two halves: the rows leave on the backend's own stream (RCCL: async_op) and the caller goes on
        launching kernels on its stream until all_to_all_v_finish waits for them.  A rank alone and exchanges of several
        rounds complete here."""
        rows, counts = send[self.rank]
        counts = [int(c) for c in counts]
        cm = self.gather_ints({self.rank: counts})
        dist = self.dist
        rows_in = self._in(rows).contiguous()
        row_bytes = rows_in.element_size()
        for d in rows_in.shape[1:]:
            row_bytes *= int(d)
        max_rows = max(1, self.MAX_PIECE_BYTES // max(row_bytes, 1))
        single = max(max(r) for r in cm) <= max_rows
        if self.world == 1 or not single:
            return ("done", {self.rank: self._exchange(rows, counts, cm)}, cm)
        out_counts = [cm[s][self.rank] for s in range(self.world)]
        out = torch.empty((sum(out_counts),) + tuple(rows_in.shape[1:]), dtype=rows_in.dtype, device=self.device)
        work = dist.all_to_all_single(out, rows_in, output_split_sizes=out_counts, input_split_sizes=counts, async_op=True)
        return ("flying", work, out, rows_in, cm)  # rows_in stays alive until the wait

    def all_to_all_v_finish(self, handle):
        if handle[0] == "done":
            return handle[1], handle[2]
        _, work, out, _rows, cm = handle
        work.wait()  # the caller's stream now waits for the exchange
        return {self.rank: self._out(out)}, cm

    def all_to_all_reply(self, send, cm):
        counts = [cm[s][self.rank] for s in range(self.world)]
        cmt = [[cm[d][s] for d in range(self.world)] for s in range(self.world)]  # the reply's own matrix
        return {self.rank: self._exchange(send[self.rank], counts, cmt)}

    def all_to_all_again(self, send, cm):
        return {self.rank: self._exchange(send[self.rank], [int(c) for c in cm[self.rank]], cm)}

    def _exchange(self, rows, counts, cm):
        dist = self.dist
        if self.world == 1:  # everything stays: the rows are where they belong already (callers give `send` up)
            return self._out(self._in(rows).contiguous())
        rows = self._in(rows).contiguous()
        out_counts = [cm[s][self.rank] for s in range(self.world)]
        row_bytes = rows.element_size()
        for d in rows.shape[1:]:
            row_bytes *= int(d)
        max_rows = max(1, self.MAX_PIECE_BYTES // max(row_bytes, 1))
        rounds = max(1, -(-max(max(r) for r in cm) // max_rows))
        starts = [0]
        for c in counts:
            starts.append(starts[-1] + c)
        if rounds == 1:  # the one collective every backend implements natively (the piece a rank keeps is a local copy in it)
            out = torch.empty((sum(out_counts),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=self.device)
            dist.all_to_all_single(out, rows, output_split_sizes=out_counts, input_split_sizes=counts)
            return self._out(out)
        # Several rounds.  The result keeps the single-round layout (all rows of source 0, then source 1, ...,
        # each in the order they were sent): the presence bits of the k-mer exchange travel back by position.
        ostarts = [0]
        for c in out_counts:
            ostarts.append(ostarts[-1] + c)
        out = torch.empty((ostarts[-1],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=self.device)
        for t in range(rounds):
            lo = t * max_rows
            ins = [rows[starts[d] + min(lo, counts[d]): starts[d] + min(lo + max_rows, counts[d])]
                   for d in range(self.world)]
            rcv = [out[ostarts[d] + min(lo, out_counts[d]): ostarts[d] + min(lo + max_rows, out_counts[d])]
                   for d in range(self.world)]
            if dist.get_backend() == "nccl":
                # RCCL: grouped send/recv on views - every piece leaves from where it lies and lands where it
                # belongs, no staging copies of a multi-GB exchange
                dist.all_to_all(rcv, ins)
            else:  # gloo has only the single-buffer form
                tmp = torch.empty((sum(x.shape[0] for x in rcv),) + tuple(rows.shape[1:]), dtype=rows.dtype,
                                  device=self.device)
                dist.all_to_all_single(tmp, torch.cat(ins, dim=0), output_split_sizes=[x.shape[0] for x in rcv],
                                       input_split_sizes=[x.shape[0] for x in ins])
                off = 0
                for x in rcv:
                    x.copy_(tmp[off: off + x.shape[0]])
                    off += x.shape[0]
        return self._out(out)

    def all_gather_v(self, send):
        dist = self.dist
        if self.world == 1:
            return {self.rank: [send[self.rank]]}
        t = self._in(send[self.rank]).contiguous()
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=self.device)
        sizes = [torch.empty_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes + [1])
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
        pad[:t.shape[0]] = t
        outs = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(outs, pad)
        return {self.rank: [self._out(outs[i][:sizes[i]]) for i in range(self.world)]}

    def all_gather_obj(self, send):
        if self.world == 1:
            return {self.rank: [send[self.rank]]}
        lst = [None] * self.world
        self.dist.all_gather_object(lst, send[self.rank])
        return {self.rank: lst}

    def all_reduce_sum(self, send):
        if self.world == 1:
            return {self.rank: send[self.rank]}
        t = self._in(send[self.rank]).clone()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return {self.rank: self._out(t)}

    def barrier(self):
        self.dist.barrier()


# ------------------------------------------------------------------------------------------------
# thin wrappers over the sharded C ABI working on torch tensors (int64 tensors carry u64 bits)

def _ptr(t):
    return C.c_void_p(t.data_ptr() if t.numel() else 0)


def _partition(ctx, rows, mode, nparts, params):
    """rows: int64 [n, width] -> (partitioned rows, counts list)."""
    n, width = rows.shape
    out = torch.empty_like(rows)
    counts = (C.c_uint64 * nparts)()
    p = None
    if params is not None:
        params = np.ascontiguousarray(params, np.uint64)
        p = capi.np_ptr(params)
    capi.check(capi.lib().bg_partition_rows(ctx.h, _ptr(rows), n, width, mode, nparts, p, _ptr(out), counts))
    counts = [int(c) for c in counts]
    return out[:sum(counts)], counts


def choose_splitters(hist, nparts):
    """Balanced contiguous bin ranges from the global leading-bases histogram: splitter j = first bin
    at which the running count reaches j/nparts of the total (ascending, may repeat)."""
    h = np.asarray(hist, np.float64)
    cum = np.cumsum(h)
    total = cum[-1] if len(cum) else 0.0
    sp = []
    for j in range(1, nparts):
        target = total * j / nparts
        b = int(np.searchsorted(cum, target, side="left")) + 1 if total > 0 else len(h)
        sp.append(min(max(b, sp[-1] if sp else 0), len(h)))
    return sp


def _next_nonempty(rows, r, step):
    q = r + step
    while 0 <= q < len(rows):
        if rows[q][-1] != 0xFFFFFFFFFFFFFFFF:
            return rows[q]
        q += step
    return None


# ------------------------------------------------------------------------------------------------

class _Phases:
    """BG_SHARD_TIMING=1: wall time per phase of sharded_create on stderr (device synchronised), with the peak of
    torch's allocator and of the library's (context `ctx`) inside the phase, in GB."""

    def __init__(self, ctx=None):
        import os
        self.on = os.environ.get("BG_SHARD_TIMING") == "1"
        self.t = None
        self.rows = []
        self.ctx = ctx

    def _mem(self):
        try:  # (a pluggable allocator keeps no statistics: the library's figure is then the whole footprint)
            tp = torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0
            torch.cuda.reset_peak_memory_stats()
        except Exception:
            tp = 0.0
        lp = self.ctx.memory()["peak"] / 1e9 if self.ctx is not None else 0.0
        if self.ctx is not None:
            self.ctx.memory(reset_peak=True)
        return tp, lp

    def mark(self, name):
        if not self.on:
            return
        import time
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        tp, lp = self._mem()
        if self.t is not None:
            self.rows.append((self.name, (now - self.t) * 1e3, tp, lp))
        self.t, self.name = now, name

    def done(self):
        if self.on:
            import sys
            self.mark("end")
            sys.stderr.write("sharded_create phases (ms | torch peak GB | library peak GB): " +
                             ", ".join("%s %.1f|%.1f|%.1f" % r for r in self.rows) + "\n")


def _relieve(ctxs, dev):
    """Two allocators share the GPU (torch for the exchange buffers, the library's arena for everything else) and
    neither can ask the other to give unused memory back.  Each returns its own cache to the driver when one of its
    own requests fails; what is left to do here is to return the library's wholly free segments when the device is
    nearly full, so that torch's next large buffer finds room.  (Emptying torch's cache as a routine cost seconds per
    build: the driver maps memory at ~30 us per MiB, and the next phase asked for all of it again.)"""
    if capi.lib().bg_torch_allocator_enabled():
        return  # one arena for both: nothing is stranded
    for r, c in ctxs.items():
        free, total = torch.cuda.mem_get_info(dev[r])
        if free < 0.08 * total:
            c.trim()


# suffix offsets at which a new round of cover candidates starts (the last round runs to the longest read):
# most reads are covered within a few bases at 30x, so early rounds are narrow and few reads reach the late ones
COVER_ROUNDS = "1,5,13,29,61,256"


def _all_sum(comm, vals):
    """Sum of one python int per local rank over all ranks."""
    return sum(row[0] for row in comm.gather_ints({r: [int(vals[r])] for r in comm.local_ranks}))


def _select_rows(ctx, rows, flags):
    """rows[flags != 0] without torch's nonzero + gather (27 ms of a chr20-sized build): one compaction and one
    coalesced copy in the library."""
    n, width = rows.shape
    cnt = int((flags != 0).sum().item()) if n else 0
    out = torch.empty((cnt, width), dtype=rows.dtype, device=rows.device)
    m = C.c_uint64()
    capi.check(capi.lib().bg_rows_select(ctx.h, _ptr(rows), n, width, _ptr(flags), _ptr(out), cnt, C.byref(m)))
    return out[:m.value]


def _skm_extract(ctx, reads_h, k, P, dev, first_room=0.5):
    """Super-k-mer records of a reads object grouped by owner rank -> (recs [n,2], org [n], counts, windows).
    The buffers are sized for first_room of bg_skm_capacity's 3x-the-expected-density worst case (tried only when an
    input beats that) and only trimmed when less than three quarters are used: they live until the presence bits are
    back.  Reads: 0.385 = 1.15x the expected density (random sequence produces 0.9x), so that the 3.4 GB of a
    chr20-sized rank are not copied to be trimmed; probe strings (short, denser) keep 1.5x."""
    L = capi.lib()
    n = C.c_uint64()
    capi.check(L.bg_kmer_instances(ctx.h, reads_h, k, C.byref(n)))
    cap = C.c_uint64()
    capi.check(L.bg_skm_capacity(ctx.h, reads_h, k, n.value, C.byref(cap)))
    counts = (C.c_uint64 * P)()
    nrec = C.c_uint64()
    if not cap.value:
        return (torch.empty((0, 2), dtype=torch.int64, device=dev), torch.empty((0,), dtype=torch.int64, device=dev),
                [0] * P, n.value)
    for room in (int(cap.value * first_room) + 1024, cap.value):
        room = min(room, cap.value)
        recs = torch.empty((room, 2), dtype=torch.int64, device=dev)
        org = torch.empty((room,), dtype=torch.int64, device=dev)
        rc = L.bg_skm_extract_by_owner(ctx.h, reads_h, k, n.value, P, _ptr(recs), _ptr(org), room, counts, C.byref(nrec))
        if rc == -3 and room < cap.value:  # BG_ENOMEM: more records than 1.5x the expected density
            del recs, org
            continue
        capi.check(rc)
        break
    if nrec.value * 4 < room * 3:  # give the unused tail back (a view would keep the whole buffer alive)
        recs, org = recs[:nrec.value].clone(), org[:nrec.value].clone()
    else:
        recs, org = recs[:nrec.value], org[:nrec.value]
    return recs, org, [int(c) for c in counts], n.value


def _correct_routed(comm, ctxs, reads, owned, skm_recs, skm_org, skm_bits, prm, dev, ph):
    """Read correction with the kept k-mers left on their owner ranks: every rank searches
    its own reads; the k-mers a search needs beyond the read's own windows travel as probe strings to their
    owners and come back as presence bits, one exchange per step, until no read misses a look-up."""
    L = capi.lib()
    P = comm.world
    ranks = comm.local_ranks
    k = prm.kmer_size
    corr = {}
    for r in ranks:
        h = C.c_void_p()
        b = skm_bits[r].contiguous().view(-1)
        capi.check(L.bg_corr_begin(ctxs[r].h, reads[r].h, k, C.byref(prm.correct), _ptr(skm_recs[r]), _ptr(skm_org[r]),
                                   _ptr(b), b.shape[0], C.byref(h)))
        corr[r] = h
    steps = 0
    while True:
        ph.mark("correct_step")
        probes, pend = {}, {}
        for r in ranks:
            npend = C.c_uint64()
            pr = C.c_void_p()
            capi.check(L.bg_corr_step(ctxs[r].h, corr[r], C.byref(npend), C.byref(pr)))
            pend[r] = npend.value
            probes[r] = pr
        steps += 1
        if _all_sum(comm, pend) == 0:
            break
        ph.mark("correct_probe_exchange")
        send, porg = {}, {}
        for r in ranks:
            if probes[r]:
                recs, org, counts, _ = _skm_extract(ctxs[r], probes[r], k, P, dev[r])
                L.bg_reads_free(probes[r])
            else:
                recs = torch.empty((0, 2), dtype=torch.int64, device=dev[r])
                org = torch.empty((0,), dtype=torch.int64, device=dev[r])
                counts = [0] * P
            send[r] = (recs, counts)
            porg[r] = org
        recv, cm = comm.all_to_all_v(send)
        back = {}
        for r in ranks:
            rows = recv[r].contiguous()
            bits = torch.empty((max(rows.shape[0], 1),), dtype=torch.int32, device=dev[r])
            # every window of a probe is looked up: a probe answers by K-MER, so a search that reaches one of its later
            # windows by another route (other substitutions further left) must find the true answer there, not the
            # "absent" an early exit at an earlier window would leave (tests/fuzz_parity.py seed 300028 found that)
            capi.check(L.bg_skm_lookup(ctxs[r].h, owned[r].h, _ptr(rows), rows.shape[0], _ptr(bits), 0))
            back[r] = bits[:rows.shape[0]].view(-1, 1)
        ans = comm.all_to_all_reply(back, cm)
        for r in ranks:
            a = ans[r].contiguous().view(-1)
            capi.check(L.bg_corr_answers(ctxs[r].h, corr[r], _ptr(porg[r]), _ptr(a), a.shape[0]))
        del send, recv, back, ans, porg
    corrected, cstats, nprobes = {}, {}, {}
    import biograph_amd as bg
    for r in ranks:
        h = C.c_void_p()
        st = capi.CorrectStats()
        capi.check(L.bg_corr_finish(ctxs[r].h, corr[r], C.byref(h), C.byref(st)))
        nprobes[r] = int(L.bg_corr_probes(corr[r]))
        L.bg_corr_free(corr[r])
        corrected[r] = bg.Reads(ctxs[r], h)
        cstats[r] = {n: int(getattr(st, n)) for n, _ in st._fields_}
    return corrected, cstats, steps, nprobes


def _records_sharded_cover(comm, ctxs, corrected, W, k, dev, ph):
    """Suffix records of a sharded build without a replicated cover index (csrc/shard_cover.hip): oriented reads
    and candidate suffixes meet on the rank that owns their leading bases.  Returns ({r: record rows}, n_records)."""
    L = capi.lib()
    P = comm.world
    ranks = comm.local_ranks
    i64 = torch.int64
    rw = W + 1
    ph.mark("cover_oriented_rows")
    rows, hists, qs = {}, {}, {}
    for r in ranks:
        n = len(corrected[r])
        rows[r] = torch.empty((2 * n, rw), dtype=i64, device=dev[r])
        capi.check(L.bg_oriented_rows(ctxs[r].h, corrected[r].h, W, _ptr(rows[r])))
        corrected[r].free()
        hists[r] = torch.empty((4 ** PREFIX_BASES,), dtype=i64, device=dev[r])
        capi.check(L.bg_rows_prefix_hist(ctxs[r].h, _ptr(rows[r]), 2 * n, rw, PREFIX_BASES, PREFIX_SAMPLE, _ptr(hists[r])))
        qh = C.c_void_p()
        capi.check(L.bg_cover_query_begin(ctxs[r].h, _ptr(rows[r]), 2 * n, W, k, C.byref(qh)))
        qs[r] = qh
    gh = comm.all_reduce_sum(hists)[ranks[0]].cpu().numpy()
    pparams = [PREFIX_BASES] + choose_splitters(gh, P)
    ph.mark("cover_index_exchange")
    send = {r: _partition(ctxs[r], rows[r], capi.PART_PREFIX, P, pparams) for r in ranks}
    del rows, hists
    recv, _ = comm.all_to_all_v(send)
    del send
    _relieve(ctxs, dev)
    ph.mark("cover_index_build")
    covers, parts = {}, {}
    for r in ranks:
        t = recv[r].contiguous()
        ch = C.c_void_p()
        capi.check(L.bg_cover_build(ctxs[r].h, _ptr(t), t.shape[0], W, k, C.byref(ch)))
        covers[r] = ch
        parts[r] = [t]  # the oriented reads a rank indexes are its j = 0 records
    del recv
    if os.environ.get("BG_COVER_MODE", "filtered") != "windows":
        return _filtered_cover_rounds(comm, ctxs, covers, qs, parts, W, k, dev, ph, pparams)
    bounds = [int(x) for x in os.environ.get("BG_COVER_ROUNDS", COVER_ROUNDS).split(",")]
    if bounds[-1] < 256:
        bounds.append(256)
    for j_lo, j_hi in zip(bounds[:-1], bounds[1:]):
        ph.mark("cover_round_%d" % j_lo)
        pend = {r: int(L.bg_cover_query_pending(qs[r])) for r in ranks}
        if _all_sum(comm, pend) == 0:
            break
        cand = {}
        pp = np.ascontiguousarray(pparams, np.uint64)
        for r in ranks:
            # candidates are generated straight into partitioned order (no unpartitioned copy); the buffer holds every
            # slot of the round, the rows actually produced (suffixes with >= K bases left) are its head
            cap = pend[r] * (j_hi - j_lo)
            t = torch.empty((max(cap, 1), rw), dtype=i64, device=dev[r])
            n = C.c_uint64()
            counts = (C.c_uint64 * P)()
            capi.check(L.bg_cover_query_candidates_split(ctxs[r].h, qs[r], j_lo, j_hi, P, capi.np_ptr(pp), _ptr(t), cap, counts,
                                                         C.byref(n)))
            cand[r] = (t[:n.value], [int(c) for c in counts])
            del t
        got, cm = comm.all_to_all_v(cand)
        back = {}
        for r in ranks:
            t = got[r].contiguous()
            got[r] = t
            cv = torch.empty((max(t.shape[0], 1),), dtype=torch.uint8, device=dev[r])
            capi.check(L.bg_cover_test(ctxs[r].h, covers[r], _ptr(t), t.shape[0], _ptr(cv)))
            back[r] = cv[:t.shape[0]].view(-1, 1)
        ans = comm.all_to_all_reply(back, cm)
        keep_send = {}
        for r in ranks:
            a = ans[r].contiguous().view(-1)
            sent = cand[r][0]
            kp = torch.empty((max(sent.shape[0], 1),), dtype=torch.uint8, device=dev[r])
            capi.check(L.bg_cover_query_answers(ctxs[r].h, qs[r], _ptr(sent), _ptr(a), sent.shape[0], _ptr(kp)))
            keep_send[r] = kp[:sent.shape[0]].view(-1, 1)
        keeps = comm.all_to_all_again(keep_send, cm)
        for r in ranks:
            parts[r].append(_select_rows(ctxs[r], got[r], keeps[r].contiguous().view(-1)))
        del cand, got, back, ans, keep_send, keeps
        _relieve(ctxs, dev)
    ph.mark("cover_plain_tails")
    pend = {r: int(L.bg_cover_query_pending(qs[r])) for r in ranks}
    if _all_sum(comm, pend) and k > 1:
        plain = {}
        for r in ranks:
            cap = pend[r] * (k - 1)
            t = torch.empty((max(cap, 1), rw), dtype=i64, device=dev[r])
            n = C.c_uint64()
            capi.check(L.bg_cover_query_plain(ctxs[r].h, qs[r], _ptr(t), cap, C.byref(n)))
            plain[r] = _partition(ctxs[r], t[:n.value], capi.PART_PREFIX, P, pparams)
            del t
        got, _ = comm.all_to_all_v(plain)
        for r in ranks:
            parts[r].append(got[r])
        del plain, got
    for r in ranks:
        L.bg_cover_query_free(qs[r])
        L.bg_cover_free(covers[r])
        parts[r] = [t.contiguous() for t in parts[r] if t.shape[0]]
    n_records = _all_sum(comm, {r: sum(t.shape[0] for t in parts[r]) for r in ranks})
    return parts, n_records  # a list of row arrays per rank: bg_records_from_row_parts takes them as they are


# hits per pending read and round of the filtered cover: one at first (nine reads in ten are settled by their first hit),
# more once few reads are left, so that the tail of reads whose candidates keep failing ends after a few rounds
COVER_SCAN_HITS = "1,2,4,16,64"
# suffix windows the records are generated in (one all_to_all each, in flight while the next window is generated): narrow
# at first, where most records are, so that generation and transfer overlap from early on
COVER_RECORD_WINDOWS = "1,3,5,9,17,33,256"


def _filtered_cover_rounds(comm, ctxs, covers, qs, parts, W, k, dev, ph, pparams):
    """The cover rounds with the owners' first-K-mer filters on every rank (csrc/shard.hip: bg_cover_filter_build,
    bg_cover_query_scan / _scan_answers / _records_split): a sender asks only about suffixes whose K-mer some read starts
    with.  Collectives: one all_gather (filters), one all_to_all + reply per scan round, one all_to_all per record window."""
    L = capi.lib()
    P = comm.world
    ranks = comm.local_ranks
    i64 = torch.int64
    rw = W + 1
    pp = np.ascontiguousarray(pparams, np.uint64)
    ph.mark("cover_filters")
    owned = [row[0] for row in comm.gather_ints({r: [sum(t.shape[0] for t in parts[r])] for r in ranks})]
    log2w = max(10, int(np.ceil(np.log2(max(max(owned), 1) / 2.0))))  # 32 bits per indexed read, in 64-bit words
    # The gathered filters are the one structure of the routed build that grows with the number of ranks (P x 2^log2w x 8
    # bytes on every rank): they stay within an eighth of the device (BG_COVER_FILTER_BUDGET) -- fewer bits per read mean more
    # candidates to verify, the records are the same (csrc/sharded.hip applies the same rule).
    try:
        budget = int(os.environ.get("BG_COVER_FILTER_BUDGET", "0")) or torch.cuda.mem_get_info(dev[ranks[0]])[1] // 8
    except Exception:
        budget = 1 << 35
    while log2w > 10 and (P << log2w) * 8 > budget:
        log2w -= 1
    bits = {}
    for r in ranks:
        bits[r] = torch.empty((1 << log2w,), dtype=i64, device=dev[r])
        capi.check(L.bg_cover_filter_build(ctxs[r].h, covers[r], log2w, _ptr(bits[r])))
    gathered = comm.all_gather_v(bits)
    filt = {r: torch.cat([g.to(dev[r]) for g in gathered[r]]).contiguous() for r in ranks}
    del bits, gathered
    hits_plan = [int(x) for x in os.environ.get("BG_COVER_SCAN_HITS", COVER_SCAN_HITS).split(",")]
    rnd = 0
    while True:
        pend = {r: int(L.bg_cover_query_pending(qs[r])) for r in ranks}
        if _all_sum(comm, pend) == 0:
            break
        hits = hits_plan[min(rnd, len(hits_plan) - 1)]
        ph.mark("cover_scan_%d" % rnd)
        rnd += 1
        cand = {}
        for r in ranks:
            cap = pend[r] * hits
            t = torch.empty((max(cap, 1), rw), dtype=i64, device=dev[r])
            n = C.c_uint64()
            counts = (C.c_uint64 * P)()
            capi.check(L.bg_cover_query_scan(ctxs[r].h, qs[r], _ptr(filt[r]), log2w, hits, P, capi.np_ptr(pp), _ptr(t), cap, counts,
                                             C.byref(n)))
            cand[r] = (t[:n.value], [int(c) for c in counts])
            del t
        got, cm = comm.all_to_all_v(cand)
        back = {}
        for r in ranks:
            t = got[r].contiguous()
            cv = torch.empty((max(t.shape[0], 1),), dtype=torch.uint8, device=dev[r])
            capi.check(L.bg_cover_test(ctxs[r].h, covers[r], _ptr(t), t.shape[0], _ptr(cv)))
            back[r] = cv[:t.shape[0]].view(-1, 1)
        del got
        ans = comm.all_to_all_reply(back, cm)
        for r in ranks:
            a = ans[r].contiguous().view(-1)
            sent = cand[r][0]
            capi.check(L.bg_cover_query_scan_answers(ctxs[r].h, qs[r], _ptr(sent), _ptr(a), sent.shape[0]))
        del cand, back, ans
    del filt
    for r in ranks:
        L.bg_cover_free(covers[r])
    covers.clear()
    _relieve(ctxs, dev)
    wins = [int(x) for x in os.environ.get("BG_COVER_RECORD_WINDOWS", COVER_RECORD_WINDOWS).split(",")]
    if wins[-1] < 256:
        wins.append(256)
    # the windows are pipelined: while the records of one window travel (all_to_all_v_start), the next window is generated
    flying = None
    for j_lo, j_hi in zip(wins[:-1], wins[1:]):
        ph.mark("cover_records_%d" % j_lo)
        live = {}
        for r in ranks:
            n = C.c_uint64()
            capi.check(L.bg_cover_query_records_live(ctxs[r].h, qs[r], j_lo, C.byref(n)))
            live[r] = n.value
        if _all_sum(comm, live) == 0:
            break
        send = {}
        for r in ranks:
            cap = live[r] * (j_hi - j_lo)
            t = torch.empty((max(cap, 1), rw), dtype=i64, device=dev[r])
            n = C.c_uint64()
            counts = (C.c_uint64 * P)()
            capi.check(L.bg_cover_query_records_split(ctxs[r].h, qs[r], j_lo, j_hi, P, capi.np_ptr(pp), _ptr(t), cap, counts, C.byref(n)))
            send[r] = (t[:n.value], [int(c) for c in counts])
            del t
        if flying is not None:
            got, _ = comm.all_to_all_v_finish(flying)
            for r in ranks:
                parts[r].append(got[r])
            del got
        flying = comm.all_to_all_v_start(send)
        del send
    if flying is not None:
        got, _ = comm.all_to_all_v_finish(flying)
        for r in ranks:
            parts[r].append(got[r])
        del got, flying
    _relieve(ctxs, dev)
    for r in ranks:
        L.bg_cover_query_free(qs[r])
        parts[r] = [t.contiguous() for t in parts[r] if t.shape[0]]
    n_records = _all_sum(comm, {r: sum(t.shape[0] for t in parts[r]) for r in ranks})
    return parts, n_records


class SeqsetSlice:
    """One rank's stretch of a sharded seqset: entries [first, first + size()) of n_total."""

    def __init__(self, ctx, seqset, first, n_total, max_len, is_last, base_pop, fixed):
        self.ctx, self.seqset, self.first, self.n_total, self.max_len = ctx, seqset, first, n_total, max_len
        self.is_last, self.base_pop, self.fixed = is_last, base_pop, fixed

    def size(self):
        return self.seqset.size()

    def free(self):
        self.seqset.free()


def _aligned_slices(comm, ctxs, shards, part_n, max_len, dev):
    """Shards hold arbitrary numbers of entries; container members are addressed in 64-bit words and 512-bit
    groups.  Every rank passes its last < 512 entries on (sizes, shared, one prev-mask byte each: a few KB through
    all_gather) so that rank r holds the entries [B_r, B_r+1) with B_r = its global start rounded down to 512."""
    import biograph_amd as bg
    L = capi.lib()
    P = comm.world
    ranks = comm.local_ranks
    off = [0]
    for n in part_n:
        off.append(off[-1] + n)
    n_total = off[-1]
    B = [(off[r] // 512) * 512 for r in range(P)] + [n_total]
    cols, tails = {}, {}
    for r in ranks:
        n = part_n[r]
        t = torch.empty((3, max(n, 1)), dtype=torch.uint8, device=dev[r])
        capi.check(L.bg_seqset_slice_masks(ctxs[r].h, shards[r].h, _ptr(t[0]), _ptr(t[1]), _ptr(t[2])))
        cols[r] = t[:, :n]
        tails[r] = cols[r][:, max(0, n - 511):].t().contiguous()  # [<=511, 3]
    gt = comm.all_gather_v(tails)
    slices, pops = {}, {}
    for r in ranks:
        pre = []
        for q in range(r):  # entries [B_r, off_r) sit in the tails of the ranks before r
            tq = gt[r][q].to(dev[r])
            start = off[q + 1] - tq.shape[0]
            lo = max(B[r] - start, 0)
            if lo < tq.shape[0]:
                pre.append(tq[lo:].t())
        n_own = part_n[r] - (off[r + 1] - B[r + 1]) if r + 1 < P else part_n[r]
        own = cols[r][:, :max(n_own, 0)]
        if n_own < 0:  # this rank's entries and some of the inherited ones belong to the next rank
            allc = torch.cat(pre + [cols[r]], dim=1) if pre else cols[r]
            allc = allc[:, :B[r + 1] - B[r]]
        else:
            allc = torch.cat(pre + [own], dim=1) if pre else own
        allc = allc.contiguous()
        m = allc.shape[1]
        assert m == B[r + 1] - B[r], (r, m, B[r], B[r + 1])
        h = C.c_void_p()
        capi.check(L.bg_seqset_slice_build(ctxs[r].h, m, _ptr(allc[0]), _ptr(allc[1]), _ptr(allc[2]), max_len if n_total else 0,
                                           C.byref(h)))
        slices[r] = bg.Seqset(ctxs[r], h)
        pc = (C.c_uint64 * 4)()
        capi.check(L.bg_seqset_slice_popcounts(h, pc))
        pops[r] = [int(x) for x in pc]
    allp = comm.all_gather_obj(pops)[ranks[0]]
    tot = [sum(allp[q][b] for q in range(P)) for b in range(4)]
    fixed = [0]
    for b in range(4):
        fixed.append(fixed[-1] + tot[b])
    if fixed[4] != n_total:
        raise RuntimeError("Invalid seqset in finalize: %d != %d" % (fixed[4], n_total))  # seqset.cpp:123-126
    out = {}
    for r in ranks:
        base = [sum(allp[q][b] for q in range(r)) for b in range(4)]
        out[r] = SeqsetSlice(ctxs[r], slices[r], B[r], n_total, max_len if n_total else 0, r == P - 1, base, fixed)
    return out


def write_slices(comm, slices, path, argv=()):
    """All ranks write ONE seqset container: rank 0 lays the file out (headers, JSON members, `fixed`, room for the
    rest), every rank then writes its bytes of every member in place.  slices = {local rank: SeqsetSlice}.  -> the uuid
    the seqset was given."""
    L = capi.lib()
    ranks = comm.local_ranks
    offs = None
    if 0 in ranks:
        s0 = slices[0]
        fx = (C.c_uint64 * 5)(*s0.fixed)
        av = (C.c_char_p * max(len(argv), 1))(*[a.encode() for a in argv])
        ho = (C.c_uint64 * 15)()
        uu = C.create_string_buffer(37)
        capi.check(L.bg_seqset_sharded_create_file(path.encode(), s0.n_total, s0.max_len, fx, len(argv), av, ho, uu))
        offs = [int(x) for x in ho] + [uu.value.decode()]
    offs = [o for o in comm.all_gather_obj({r: offs if r == 0 else None for r in ranks})[ranks[0]] if o is not None][0]
    uuid, offs = offs[-1], offs[:-1]
    comm.barrier()
    ho = (C.c_uint64 * 15)(*offs)
    for r in ranks:
        s = slices[r]
        bp = (C.c_uint64 * 4)(*s.base_pop)
        capi.check(L.bg_seqset_slice_write(s.ctx.h, s.seqset.h, path.encode(), ho, s.first, int(s.is_last), bp))
    comm.barrier()
    return uuid  # the seqset's uuid (every rank): what the sample's readmap is bound to (readmap.json)


def sharded_create(comm, ctxs, inputs, params=None, device=None, mode=None, gather=True, keep_entry_rows=False,
                   keep_corrected=False, readmap=None):
    """comm: communicator; ctxs[r]: biograph_amd.Context of local rank r (its HIP stream must be torch's current
    stream on that device: the exchange buffers are torch tensors handed to the library by pointer); inputs[r] =
    (seq uint8 tensor, offs int64 tensor, n_reads) resident on that rank's device.  gather=True (small outputs,
    tests): returns {r: (Seqset | None, stats)}, the assembled seqset lives on rank 0.  gather=False (what scales):
    returns {r: (SeqsetSlice, stats)} — every rank keeps a 512-aligned stretch of the entries and write_slices()
    lets all ranks write one container in place.

    mode "routed" (default; k >= 14): nothing is replicated except the owners' first-K-mer bit filters of the cover rounds
    (32 bits per indexed read of every rank, capped at an eighth of the device: _filtered_cover_rounds) — the kept k-mers stay with the rank that counted
    them and the cover index is range-partitioned like the records; "replicated" (BG_SHARD_REPLICATED=1, and
    what k < 14 falls back to): kept k-mers and corrected reads are all-gathered on every rank.

    readmap="paired" | "unpaired": also the sample's readmap, sharded like the seqset (sharded_readmap below); every
    slice (or, gather=True, the seqset on rank 0) comes back with .readmap_columns = ReadmapColumns of its rank."""
    import biograph_amd as bg
    L = capi.lib()
    P = comm.world
    ranks = comm.local_ranks
    any_ctx = ctxs[ranks[0]]
    if readmap not in (None, "paired", "unpaired"):
        raise ValueError("sharded_create: readmap takes None, 'paired' or 'unpaired'")
    prm = params or any_ctx.default_params()
    k = prm.kmer_size
    dev = {r: inputs[r][0].device for r in ranks}
    i64 = torch.int64
    for r in ranks:
        if dev[r].type == "cuda" and ctxs[r].stream is not None:
            cur = torch.cuda.current_stream(dev[r]).cuda_stream
            if int(ctxs[r].stream or 0) != int(cur or 0):
                raise ValueError("sharded_create: the Context of rank %d runs on another HIP stream than torch's current "
                                 "one — the exchange buffers would be read before they are complete" % r)

    ph = _Phases(ctxs[ranks[0]])
    ph.mark("pack")
    # ---- S1 pack (local)
    reads = {r: ctxs[r].pack_reads_device(inputs[r][0].data_ptr(), inputs[r][1].data_ptr(), inputs[r][2])
             for r in ranks}
    n_bases = comm.all_gather_obj({r: int(reads[r].total_bases) for r in ranks})[ranks[0]]

    ph.mark("kmer_extract_partition")
    # ---- S2 windows -> owner ranks: super-k-mer records (16 B per ~10 windows) when the k-mer size allows a
    # 12-base minimiser, else k-mer instances (8 B per window)
    use_skm = bool(L.bg_skm_supported(k)) and not os.environ.get("BG_KMER_SKM") == "0"
    if mode is None:
        mode = "replicated" if os.environ.get("BG_SHARD_REPLICATED") == "1" else "routed"
    if not use_skm:
        mode = "replicated"
    send = {}
    n_inst = {}
    skm_org = {}
    skm_recs = {}
    for r in ranks:
        counts = (C.c_uint64 * P)()
        if use_skm:
            recs, org, cnts, n = _skm_extract(ctxs[r], reads[r].h, k, P, dev[r], first_room=0.385)
            n_inst[r] = n
            skm_recs[r] = recs
            skm_org[r] = org
            send[r] = (recs, cnts)
        else:
            n = C.c_uint64()
            capi.check(L.bg_kmer_instances(ctxs[r].h, reads[r].h, k, C.byref(n)))
            n_inst[r] = n.value
            keys = torch.empty((n.value, 1), dtype=i64, device=dev[r])
            # extraction + owner partition in one pass: whole hash-byte buckets go to their owner rank
            capi.check(L.bg_kmer_extract_by_owner(ctxs[r].h, reads[r].h, k, P, _ptr(keys), n.value, counts))
            send[r] = (keys, [int(c) for c in counts])
            del keys
    recs = org = None  # loop locals would keep the last rank's buffers alive to the end of the build
    ph.mark("kmer_all_to_all")
    recv, kcm = comm.all_to_all_v(send)  # kcm[source][owner]
    del send
    ph.mark("kmer_count")
    kept = {}
    owned = {}
    distinct = {}
    n_kept_local = {}
    bits_back = {}
    for r in ranks:
        keys = recv[r].contiguous()
        h = C.c_void_p()
        if use_skm:
            bits = torch.empty((max(keys.shape[0], 1),), dtype=torch.int32, device=dev[r])
            capi.check(L.bg_skm_count(ctxs[r].h, _ptr(keys), keys.shape[0], k, prm.min_kmer_count, C.byref(h), _ptr(bits)))
            # the presence bits of every record go back to the rank it came from
            bits_back[r] = bits[:keys.shape[0]].view(-1, 1)
        else:
            capi.check(L.bg_kmer_count_keys(ctxs[r].h, _ptr(keys), keys.shape[0], k, prm.min_kmer_count, C.byref(h)))
        ks = bg.KmerSet(ctxs[r], h, k)
        distinct[r] = ks.distinct
        n_kept_local[r] = len(ks)
        if mode == "routed":
            # the owner keeps its k-mers (hash set) and settles the records its counting tables gave up on
            capi.check(L.bg_kmerset_build_hash(ctxs[r].h, ks.h))
            capi.check(L.bg_skm_lookup(ctxs[r].h, ks.h, _ptr(keys), keys.shape[0], _ptr(bits), 1))
            owned[r] = ks
        else:
            t = torch.empty((len(ks),), dtype=i64, device=dev[r])
            capi.check(L.bg_kmerset_export(ctxs[r].h, ks.h, _ptr(t)))
            kept[r] = t
            ks.free()
    del recv
    keys = bits = t = ks = None
    skm_bits = comm.all_to_all_reply(bits_back, kcm) if use_skm else None
    del bits_back
    tot_inst = sum(comm.all_gather_obj(n_inst)[ranks[0]])
    tot_distinct = sum(comm.all_gather_obj(distinct)[ranks[0]])
    n_kept = _all_sum(comm, n_kept_local)
    extra_stats = {"mode": mode}

    if mode == "routed":
        _relieve(ctxs, dev)
        corrected, cstats, steps, nprobes = _correct_routed(comm, ctxs, reads, owned, skm_recs, skm_org, skm_bits, prm,
                                                            dev, ph)
        for r in ranks:
            reads[r].free()
            owned[r].free()
        del skm_bits, skm_recs, skm_org
        extra_stats["correct_steps"] = steps
        extra_stats["correct_probes"] = _all_sum(comm, nprobes)
    else:
        ph.mark("kmerset_all_gather_build")
        _relieve(ctxs, dev)
        gathered = comm.all_gather_v(kept)
        ksets = {}
        for r in ranks:
            allk = torch.cat([g.to(dev[r]) for g in gathered[r]]).contiguous()
            h = C.c_void_p()
            capi.check(L.bg_kmerset_from_kmers(ctxs[r].h, _ptr(allk), allk.shape[0], k, reads[r].h, C.byref(h)))
            ksets[r] = bg.KmerSet(ctxs[r], h, k)
            if use_skm:  # window presence of this rank's reads: the correction looks nothing up
                b = skm_bits[r].contiguous().view(-1)
                capi.check(L.bg_kmerset_attach_presence(ctxs[r].h, h, reads[r].h, _ptr(skm_recs[r]), _ptr(skm_org[r]),
                                                        _ptr(b), b.shape[0]))
        del gathered, kept, skm_bits, skm_recs, skm_org

        ph.mark("correct")
        # ---- S3 correction (local, replicated k-mer set)
        corrected = {}
        cstats = {}
        for r in ranks:
            corrected[r], cstats[r] = ctxs[r].correct_reads(reads[r], ksets[r], prm.correct.max_corrections,
                                                            prm.correct.min_good_run, prm.correct.trim_after_portion)
            reads[r].free()
            ksets[r].free()
    all_cs = comm.all_gather_obj(cstats)[ranks[0]]
    max_len = max(comm.all_gather_obj({r: int(corrected[r].max_len) for r in ranks})[ranks[0]])
    W = max(1, (max_len + 31) // 32)

    kept_reads = {}
    if keep_corrected or readmap:  # this rank's corrected reads, packed (the readmaps are built from them)
        for r in ranks:
            kept_reads[r] = export_reads(ctxs[r], corrected[r], W, dev[r]) + (max_len,)
    _relieve(ctxs, dev)
    if mode == "routed":
        recv, n_records = _records_sharded_cover(comm, ctxs, corrected, W, k, dev, ph)
    else:
        ph.mark("reads_all_gather")
        # ---- corrected reads replicated (cover index spans all reads)
        words, lens = {}, {}
        for r in ranks:
            n = len(corrected[r])
            words[r] = torch.empty((n, W), dtype=i64, device=dev[r])
            lens[r] = torch.empty((n, 2), dtype=torch.uint8, device=dev[r])  # u16 lengths as bytes (RCCL has no int16)
            capi.check(L.bg_reads_export(ctxs[r].h, corrected[r].h, W, _ptr(words[r]), _ptr(lens[r])))
            corrected[r].free()
        gw = comm.all_gather_v(words)
        gl = comm.all_gather_v(lens)
        del words, lens
        ph.mark("suffix_records")
        send = {}
        hists = {}
        for r in ranks:
            aw = torch.cat([g.to(dev[r]) for g in gw[r]]).contiguous()
            al = torch.cat([g.to(dev[r]) for g in gl[r]]).contiguous()
            counts = [g.shape[0] for g in gl[r]]
            lo = sum(counts[:r])
            hi = lo + counts[r]
            h = C.c_void_p()
            capi.check(L.bg_reads_import(ctxs[r].h, aw.shape[0], W, max_len, _ptr(aw), _ptr(al), C.byref(h)))
            all_reads = bg.Reads(ctxs[r], h)
            del aw, al
            # ---- S4 suffix records of own reads
            rh = C.c_void_p()
            capi.check(L.bg_suffix_records(ctxs[r].h, all_reads.h, lo, hi, k, C.byref(rh)))
            all_reads.free()
            n_rec = L.bg_records_count(rh)
            rows = torch.empty((n_rec, W + 1), dtype=i64, device=dev[r])
            capi.check(L.bg_records_rows(ctxs[r].h, rh, _ptr(rows)))
            hists[r] = torch.empty((4 ** PREFIX_BASES + 1,), dtype=i64, device=dev[r])
            capi.check(L.bg_records_prefix_hist(ctxs[r].h, rh, PREFIX_BASES, PREFIX_SAMPLE, _ptr(hists[r])))
            hists[r][-1] = n_rec  # the record total rides along with the bins
            L.bg_records_free(rh)
            send[r] = rows
        del gw, gl
        ph.mark("records_partition")
        _relieve(ctxs, dev)
        ghist = comm.all_reduce_sum(hists)
        gh = ghist[ranks[0]].cpu().numpy()
        splitters = choose_splitters(gh[:-1], P)
        n_records = int(gh[-1])
        for r in ranks:
            send[r] = _partition(ctxs[r], send[r], capi.PART_PREFIX, P, [PREFIX_BASES] + splitters)
        ph.mark("records_all_to_all")
        recv, _ = comm.all_to_all_v(send)
        del send
    stats = {"n_reads": sum(row[0] for row in comm.gather_ints({r: [int(inputs[r][2])] for r in ranks})),
             "n_bases": sum(n_bases), "n_kmers": tot_inst, "n_distinct_kmers": tot_distinct, "n_kept_kmers": n_kept,
             "n_corrected_reads": sum(c["corrected_reads"] for c in all_cs),
             "n_corrected_bases": sum(c["corrected_bases"] for c in all_cs), "n_records": n_records}
    stats.update(extra_stats)
    out = _records_to_output(comm, ctxs, recv, W, max_len, dev, ph, gather, stats, keep_entry_rows=keep_entry_rows,
                             readmap=(kept_reads, readmap == "paired") if readmap else None)
    for r in kept_reads:
        if out[r][0] is not None and keep_corrected:
            out[r][0].corrected = kept_reads[r]
    return out


def export_reads(ctx, reads, W, device):
    """Packed reads as tensors: (words int64 [n, W], lengths uint8 [n, 2] = little-endian u16)."""
    n = len(reads)
    words = torch.empty((n, W), dtype=torch.int64, device=device)
    lens = torch.empty((n, 2), dtype=torch.uint8, device=device)
    capi.check(capi.lib().bg_reads_export(ctx.h, reads.h, W, _ptr(words), _ptr(lens)))
    return words, lens


def import_reads(ctx, words, lens, max_len):
    """The inverse of export_reads -> Reads."""
    import biograph_amd as bg
    h = C.c_void_p()
    words = words.contiguous()
    lens = lens.contiguous()
    capi.check(capi.lib().bg_reads_import(ctx.h, words.shape[0], words.shape[1], max_len, _ptr(words), _ptr(lens), C.byref(h)))
    return bg.Reads(ctx, h)


class ReadmapColumns:
    """One rank's stretch of a sharded readmap: the per-read columns of the read ids [first, first + n) of n_total --
    entry (GLOBAL seqset entry id, int64), read_len / mate_loop_ptr (int32; the pointers are global read ids), is_forward
    (uint8), all on the rank's device -- and what the container needs beside them."""

    def __init__(self, first, n_total, n_entries, max_read_len, entry, read_len, mate_loop_ptr, is_forward):
        self.first, self.n_total, self.n_entries, self.max_read_len = first, n_total, n_entries, max_read_len
        self.entry, self.read_len, self.mate_loop_ptr, self.is_forward = entry, read_len, mate_loop_ptr, is_forward


def sharded_readmap(comm, ctxs, shards, part_n, reads, paired, W, dev, ph=None):
    """make_readmap::create_from_reads (make_readmap.cpp:120-392) for ONE sample whose seqset is range-partitioned:
    shards[r] = rank r's entries with their keys resident (global entries [off_r, off_r+1), off = running sum of part_n),
    reads[r] = (words, lens, max_len) = rank r's corrected reads (whole pairs when `paired`).  Every structure has one
    owner: a read's entry is found on the shard its key falls into, a row lives with its entry, a claim of the mate loop
    is settled where the claimed group lives.  -> {r: ReadmapColumns}.  Collectives: one all_to_all + reply for the entry
    ids, one all_to_all for the rows, one all_to_all + reply per claim hop (3)."""
    import biograph_amd as bg
    L = capi.lib()
    P = comm.world
    ranks = comm.local_ranks
    i64 = torch.int64
    mark = ph.mark if ph is not None else (lambda name: None)
    off = [0]
    for n in part_n:
        off.append(off[-1] + n)
    h_off = (C.c_uint64 * (P + 1))(*off)
    mark("readmap_lookup")
    # ---- 1. entry ids: oriented rows (tagged with their index) to the shard their key falls into
    bounds = {}
    for r in ranks:
        la = np.zeros(W + 1, np.uint64)
        fi = np.zeros(W + 1, np.uint64)
        capi.check(L.bg_seqset_boundary_rows(ctxs[r].h, shards[r].h, capi.np_ptr(fi), capi.np_ptr(la)))
        bounds[r] = [int(x) for x in la]
    lasts = comm.all_gather_obj(bounds)[ranks[0]]
    kb = np.ascontiguousarray(np.array(lasts, np.uint64).reshape(P, W + 1))
    handles, send, n_or = {}, {}, {}
    for r in ranks:
        words, lens, mlen = reads[r]
        handles[r] = import_reads(ctxs[r], words, lens, mlen)
        n = len(handles[r])
        n_or[r] = 2 * n
        rows = torch.empty((2 * n, W + 1), dtype=i64, device=dev[r])
        capi.check(L.bg_oriented_rows(ctxs[r].h, handles[r].h, W, _ptr(rows)))
        rows[:, W] |= torch.arange(2 * n, dtype=i64, device=dev[r]) << 8  # the row's way home
        if paired and n % 2:
            raise ValueError("sharded_readmap: a rank holds an odd number of reads of a paired sample")
        # a dropped read (length 0) needs no look-up but keeps its place: it travels to the first shard and comes back as ~0
        send[r] = _partition(ctxs[r], rows, capi.PART_KEY_BOUNDS, P, kb)
        del rows
    recv, cm = comm.all_to_all_v(send)
    back = {}
    for r in ranks:
        rows = recv[r].contiguous()
        ent = torch.empty((max(rows.shape[0], 1),), dtype=i64, device=dev[r])
        capi.check(L.bg_seqset_find_rows(ctxs[r].h, shards[r].h, _ptr(rows), rows.shape[0], W, off[r], _ptr(ent)))
        back[r] = ent[:rows.shape[0]].view(-1, 1)
    del recv
    answers = comm.all_to_all_reply(back, cm)
    del back
    mark("readmap_rows")
    # ---- 2. rows to the owner of their entry; sorted there
    rsend = {}
    for r in ranks:
        sent = send[r][0]
        eid = torch.full((max(n_or[r], 1),), -1, dtype=i64, device=dev[r])
        if sent.shape[0]:
            eid[sent[:, W] >> 8] = answers[r].contiguous().view(-1)
        n = n_or[r] // 2
        n_slots = (n // 2) * 4 if paired else n * 2
        rows = torch.empty((max(n_slots, 1), 3), dtype=i64, device=dev[r])
        m = C.c_uint64()
        capi.check(L.bg_readmap_rows(ctxs[r].h, handles[r].h, int(bool(paired)), _ptr(eid), P, h_off, _ptr(rows), n_slots, C.byref(m)))
        handles[r].free()
        rsend[r] = _partition(ctxs[r], rows[:m.value], capi.PART_ROW_DEST, P, None)
        del rows, eid
    del send, answers
    recv, _ = comm.all_to_all_v(rsend)
    del rsend
    rsh = {}
    for r in ranks:
        rows = recv[r].contiguous()
        h = C.c_void_p()
        capi.check(L.bg_readmap_shard_build(ctxs[r].h, _ptr(rows), rows.shape[0], C.byref(h)))
        rsh[r] = h
    del recv
    try:
        sizes = [row[0] for row in comm.gather_ints({r: [int(L.bg_readmap_shard_size(rsh[r]))] for r in ranks})]
        base = [0]
        for n in sizes:
            base.append(base[-1] + n)
        n_total = base[-1]
        if n_total >= 1 << 32:
            raise RuntimeError("sharded_readmap: more than 2^32 read ids (readmap.h holds them in 32 bits)")
        mark("readmap_mate_loops")
        # ---- 3. the claim hops (make_readmap.cpp:283-392)
        claims = {}
        for r in ranks:
            n = sizes[r]
            t = torch.empty((max(n, 1), 4), dtype=i64, device=dev[r])
            m = C.c_uint64()
            capi.check(L.bg_readmap_shard_first_claims(ctxs[r].h, rsh[r], base[r], P, h_off, _ptr(t), n, C.byref(m)))
            claims[r] = t[:m.value]
        for hop in (1, 2, 3):
            if not _all_sum(comm, {r: int(claims[r].shape[0]) for r in ranks}):
                break
            csend = {r: _partition(ctxs[r], claims[r].contiguous(), capi.PART_ROW_DEST, P, None) for r in ranks}
            crecv, ccm = comm.all_to_all_v(csend)
            reply, nxt = {}, {}
            for r in ranks:
                got = crecv[r].contiguous()
                n = got.shape[0]
                rep = torch.empty((max(n, 1),), dtype=i64, device=dev[r])
                nx = torch.empty((max(n, 1), 4), dtype=i64, device=dev[r])
                m = C.c_uint64()
                capi.check(L.bg_readmap_shard_apply_claims(ctxs[r].h, rsh[r], base[r], hop, _ptr(got), n, P, h_off, _ptr(rep), _ptr(nx),
                                                           C.byref(m)))
                reply[r] = rep[:n].view(-1, 1)
                nxt[r] = nx[:m.value]
            del crecv
            home = comm.all_to_all_reply(reply, ccm)
            for r in ranks:
                sent = csend[r][0].contiguous()
                rep = home[r].contiguous()
                capi.check(L.bg_readmap_shard_take_replies(ctxs[r].h, rsh[r], base[r], _ptr(sent), _ptr(rep), sent.shape[0]))
            claims = nxt
            del csend, reply, home
        mark("readmap_columns")
        mlen = max(x[0] for x in comm.gather_ints({r: [int(reads[r][2])] for r in ranks}))
        out = {}
        for r in ranks:
            n = sizes[r]
            e = torch.empty((max(n, 1),), dtype=i64, device=dev[r])
            rl = torch.empty((max(n, 1),), dtype=torch.int32, device=dev[r])
            ml = torch.empty((max(n, 1),), dtype=torch.int32, device=dev[r])
            fw = torch.empty((max(n, 1),), dtype=torch.uint8, device=dev[r])
            capi.check(L.bg_readmap_shard_columns(ctxs[r].h, rsh[r], _ptr(e), _ptr(rl), _ptr(ml), _ptr(fw)))
            out[r] = ReadmapColumns(base[r], n_total, off[-1], mlen, e[:n], rl[:n], ml[:n], fw[:n])
        return out
    finally:
        for h in rsh.values():
            L.bg_readmap_shard_free(h)


def gather_readmap(comm, ctxs, cols, seqset_uuid, writer=0):
    """The ranks' ReadmapColumns put together on rank `writer` as one resident Readmap (bg_readmap_from_arrays: the
    sparse_multi bit vectors and their bitcounts are derived there); None on the other ranks.  17 bytes per read id
    cross the links once."""
    ranks = comm.local_ranks
    r0 = ranks[0]
    c0 = cols[r0]
    parts = {}
    for name in ("entry", "read_len", "mate_loop_ptr", "is_forward"):
        parts[name] = comm.all_gather_v({r: getattr(cols[r], name).contiguous() for r in ranks})
    out = {}
    for r in ranks:
        if r != writer:
            out[r] = None
            continue
        dev = cols[r].entry.device
        full = {name: torch.cat([g.to(dev) for g in parts[name][r]]).contiguous() for name in parts}
        out[r] = ctxs[r].readmap_from_arrays(c0.n_entries, c0.max_read_len, seqset_uuid, full["entry"], full["read_len"],
                                             full["mate_loop_ptr"], full["is_forward"])
    return out


def _records_to_output(comm, ctxs, recv, W, max_len, dev, ph, gather, stats, keep_entry_rows=False, mark_inputs=0, readmap=None):
    """The back half of a sharded build, shared with sharded_merge: per-rank record rows (already on the rank that owns
    their leading bases) -> sorted prefix-free entries -> cross-shard boundaries -> prev bits through the tails
    exchange -> 512-aligned slices (or, gather=True, one seqset assembled on rank 0).  recv[r] = a rows tensor or a list
    of row tensors.  stats is completed in place (n_entries, shard_entries, tails_sent_in_full).
    mark_inputs=n (sharded merge): the rows carry their input's number (last word, bits 8..15) and every slice comes
    back with input_marks = (uint8 [n, shard entries], global index of the shard's first entry): the image of each
    input's entries under the merge map, by shard (before the slices were re-aligned)."""
    import biograph_amd as bg
    L = capi.lib()
    P = comm.world
    ranks = comm.local_ranks
    i64 = torch.int64
    extra_stats = {}
    ph.mark("entries_from_records")
    shards, kept_rows = {}, {}
    for r in ranks:
        rh = C.c_void_p()
        if isinstance(recv[r], list):
            np_ = len(recv[r])
            ptrs = (C.c_void_p * max(np_, 1))(*[t.data_ptr() for t in recv[r]])
            cnts = (C.c_uint64 * max(np_, 1))(*[t.shape[0] for t in recv[r]])
            capi.check(L.bg_records_from_row_parts(ctxs[r].h, np_, ptrs, cnts, W, C.byref(rh)))
        else:
            rows = recv[r].contiguous()
            capi.check(L.bg_records_from_rows(ctxs[r].h, _ptr(rows), rows.shape[0], W, C.byref(rh)))
        if mark_inputs:
            kept_rows[r] = recv[r] if not isinstance(recv[r], list) else list(recv[r])
        recv[r] = None  # the rows are in the library now
        h = C.c_void_p()
        st = BuildStats()
        capi.check(L.bg_entries_from_records(ctxs[r].h, rh, C.byref(h), C.byref(st)))
        L.bg_records_free(rh)
        shards[r] = bg.Seqset(ctxs[r], h)
    del recv

    ph.mark("boundaries")
    _relieve(ctxs, dev)
    # ---- shard boundaries
    def boundary(r):
        f = np.zeros(W + 1, np.uint64)
        la = np.zeros(W + 1, np.uint64)
        capi.check(L.bg_seqset_boundary_rows(ctxs[r].h, shards[r].h, capi.np_ptr(f), capi.np_ptr(la)))
        return [int(x) for x in f], [int(x) for x in la]

    firsts = comm.all_gather_obj({r: boundary(r)[0] for r in ranks})[ranks[0]]
    for r in ranks:
        nxt = _next_nonempty(firsts, r, +1)
        if nxt is not None:
            a = np.array(nxt, np.uint64)
            capi.check(L.bg_seqset_fix_boundary(ctxs[r].h, shards[r].h, None, capi.np_ptr(a)))
    lasts = comm.all_gather_obj({r: boundary(r)[1] for r in ranks})[ranks[0]]
    for r in ranks:
        prv = _next_nonempty(lasts, r, -1)
        if prv is not None:
            a = np.array(prv, np.uint64)
            capi.check(L.bg_seqset_fix_boundary(ctxs[r].h, shards[r].h, capi.np_ptr(a), None))

    marks = {}
    if mark_inputs:
        ph.mark("merge_marks")
        carry = {}
        for r in ranks:
            n = shards[r].size()
            marks[r] = torch.empty((mark_inputs, n), dtype=torch.uint8, device=dev[r])
            carry[r] = [0] * mark_inputs
            parts = kept_rows[r] if isinstance(kept_rows[r], list) else [kept_rows[r]]
            first = True
            for t in parts:
                t = t.contiguous()
                m = marks[r] if first else torch.empty_like(marks[r])
                hc = (C.c_uint32 * mark_inputs)()
                capi.check(L.bg_seqset_mark_inputs(ctxs[r].h, shards[r].h, _ptr(t), t.shape[0], mark_inputs, _ptr(m), hc))
                if not first:
                    marks[r] |= m
                first = False
                carry[r] = [a | int(b) for a, b in zip(carry[r], hc)]
            kept_rows[r] = None
        # an entry dropped at a shard's end as a prefix of the next shard's first entry maps to that entry
        every = comm.gather_ints(carry)
        sizes = [row[0] for row in comm.gather_ints({r: [int(shards[r].size())] for r in ranks})]
        for r in ranks:
            if not sizes[r]:
                continue
            q = r - 1
            while q >= 0:
                for i in range(mark_inputs):
                    if every[q][i]:
                        marks[r][i, 0] = 1
                if sizes[q]:
                    break
                q -= 1
        if any(any(every[q]) for q in range(P) if not any(sizes[q + 1:])):
            raise RuntimeError("sharded merge: an input entry has no extension among the merged entries")
    del kept_rows

    ph.mark("tails_partition")
    # ---- S5 prev bits: every entry's tail goes to the shard holding lower_bound(tail).  Short form first (the
    # tail's two leading words; the destination is settled by the sender, who has the whole key): the receiving
    # shard's two-word prefixes settle nearly all of them; it asks for the whole tail of the rest.
    bounds = np.ascontiguousarray(np.array(lasts, np.uint64).reshape(P, W + 1))
    send, short = {}, {}
    for r in ranks:
        n = shards[r].size()
        rows = torch.empty((n, 3), dtype=i64, device=dev[r])
        capi.check(L.bg_seqset_tail_rows2(ctxs[r].h, shards[r].h, P, capi.np_ptr(bounds), _ptr(rows)))
        send[r] = _partition(ctxs[r], rows, capi.PART_ROW_DEST, P, None)
        short[r] = send[r][0]
        del rows
    ph.mark("tails_all_to_all")
    recv, tcm = comm.all_to_all_v(send)
    del send
    ph.mark("apply_tails")
    back = {}
    for r in ranks:
        rows = recv[r].contiguous()
        need = torch.empty((max(rows.shape[0], 1),), dtype=torch.uint8, device=dev[r])
        capi.check(L.bg_seqset_apply_tails2(ctxs[r].h, shards[r].h, _ptr(rows), rows.shape[0], _ptr(need)))
        back[r] = need[:rows.shape[0]].view(-1, 1)
    del recv
    needs = comm.all_to_all_reply(back, tcm)
    del back
    ph.mark("tails_full")
    full = {}
    n_full = {}
    for r in ranks:
        nd = needs[r].contiguous().view(-1)
        cap = int(nd.sum().item()) if nd.shape[0] else 0
        t = torch.empty((max(cap, 1), W + 1), dtype=i64, device=dev[r])
        m = C.c_uint64()
        capi.check(L.bg_seqset_tail_rows_full(ctxs[r].h, shards[r].h, _ptr(short[r]), _ptr(nd), nd.shape[0], _ptr(t), cap,
                                              C.byref(m)))
        n_full[r] = m.value
        full[r] = _partition(ctxs[r], t[:m.value], capi.PART_ROW_DEST, P, None)
        del t
    del short, needs
    if _all_sum(comm, n_full):
        recv, _ = comm.all_to_all_v(full)
        for r in ranks:
            rows = recv[r].contiguous()
            capi.check(L.bg_seqset_apply_tails(ctxs[r].h, shards[r].h, _ptr(rows), rows.shape[0]))
        del recv
    del full
    extra_stats["tails_sent_in_full"] = _all_sum(comm, n_full)
    part_n = [row[0] for row in comm.gather_ints({r: [int(shards[r].size())] for r in ranks})]
    n_entries = sum(part_n)
    stats.update({"n_entries": n_entries, "max_len": max_len, "shard_entries": part_n})
    stats.update(extra_stats)
    rm_cols = {}
    if readmap is not None:  # (packed corrected reads per rank, paired): the sample's readmap while the keys are resident
        rm_cols = sharded_readmap(comm, ctxs, shards, part_n, readmap[0], readmap[1], W, dev, ph)
        stats["n_read_ids"] = rm_cols[ranks[0]].n_total
    entry_rows = {}
    if keep_entry_rows:  # the shard's entries as rows (W key words + length): what sharded_merge takes
        for r in ranks:
            n = shards[r].size()
            t = torch.empty((n, W + 1), dtype=i64, device=dev[r])
            capi.check(L.bg_seqset_entry_rows(ctxs[r].h, shards[r].h, _ptr(t)))
            entry_rows[r] = t
    out = {}
    if not gather:
        ph.mark("slices")
        sl = _aligned_slices(comm, ctxs, shards, part_n, max_len, dev)
        for r in ranks:
            shards[r].free()
            sl[r].entry_rows = entry_rows.get(r)
            sl[r].input_marks = (marks[r], sum(part_n[:r])) if mark_inputs else None
            sl[r].readmap_columns = rm_cols.get(r)
            out[r] = (sl[r], stats)
        ph.done()
        return out

    ph.mark("slice_export")
    sl_sizes, sl_shared, sl_bits = {}, {}, {}
    for r in ranks:
        n = shards[r].size()
        nw = (n + 63) // 64
        sl_sizes[r] = torch.empty((n,), dtype=torch.uint8, device=dev[r])
        sl_shared[r] = torch.empty((n,), dtype=torch.uint8, device=dev[r])
        sl_bits[r] = torch.empty((4 * nw,), dtype=i64, device=dev[r])
        capi.check(L.bg_seqset_slice_export(ctxs[r].h, shards[r].h, _ptr(sl_sizes[r]), _ptr(sl_shared[r]),
                                            _ptr(sl_bits[r])))
        shards[r].free()

    ph.mark("gather_assemble")
    _relieve(ctxs, dev)
    # ---- small outputs / tests: concatenation in rank order + finalize on rank 0
    g_sizes = comm.all_gather_v(sl_sizes)
    g_shared = comm.all_gather_v(sl_shared)
    g_bits = comm.all_gather_v(sl_bits)
    for r in ranks:
        if r != 0:
            out[r] = (None, stats)
            continue
        sz = torch.cat([g.to(dev[r]) for g in g_sizes[r]]).contiguous()
        sh = torch.cat([g.to(dev[r]) for g in g_shared[r]]).contiguous()
        bt = torch.cat([g.to(dev[r]) for g in g_bits[r]]).contiguous()
        pn = (C.c_uint64 * P)(*part_n)
        h = C.c_void_p()
        capi.check(L.bg_seqset_assemble(ctxs[r].h, P, pn, _ptr(sz), _ptr(sh), _ptr(bt), max_len, C.byref(h)))
        out[r] = (bg.Seqset(ctxs[r], h), stats)
        out[r][0].readmap_columns = rm_cols.get(r)
    if rm_cols:
        stats["readmap_columns"] = rm_cols  # gather=True: the other ranks have no seqset object to hang theirs on
    ph.done()
    return out


def sharded_merge(comm, ctxs, entry_rows, max_len, gather=False):
    """`biograph merge` of several samples' seqsets sharded over the ranks (SURVEY.md section 8e, configs[4];
    biograph_merge.cpp:189-290): the merged entry set is the prefix-free maxima of the union of the inputs' entries, so
    the inputs' entries (entry_rows[r] = list of row tensors [n, W+1], one per input: the slices' entry_rows of
    sharded_create(keep_entry_rows=True)) go through the records exchange with common splitters and the back half of
    the build.  Returns what sharded_create returns; with gather=False every slice also carries input_marks (see
    _records_to_output): which merged entries are the images of each input's entries, what the readmaps of the inputs
    need to move onto the merged seqset (population.gathered_readmap)."""
    if len(entry_rows[comm.local_ranks[0]]) > 256:
        raise ValueError("sharded_merge: more than 256 inputs")
    L = capi.lib()
    P = comm.world
    ranks = comm.local_ranks
    dev = {r: entry_rows[r][0].device for r in ranks}
    ph = _Phases(ctxs[ranks[0]])
    ph.mark("merge_partition")
    rw = max(int(t.shape[1]) for r in ranks for t in entry_rows[r])
    rw = max(row[0] for row in comm.gather_ints({r: [rw] for r in ranks}))  # seqset_merger.cpp:57-60: the longest input
    W = rw - 1

    def widen(t):  # an input of shorter entries: zero key words before the length word
        if t.shape[1] == rw:
            return t
        pad = torch.zeros((t.shape[0], rw - t.shape[1]), dtype=t.dtype, device=t.device)
        return torch.cat([t[:, :-1], pad, t[:, -1:]], dim=1)

    hists, rows = {}, {}
    for r in ranks:
        rows[r] = torch.cat([widen(t) for t in entry_rows[r]], dim=0).contiguous()
        at = 0
        for i, t in enumerate(entry_rows[r]):  # the input's number rides in the length word (whose low byte is read)
            rows[r][at:at + t.shape[0], -1] |= i << 8
            at += t.shape[0]
        hists[r] = torch.empty((4 ** PREFIX_BASES,), dtype=torch.int64, device=dev[r])
        capi.check(L.bg_rows_prefix_hist(ctxs[r].h, _ptr(rows[r]), rows[r].shape[0], rw, PREFIX_BASES, PREFIX_SAMPLE, _ptr(hists[r])))
    gh = comm.all_reduce_sum(hists)[ranks[0]].cpu().numpy()
    pparams = [PREFIX_BASES] + choose_splitters(gh, P)
    send = {r: _partition(ctxs[r], rows[r], capi.PART_PREFIX, P, pparams) for r in ranks}
    n_in = _all_sum(comm, {r: rows[r].shape[0] for r in ranks})
    del rows, hists
    ph.mark("merge_all_to_all")
    recv, _ = comm.all_to_all_v(send)
    del send
    stats = {"n_input_entries": n_in, "n_records": n_in, "mode": "merge"}
    return _records_to_output(comm, ctxs, recv, W, max_len, dev, ph, gather, stats,
                              mark_inputs=0 if gather else len(entry_rows[ranks[0]]))


# ------------------------------------------------------------------------------------------------
# The same build as ONE C call per rank: bg_create_sharded over bg_comm (csrc/sharded.hip, csrc/comm.hip) -- what
# `bgbinary create --gpus N` runs.  sharded_create above is its twin for simulated ranks (LocalSimComm: several ranks in
# one process on one GPU) and for the merge / population flows.

class NativeComm:
    """bg_comm: rank `rank` of `world` on the context's device.  backend "rccl": `ident` = the 128 bytes of
    NativeComm.unique_id() drawn on one rank and handed to the others by the launcher; backend "shm": `ident` = a short
    name shared by the ranks (several processes on one GPU)."""

    def __init__(self, ctx, rank, world, ident, backend="rccl"):
        self.ctx, self.rank, self.world, self.backend = ctx, rank, world, backend
        self.h = C.c_void_p()
        if isinstance(ident, str):
            ident = ident.encode()
        buf = C.create_string_buffer(bytes(ident), 128)
        capi.check(capi.lib().bg_comm_init(ctx.h, rank, world, buf, backend.encode(), C.byref(self.h)))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        capi.check(capi.lib().bg_comm_unique_id(buf))
        return buf.raw

    def stats(self):
        st = capi.CommStats()
        capi.check(capi.lib().bg_comm_get_stats(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def barrier(self):
        capi.check(capi.lib().bg_comm_barrier(self.h))

    def free(self):
        if self.h:
            capi.lib().bg_comm_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class NativeSharded:
    """bg_sharded: one rank's 512-aligned slice of the seqset and its stretch of the readmap's columns."""

    def __init__(self, ctx, h, stats):
        self.ctx, self.h, self.stats = ctx, h, stats

    def write(self, comm, seqset_path, readmap_path=None, argv=()):
        """All ranks: the seqset container written in place, the readmap (when built) gathered on and written by rank 0.
        -> (seqset uuid, readmap sha1 or "")."""
        av = (C.c_char_p * max(len(argv), 1))(*[a.encode() for a in argv])
        uu = C.create_string_buffer(40)
        sha = C.create_string_buffer(44)
        capi.check(capi.lib().bg_sharded_write(self.ctx.h, comm.h, self.h, seqset_path.encode(),
                                               readmap_path.encode() if readmap_path else None, len(argv), av, uu, sha))
        return uu.value.decode(), sha.value.decode()

    def free(self):
        if self.h:
            capi.lib().bg_sharded_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def native_create(ctx, comm, d_seq, d_offs, n_reads, params=None, readmap=None):
    """bg_create_sharded: this rank's share of ONE sample (device pointers as ints) -> NativeSharded (its .stats are the
    whole sample's).  readmap: None | "unpaired" | "paired"."""
    p = params or ctx.default_params()
    st = capi.ShardedStats()
    h = C.c_void_p()
    capi.check(capi.lib().bg_create_sharded(ctx.h, comm.h, C.c_void_p(d_seq), C.c_void_p(d_offs), n_reads, C.byref(p),
                                            {None: 0, "unpaired": 1, "paired": 2}[readmap], C.byref(h), C.byref(st)))
    return NativeSharded(ctx, h, st.as_dict())
