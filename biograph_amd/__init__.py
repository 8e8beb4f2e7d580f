"""biograph_amd — MI355X-native seqset construction and overlap walk (BioGraph hot path).

Host-side mirror of the reference's seams for this path:

  * `Context.create(...)`            ~ `bgbinary create` data path (SEQSETMain::run,
                                       modules/biograph/biograph_create.cpp:431-783)
  * `Context.pack_reads/kmer_count/correct_reads/build_seqset`
                                     ~ read_importer / kmerizer::run / correct_reads::correct /
                                       expander+builder (modules/build_seqset/*)
  * `Seqset`, `SeqsetEntry`          ~ `biograph.Seqset` / `biograph.SeqsetEntry`
                                       (python/biograph/seqset.cpp:46-183)

Everything computes through libbgseq.so (HIP, gfx950).  No CPU fallback exists.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import BgError, BuildStats, CorrectParams, CorrectStats, CreateParams  # noqa: F401

__all__ = ["Context", "Reads", "KmerSet", "Seqset", "SeqsetEntry", "BgError"]


def _pack_strings(strs):
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strs]
    offs = np.zeros(len(bs) + 1, np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    seq = np.frombuffer(b"".join(bs), np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return seq, offs


def _buf(a):
    """ctypes pointer of a numpy array (a 1-element dummy when empty)."""
    if a.size == 0:
        a = np.zeros(1, a.dtype)
    return a, capi.np_ptr(a)


def use_shared_arena():
    """Makes torch allocate device memory from the library's arena (one per device, shared by every Context created
    afterwards) instead of its own caching allocator: the multi-GPU driver's exchange buffers and the library's
    buffers then never starve each other.  Must run before torch touches the GPU; returns False (and changes
    nothing) when it is too late for that."""
    import torch
    try:
        alloc = torch.cuda.memory.CUDAPluggableAllocator(capi.LIB_PATH, "bg_torch_malloc", "bg_torch_free")
        torch.cuda.memory.change_current_allocator(alloc)
    except Exception:
        return False
    capi.lib().bg_torch_allocator_enable()
    return True


class DeviceArray:
    """hipMalloc'ed buffer owned through the C ABI (for hosts without their own allocator)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        capi.check(capi.lib().bg_dev_alloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, ctx, a):
        a = np.ascontiguousarray(a)
        d = cls(ctx, a.nbytes)
        if a.nbytes:
            capi.check(capi.lib().bg_dev_upload(ctx.h, d.ptr, capi.np_ptr(a), a.nbytes))
        return d

    def to_numpy(self, dtype, count):
        out = np.zeros(count, dtype)
        if out.nbytes:
            capi.check(capi.lib().bg_dev_download(self.ctx.h, capi.np_ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr and self.ctx.h:
            capi.lib().bg_dev_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream (bg_ctx)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        capi.check(capi.lib().bg_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(h)))
        self.h = h
        self.device = device
        self.stream = stream or 0  # the HIP stream every call of this context is issued on (0 = the null stream)

    def close(self):
        if self.h:
            capi.lib().bg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trim(self):
        capi.check(capi.lib().bg_ctx_trim(self.h))

    def memory(self, reset_peak=False):
        """Device bytes of this context's allocator: {'live', 'cached', 'peak'} (peak = high-water mark of live)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        capi.check(capi.lib().bg_ctx_memory(self.h, C.byref(a), C.byref(b), C.byref(c), int(bool(reset_peak))))
        return {"live": a.value, "cached": b.value, "peak": c.value}

    def set_profiling(self, on=True):
        capi.check(capi.lib().bg_ctx_set_profiling(self.h, int(bool(on))))

    def timings(self):
        n = capi.lib().bg_ctx_timings(self.h, None, 0)
        arr = (capi.Timing * max(n, 1))()
        n = capi.lib().bg_ctx_timings(self.h, arr, n)
        return [{"name": arr[i].name.decode(), "ms": arr[i].ms, "launches": int(arr[i].launches),
                 "bytes": int(arr[i].bytes)} for i in range(n)]

    # ---- staged build -------------------------------------------------------------------------
    def pack_reads(self, reads):
        """reads: list of str, or (uint8 concat, uint64 offsets) host arrays."""
        seq, offs = _pack_strings(reads) if isinstance(reads, (list, tuple)) and (
            not reads or isinstance(reads[0], (str, bytes))) else reads
        seq = np.ascontiguousarray(seq, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        h = C.c_void_p()
        _s, ps = _buf(seq)
        capi.check(capi.lib().bg_pack_reads_host(self.h, ps, capi.np_ptr(offs), len(offs) - 1, C.byref(h)))
        return Reads(self, h)

    def import_fastq(self, texts, paired_files=False, interleaved=False, cut=None, pinned=True):
        """FASTQ import on the device (bg_fastq_import): texts = the bytes of every file (plain text; mate files
        alternate when paired_files).  Returns (uint8 concatenated bases, uint64 offsets) on the host — mates are
        reads 2i, 2i+1 when paired."""
        L = capi.lib()
        bufs = []
        try:
            for t in texts:
                a = np.frombuffer(bytes(t), np.uint8)
                d = DeviceArray(self, max(a.nbytes, 1))
                if a.nbytes:
                    if pinned:
                        capi.check(L.bg_dev_upload_pinned(self.h, d.ptr, capi.np_ptr(a), a.nbytes, 0))
                    else:
                        capi.check(L.bg_dev_upload(self.h, d.ptr, capi.np_ptr(a), a.nbytes))
                bufs.append((d, a.nbytes))
            n = len(bufs)
            ptrs = (C.c_void_p * max(n, 1))(*[b[0].ptr for b in bufs])
            sizes = (C.c_uint64 * max(n, 1))(*[b[1] for b in bufs])
            d_seq, d_offs = C.c_void_p(), C.c_void_p()
            n_reads, n_bases = C.c_uint64(), C.c_uint64()
            cs, ce = cut if cut else (0, 0)
            capi.check(L.bg_fastq_import(self.h, n, ptrs, sizes, int(bool(paired_files)), int(bool(interleaved)), cs, ce,
                                         C.byref(d_seq), C.byref(d_offs), C.byref(n_reads), C.byref(n_bases)))
            try:
                seq = np.zeros(max(n_bases.value, 1), np.uint8)
                offs = np.zeros(n_reads.value + 1, np.uint64)
                if n_bases.value:
                    capi.check(L.bg_dev_download(self.h, capi.np_ptr(seq), d_seq, n_bases.value))
                capi.check(L.bg_dev_download(self.h, capi.np_ptr(offs), d_offs, offs.nbytes))
            finally:
                L.bg_dev_free(self.h, d_seq)
                L.bg_dev_free(self.h, d_offs)
            return seq[:n_bases.value], offs
        finally:
            for d, _ in bufs:
                d.free()

    def pack_reads_device(self, d_seq, d_offs, n_reads):
        h = C.c_void_p()
        capi.check(capi.lib().bg_pack_reads(self.h, C.c_void_p(d_seq), C.c_void_p(d_offs), n_reads, C.byref(h)))
        return Reads(self, h)

    def kmer_count(self, reads, k=30, min_count=5):
        h = C.c_void_p()
        capi.check(capi.lib().bg_kmer_count(self.h, reads.h, k, min_count, C.byref(h)))
        return KmerSet(self, h, k)

    def correct_reads(self, reads, kmerset, max_corrections=8, min_good_run=2, trim_after_portion=0.7):
        p = CorrectParams(max_corrections, min_good_run, trim_after_portion)
        st = CorrectStats()
        h = C.c_void_p()
        capi.check(capi.lib().bg_correct_reads(self.h, reads.h, kmerset.h, C.byref(p), C.byref(h), C.byref(st)))
        return Reads(self, h), {n: int(getattr(st, n)) for n, _ in st._fields_}

    def build_seqset(self, corrected, cover_k=30):
        st = BuildStats()
        h = C.c_void_p()
        capi.check(capi.lib().bg_seqset_build(self.h, corrected.h, cover_k, C.byref(h), C.byref(st)))
        return Seqset(self, h), st.as_dict()

    def build_readmap(self, seqset, corrected, is_paired=True, marks=None):
        """make_readmap::create_from_reads (bio_mapred/make_readmap.cpp:120-392) on the GPU.  marks (device pointer or
        a torch uint8 tensor, one byte per entry): `seqset` is a merged seqset and the result is the sample's readmap
        as create_from_fast_migrate would give it (bg_readmap_build_marked)."""
        h = C.c_void_p()
        mp = None if marks is None else C.c_void_p(marks if isinstance(marks, int) else marks.data_ptr())
        capi.check(capi.lib().bg_readmap_build_marked(self.h, seqset.h, corrected.h, int(bool(is_paired)), mp, C.byref(h)))
        return Readmap(self, h)

    # ---- whole create path --------------------------------------------------------------------
    def default_params(self, **kw):
        p = CreateParams()
        capi.lib().bg_create_params_default(C.byref(p))
        for k, v in kw.items():
            if k in ("max_corrections", "min_good_run", "trim_after_portion"):
                setattr(p.correct, k, v)
            else:
                setattr(p, k, v)
        return p

    def create_device(self, d_seq, d_offs, n_reads, params=None):
        """Whole create data path on device-resident ASCII reads (device pointers as ints)."""
        p = params or self.default_params()
        st = BuildStats()
        h = C.c_void_p()
        capi.check(capi.lib().bg_create(self.h, C.c_void_p(d_seq), C.c_void_p(d_offs), n_reads, C.byref(p),
                                        C.byref(h), C.byref(st)))
        return Seqset(self, h), st.as_dict()

    def create(self, reads, **kw):
        """reads: list of str (host).  Returns (Seqset, stats)."""
        seq, offs = _pack_strings(reads)
        dseq = DeviceArray.from_numpy(self, seq)
        doffs = DeviceArray.from_numpy(self, offs)
        try:
            return self.create_device(dseq.ptr, doffs.ptr, len(reads), self.default_params(**kw))
        finally:
            dseq.free()
            doffs.free()

    def merge_seqsets(self, seqsets, want_maps=True):
        """`biograph merge` of resident seqsets -> (merged Seqset, [np.uint64 map per input] or None)."""
        n = len(seqsets)
        arr = (C.c_void_p * n)(*[s.h for s in seqsets])
        maps = [DeviceArray(self, max(s.size(), 1) * 8) for s in seqsets] if want_maps else None
        marr = (C.c_void_p * n)(*[m.ptr for m in maps]) if want_maps else None
        h = C.c_void_p()
        try:
            capi.check(capi.lib().bg_seqset_merge(self.h, n, arr, C.byref(h), marr))
            out = [m.to_numpy(np.uint64, s.size()) for m, s in zip(maps, seqsets)] if want_maps else None
        finally:
            for m in maps or []:
                m.free()
        return Seqset(self, h), out

    def migrate_readmap(self, readmap, entry_map, merged):
        """make_readmap::create_from_fast_migrate: the sample's readmap on the merged seqset."""
        d = DeviceArray.from_numpy(self, np.ascontiguousarray(entry_map, np.uint64))
        h = C.c_void_p()
        try:
            capi.check(capi.lib().bg_readmap_migrate(self.h, readmap.h, C.c_void_p(d.ptr), merged.h, C.byref(h)))
        finally:
            d.free()
        return Readmap(self, h)

    def open_readmap(self, path):
        h = C.c_void_p()
        capi.check(capi.lib().bg_readmap_open(self.h, str(path).encode(), C.byref(h)))
        return Readmap(self, h)

    def readmap_from_arrays(self, n_entries, max_read_len, seqset_uuid, entry_id, read_len, mate_loop_ptr, is_forward):
        """bg_readmap_from_arrays: a readmap from its per-read columns (torch DEVICE tensors: uint64-as-int64 entry ids,
        int32 lengths and loop pointers, uint8 orientation; or numpy arrays, which are uploaded)."""
        import torch
        dev = "cuda:%d" % self.device

        def dev_t(x, dt):
            if isinstance(x, torch.Tensor):
                return x.to(device=dev, dtype=dt).contiguous()
            return torch.from_numpy(np.ascontiguousarray(x).astype({torch.int64: np.int64, torch.int32: np.int32, torch.uint8: np.uint8}[dt],
                                                                       copy=False)).to(dev)
        e, rl = dev_t(entry_id, torch.int64), dev_t(read_len, torch.int32)
        ml, f = dev_t(mate_loop_ptr, torch.int32), dev_t(is_forward, torch.uint8)
        n = int(e.shape[0])
        torch.cuda.current_stream(dev).synchronize()
        h = C.c_void_p()
        capi.check(capi.lib().bg_readmap_from_arrays(self.h, int(n_entries), int(max_read_len), str(seqset_uuid).encode(), n,
                                                     e.data_ptr(), rl.data_ptr(), ml.data_ptr(), f.data_ptr(), C.byref(h)))
        return Readmap(self, h)

    def open_seqset(self, path):
        h = C.c_void_p()
        capi.check(capi.lib().bg_seqset_open(self.h, str(path).encode(), C.byref(h)))
        return Seqset(self, h)

    def seqset_from_arrays(self, sizes, shared, bits):
        sizes = np.ascontiguousarray(sizes, np.uint32)
        shared = np.ascontiguousarray(shared, np.uint32)
        bb = [_buf(np.ascontiguousarray(b, np.uint64)) for b in bits]
        arr = (C.c_void_p * 4)(*[p for _, p in bb])
        h = C.c_void_p()
        _a, ps = _buf(sizes)
        _b, psh = _buf(shared)
        capi.check(capi.lib().bg_seqset_from_arrays(self.h, len(sizes), ps, psh, arr, C.byref(h)))
        return Seqset(self, h)


class _Handle:
    _free = None

    def __init__(self, ctx, h):
        self.ctx = ctx
        self.h = h

    def free(self):
        # objects must be freed before their context (bgseq.h); one that outlives it is simply dropped
        if self.h and self.ctx is not None and self.ctx.h:
            getattr(capi.lib(), self._free)(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Reads(_Handle):
    _free = "bg_reads_free"

    def __len__(self):
        return capi.lib().bg_reads_count(self.h)

    @property
    def total_bases(self):
        return capi.lib().bg_reads_total_bases(self.h)

    @property
    def max_len(self):
        return capi.lib().bg_reads_max_len(self.h)

    def download(self):
        """-> (uint8 ASCII concat, uint64 offsets); dropped reads have length 0."""
        n = len(self)
        offs = np.zeros(n + 1, np.uint64)
        seq = np.zeros(max(int(self.total_bases), 1), np.uint8)
        capi.check(capi.lib().bg_reads_download(self.ctx.h, self.h, capi.np_ptr(seq), capi.np_ptr(offs)))
        return seq[:int(offs[-1])], offs


class KmerSet(_Handle):
    _free = "bg_kmerset_free"

    def __init__(self, ctx, h, k):
        super().__init__(ctx, h)
        self.k = k

    def __len__(self):
        return capi.lib().bg_kmerset_size(self.h)

    @property
    def distinct(self):
        return capi.lib().bg_kmerset_distinct(self.h)

    @property
    def instances(self):
        return capi.lib().bg_kmerset_instances(self.h)

    def download(self):
        """-> (sorted canonical kmers u64, flags u8, total counts u32)."""
        n = len(self)
        k = np.zeros(max(n, 1), np.uint64)
        f = np.zeros(max(n, 1), np.uint8)
        c = np.zeros(max(n, 1), np.uint32)
        capi.check(capi.lib().bg_kmerset_download(self.ctx.h, self.h, capi.np_ptr(k), capi.np_ptr(f), capi.np_ptr(c)))
        return k[:n], f[:n], c[:n]


PART_NAMES = ["fixed", "entry_sizes/elements", "shared/elements"] + [
    "prev_%s/%s" % (b, w) for b in "ACGT" for w in ("bits", "subaccum", "accum")]


READMAP_PART_NAMES = ["read_ids/%s/%s" % (s, w) for s in ("source_to_mid", "dest_to_mid")
                      for w in ("bits", "subaccum", "accum")] + [
    "read_lengths/elements", "mate_loop_ptr/elements", "is_forward/packed_data"]


class Readmap(_Handle):
    """Resident readmap of one sample (modules/bio_base/readmap.h); built by Context.build_readmap."""
    _free = "bg_readmap_free"

    def size(self):
        return capi.lib().bg_readmap_size(self.h)

    __len__ = size

    @property
    def max_read_len(self):
        return capi.lib().bg_readmap_max_read_len(self.h)

    def part(self, name):
        sz = C.c_uint64(0)
        capi.check(capi.lib().bg_readmap_part(self.ctx.h, self.h, name.encode(), None, C.byref(sz)))
        buf = np.zeros(max(sz.value, 1), np.uint8)
        capi.check(capi.lib().bg_readmap_part(self.ctx.h, self.h, name.encode(), capi.np_ptr(buf), C.byref(sz)))
        return buf[:sz.value].tobytes()

    def parts(self):
        return {n: self.part(n) for n in READMAP_PART_NAMES}

    def arrays(self):
        """(entry id, read length, mate loop pointer, is_forward) per read id."""
        n = self.size()
        e = np.zeros(max(n, 1), np.uint64)
        rl, ml = (np.zeros(max(n, 1), np.uint32) for _ in range(2))
        f = np.zeros(max(n, 1), np.uint8)
        capi.check(capi.lib().bg_readmap_arrays(self.ctx.h, self.h, capi.np_ptr(e), capi.np_ptr(rl), capi.np_ptr(ml),
                                                capi.np_ptr(f)))
        return e[:n], rl[:n], ml[:n], f[:n]

    def write(self, path, argv=()):
        """Writes the container and returns its SHA-1 (the name it takes under coverage/)."""
        arr = (C.c_char_p * max(len(argv), 1))(*[a.encode() for a in argv])
        sha = C.create_string_buffer(41)
        capi.check(capi.lib().bg_readmap_write(self.ctx.h, self.h, str(path).encode(), len(argv), arr, sha))
        return sha.value.decode()


class Seqset(_Handle):
    """Resident seqset (modules/bio_base/seqset.h:35-128)."""
    _free = "bg_seqset_free"

    def size(self):
        return capi.lib().bg_seqset_size(self.h)

    __len__ = size

    @property
    def max_len(self):
        return capi.lib().bg_seqset_max_len(self.h)

    def part(self, name):
        sz = C.c_uint64(0)
        capi.check(capi.lib().bg_seqset_part(self.ctx.h, self.h, name.encode(), None, C.byref(sz)))
        buf = np.zeros(max(sz.value, 1), np.uint8)
        capi.check(capi.lib().bg_seqset_part(self.ctx.h, self.h, name.encode(), capi.np_ptr(buf), C.byref(sz)))
        return buf[:sz.value].tobytes()

    def parts(self):
        """Payload parts keyed like the spiral zip members (everything but the JSON members)."""
        return {n: self.part(n) for n in PART_NAMES}

    def arrays(self):
        n = self.size()
        s = np.zeros(max(n, 1), np.uint32)
        sh = np.zeros(max(n, 1), np.uint32)
        capi.check(capi.lib().bg_seqset_arrays(self.ctx.h, self.h, capi.np_ptr(s), capi.np_ptr(sh)))
        return s[:n], sh[:n]

    def entries(self, first=0, count=None):
        """Entry sequences as list of str."""
        n = self.size()
        count = n - first if count is None else count
        offs = np.zeros(count + 1, np.uint64)
        capi.check(capi.lib().bg_seqset_entries(self.ctx.h, self.h, first, count, None, capi.np_ptr(offs)))
        seq = np.zeros(max(int(offs[-1]), 1), np.uint8)
        capi.check(capi.lib().bg_seqset_entries(self.ctx.h, self.h, first, count, capi.np_ptr(seq), capi.np_ptr(offs)))
        b = seq.tobytes()
        return [b[int(offs[i]):int(offs[i + 1])].decode() for i in range(count)]

    def write(self, path, argv=()):
        arr = (C.c_char_p * max(len(argv), 1))(*[a.encode() for a in argv])
        capi.check(capi.lib().bg_seqset_write(self.ctx.h, self.h, str(path).encode(), len(argv), arr))

    # ---- batched walk (host numpy in/out; the C ABI itself takes device pointers) ---------------
    def _walk(self, fn, begin, end, size, base, min_ctx=None):
        ctx = self.ctx
        n = len(begin)
        ins = [DeviceArray.from_numpy(ctx, np.ascontiguousarray(begin, np.uint64)),
               DeviceArray.from_numpy(ctx, np.ascontiguousarray(end, np.uint64)),
               DeviceArray.from_numpy(ctx, np.ascontiguousarray(size, np.uint32)),
               DeviceArray.from_numpy(ctx, np.ascontiguousarray(base, np.uint8))]
        if min_ctx is not None:
            ins.append(DeviceArray.from_numpy(ctx, np.ascontiguousarray(min_ctx, np.uint32)))
        outs = [DeviceArray(ctx, n * 8), DeviceArray(ctx, n * 8), DeviceArray(ctx, n * 4)]
        try:
            capi.check(fn(ctx.h, self.h, n, *[C.c_void_p(a.ptr) for a in ins + outs]))
            return (outs[0].to_numpy(np.uint64, n), outs[1].to_numpy(np.uint64, n), outs[2].to_numpy(np.uint32, n))
        finally:
            for a in ins + outs:
                a.free()

    def push_front_batch(self, begin, end, size, base):
        return self._walk(capi.lib().bg_push_front_batch, begin, end, size, base)

    def push_front_drop_batch(self, begin, end, size, base, min_ctx):
        return self._walk(capi.lib().bg_push_front_drop_batch, begin, end, size, base, min_ctx)

    def pop_front_batch(self, begin, end, size):
        """seqset_range::pop_front for n ranges -> (begin, end, size) of the sequences without their first base."""
        ctx = self.ctx
        n = len(begin)
        ins = [DeviceArray.from_numpy(ctx, np.ascontiguousarray(begin, np.uint64)),
               DeviceArray.from_numpy(ctx, np.ascontiguousarray(end, np.uint64)),
               DeviceArray.from_numpy(ctx, np.ascontiguousarray(size, np.uint32))]
        outs = [DeviceArray(ctx, n * 8), DeviceArray(ctx, n * 8), DeviceArray(ctx, n * 4)]
        try:
            capi.check(capi.lib().bg_pop_front_batch(ctx.h, self.h, n, *[C.c_void_p(a.ptr) for a in ins + outs]))
            return (outs[0].to_numpy(np.uint64, n), outs[1].to_numpy(np.uint64, n), outs[2].to_numpy(np.uint32, n))
        finally:
            for a in ins + outs:
                a.free()

    def find_batch(self, seqs):
        seq, offs = _pack_strings(seqs)
        ctx = self.ctx
        n = len(seqs)
        dseq = DeviceArray.from_numpy(ctx, seq)
        doffs = DeviceArray.from_numpy(ctx, offs)
        ob, oe = DeviceArray(ctx, n * 8), DeviceArray(ctx, n * 8)
        try:
            capi.check(capi.lib().bg_find_batch(ctx.h, self.h, n, C.c_void_p(dseq.ptr), C.c_void_p(doffs.ptr),
                                                C.c_void_p(ob.ptr), C.c_void_p(oe.ptr)))
            return ob.to_numpy(np.uint64, n), oe.to_numpy(np.uint64, n)
        finally:
            for a in (dseq, doffs, ob, oe):
                a.free()

    # ---- frontier walks (graph_discover rows G2, G3; ref_map G7) --------------------------------------
    @staticmethod
    def _flatten_frontiers(frontiers):
        foff = np.zeros(len(frontiers) + 1, np.uint32)
        foff[1:] = np.cumsum([len(f) for f in frontiers])
        flat = [t for f in frontiers for t in f]
        b = np.array([t[0] for t in flat], np.uint64)
        e = np.array([t[1] for t in flat], np.uint64)
        s = np.array([t[2] for t in flat], np.uint32)
        return foff, b, e, s

    def trace_one_base_batch(self, frontiers, min_ctx):
        """frontiers: list of lists of (begin,end,size); min_ctx: per frontier.
        -> list of (best rc base or -1, [(range, dropped)], [next ranges])."""
        ctx, n = self.ctx, len(frontiers)
        foff, b, e, s = self._flatten_frontiers(frontiers)
        D = DeviceArray
        ins = [D.from_numpy(ctx, foff), D.from_numpy(ctx, b), D.from_numpy(ctx, e), D.from_numpy(ctx, s),
               D.from_numpy(ctx, np.ascontiguousarray(min_ctx, np.uint32))]
        outs = [D(ctx, n), D(ctx, n), D(ctx, n * 128), D(ctx, n * 128), D(ctx, n * 64), D(ctx, n * 16), D(ctx, n),
                D(ctx, n * 128), D(ctx, n * 128), D(ctx, n * 64)]
        try:
            capi.check(capi.lib().bg_trace_one_base_batch(ctx.h, self.h, n, *[C.c_void_p(a.ptr) for a in ins + outs]))
            best = outs[0].to_numpy(np.uint8, n)
            nc = outs[1].to_numpy(np.uint8, n)
            cb, ce = outs[2].to_numpy(np.uint64, n * 16), outs[3].to_numpy(np.uint64, n * 16)
            cs, cd = outs[4].to_numpy(np.uint32, n * 16), outs[5].to_numpy(np.uint8, n * 16)
            nn = outs[6].to_numpy(np.uint8, n)
            nb, ne = outs[7].to_numpy(np.uint64, n * 16), outs[8].to_numpy(np.uint64, n * 16)
            ns = outs[9].to_numpy(np.uint32, n * 16)
        finally:
            for a in ins + outs:
                a.free()
        res = []
        for f in range(n):
            cands = [((int(cb[f * 16 + i]), int(ce[f * 16 + i]), int(cs[f * 16 + i])), bool(cd[f * 16 + i]))
                     for i in range(int(nc[f]))]
            nxt = [(int(nb[f * 16 + i]), int(ne[f * 16 + i]), int(ns[f * 16 + i])) for i in range(int(nn[f]))]
            res.append((-1 if best[f] == 255 else int(best[f]), cands, nxt))
        return res

    def branch_trace_batch(self, seqs, frontiers, min_overlap, ev_cap=256):
        """seqs: list of uint8 code arrays (0..3) -> list of (n_events, [(offset, base)], final frontier)."""
        ctx, n = self.ctx, len(seqs)
        soff = np.zeros(n + 1, np.uint64)
        soff[1:] = np.cumsum([len(x) for x in seqs])
        codes = np.concatenate([np.asarray(x, np.uint8) for x in seqs]) if n else np.zeros(0, np.uint8)
        foff, b, e, s = self._flatten_frontiers(frontiers)
        D = DeviceArray
        ins = [D.from_numpy(ctx, soff), D.from_numpy(ctx, codes), D.from_numpy(ctx, foff), D.from_numpy(ctx, b),
               D.from_numpy(ctx, e), D.from_numpy(ctx, s)]
        outs = [D(ctx, n * ev_cap * 4), D(ctx, n * ev_cap), D(ctx, n * 4), D(ctx, n), D(ctx, n * 128), D(ctx, n * 128),
                D(ctx, n * 64)]
        try:
            args = [C.c_void_p(a.ptr) for a in ins] + [min_overlap, ev_cap] + [C.c_void_p(a.ptr) for a in outs]
            capi.check(capi.lib().bg_branch_trace_batch(ctx.h, self.h, n, *args))
            eo = outs[0].to_numpy(np.uint32, n * ev_cap)
            eb = outs[1].to_numpy(np.uint8, n * ev_cap)
            nev = outs[2].to_numpy(np.uint32, n)
            no = outs[3].to_numpy(np.uint8, n)
            ob, oe = outs[4].to_numpy(np.uint64, n * 16), outs[5].to_numpy(np.uint64, n * 16)
            os_ = outs[6].to_numpy(np.uint32, n * 16)
        finally:
            for a in ins + outs:
                a.free()
        res = []
        for a in range(n):
            ev = [(int(eo[a * ev_cap + i]), int(eb[a * ev_cap + i])) for i in range(min(int(nev[a]), ev_cap))]
            cur = [(int(ob[a * 16 + i]), int(oe[a * 16 + i]), int(os_[a * 16 + i])) for i in range(int(no[a]))]
            res.append((int(nev[a]), ev, cur))
        return res

    def ref_map_build(self, ref_codes, ext_start, ext_size, chain_bases=0):
        """ref_codes: uint8 array of bases 0..3; extents = N-free stretches.  -> uint8[n] ref_map cells."""
        ctx = self.ctx
        dref = DeviceArray.from_numpy(ctx, np.ascontiguousarray(ref_codes, np.uint8))
        n = self.size()
        out = DeviceArray(ctx, max(n, 1))
        es = np.ascontiguousarray(ext_start, np.uint64)
        ez = np.ascontiguousarray(ext_size, np.uint64)
        try:
            capi.check(capi.lib().bg_ref_map_build(ctx.h, self.h, C.c_void_p(dref.ptr), len(es), capi.np_ptr(es),
                                                   capi.np_ptr(ez), chain_bases, C.c_void_p(out.ptr)))
            return out.to_numpy(np.uint8, n)
        finally:
            dref.free()
            out.free()

    def push_to_pair_trace_batch(self, frontiers, min_overlap, max_steps=512, range_cap=None, seen_cap=256):
        """Extension traces of push_to_pair_discover::on_trace -> list of (reason, [bases], [frontier per step])."""
        ctx, n = self.ctx, len(frontiers)
        range_cap = range_cap or 4 * max_steps
        foff, b, e, s = self._flatten_frontiers(frontiers)
        D = DeviceArray
        ins = [D.from_numpy(ctx, foff), D.from_numpy(ctx, b), D.from_numpy(ctx, e), D.from_numpy(ctx, s)]
        m = max(n, 1)
        outs = [D(ctx, m * 4), D(ctx, m), D(ctx, m * max_steps), D(ctx, m * max_steps), D(ctx, m * 4),
                D(ctx, m * range_cap * 8), D(ctx, m * range_cap * 8), D(ctx, m * range_cap * 4)]
        try:
            vp = C.c_void_p
            capi.check(capi.lib().bg_push_to_pair_trace_batch(ctx.h, self.h, n, *[vp(a.ptr) for a in ins], min_overlap,
                                                              max_steps, range_cap, seen_cap,
                                                              *[vp(a.ptr) for a in outs]))
            ns = outs[0].to_numpy(np.uint32, n)
            why = outs[1].to_numpy(np.uint8, n)
            bases = outs[2].to_numpy(np.uint8, n * max_steps)
            counts = outs[3].to_numpy(np.uint8, n * max_steps)
            rb, re_ = outs[5].to_numpy(np.uint64, n * range_cap), outs[6].to_numpy(np.uint64, n * range_cap)
            rs = outs[7].to_numpy(np.uint32, n * range_cap)
        finally:
            for a in ins + outs:
                a.free()
        res = []
        for a in range(n):
            k = int(ns[a])
            fr, pos = [], a * range_cap
            for t in range(k):
                c = int(counts[a * max_steps + t])
                fr.append([(int(rb[pos + i]), int(re_[pos + i]), int(rs[pos + i])) for i in range(c)])
                pos += c
            res.append((int(why[a]), [int(x) for x in bases[a * max_steps:a * max_steps + k]], fr))
        return res

    # ---- row G4: readmap queries and propagate_from_end ------------------------------------------
    def prefix_reads_batch(self, readmap, ranges, read_len_limit=0, cap=256):
        """readmap::get_prefix_reads for each (begin, end, size) -> list of read-id lists (iteration order)."""
        ctx, n = self.ctx, len(ranges)
        b = np.array([r[0] for r in ranges], np.uint64)
        e = np.array([r[1] for r in ranges], np.uint64)
        sz = np.array([r[2] for r in ranges], np.uint32)
        D = DeviceArray
        ins = [D.from_numpy(ctx, b), D.from_numpy(ctx, e), D.from_numpy(ctx, sz)]
        outs = [D(ctx, max(n * cap, 1) * 4), D(ctx, max(n, 1) * 4)]
        try:
            capi.check(capi.lib().bg_prefix_reads_batch(ctx.h, self.h, readmap.h, n, *[C.c_void_p(a.ptr) for a in ins],
                                                        read_len_limit, cap, C.c_void_p(outs[0].ptr),
                                                        C.c_void_p(outs[1].ptr)))
            ids = outs[0].to_numpy(np.uint32, n * cap)
            cnt = outs[1].to_numpy(np.uint32, n)
        finally:
            for a in ins + outs:
                a.free()
        if (cnt > cap).any():
            raise BgError(-1, "prefix_reads_batch: more than cap=%d reads for a range" % cap)
        return [[int(x) for x in ids[q * cap:q * cap + int(cnt[q])]] for q in range(n)]

    def read_info_batch(self, readmap, read_ids):
        """-> list of (has_mate, mate_rc id or None, rev_comp id, (begin, end, size) of get_seqset_entry())."""
        ctx, n = self.ctx, len(read_ids)
        D = DeviceArray
        ins = [D.from_numpy(ctx, np.ascontiguousarray(read_ids, np.uint32))]
        outs = [D(ctx, max(n, 1)), D(ctx, max(n, 1) * 4), D(ctx, max(n, 1) * 4), D(ctx, max(n, 1) * 8),
                D(ctx, max(n, 1) * 8), D(ctx, max(n, 1) * 4)]
        try:
            capi.check(capi.lib().bg_read_info_batch(ctx.h, self.h, readmap.h, n,
                                                     *[C.c_void_p(a.ptr) for a in ins + outs]))
            hm = outs[0].to_numpy(np.uint8, n)
            mrc, rc = outs[1].to_numpy(np.uint32, n), outs[2].to_numpy(np.uint32, n)
            b, e, sz = outs[3].to_numpy(np.uint64, n), outs[4].to_numpy(np.uint64, n), outs[5].to_numpy(np.uint32, n)
        finally:
            for a in ins + outs:
                a.free()
        return [(bool(hm[i]), int(mrc[i]) if hm[i] else None, int(rc[i]), (int(b[i]), int(e[i]), int(sz[i])))
                for i in range(n)]

    def propagate_from_end_batch(self, seqs, ends, readmap=None, max_pairs_per_read=20, marks=None, mate_cap=4096):
        """seqset_path::propagate_from_end for many assemblies.  seqs[a] = codes of the `seq` argument
        (pushed from its last base to its first); ends[a] = list of (begin, end, size); marks[a] = offsets
        strictly inside (0, len) whose sets are wanted.  -> list of ({offset: [ranges]}, sorted mate set)."""
        ctx, n = self.ctx, len(seqs)
        soff = np.zeros(n + 1, np.uint64)
        soff[1:] = np.cumsum([len(x) for x in seqs])
        codes = np.concatenate([np.asarray(x, np.uint8) for x in seqs]) if n else np.zeros(0, np.uint8)
        foff, b, e, s = self._flatten_frontiers(ends)
        marks = [sorted(m) for m in marks] if marks is not None else [[] for _ in range(n)]
        moff = np.zeros(n + 1, np.uint32)
        moff[1:] = np.cumsum([len(m) for m in marks])
        mflat = np.array([x for m in marks for x in m], np.uint32)
        nmk = len(mflat)
        D = DeviceArray
        ins = [D.from_numpy(ctx, soff), D.from_numpy(ctx, codes), D.from_numpy(ctx, foff), D.from_numpy(ctx, b),
               D.from_numpy(ctx, e), D.from_numpy(ctx, s)]
        mk = [D.from_numpy(ctx, moff), D.from_numpy(ctx, mflat)]
        mouts = [D(ctx, max(nmk, 1)), D(ctx, max(nmk, 1) * 128), D(ctx, max(nmk, 1) * 128), D(ctx, max(nmk, 1) * 64)]
        outs = [D(ctx, max(n, 1)), D(ctx, max(n, 1) * 128), D(ctx, max(n, 1) * 128), D(ctx, max(n, 1) * 64)]
        mates = [D(ctx, max(n, 1) * 4), D(ctx, max(n * mate_cap, 1) * 8), D(ctx, max(n * mate_cap, 1) * 8),
                 D(ctx, max(n * mate_cap, 1) * 4)]
        try:
            vp = C.c_void_p
            capi.check(capi.lib().bg_propagate_from_end_batch(
                ctx.h, self.h, readmap.h if readmap is not None else None, n, *[vp(a.ptr) for a in ins],
                max_pairs_per_read, *[vp(a.ptr) for a in mk + mouts + outs], mate_cap, *[vp(a.ptr) for a in mates]))
            mn = mouts[0].to_numpy(np.uint8, nmk)
            mb, me = mouts[1].to_numpy(np.uint64, nmk * 16), mouts[2].to_numpy(np.uint64, nmk * 16)
            ms = mouts[3].to_numpy(np.uint32, nmk * 16)
            no = outs[0].to_numpy(np.uint8, n)
            ob, oe = outs[1].to_numpy(np.uint64, n * 16), outs[2].to_numpy(np.uint64, n * 16)
            os_ = outs[3].to_numpy(np.uint32, n * 16)
            nm = mates[0].to_numpy(np.uint32, n)
            tb, te = mates[1].to_numpy(np.uint64, n * mate_cap), mates[2].to_numpy(np.uint64, n * mate_cap)
            ts = mates[3].to_numpy(np.uint32, n * mate_cap)
        finally:
            for a in ins + mk + mouts + outs + mates:
                a.free()
        if (nm > mate_cap).any():
            raise BgError(-1, "propagate_from_end_batch: more than mate_cap=%d mates for an assembly" % mate_cap)
        res = []
        for a in range(n):
            ent = {0: [(int(ob[a * 16 + i]), int(oe[a * 16 + i]), int(os_[a * 16 + i])) for i in range(int(no[a]))]}
            for k in range(int(moff[a]), int(moff[a + 1])):
                ent[int(mflat[k])] = [(int(mb[k * 16 + i]), int(me[k * 16 + i]), int(ms[k * 16 + i]))
                                      for i in range(int(mn[k]))]
            ms_set = sorted({(int(tb[a * mate_cap + i]), int(te[a * mate_cap + i]), int(ts[a * mate_cap + i]))
                             for i in range(int(nm[a]))}, key=lambda r: (r[0], -r[1], r[2]))
            res.append((ent, ms_set))
        return res

    # ---- SDK surface of python/biograph/seqset.cpp:46-160 -------------------------------------
    def empty_entry(self):
        return SeqsetEntry(self, 0, self.size(), 0)

    def find(self, seq):
        seq = str(seq)
        b, e = self.find_batch([seq])
        return SeqsetEntry(self, int(b[0]), int(e[0]), len(seq)) if e[0] > b[0] else None

    def get_entry_by_id(self, i):
        if i < 0:
            raise TypeError("entry id must be non-negative")
        if i >= self.size():
            raise RuntimeError("entry id out of range")
        s, _ = self.arrays()
        return SeqsetEntry(self, i, i + 1, int(s[i]))


_B = {"A": 0, "C": 1, "G": 2, "T": 3}


class SeqsetEntry:
    """A seqset range (seqset_range, bio_base/seqset.h:207-334); invalid results are None
    (valid_or_none, python/biograph/seqset.cpp:20-42)."""

    def __init__(self, ss, begin, end, size):
        self.ss, self.begin, self.end, self.size = ss, begin, end, size

    def __len__(self):
        return self.size

    def __eq__(self, o):
        return isinstance(o, SeqsetEntry) and (self.begin, self.end, self.size) == (o.begin, o.end, o.size)

    def get_begin_entry_id(self):
        return self.begin

    def get_end_entry_id(self):
        return self.end

    def sequence(self, n=None):
        s = self.ss.entries(self.begin, 1)[0][:self.size] if self.end > self.begin else ""
        return s if n is None else s[:n]

    def __repr__(self):
        return "<SeqsetEntry %d-%d: %s>" % (self.begin, self.end, self.sequence())

    def push_front(self, base):
        b, e, s = self.ss.push_front_batch([self.begin], [self.end], [self.size], [_B[base]])
        return SeqsetEntry(self.ss, int(b[0]), int(e[0]), int(s[0])) if e[0] > b[0] else None

    def push_front_drop(self, base, min_ctx=0):
        b, e, s = self.ss.push_front_drop_batch([self.begin], [self.end], [self.size], [_B[base]], [min_ctx])
        return SeqsetEntry(self.ss, int(b[0]), int(e[0]), int(s[0])) if e[0] > b[0] else None

    def truncate(self, n):
        if n >= self.size:
            return self
        return self.ss.find(self.sequence()[:n])


class Reference:
    """The reference the discovery walks along (python/biograph/reference.cpp `biograph.Reference`): a BioGraph
    reference directory (karyotype.json + reference.ref | reference.fasta), a bare .ref (SPR000) or a FASTA.
    Host-only; needs no Context.  scaffolds = chromosomes in order; supercontigs = their N-free stretches in
    flat order as (scaffold name, start, end) with 0-based [start, end)."""

    def __init__(self, path):
        self.path = str(path)
        self.h = C.c_void_p()
        capi.check(capi.lib().bg_reference_open(self.path.encode(), C.byref(self.h)))
        L = capi.lib()
        self.scaffolds = [L.bg_reference_scaffold_name(self.h, i).decode() for i in range(L.bg_reference_scaffold_count(self.h))]
        self.scaffold_lens = {n: int(L.bg_reference_scaffold_len(self.h, i)) for i, n in enumerate(self.scaffolds)}
        self._sc = []
        for i in range(L.bg_reference_supercontig_count(self.h)):
            s, o, n, f = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
            capi.check(L.bg_reference_supercontig(self.h, i, C.byref(s), C.byref(o), C.byref(n), C.byref(f)))
            self._sc.append((s.value, o.value, n.value, f.value))
        self.supercontigs = [(self.scaffolds[s], o, o + n) for s, o, n, f in self._sc]

    def size(self):
        return int(capi.lib().bg_reference_size(self.h))

    def flat_bases(self, pos, n):
        buf = C.create_string_buffer(max(n, 1))
        capi.check(capi.lib().bg_reference_bases(self.h, pos, n, buf))
        return buf.raw[:n].decode()

    def sequence(self, scaffold, start, end):
        """Bases [start, end) of a scaffold; the range must lie inside one supercontig (reference.cpp make_range)."""
        for (s, o, n, f) in self._sc:
            if self.scaffolds[s] == scaffold and o <= start and end <= o + n:
                return self.flat_bases(f + (start - o), end - start)
        raise ValueError("%s:[%d-%d) is not inside one supercontig" % (scaffold, start, end))

    def close(self):
        if self.h:
            capi.lib().bg_reference_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        return "<biograph_amd.Reference %s>" % self.path
