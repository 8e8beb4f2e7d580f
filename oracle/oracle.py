"""ctypes front-end of the CPU oracle (oracle/bg_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (biograph_amd/) never imports this module.

Also holds the pure-numpy readers used to decode the reference's committed golden `.bg/seqset`
containers (spiral zip; modules/io/spiral_file_mmap.cpp:372-443) into arrays.
"""
import ctypes as C
import json
import os
import subprocess
import zipfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libbgoracle.so")
    src = os.path.join(_HERE, "bg_oracle.cpp")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p
    L.orc_kmer_count.restype = vp
    L.orc_kmer_count.argtypes = [u8p, u64p, C.c_uint64, C.c_int, C.c_uint32, C.c_int]
    L.orc_kmerset_size.restype = C.c_uint64
    L.orc_kmerset_size.argtypes = [vp]
    L.orc_kmerset_get.argtypes = [vp, u64p, u8p, u32p, u32p]
    L.orc_kmerset_from_arrays.restype = vp
    L.orc_kmerset_from_arrays.argtypes = [u64p, u8p, C.c_uint64, C.c_int]
    L.orc_kmerset_free.argtypes = [vp]
    L.orc_correct_reads.restype = C.c_int
    L.orc_correct_reads.argtypes = [vp, u8p, u64p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, u8p, u64p, u8p,
                                    u64p]
    L.orc_build_entries.restype = vp
    L.orc_build_entries.argtypes = [u8p, u64p, C.c_uint64, C.c_int, C.c_int]
    L.orc_entries_from_strings.restype = vp
    L.orc_entries_from_strings.argtypes = [u8p, u64p, C.c_uint64]
    for f in ("orc_entries_count", "orc_entries_records", "orc_entries_total_bases"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [vp]
    L.orc_entries_get.argtypes = [vp, u8p, u64p]
    L.orc_entries_free.argtypes = [vp]
    L.orc_seqset_from_entries.restype = vp
    L.orc_seqset_from_entries.argtypes = [vp]
    L.orc_hardware_threads.restype = C.c_uint
    L.orc_kmer_count_mt.restype = vp
    L.orc_kmer_count_mt.argtypes = [u8p, u64p, C.c_uint64, C.c_int, C.c_uint32, C.c_int, C.c_uint]
    L.orc_correct_reads_mt.restype = C.c_int
    L.orc_correct_reads_mt.argtypes = [vp, u8p, u64p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, u8p, u64p, u8p,
                                       u64p, C.c_uint]
    L.orc_build_entries_mt.restype = vp
    L.orc_build_entries_mt.argtypes = [u8p, u64p, C.c_uint64, C.c_int, C.c_uint]
    L.orc_seqset_from_entries_mt.restype = vp
    L.orc_seqset_from_entries_mt.argtypes = [vp, C.c_uint]
    L.orc_seqset_from_arrays.restype = vp
    L.orc_seqset_from_arrays.argtypes = [C.c_uint64, u32p, u32p, u64p, u64p, u64p, u64p]
    L.orc_seqset_error.restype = C.c_char_p
    L.orc_seqset_error.argtypes = [vp]
    L.orc_seqset_size.restype = C.c_uint64
    L.orc_seqset_size.argtypes = [vp]
    L.orc_seqset_max_len.restype = C.c_uint32
    L.orc_seqset_max_len.argtypes = [vp]
    L.orc_seqset_fixed.argtypes = [vp, u64p]
    L.orc_seqset_arrays.argtypes = [vp, u32p, u32p]
    L.orc_seqset_prev.restype = C.c_uint64
    L.orc_seqset_prev.argtypes = [vp, C.c_int, C.c_int, C.c_void_p]
    L.orc_seqset_packed.restype = C.c_uint64
    L.orc_seqset_packed.argtypes = [vp, C.c_int, C.c_void_p]
    L.orc_seqset_count.restype = C.c_uint64
    L.orc_seqset_count.argtypes = [vp, C.c_int, C.c_uint64]
    L.orc_seqset_find.argtypes = [vp, C.c_char_p, C.c_uint64, u64p]
    L.orc_push_front_drop.argtypes = [vp, C.c_uint64, u64p, u64p, u32p, u8p, u32p, u64p, u64p, u32p]
    L.orc_push_front.argtypes = [vp, C.c_uint64, u64p, u64p, u32p, u8p, u64p, u64p, u32p]
    L.orc_seqset_free.argtypes = [vp]
    L.orc_trace_one_base.restype = C.c_int
    L.orc_trace_one_base.argtypes = [vp, u64p, C.c_uint64, C.c_uint32, C.c_uint64, u64p, u8p, u64p, u64p, u64p]
    L.orc_branch_trace.argtypes = [vp, u8p, C.c_uint64, u64p, C.c_uint64, C.c_uint32, C.c_uint64, u32p, u8p, u64p,
                                   u64p, u64p]
    L.orc_push_to_pair_trace.restype = C.c_int
    L.orc_push_to_pair_trace.argtypes = [vp, u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, u8p, u64p, u64p,
                                         u64p]
    L.orc_ref_map_build.argtypes = [vp, u8p, u64p, u64p, C.c_uint64, u8p]
    L.orc_readmap_build.restype = vp
    L.orc_readmap_build.argtypes = [vp, vp, u8p, u64p, C.c_uint64, u64p, C.c_uint64, C.c_int, C.c_int]
    L.orc_readmap_build_mt.restype = vp
    L.orc_readmap_build_mt.argtypes = [vp, vp, u8p, u64p, C.c_uint64, u64p, C.c_uint64, C.c_int, C.c_int, C.c_uint]
    L.orc_readmap_error.restype = C.c_char_p
    L.orc_readmap_size.restype = C.c_uint64
    L.orc_readmap_size.argtypes = [vp]
    L.orc_readmap_max_len.restype = C.c_uint32
    L.orc_readmap_max_len.argtypes = [vp]
    L.orc_readmap_get.argtypes = [vp, C.c_int, u64p]
    L.orc_readmap_bitcount.restype = C.c_uint64
    L.orc_readmap_bitcount.argtypes = [vp, C.c_int, C.c_int, C.c_void_p]
    L.orc_pack_bits.restype = C.c_uint64
    L.orc_pack_bits.argtypes = [u64p, C.c_uint64, C.c_int, C.c_void_p]
    L.orc_readmap_free.argtypes = [vp]
    L.orc_readmap_from_arrays.restype = vp
    L.orc_readmap_from_arrays.argtypes = [C.c_uint64, u64p, u64p, u64p, u64p, C.c_uint32]
    L.orc_readmap_prefix_reads.restype = C.c_uint64
    L.orc_readmap_prefix_reads.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_uint64, u64p]
    L.orc_readmap_read_info.argtypes = [vp, vp, C.c_uint64, u64p]
    L.orc_propagate_from_end.restype = C.c_uint64
    L.orc_propagate_from_end.argtypes = [vp, vp, u8p, C.c_uint64, u64p, C.c_uint64, C.c_uint32, u32p, C.c_uint64,
                                         C.c_uint64, u64p, u64p, C.c_uint64, u64p]
    L.orc_merge.restype = vp
    L.orc_merge.argtypes = [C.POINTER(vp), C.c_uint64]
    L.orc_merge_entries.restype = vp
    L.orc_merge_entries.argtypes = [vp]
    L.orc_merge_map.restype = C.c_uint64
    L.orc_merge_map.argtypes = [vp, C.c_uint64, C.c_void_p]
    L.orc_merge_free.argtypes = [vp]
    L.orc_readmap_migrate.restype = vp
    L.orc_readmap_migrate.argtypes = [vp, u64p, C.c_uint64]
    _LIB = L
    return L


# ------------------------------------------------------------------------------------------------
# helpers: reads <-> (concatenated uint8, offsets uint64)

def pack_strings(strs):
    """list[str|bytes] -> (uint8 concat, uint64 offsets[n+1])."""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strs]
    offs = np.zeros(len(bs) + 1, np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    seq = np.frombuffer(b"".join(bs), np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if seq.size == 0:
        seq = np.zeros(1, np.uint8)[:0].copy()
    return np.ascontiguousarray(seq), offs


def unpack_strings(seq, offs):
    b = seq.tobytes()
    return [b[int(offs[i]):int(offs[i + 1])].decode() for i in range(len(offs) - 1)]


def read_fastq(path):
    """Minimal 4-line FASTQ reader -> list[str] of sequences (qualities dropped,
    modules/build_seqset/read_importer.cpp:22 k_keep_quality_scores=false)."""
    out = []
    with open(path, "rb") as f:
        for i, line in enumerate(f):
            if i % 4 == 1:
                out.append(line.strip().decode())
    return out


def _pad(a):
    return a if a.size else np.zeros(1, a.dtype)


# ------------------------------------------------------------------------------------------------

class KmerSet:
    def __init__(self, h):
        self.h = h

    @classmethod
    def count(cls, seq, offs, k=30, min_count=5, keep_all=False, threads=0):
        """threads > 0: the same count on a std::thread pool (CPU baseline; same result)."""
        if threads:
            h = lib().orc_kmer_count_mt(_pad(seq), offs, len(offs) - 1, k, min_count, int(keep_all), threads)
        else:
            h = lib().orc_kmer_count(_pad(seq), offs, len(offs) - 1, k, min_count, int(keep_all))
        if not h:
            raise ValueError("invalid base in reads")
        return cls(h)

    @classmethod
    def from_arrays(cls, kmers, flags, k):
        return cls(lib().orc_kmerset_from_arrays(_pad(np.ascontiguousarray(kmers, np.uint64)),
                                                 _pad(np.ascontiguousarray(flags, np.uint8)), len(kmers), k))

    def arrays(self):
        n = lib().orc_kmerset_size(self.h)
        k = np.zeros(max(n, 1), np.uint64)
        fl = np.zeros(max(n, 1), np.uint8)
        fw = np.zeros(max(n, 1), np.uint32)
        rv = np.zeros(max(n, 1), np.uint32)
        lib().orc_kmerset_get(self.h, k, fl, fw, rv)
        return k[:n], fl[:n], fw[:n], rv[:n]

    def __len__(self):
        return lib().orc_kmerset_size(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kmerset_free(self.h)
            self.h = None


STAT_NAMES = ["corrected_reads", "corrected_bases", "failed", "modified", "modified_bases", "truncated",
              "dropped_bases", "ref_records"]


def hardware_threads():
    return int(lib().orc_hardware_threads())


def correct_reads(ks, seq, offs, max_corrections=8, min_good_run=2, trim_after_portion=0.7, threads=0):
    """-> (out_seq uint8, out_offs uint64[n+1], keep uint8[n], stats dict)."""
    n = len(offs) - 1
    out_seq = np.zeros(max(int(offs[-1]), 1), np.uint8)
    out_offs = np.zeros(n + 1, np.uint64)
    keep = np.zeros(max(n, 1), np.uint8)
    stats = np.zeros(8, np.uint64)
    if threads:
        lib().orc_correct_reads_mt(ks.h, _pad(seq), offs, n, max_corrections, min_good_run, trim_after_portion, out_seq,
                                   out_offs, keep, stats, threads)
    else:
        lib().orc_correct_reads(ks.h, _pad(seq), offs, n, max_corrections, min_good_run, trim_after_portion, out_seq,
                                out_offs, keep, stats)
    return out_seq[:int(out_offs[-1])], out_offs, keep[:n], dict(zip(STAT_NAMES, (int(x) for x in stats)))


class Entries:
    def __init__(self, h):
        self.h = h

    @classmethod
    def build(cls, seq, offs, mode="simple", cover_k=30, threads=0):
        if threads:
            return cls(lib().orc_build_entries_mt(_pad(seq), offs, len(offs) - 1, cover_k, threads))
        return cls(lib().orc_build_entries(_pad(seq), offs, len(offs) - 1, 0 if mode == "simple" else 1, cover_k))

    @classmethod
    def from_strings(cls, strs):
        seq, offs = pack_strings(strs)
        return cls(lib().orc_entries_from_strings(_pad(seq), offs, len(strs)))

    def __len__(self):
        return lib().orc_entries_count(self.h)

    @property
    def n_records(self):
        return lib().orc_entries_records(self.h)

    def get(self):
        n = len(self)
        seq = np.zeros(max(lib().orc_entries_total_bases(self.h), 1), np.uint8)
        offs = np.zeros(n + 1, np.uint64)
        lib().orc_entries_get(self.h, seq, offs)
        return seq[:int(offs[-1])], offs

    def strings(self):
        return unpack_strings(*self.get())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_entries_free(self.h)
            self.h = None


def merge_entries(entry_sets):
    """make_mergemap over whole inputs: list of Entries -> (merged Entries, [map per input])."""
    arr = (C.c_void_p * len(entry_sets))(*[e.h for e in entry_sets])
    h = lib().orc_merge(arr, len(entry_sets))
    try:
        merged = Entries(lib().orc_merge_entries(h))
        maps = []
        for i in range(len(entry_sets)):
            n = lib().orc_merge_map(h, i, None)
            m = np.zeros(max(n, 1), np.uint64)
            lib().orc_merge_map(h, i, m.ctypes.data)
            maps.append(m[:n])
    finally:
        lib().orc_merge_free(h)
    return merged, maps


class Seqset:
    """Oracle seqset: arrays + FM navigation (modules/bio_base/seqset.cpp)."""

    def __init__(self, h):
        self.h = h
        err = lib().orc_seqset_error(h)
        if err:
            raise RuntimeError(err.decode())

    @classmethod
    def from_entries(cls, entries, threads=0):
        if threads:
            return cls(lib().orc_seqset_from_entries_mt(entries.h, threads))
        return cls(lib().orc_seqset_from_entries(entries.h))

    @classmethod
    def from_arrays(cls, sizes, shared, bits):
        n = len(sizes)
        bb = [_pad(np.ascontiguousarray(b, np.uint64)) for b in bits]
        return cls(lib().orc_seqset_from_arrays(n, _pad(np.ascontiguousarray(sizes, np.uint32)),
                                                _pad(np.ascontiguousarray(shared, np.uint32)), *bb))

    def __len__(self):
        return lib().orc_seqset_size(self.h)

    @property
    def max_len(self):
        return lib().orc_seqset_max_len(self.h)

    def fixed(self):
        out = np.zeros(5, np.uint64)
        lib().orc_seqset_fixed(self.h, out)
        return out

    def arrays(self):
        n = len(self)
        s = np.zeros(max(n, 1), np.uint32)
        sh = np.zeros(max(n, 1), np.uint32)
        lib().orc_seqset_arrays(self.h, s, sh)
        return s[:n], sh[:n]

    def prev(self, b, which):
        """which in {'bits','subaccum','accum'} -> uint64 array."""
        w = {"bits": 0, "subaccum": 1, "accum": 2}[which]
        n = lib().orc_seqset_prev(self.h, b, w, None)
        out = np.zeros(max(n, 1), np.uint64)
        lib().orc_seqset_prev(self.h, b, w, out.ctypes.data)
        return out[:n]

    def packed(self, which):
        w = {"entry_sizes": 0, "shared": 1}[which]
        n = lib().orc_seqset_packed(self.h, w, None)
        out = np.zeros(max(n, 1), np.uint64)
        lib().orc_seqset_packed(self.h, w, out.ctypes.data)
        return out[:n]

    def parts(self):
        """Payload parts of the spiral container, keyed like the zip member names (format v1.1.0;
        SURVEY row F3).  file_info.json is never comparable."""
        n = len(self)
        p = {"fixed": self.fixed().tobytes(),
             "seqset.json": json.dumps({"num_entries": n}, separators=(",", ":")).encode(),
             "entry_sizes/elements": self.packed("entry_sizes").tobytes(),
             "shared/elements": self.packed("shared").tobytes()}
        for b, ch in enumerate("ACGT"):
            for w in ("bits", "subaccum", "accum"):
                p["prev_%s/%s" % (ch, w)] = self.prev(b, w).tobytes()
        return p

    def count(self, b, i):
        return lib().orc_seqset_count(self.h, b, i)

    def find(self, s):
        out = np.zeros(3, np.uint64)
        lib().orc_seqset_find(self.h, s.encode(), len(s), out)
        return int(out[0]), int(out[1]), int(out[2])

    def push_front_drop(self, begin, end, size, base, min_ctx):
        n = len(begin)
        ob = np.zeros(max(n, 1), np.uint64)
        oe = np.zeros(max(n, 1), np.uint64)
        osz = np.zeros(max(n, 1), np.uint32)
        lib().orc_push_front_drop(self.h, n, _pad(np.ascontiguousarray(begin, np.uint64)),
                                  _pad(np.ascontiguousarray(end, np.uint64)),
                                  _pad(np.ascontiguousarray(size, np.uint32)),
                                  _pad(np.ascontiguousarray(base, np.uint8)),
                                  _pad(np.ascontiguousarray(min_ctx, np.uint32)), ob, oe, osz)
        return ob[:n], oe[:n], osz[:n]

    def push_front(self, begin, end, size, base):
        n = len(begin)
        ob = np.zeros(max(n, 1), np.uint64)
        oe = np.zeros(max(n, 1), np.uint64)
        osz = np.zeros(max(n, 1), np.uint32)
        lib().orc_push_front(self.h, n, _pad(np.ascontiguousarray(begin, np.uint64)),
                             _pad(np.ascontiguousarray(end, np.uint64)), _pad(np.ascontiguousarray(size, np.uint32)),
                             _pad(np.ascontiguousarray(base, np.uint8)), ob, oe, osz)
        return ob[:n], oe[:n], osz[:n]

    # ---- walk drivers (rows G2, G3, G7)
    def trace_one_base(self, ranges, min_ctx, cap=64):
        """ranges: list of (begin,end,size).  -> (best rc base or -1, [(range, dropped)], [next ranges])."""
        tri = _pad(np.array(ranges, np.uint64).reshape(-1))
        oc = np.zeros(3 * cap, np.uint64)
        od = np.zeros(cap, np.uint8)
        on = np.zeros(3 * cap, np.uint64)
        nc = np.zeros(1, np.uint64)
        nn = np.zeros(1, np.uint64)
        best = lib().orc_trace_one_base(self.h, tri, len(ranges), min_ctx, cap, oc, od, nc, on, nn)
        cands = [((int(oc[3 * i]), int(oc[3 * i + 1]), int(oc[3 * i + 2])), bool(od[i])) for i in range(int(nc[0]))]
        nxt = [(int(on[3 * i]), int(on[3 * i + 1]), int(on[3 * i + 2])) for i in range(int(nn[0]))]
        return best, cands, nxt

    def branch_trace(self, seq_codes, ranges, min_overlap, cap=4096):
        tri = _pad(np.array(ranges, np.uint64).reshape(-1))
        seq_codes = np.ascontiguousarray(seq_codes, np.uint8)
        eo = np.zeros(cap, np.uint32)
        eb = np.zeros(cap, np.uint8)
        ne = np.zeros(1, np.uint64)
        oc = np.zeros(3 * 64, np.uint64)
        no = np.zeros(1, np.uint64)
        lib().orc_branch_trace(self.h, _pad(seq_codes), len(seq_codes), tri, len(ranges), min_overlap, cap, eo, eb, ne,
                               oc, no)
        ev = [(int(eo[i]), int(eb[i])) for i in range(min(int(ne[0]), cap))]
        cur = [(int(oc[3 * i]), int(oc[3 * i + 1]), int(oc[3 * i + 2])) for i in range(int(no[0]))]
        return int(ne[0]), ev, cur

    def push_to_pair_trace(self, ranges, min_overlap, max_steps=1000, cap=16):
        """Extension loop of push_to_pair_discover::on_trace -> (reason, [base codes], [frontier per step])."""
        tri = _pad(np.array(ranges, np.uint64).reshape(-1))
        bases = np.zeros(max(max_steps, 1), np.uint8)
        counts = np.zeros(max(max_steps, 1), np.uint64)
        out = np.zeros(max(max_steps, 1) * cap * 3, np.uint64)
        ns = np.zeros(1, np.uint64)
        reason = lib().orc_push_to_pair_trace(self.h, tri, len(ranges), min_overlap, max_steps, cap, bases, counts, out,
                                              ns)
        n = int(ns[0])
        fronts = []
        for st in range(n):
            c = int(counts[st])
            assert c <= cap
            fronts.append([tuple(int(x) for x in out[(st * cap + i) * 3:(st * cap + i) * 3 + 3]) for i in range(c)])
        return reason, [int(b) for b in bases[:n]], fronts

    def propagate_from_end(self, seq_codes, ends, readmap=None, max_pairs_per_read=20, marks=(), cap=64,
                           mate_cap=65536):
        """seqset_path::propagate_from_end (variants/assemble.cpp:1345-1418).  seq_codes = the `seq`
        argument (bases pushed from its end to its start); returns ({offset: [ranges]}, [mates])."""
        tri = _pad(np.array(ends, np.uint64).reshape(-1))
        seq_codes = np.ascontiguousarray(seq_codes, np.uint8)
        marks = np.ascontiguousarray(sorted(marks), np.uint32)
        ns = len(marks) + 2
        counts = np.zeros(ns, np.uint64)
        out = np.zeros(ns * cap * 3, np.uint64)
        mates = np.zeros(mate_cap * 3, np.uint64)
        nm = lib().orc_propagate_from_end(self.h, readmap.h if readmap is not None else None, _pad(seq_codes),
                                          len(seq_codes), tri, len(ends), max_pairs_per_read, _pad(marks), len(marks),
                                          cap, counts, out, mate_cap, mates)
        slots = [0] + [int(m) for m in marks] + [len(seq_codes)]
        ent = {}
        for i, off in enumerate(slots):
            c = int(counts[i])
            assert c <= cap
            o = out[i * cap * 3:(i * cap + c) * 3].reshape(-1, 3)
            ent[off] = [tuple(int(x) for x in row) for row in o]
        assert nm <= mate_cap
        return ent, [tuple(int(x) for x in mates[3 * i:3 * i + 3]) for i in range(int(nm))]

    def ref_map_build(self, ref_codes, ext_start, ext_size):
        out = np.zeros(max(len(self), 1), np.uint8)
        lib().orc_ref_map_build(self.h, np.ascontiguousarray(ref_codes, np.uint8),
                                np.ascontiguousarray(ext_start, np.uint64), np.ascontiguousarray(ext_size, np.uint64),
                                len(ext_start), out)
        return out[:len(self)]

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_seqset_free(self.h)
            self.h = None


def build_seqset_from_fastq_reads(reads, k=30, min_count=5, max_corrections=8, min_good_run=2,
                                  trim_after_portion=0.7, mode="simple"):
    """Whole create path on the CPU: count -> filter -> correct -> entries -> arrays.
    Returns (Seqset, Entries, info dict)."""
    seq, offs = pack_strings(reads)
    ks = KmerSet.count(seq, offs, k, min_count)
    cseq, coffs, keep, stats = correct_reads(ks, seq, offs, max_corrections, min_good_run, trim_after_portion)
    ent = Entries.build(cseq, coffs, mode, k)
    ss = Seqset.from_entries(ent)
    info = dict(stats)
    info["kept_kmers"] = len(ks)
    info["n_records"] = ent.n_records
    info["entries"] = len(ent)
    return ss, ent, info


# ------------------------------------------------------------------------------------------------
# Golden container decoding (numpy only).

def unpack_varbit(buf, n, bits):
    """modules/io/packed_varbit_vector.cpp:42-78 get(): LSB-first bit-packed ints."""
    if bits == 0:
        return np.zeros(n, np.uint32)
    raw = np.frombuffer(buf, np.uint8)
    allbits = np.unpackbits(raw, bitorder="little")[: n * bits].reshape(n, bits)
    w = (1 << np.arange(bits, dtype=np.uint64))
    return (allbits.astype(np.uint64) * w).sum(axis=1).astype(np.uint32)


def read_spiral_members(path):
    """Member name -> bytes of a spiral zip.  Members written through the mmap path carry CRC 0
    (spiral_file_mmap.cpp:429), so the CRC check of zipfile.read() must be bypassed: payloads are
    read straight from behind their local headers (method 0 = stored)."""
    import struct
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    for zi in zipfile.ZipFile(path).infolist():
        o = zi.header_offset
        assert data[o:o + 4] == b"PK\x03\x04" and zi.compress_type == 0
        nlen, elen = struct.unpack("<HH", data[o + 26:o + 30])
        start = o + 30 + nlen + elen
        out[zi.filename] = data[start:start + zi.file_size]
    return out


def load_seqset_zip(path):
    """Decode a reference-written `seqset` file (either container generation) into a dict of
    arrays: n, sizes, shared, bits[4], subaccum[4], accum[4], fixed, raw (member bytes)."""
    raw = read_spiral_members(path)
    names = set(raw)
    n = json.loads(raw["seqset.json"])["num_entries"]
    out = {"n": n, "raw": raw, "fixed": np.frombuffer(raw["fixed"], np.uint64).copy()}
    for key in ("entry_sizes", "shared"):
        if key in names:  # old layout: raw uint8 (int_map_interface.cpp detect_subpart_or_uint8_membuf)
            out["sizes" if key == "entry_sizes" else "shared"] = np.frombuffer(raw[key], np.uint8).astype(np.uint32)
        else:
            md = json.loads(raw[key + "/packed_varbit_vector.json"])
            arr = unpack_varbit(raw[key + "/elements"], md["element_count"], md["bits_per_value"])
            out["sizes" if key == "entry_sizes" else "shared"] = arr
            out[key + "_meta"] = md
    for w in ("bits", "subaccum", "accum"):
        out[w] = [np.frombuffer(raw["prev_%s/%s" % (ch, w)], np.uint64).copy() for ch in "ACGT"]
    return out


def decode_entries(d):
    """Reconstruct every entry string of a decoded seqset by FM walking (pop_front = select on
    prev_b; modules/bio_base/seqset.cpp:274-311 populate_pop_front_cache)."""
    n = d["n"]
    pf = np.zeros(n, np.int64)  # pop_front target of each entry
    first = np.zeros(n, np.uint8)
    pos = 0
    for b in range(4):
        bits = np.unpackbits(d["bits"][b].view(np.uint8), bitorder="little")[:n]
        tgt = np.flatnonzero(bits)
        pf[pos:pos + len(tgt)] = tgt
        first[pos:pos + len(tgt)] = b
        pos += len(tgt)
    assert pos == n
    sizes = d["sizes"].astype(np.int64)
    maxlen = int(sizes.max()) if n else 0
    chars = np.zeros((n, maxlen), np.uint8)
    cur = np.arange(n)
    letters = np.frombuffer(b"ACGT", np.uint8)
    for j in range(maxlen):
        chars[:, j] = letters[first[cur]]
        cur = pf[cur]
    return [chars[i, :sizes[i]].tobytes().decode() for i in range(n)]


# dna_testutil.cpp:14-33 — tseq("abc"): each char -> 'C' + 8 x {A,T by bit} + 'C'.
def tseq(s):
    out = []
    for c in s.encode():
        out.append("C" + "".join("T" if (c >> i) & 1 else "A" for i in range(8)) + "C")
    return "".join(out)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def tseq_rc(s):
    return revcomp(tseq(s))


# ------------------------------------------------------------------------------------------------
# readmap

def bits_for_value(v):
    b = 0
    while v:
        b += 1
        v >>= 1
    return b


def pack_bits(vals, bits):
    vals = np.ascontiguousarray(vals, np.uint64)
    nw = lib().orc_pack_bits(_pad(vals), len(vals), bits, None)
    out = np.zeros(max(nw, 1), np.uint64)
    lib().orc_pack_bits(_pad(vals), len(vals), bits, out.ctypes.data)
    return out[:nw]


class Readmap:
    """Oracle readmap: records = list of 1- or 2-tuples of corrected read strings."""

    def __init__(self, seqset, entries, records, is_paired=True, canonicalize=True):
        reads = [r for rec in records for r in rec]
        rec_first = np.zeros(len(records) + 1, np.uint64)
        rec_first[1:] = np.cumsum([len(rec) for rec in records])
        seq, offs = pack_strings(reads)
        self.h = lib().orc_readmap_build(seqset.h, entries.h, _pad(seq), offs, len(reads), rec_first, len(records),
                                         int(is_paired), int(canonicalize))
        if not self.h:
            raise RuntimeError(lib().orc_readmap_error().decode())
        self.n_entries = len(seqset)

    @classmethod
    def from_packed(cls, seqset, entries, seq, offs, is_paired=True, canonicalize=True, threads=1):
        """The same from packed corrected reads (bases + offsets; an empty read = dropped by the correction): the records
        are the consecutive pairs (or single reads) with their dropped members left out -- what biograph_create.cpp:880-889
        hands make_readmap -- put together with numpy instead of Python strings."""
        offs = np.ascontiguousarray(offs, np.uint64)
        lens = (offs[1:] - offs[:-1]).astype(np.int64)
        n = len(lens)
        per = 2 if is_paired else 1
        if n % per:
            raise ValueError("from_packed: odd number of reads in a paired sample")
        alive = lens > 0
        per_rec = alive.reshape(-1, per).sum(axis=1)
        rec_sizes = per_rec[per_rec > 0]
        rec_first = np.zeros(len(rec_sizes) + 1, np.uint64)
        rec_first[1:] = np.cumsum(rec_sizes)
        keep_offs = np.zeros(int(alive.sum()) + 1, np.uint64)
        keep_offs[1:] = np.cumsum(lens[alive])
        # dropped reads are empty: the bases of the kept reads are the packed bases as they are
        self = cls.__new__(cls)
        if threads != 1:  # (0 = every hardware thread) look-ups, row sort and first pass on threads; the same readmap
            self.h = lib().orc_readmap_build_mt(seqset.h, entries.h, _pad(np.ascontiguousarray(seq)), keep_offs, int(alive.sum()), rec_first,
                                                len(rec_sizes), int(is_paired), int(canonicalize), int(threads))
        else:
            self.h = lib().orc_readmap_build(seqset.h, entries.h, _pad(np.ascontiguousarray(seq)), keep_offs, int(alive.sum()), rec_first,
                                             len(rec_sizes), int(is_paired), int(canonicalize))
        if not self.h:
            raise RuntimeError(lib().orc_readmap_error().decode())
        self.n_entries = len(seqset)
        return self

    @classmethod
    def from_arrays(cls, n_entries, entry_id, read_len, mate_loop_ptr, is_forward, max_read_len):
        """A readmap read back from its container (decode_readmap) — for the query functions."""
        self = cls.__new__(cls)
        a = [np.ascontiguousarray(x, np.uint64) for x in (entry_id, read_len, mate_loop_ptr, is_forward)]
        self.h = lib().orc_readmap_from_arrays(len(a[0]), _pad(a[0]), _pad(a[1]), _pad(a[2]), _pad(a[3]),
                                               int(max_read_len))
        self.n_entries = n_entries
        return self

    def migrate(self, entry_map, n_new_entries):
        """make_readmap::create_from_fast_migrate: same read ids on the merged seqset."""
        new = Readmap.__new__(Readmap)
        m = np.ascontiguousarray(entry_map, np.uint64)
        new.h = lib().orc_readmap_migrate(self.h, _pad(m), n_new_entries)
        new.n_entries = n_new_entries
        return new

    # ---- queries (bio_base/readmap.cpp:207-290,434-447,538-649; readmap.h:249-251) ----------------
    def prefix_reads(self, seqset, rng, read_len_limit=0, cap=4096):
        """Read ids of get_prefix_reads(range) in iteration order (forward, then backward)."""
        out = np.zeros(cap, np.uint64)
        n = lib().orc_readmap_prefix_reads(self.h, seqset.h, rng[0], rng[1], rng[2], read_len_limit, cap, out)
        return [int(x) for x in out[:min(n, cap)]]

    def read_info(self, seqset, read_id):
        """(has_mate, mate_rc id or None, rev_comp id, (begin, end, size) of read.get_seqset_entry())."""
        o = np.zeros(6, np.uint64)
        lib().orc_readmap_read_info(self.h, seqset.h, read_id, o)
        return bool(o[0]), (int(o[1]) if o[0] else None), int(o[2]), (int(o[3]), int(o[4]), int(o[5]))

    def __len__(self):
        return lib().orc_readmap_size(self.h)

    @property
    def max_read_len(self):
        return lib().orc_readmap_max_len(self.h)

    def get(self, which):
        w = {"read_len": 0, "mate_loop_ptr": 1, "is_forward": 2, "entry_id": 3}[which]
        out = np.zeros(max(len(self), 1), np.uint64)
        lib().orc_readmap_get(self.h, w, out)
        return out[:len(self)]

    def bitcount(self, sel, which):
        s = {"source_to_mid": 0, "dest_to_mid": 1}[sel]
        w = {"bits": 0, "subaccum": 1, "accum": 2}[which]
        n = lib().orc_readmap_bitcount(self.h, s, w, None)
        out = np.zeros(max(n, 1), np.uint64)
        lib().orc_readmap_bitcount(self.h, s, w, out.ctypes.data)
        return out[:n]

    def parts(self):
        """Payload members of the readmap container (make_readmap.cpp:221-260, format 1.2.0)."""
        n = len(self)
        p = {}
        for sel in ("source_to_mid", "dest_to_mid"):
            for w in ("bits", "subaccum", "accum"):
                p["read_ids/%s/%s" % (sel, w)] = self.bitcount(sel, w).tobytes()
        p["read_lengths/elements"] = pack_bits(self.get("read_len"), bits_for_value(self.max_read_len)).tobytes()
        p["mate_loop_ptr/elements"] = pack_bits(self.get("mate_loop_ptr"), bits_for_value(n)).tobytes()
        p["is_forward/packed_data"] = pack_bits(self.get("is_forward"), 1).tobytes()
        return p

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_readmap_free(self.h)
            self.h = None


def decode_readmap(path, seqset_dict):
    """Decode a reference-written readmap into arrays + the records (pairs / singles of read strings) it
    was built from (reads are entry prefixes; pairing from the mate loops)."""
    m = read_spiral_members(path)
    n = seqset_dict["n"]
    nr = json.loads(m["read_ids/dest_to_mid/bitcount.json"])["nbits"]

    def bits(buf, nb):
        return np.unpackbits(np.frombuffer(buf, np.uint8), bitorder="little")[:nb]
    s2m = bits(m["read_ids/source_to_mid/bits"], n)
    d2m = bits(m["read_ids/dest_to_mid/bits"], nr)
    if "read_lengths" in m:  # readmap 1.1.0 (v3.x): raw uint8 lengths, 32-bit packed_vector loop pointers
        rl = np.frombuffer(m["read_lengths"], np.uint8).astype(np.uint64)
        if "mate_loop_ptr/packed_data" in m:
            mlp = np.frombuffer(m["mate_loop_ptr/packed_data"], np.uint32)[:nr].astype(np.uint64)
        else:  # even older: mate_pair_ptr (no loop table); the reads are still entry prefixes
            mlp = None
    else:
        rl = unpack_varbit(m["read_lengths/elements"],
                           nr, json.loads(m["read_lengths/packed_varbit_vector.json"])["bits_per_value"])
        mlp = unpack_varbit(m["mate_loop_ptr/elements"],
                            nr, json.loads(m["mate_loop_ptr/packed_varbit_vector.json"])["bits_per_value"])
    isf = bits(m["is_forward/packed_data"], nr)
    entry = np.flatnonzero(s2m)[np.cumsum(d2m) - 1]
    ent = decode_entries(seqset_dict)
    records = []
    seen = np.zeros(nr, bool)
    if mlp is None:
        return {"raw": m, "n_reads": nr, "entry": entry, "read_len": rl, "mate_loop_ptr": None, "is_forward": isf,
                "records": None, "read_seqs": [ent[int(e)][:int(l)] for e, l in zip(entry, rl)]}
    for i in range(nr):
        if seen[i] or not isf[i]:
            continue
        cyc = [i]
        j = int(mlp[i])
        while j != i:
            cyc.append(j)
            j = int(mlp[j])
        for c in cyc:
            seen[c] = True
        fw = [c for c in cyc if isf[c]]
        records.append(tuple(ent[entry[c]][:rl[c]] for c in fw))
    return {"raw": m, "n_reads": nr, "entry": entry, "read_len": rl, "mate_loop_ptr": mlp, "is_forward": isf,
            "records": records}
