"""Graph-discover pipeline steps of the reference's pybind11 generators (python/biograph/variants/graph_discover.cpp:109-133:
make_ref_assemblies, graph_trim_ref, update_rc_seqset_entries, discover_branch, discover_push_to_pair) through the C++
host drivers of libbgseq.so (csrc/discover.cpp) over the batched GPU walks.  An AssemblyList is the assembly stream the
steps pass along (variants/assemble.h:285-420), resident on the C++ side.  (tests/variants_twin.py holds the same logic in
Python; the tests require the two to agree.)"""
import numpy as np


def _omin(a, b):  # optional_aoffset min/max (assemble.cpp:1321-1339)
    return b if a is None else a if b is None else min(a, b)


def _omax(a, b):
    return b if a is None else a if b is None else max(a, b)


class Assembly:
    """The fields of variants::assembly (variants/assemble.h:285-420) this path touches."""
    __slots__ = ("left_offset", "right_offset", "seq", "matches_reference", "tags", "rc_entries", "rc_mates",
                 "assembly_id")
    _next_id = [1]

    def __init__(self, left_offset, right_offset, seq, matches_reference=False, tags=()):
        self.left_offset = left_offset      # None = optional_aoffset::none
        self.right_offset = right_offset
        self.seq = seq
        self.matches_reference = matches_reference
        self.tags = frozenset(tags)
        self.rc_entries = {}                # seqset_path::m_entries: offset -> sorted list of ranges
        self.rc_mates = set()               # seqset_path::m_mates
        self.assembly_id = Assembly._next_id[0]
        Assembly._next_id[0] += 1

    # seqset_path accessors (assemble.cpp:1425-1445)
    def rc_starts(self):
        return self.rc_entries[min(self.rc_entries)] if self.rc_entries else []

    def rc_ends(self):
        return self.rc_entries[max(self.rc_entries)] if self.rc_entries else []

    def min_offset(self):
        return _omin(self.left_offset, self.right_offset)

    def max_offset(self):
        return _omax(self.left_offset, self.right_offset)

    def __repr__(self):
        return "Assembly(%s, %r, %s%s, %s)" % (self.left_offset, self.seq if len(self.seq) < 40 else
                                              self.seq[:18] + ".." + self.seq[-18:], self.right_offset,
                                              ", ref" if self.matches_reference else "", sorted(self.tags))



class AssemblyList:
    """bg_asm_list: the assembly stream of the pipeline, resident on the C++ side."""
    NONE = -(1 << 63)

    def __init__(self, asms=()):
        import ctypes as C
        from . import capi
        self._C, self._capi = C, capi
        self.h = C.c_void_p()
        capi.check(capi.lib().bg_asm_list_create(C.byref(self.h)))
        for a in asms:
            self.add(a.left_offset, a.right_offset, a.seq, a.matches_reference, a.tags)
        if asms:
            capi.check(capi.lib().bg_asm_list_sort(self.h))

    def add(self, left, right, seq, matches_reference=False, tags=()):
        n = self.NONE
        self._capi.check(self._capi.lib().bg_asm_list_add(self.h, n if left is None else left, n if right is None else right,
                                                          seq.encode(), len(seq), int(matches_reference),
                                                          ",".join(sorted(tags)).encode()))

    def __len__(self):
        return self._capi.lib().bg_asm_list_size(self.h)

    def to_python(self, with_paths=True):
        C, L = self._C, self._capi.lib()
        out = []
        for i in range(len(self)):
            lo, ro, ln, mr = C.c_int64(), C.c_int64(), C.c_uint64(), C.c_int()
            sq, tg = C.c_char_p(), C.c_char_p()
            self._capi.check(L.bg_asm_list_get(self.h, i, C.byref(lo), C.byref(ro), C.byref(sq), C.byref(ln), C.byref(mr),
                                               C.byref(tg)))
            a = Assembly(None if lo.value == self.NONE else lo.value, None if ro.value == self.NONE else ro.value,
                         (sq.value or b"").decode(), bool(mr.value), [t for t in (tg.value or b"").decode().split(",") if t])
            if with_paths:
                no, nr = C.c_uint64(), C.c_uint64()
                self._capi.check(L.bg_asm_list_rc_path(self.h, i, C.byref(no), C.byref(nr), None, None, None))
                offs = np.zeros(max(no.value, 1), np.int64)
                cnts = np.zeros(max(no.value, 1), np.uint32)
                tri = np.zeros(max(nr.value, 1) * 3, np.uint64)
                self._capi.check(L.bg_asm_list_rc_path(self.h, i, C.byref(no), C.byref(nr), self._capi.np_ptr(offs),
                                                       self._capi.np_ptr(cnts), self._capi.np_ptr(tri)))
                p = 0
                for k in range(no.value):
                    a.rc_entries[int(offs[k])] = [tuple(int(x) for x in tri[3 * (p + j):3 * (p + j) + 3])
                                                  for j in range(int(cnts[k]))]
                    p += int(cnts[k])
            out.append(a)
        return out

    def _ext(self, extents):
        C = self._C
        n = len(extents)
        offs = (C.c_int64 * n)(*[e[0] for e in extents])
        seqs = (C.c_char_p * n)(*[e[1].encode() for e in extents])
        lens = (C.c_uint64 * n)(*[len(e[1]) for e in extents])
        return n, offs, seqs, lens

    def make_ref_assemblies(self, extents, start, end, max_chunk_size=100):
        self._capi.check(self._capi.lib().bg_make_ref_assemblies(self.h, *self._ext(extents), start, end, max_chunk_size))
        return self

    def graph_trim_ref(self, extents, max_backtrack=300):
        self._capi.check(self._capi.lib().bg_graph_trim_ref(self.h, *self._ext(extents), max_backtrack))
        return self

    def dedup(self):
        self._capi.check(self._capi.lib().bg_dedup_assemblies(self.h))
        return self

    def update_rc_seqset_entries(self, seqset, readmap=None, max_pairs_per_read=20):
        self._capi.check(self._capi.lib().bg_update_rc_seqset_entries(seqset.ctx.h, seqset.h, readmap.h if readmap else None,
                                                                      self.h, max_pairs_per_read))
        return self

    def discover_branch(self, seqset, readmap=None, min_overlap=70, tag="GRAPH_BRANCH", discover_tags=()):
        self._capi.check(self._capi.lib().bg_discover_branch(seqset.ctx.h, seqset.h, readmap.h if readmap else None, self.h,
                                                             min_overlap, tag.encode(), ",".join(discover_tags).encode()))
        return self

    def discover_push_to_pair(self, seqset, readmap=None, min_overlap=70, max_pair_distance=1000,
                              tag="GRAPH_PUSH_TO_PAIR", discover_tags=()):
        self._capi.check(self._capi.lib().bg_discover_push_to_pair(seqset.ctx.h, seqset.h, readmap.h if readmap else None,
                                                                   self.h, min_overlap, max_pair_distance, tag.encode(),
                                                                   ",".join(discover_tags).encode()))
        return self

    def free(self):
        if self.h:
            self._capi.lib().bg_asm_list_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
