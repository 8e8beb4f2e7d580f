"""TEST ONLY — the readable Python twin of the C++ host drivers (biograph_amd/csrc/discover.cpp), kept beside the tests
that require the two to agree assembly for assembly.  The product path is biograph_amd.variants.AssemblyList (C++
over the batched GPU walks); nothing in the package imports this file.

Host drivers of the graph-discover path (SURVEY §8 rows G5 and the host halves of G2-G4), mirroring the
pybind11 generators of python/biograph/variants/graph_discover.cpp:109-133:

    make_ref_assemblies, graph_trim_ref, update_rc_seqset_entries, discover_branch, discover_push_to_pair

Every walk over the seqset / readmap is a batched GPU kernel behind `Walker` (bg_propagate_from_end_batch,
bg_branch_trace_batch, bg_push_to_pair_trace_batch); this module is the bookkeeping the reference keeps on
the host: the reference-offset sweep of graph_discover (graph_discover/discover.cpp:84-150), the mate and
anchor maps of push_to_pair_discover (push_to_pair.cpp:41-157,393-526), apply_edges_step
(variants/apply_edges.cpp), graph_trim_ref (graph_trim_ref.cpp) and the canonical output order
(variants/assemble.cpp:1029-1085).  A step here takes a list of assemblies in canonical order and returns
one; the reference streams them with flush points only to bound memory — the order handed to the next
step is the same (sorted_output_pipeline_step, assemble.cpp:297-345).

The traces do not depend on the sweep's state (the mate / anchor checks only CUT a trace), so they are
computed for all assemblies of a step in one kernel launch before the sweep consumes them.
"""
import bisect

import numpy as np

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
BASES = "ACGT"
_COMP = str.maketrans("ACGT", "TGCA")
INF = float("inf")


def rev_comp(s):
    return s.translate(_COMP)[::-1]


def _codes(s):
    return np.frombuffer(s.encode(), np.uint8) if False else np.array([CODE[c] for c in s], np.uint8)


def range_key(r):  # seqset_range::operator< (bio_base/seqset.h:211-218)
    return (r[0], -r[1], r[2])


def dedup_prefixes(rs):
    """seqset_set_dedup_prefixes (variants/assemble.cpp:1473-1498): drop every range that contains another."""
    rs = sorted(set(rs), key=range_key)
    out = []
    for r in rs:
        if any(o is not r and o != r and o[0] >= r[0] and o[1] <= r[1] for o in rs):
            continue
        out.append(r)
    return out


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


def _omin(a, b):  # optional_aoffset min/max (assemble.cpp:1321-1339)
    return b if a is None else a if b is None else min(a, b)


def _omax(a, b):
    return b if a is None else a if b is None else max(a, b)


def canon_key(a):
    """canon_assembly_order, GRAPH_DISCOVER flavour (assemble.cpp:1029-1085)."""
    return (a.min_offset(), a.matches_reference, a.left_offset is None, a.right_offset is None, -a.max_offset(),
            a.seq, tuple(sorted(a.tags)))


def _canon_sorted(asms):
    return sorted(asms, key=canon_key)   # stable, like the multimap the reference queues output in


# -------------------------------------------------------------------------------------------------
class Walker:
    """What the drivers need from the seqset / readmap.  `seqset` and `readmap` are biograph_amd.Seqset /
    Readmap (kernels through the C ABI); tests pass stand-ins with the same three batch methods."""

    def __init__(self, seqset, readmap=None):
        self.ss = seqset
        self.rm = readmap
        self.n = seqset.size()
        self._shared = None

    def ctx_begin(self):
        return (0, self.n, 0)

    @property
    def shared(self):
        if self._shared is None:
            self._shared = np.asarray(self.ss.arrays()[1])
        return self._shared

    def truncate(self, r, count):  # seqset_range::truncate / pop_back (seqset.cpp:630-661)
        if r[2] <= count:
            return r
        sh, b, e = self.shared, r[0], r[1]
        while b and sh[b] >= count:
            b -= 1
        while e < self.n and sh[e] >= count:
            e += 1
        return (b, e, count)

    def shared_prefix_length(self, a, b):  # seqset.cpp:723-748
        s = min(a[2], b[2])
        if a[1] > b[0] and b[1] > a[0]:
            return s
        lo, hi = (b[1], a[0]) if a[1] > b[0] else (a[1], b[0])
        return int(min(s, int(self.shared[lo:hi + 1].min()))) if hi >= lo else s

    def propagate(self, seqs_rc, ends, marks, with_mates, max_pairs):
        return self.ss.propagate_from_end_batch([_codes(s) for s in seqs_rc], ends,
                                                self.rm if with_mates else None, max_pairs, marks)

    def branch_trace(self, seqs, fronts, min_overlap):
        return self.ss.branch_trace_batch([_codes(s) for s in seqs], fronts, min_overlap,
                                          ev_cap=max([len(s) for s in seqs] + [1]))

    def push_trace(self, fronts, min_overlap, max_steps):
        return self.ss.push_to_pair_trace_batch(fronts, min_overlap, max_steps=max_steps)


# -------------------------------------------------------------------------------------------------
def make_ref_assemblies(extents, start_offset, end_offset, max_chunk_size=100):
    """graph_discover/make_ref.cpp:9-35.  extents = [(offset, sequence)] of the scaffold."""
    out = []
    for ext_off, ext_seq in extents:
        lo, hi = max(ext_off, start_offset), min(ext_off + len(ext_seq), end_offset)
        pos = lo
        while pos < hi:
            chunk = hi - pos
            if max_chunk_size and chunk > max_chunk_size:
                chunk = max_chunk_size
            out.append(Assembly(pos, pos + chunk, ext_seq[pos - ext_off:pos - ext_off + chunk], True, ["MAKE_REF"]))
            pos += chunk
    return out


def _split_extent_at(extents, start):  # scaffold::split_extent_at (variants/scaffold.cpp:189-203)
    for off, seq in extents:
        if off > start:
            break
        if off + len(seq) < start:
            continue
        return seq[:start - off], seq[start - off:]
    return "", ""


def _shared_prefix(a, b):
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    return i


def graph_trim_ref(asms, extents, max_backtrack=300):
    """graph_discover/graph_trim_ref.cpp: trim the reference-matching ends off non-reference assemblies and
    cut reference assemblies where the trimmed ones attach."""
    out = []
    ref_asms = []
    ref_stops = set()

    def split_and_output_ref(a):
        for stop in sorted(ref_stops):
            if stop <= a.left_offset:
                continue
            if stop >= a.right_offset:
                break
            k = stop - a.left_offset
            left = Assembly(a.left_offset, stop, a.seq[:k], True, a.tags)
            out.append(left)
            a = Assembly(stop, a.right_offset, a.seq[k:], True, a.tags)
        out.append(a)

    def advance_to(offset):
        while ref_asms and ref_asms[0].right_offset <= offset:
            split_and_output_ref(ref_asms.pop(0))
        # m_ref_stops below the flush point are dropped in the reference; harmless to keep

    for a in asms:
        advance_to(a.min_offset() - max_backtrack)
        if a.matches_reference:
            ref_asms.append(a)
            continue
        shared_left = shared_right = 0
        if a.left_offset is not None:
            shared_left = _shared_prefix(_split_extent_at(extents, a.left_offset)[1], a.seq)
        if a.right_offset is not None:
            shared_right = _shared_prefix(rev_comp(_split_extent_at(extents, a.right_offset)[0]), rev_comp(a.seq))
        max_anchor = len(a.seq)
        if a.left_offset is not None and a.right_offset is not None:
            max_anchor = min(max_anchor, a.right_offset - a.left_offset)
        if a.left_offset is None:
            max_anchor = min(max_anchor, max_backtrack)
        shared_right = min(shared_right, max_anchor)
        if shared_right + shared_left > max_anchor:
            shared_left = max_anchor - shared_right
        lo = None if a.left_offset is None else a.left_offset + shared_left
        ro = None if a.right_offset is None else a.right_offset - shared_right
        seq = a.seq[shared_left:len(a.seq) - shared_right]
        if len(seq) == 0 and (lo == ro or lo is None or ro is None):
            continue   # entirely reference
        b = Assembly(lo, ro, seq, False, a.tags)
        if lo is not None:
            ref_stops.add(lo)
        if ro is not None:
            ref_stops.add(ro)
        out.append(b)
    advance_to(INF)
    return _canon_sorted(out)


def dedup_assemblies(asms):
    """The dedup of graph_discover_test.py:23-35 (same offsets and sequence, adjacent in canonical order)."""
    out = []
    for a in asms:
        if out and out[-1].left_offset == a.left_offset and out[-1].right_offset == a.right_offset and \
                out[-1].seq == a.seq:
            continue
        out.append(a)
    return out


# -------------------------------------------------------------------------------------------------
def apply_edges(asms, on_edges):
    """apply_edges_step (variants/apply_edges.cpp:17-125) over a canonical list: at every reference position
    call on_edges(pos, left_edges, inserts, right_edges) with the assemblies that end there, are inserted
    there and begin there."""
    out = []
    state = {"cur": -INF}
    active = []          # multimap right_offset -> assembly, kept sorted by key, stable
    cur_inserts, cur_non_inserts, cur_left_unanchored = [], [], []

    def flush_active_to_here():
        cur_left = []
        if cur_left_unanchored:
            on_edges(None, [], [], list(cur_left_unanchored))
            cur_left = list(cur_left_unanchored)
            del cur_left_unanchored[:]
        while active and active[0][0] == state["cur"]:
            cur_left.append(active.pop(0)[1])
        if cur_left or cur_inserts or cur_non_inserts:
            on_edges(state["cur"], cur_left, list(cur_inserts), list(cur_non_inserts))
        out.extend(cur_left)
        out.extend(cur_inserts)
        del cur_inserts[:]
        right_unanchored = []
        for a in cur_non_inserts:
            if a.right_offset is not None:
                keys = [k for k, _ in active]
                active.insert(bisect.bisect_right(keys, a.right_offset), (a.right_offset, a))
            else:
                right_unanchored.append(a)
        del cur_non_inserts[:]
        if right_unanchored:
            on_edges(None, right_unanchored, [], [])
            out.extend(right_unanchored)

    def advance_to(offset):
        while state["cur"] < offset:
            flush_active_to_here()
            target = offset
            if active and target > active[0][0]:
                target = active[0][0]
            state["cur"] = target

    for a in asms:
        advance_to(a.min_offset())
        if a.left_offset == a.right_offset:
            cur_inserts.append(a)
        elif a.left_offset is not None:
            cur_non_inserts.append(a)
        else:
            cur_left_unanchored.append(a)
    advance_to(INF)
    assert not active and not cur_inserts and not cur_non_inserts and not cur_left_unanchored
    return _canon_sorted(out)


def update_rc_seqset_entries(asms, walker, with_mates=False, max_pairs_per_read=20):
    """graph_discover/update_rc_seqset_entries.cpp: give every assembly its rc range path, propagating from
    the assemblies that start where it ends (seqset_path::propagate_from_end on the GPU, all targets of one
    reference position in one launch)."""
    def propagate(incoming, targets):
        todo = []
        for t in targets:
            inc = dedup_prefixes(incoming)
            if t.rc_entries and sorted(inc, key=range_key) == t.rc_ends() and len(t.seq) == max(t.rc_entries):
                continue   # "Skipping, already done" (assemble.cpp:1353-1358)
            todo.append(t)
        if not todo:
            return
        seqs = [rev_comp(t.seq) for t in todo]
        marks = [[o for o in t.rc_entries if 0 < o < len(t.seq)] for t in todo]
        res = walker.propagate(seqs, [list(incoming)] * len(todo), marks, with_mates, max_pairs_per_read)
        for t, (ent, mates) in zip(todo, res):
            new = {len(t.seq): dedup_prefixes(incoming)}
            new.update(ent)
            if len(t.seq) == 0:
                new[0] = dedup_prefixes(incoming)
            t.rc_entries = new
            t.rc_mates |= set(mates)

    def on_edges(pos, left_edges, inserts, right_edges):
        incoming = []
        for a in left_edges:
            assert a.rc_entries, "Should have already generated rc_seqset_entries for %r" % a
            incoming.extend(a.rc_starts())
        incoming = sorted(set(incoming), key=range_key)
        if not incoming:
            incoming = [walker.ctx_begin()]
        if inserts:
            propagate(incoming, inserts)
            for a in inserts:
                incoming.extend(a.rc_starts())
            incoming = sorted(set(incoming), key=range_key)
        propagate(incoming, right_edges)

    return apply_edges(asms, on_edges)


# -------------------------------------------------------------------------------------------------
class _GraphDiscover:
    """The sweep of graph_discover (graph_discover/discover.cpp:84-150): assemblies enter `read_ahead_distance`
    before the trace position (on_walk, on_readahead), leave the read-ahead when the trace passes their left
    end (on_readahead_done) and are traced when it reaches their right end (on_trace)."""

    def __init__(self, walker, min_overlap, tag, discover_tags=(), read_ahead_distance=100000,
                 max_pair_distance=1000):
        self.w = walker
        self.min_overlap = min_overlap
        self.tag = tag
        self.discover_tags = frozenset(discover_tags)
        self.read_ahead_distance = read_ahead_distance
        self.max_pair_distance = max_pair_distance
        self.out = []
        self.readahead_done = []   # multimap min offset -> act
        self.trace_pending = []    # multimap max offset -> act
        self.trace_offset = -INF

    # hooks
    def make_active(self, a):
        return {"a": a}

    def on_walk(self, act): pass
    def on_readahead(self, act): pass
    def on_readahead_done(self, act): pass
    def on_advance_trace(self, offset): pass
    def on_trace(self, act): raise NotImplementedError
    def precompute(self, asms): pass

    def wants_trace(self, a):
        return not self.discover_tags or bool(self.discover_tags & a.tags)

    @staticmethod
    def _mm_insert(mm, key, val):
        keys = [k for k, _ in mm]
        mm.insert(bisect.bisect_right(keys, key), (key, val))

    def run(self, asms):
        self.precompute(asms)
        for a in asms:
            self.advance_trace_to(a.min_offset() - self.read_ahead_distance)
            act = self.make_active(a)
            self.on_walk(act)
            self.on_readahead(act)
            self._mm_insert(self.readahead_done, a.min_offset(), act)
        self.advance_trace_to(INF)
        return _canon_sorted(self.out)

    def advance_trace_to(self, pos):
        while self.trace_offset < pos:
            self.advance_trace_towards(pos)

    def advance_trace_towards(self, pos):
        if self.trace_pending and self.trace_pending[0][0] < pos:
            pos = self.trace_pending[0][0]
        while self.readahead_done and self.readahead_done[0][0] < pos:
            act = self.readahead_done.pop(0)[1]
            self.on_readahead_done(act)
            mx = act["a"].max_offset()
            self._mm_insert(self.trace_pending, mx, act)
            if mx < pos:
                pos = mx
        self.trace_offset = pos
        self.on_advance_trace(pos)
        while self.trace_pending and self.trace_pending[0][0] == pos:
            act = self.trace_pending.pop(0)[1]
            if self.wants_trace(act["a"]):
                self.on_trace(act)
            self.out.append(act["a"])

    # discover.cpp:157-197
    def extend_right(self, act, offset, seq, rc_path=None):
        a = Assembly(act["a"].left_offset, None, act["a"].seq[:offset] + seq, False, [self.tag])
        if rc_path:
            a.rc_entries = rc_path
        return a

    def anchor(self, act, offset, seq, anc, rc_path=None):
        other = anc["act"]["a"]
        ext = seq + other.seq[anc["offset"]:]
        a = Assembly(act["a"].left_offset, other.right_offset, act["a"].seq[:offset] + ext, False, [self.tag])
        if rc_path:
            a.rc_entries = rc_path
        return a


class _BranchDiscover(_GraphDiscover):
    """graph_discover/branch.cpp:20-80."""

    def precompute(self, asms):
        todo = [a for a in asms if self.wants_trace(a) and a.left_offset is not None and len(a.seq)]
        self.events = {}
        if todo:
            res = self.w.branch_trace([a.seq for a in todo], [a.rc_ends() for a in todo], self.min_overlap)
            for a, (nev, ev, cur) in zip(todo, res):
                assert nev == len(ev)
                assert sorted(cur, key=range_key) == a.rc_starts(), (cur, a.rc_starts())   # CHECK at :77-78
                self.events[a.assembly_id] = ev

    def on_trace(self, act):
        a = act["a"]
        if a.left_offset is None:
            return
        for off, base in self.events.get(a.assembly_id, []):
            self.out.append(self.extend_right(act, off, BASES[base]))


def discover_branch(asms, walker, min_overlap=70, tag="GRAPH_BRANCH", discover_tags=()):
    return _BranchDiscover(walker, min_overlap, tag, discover_tags).run(asms)


class _PushToPair(_GraphDiscover):
    """graph_discover/push_to_pair.cpp."""
    MAX_STEPS = 512

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.mates = {}            # m_mates: range -> refcount
        self.mate_expiry = {}      # offset -> set of ranges
        self.rc_entry_anchors = {}  # truncated range -> list of (orig range, anchor) sorted by orig
        self.rc_anchor_keys = []   # sorted keys of rc_entry_anchors

    def precompute(self, asms):
        todo = [a for a in asms if self.wants_trace(a) and a.right_offset is None]
        self.traces = {}
        if todo:
            res = self.w.push_trace([a.rc_starts() for a in todo], self.min_overlap, self.MAX_STEPS)
            for a, r in zip(todo, res):
                self.traces[a.assembly_id] = r

    def make_active(self, a):
        return {"a": a, "mates": set(), "rc_anchors": []}

    def on_walk(self, act):  # :41-73
        a = act["a"]
        act["mates"] = set(a.rc_mates)
        n = len(a.seq)
        if a.right_offset is not None:
            for off in sorted(a.rc_entries):
                if off == n:
                    continue
                for r in a.rc_entries[off]:
                    if r[2] < self.min_overlap:
                        continue
                    act["rc_anchors"].append({"truncated": self.w.truncate(r, self.min_overlap), "orig": r,
                                              "anchor": {"act": act, "offset": n - off}})

    def on_advance_trace(self, offset):  # :75-103
        for exp in sorted(k for k in self.mate_expiry if k < offset):
            for m in self.mate_expiry.pop(exp):
                self.mates[m] -= 1
                if not self.mates[m]:
                    del self.mates[m]

    def on_readahead(self, act):  # :105-131
        a = act["a"]
        expiry = self.mate_expiry.setdefault(a.max_offset() + self.max_pair_distance, set())
        for info in act["rc_anchors"]:
            lst = self.rc_entry_anchors.get(info["truncated"])
            if lst is None:
                lst = self.rc_entry_anchors[info["truncated"]] = []
                bisect.insort(self.rc_anchor_keys, (range_key(info["truncated"]), info["truncated"]))
            keys = [range_key(o) for o, _ in lst]
            lst.insert(bisect.bisect_right(keys, range_key(info["orig"])), (info["orig"], info["anchor"]))
        for m in act["mates"]:
            if m not in expiry:
                expiry.add(m)
                self.mates[m] = self.mates.get(m, 0) + 1

    def on_readahead_done(self, act):  # :133-157
        if act["a"].right_offset is None:
            return
        for info in act["rc_anchors"]:
            lst = self.rc_entry_anchors[info["truncated"]]
            for i, (orig, anc) in enumerate(lst):
                if orig == info["orig"] and anc["act"] is info["anchor"]["act"] and anc["offset"] == info["anchor"]["offset"]:
                    del lst[i]
                    break
            else:
                raise AssertionError("Could not find entry anchor to delete")
            if not lst:
                del self.rc_entry_anchors[info["truncated"]]
                self.rc_anchor_keys.remove((range_key(info["truncated"]), info["truncated"]))

    # :393-457
    def get_entry_anchors(self, rs, approx_ref_offset):
        anchors = []
        best = [INF]

        def add(orig_r, r, anc):
            other = anc["act"]["a"]
            anchor_offset = other.right_offset - len(other.seq) + anc["offset"]
            svlen = abs(anchor_offset - approx_ref_offset)
            if svlen > best[0]:
                return
            best[0] = svlen
            anchors.append({"act": anc["act"], "offset": anc["offset"], "r": r,
                            "shared_bases": self.w.shared_prefix_length(r, orig_r)})

        searches = []
        for r in rs:
            if r[2] < self.min_overlap:
                continue
            i = bisect.bisect_right(self.rc_anchor_keys, (range_key(r), r))   # upper_bound(r)
            if i == 0:
                continue
            trunc = self.rc_anchor_keys[i - 1][1]
            if trunc[1] <= r[0]:
                continue
            lst = self.rc_entry_anchors[trunc]
            keys = [range_key(o) for o, _ in lst]
            f = bisect.bisect_right(keys, range_key(r))
            searches.append({"lst": lst, "rev": f, "fwd": f, "r": r})
        while searches:
            nxt = []
            for s in searches:
                more = False
                if s["rev"] != 0:
                    s["rev"] -= 1
                    add(s["r"], s["lst"][s["rev"]][0], s["lst"][s["rev"]][1])
                    more = True
                if s["fwd"] != len(s["lst"]):
                    add(s["r"], s["lst"][s["fwd"]][0], s["lst"][s["fwd"]][1])
                    s["fwd"] += 1
                    more = True
                if more:
                    nxt.append(s)
            searches = nxt
        return anchors

    @staticmethod
    def path_entries_to_rc_path(path_entries, end_pos):  # :459-470
        rc = {}
        for off, rs in path_entries.items():
            rc[end_pos - off] = dedup_prefixes(rc.get(end_pos - off, []) + list(rs))
        return rc

    def save_anchors(self, act, rs, seq, path_entries, best):  # :472-526
        a = act["a"]
        ref_offset = a.left_offset + len(a.seq) + len(seq)
        anchors = self.get_entry_anchors(rs, ref_offset)
        if not anchors:
            return 0

        def abs_svlen(anc):
            other = anc["act"]["a"]
            return abs(other.right_offset - len(other.seq) + anc["offset"] - ref_offset)

        end_pos = len(a.seq) + len(seq)
        new_rc_path = self.path_entries_to_rc_path(path_entries, end_pos)
        new_rc_path[end_pos] = dedup_prefixes(new_rc_path.get(end_pos, []) + list(rs))
        anchors.sort(key=lambda x: (-x["shared_bases"], abs_svlen(x)))   # std::sort with a strict weak order
        shared_here = 0
        for anc in anchors:
            if anc["act"]["a"].right_offset is None:
                continue
            cur = abs_svlen(anc)
            if cur < best["abs_svlen"] or anc["shared_bases"] > best["shared_bases"]:
                if cur < best["abs_svlen"]:
                    best["abs_svlen"] = cur
                    best["abs_svlen_a"] = self.anchor(act, len(a.seq), seq, anc, dict(new_rc_path))
                if anc["shared_bases"] > best["shared_bases"]:
                    if not shared_here:
                        shared_here = anc["shared_bases"]
                    best["shared_bases"] = anc["shared_bases"]
                    best["shared_bases_a"] = self.anchor(act, len(a.seq), seq, anc, dict(new_rc_path))
        return shared_here

    def on_trace(self, act):  # :242-343
        a = act["a"]
        if a.right_offset is not None:
            return
        rs = a.rc_starts()
        n = len(a.seq)
        path_entries = {n - off: list(v) for off, v in a.rc_entries.items()}
        best = {"abs_svlen": INF, "abs_svlen_a": None, "shared_bases": 0, "shared_bases_a": None}
        reason, bases, fronts = self.traces.get(a.assembly_id, (0, [], []))
        seq = ""
        step = 0
        while rs:
            if step >= len(bases):
                if reason in (2, 3, 4) and rs:   # a bounded buffer ended the kernel's trace: resume from here
                    reason, bases, fronts = self.w.push_trace([rs], self.min_overlap, self.MAX_STEPS)[0]
                    step = 0
                    if bases:
                        continue
                path_entries[n + len(seq)] = list(rs)   # nothing found pushing here (:276-283)
                break
            seq += BASES[bases[step]]
            rs = fronts[step]
            step += 1
            found_mate = any(r in self.mates for r in rs)
            max_size = max(r[2] for r in rs)
            shared_here = self.save_anchors(act, rs, seq, path_entries, best)
            found_anchor = bool(shared_here)
            if found_mate or found_anchor:
                path_entries[n + len(seq)] = list(rs)
            if found_anchor and shared_here == max_size:
                break
        if best["shared_bases_a"] is not None:
            self.out.append(best["shared_bases_a"])
        if best["abs_svlen_a"] is not None:
            self.out.append(best["abs_svlen_a"])
        if path_entries:
            self.out.append(self.extend_right(act, n, seq, self.path_entries_to_rc_path(path_entries, n + len(seq))))


def discover_push_to_pair(asms, walker, min_overlap=70, max_pair_distance=1000, tag="GRAPH_PUSH_TO_PAIR",
                          discover_tags=()):
    return _PushToPair(walker, min_overlap, tag, discover_tags, max_pair_distance=max_pair_distance).run(asms)
