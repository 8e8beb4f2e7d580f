"""ORACLE (test infrastructure only; never imported by the product path): a CPU restatement of the reference's pop
tracer and of the wiring that feeds it --

  pop_tracer               modules/variants/pop_tracer.cpp:55-653   re-traces what the push tracer gave up on: the anchor
                                                                   drops (assemblies anchored on one side only) and the
                                                                   mates of the reads along the reference are "fronts";
                                                                   "poppers" lose their first base pass by pass
                                                                   (seqset_range::pop_front) until what is left is a
                                                                   prefix of some front
  reversable_tracer        modules/variants/reversable_tracer.cpp:49-74,118-126  direction wrapper of the pop tracer:
                                                                   pop tracer -> sorter(left offset) -> pair_counter
  trace_ref::execute_work  modules/variants/trace_ref.cpp:141-221  one work item: push forward, push reverse (their anchor
                                                                   drops reported to the pop tracers), pop forward, pop
                                                                   reverse

`biograph discovery --enable-pop-tracer` is on by default in the current reference (biograph_variants.cpp:188).
trace_ref.cpp:226-262 also hands the reads of the push tracer's non-reference assemblies to the pop tracers
(add_approx_read) -- but those assemblies come out of pair_counter, which has moved their rc_read_ids away
(pair_counter.cpp:147-148), so the loop never runs; not restated.

Two places depend on libstdc++'s std::sort (pop_tracer.cpp:484 over the matches, sort.cpp:7 over the assemblies of a
pass): introsort is not stable, so `std_sort` below restates bits/stl_algo.h (median-of-three quicksort down to runs of
16, then insertion sort) -- equal keys come out in the order the reference's binary gives them.

Pinned on modules/variants/trace_ref_test.cpp:230-300 (trace_ref_pop_test: simple, additional_inside_read; both
directions): tests/test_oracle_pop_tracer.py."""
import heapq  # noqa: F401  (the popper heap is libstdc++'s, restated below; heapq is not used for it)

from .tracer import (COMP, INT_MAX, Assembly, PairCounter, Scaffold, Tracer, _ScaffoldIt, allocate_assembly_id,
                     reverse_assembly_in_place, revcomp)

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


# ---- libstdc++ ------------------------------------------------------------------------------------------------------------
def std_sort(v, less):
    """std::sort (bits/stl_algo.h: __introsort_loop + __final_insertion_sort) in place."""
    n = len(v)
    if n < 2:
        return

    def move_median_to_first(result, a, b, c):
        if less(v[a], v[b]):
            if less(v[b], v[c]):
                m = b
            elif less(v[a], v[c]):
                m = c
            else:
                m = a
        elif less(v[a], v[c]):
            m = a
        elif less(v[b], v[c]):
            m = c
        else:
            m = b
        v[result], v[m] = v[m], v[result]

    def unguarded_partition(first, last, pivot):
        while True:
            while less(v[first], v[pivot]):
                first += 1
            last -= 1
            while less(v[pivot], v[last]):
                last -= 1
            if not first < last:
                return first
            v[first], v[last] = v[last], v[first]
            first += 1

    def introsort_loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                raise NotImplementedError("std::sort fell back to heapsort (not restated)")
            depth -= 1
            mid = first + (last - first) // 2
            move_median_to_first(first, first + 1, mid, last - 1)
            cut = unguarded_partition(first + 1, last, first)
            introsort_loop(cut, last, depth)
            last = cut

    def unguarded_linear_insert(last):
        val = v[last]
        nxt = last - 1
        while less(val, v[nxt]):
            v[last] = v[nxt]
            last = nxt
            nxt -= 1
        v[last] = val

    def insertion_sort(first, last):
        for i in range(first + 1, last):
            if less(v[i], v[first]):
                val = v[i]
                v[first + 1:i + 1] = v[first:i]
                v[first] = val
            else:
                unguarded_linear_insert(i)

    introsort_loop(0, n, 2 * (n.bit_length() - 1))
    if n > 16:
        insertion_sort(0, 16)
        for i in range(16, n):
            unguarded_linear_insert(i)
    else:
        insertion_sort(0, n)


def _push_heap(h, less):  # std::push_heap (bits/stl_heap.h __push_heap), the new element is h[-1]
    hole = len(h) - 1
    val = h[hole]
    parent = (hole - 1) // 2
    while hole > 0 and less(h[parent], val):
        h[hole] = h[parent]
        hole = parent
        parent = (hole - 1) // 2
    h[hole] = val


def _adjust_heap(h, hole, n, val, less):  # bits/stl_heap.h __adjust_heap
    top = hole
    child = hole
    while child < (n - 1) // 2:
        child = 2 * (child + 1)
        if less(h[child], h[child - 1]):
            child -= 1
        h[hole] = h[child]
        hole = child
    if (n & 1) == 0 and child == (n - 2) // 2:
        child = 2 * (child + 1)
        h[hole] = h[child - 1]
        hole = child - 1
    parent = (hole - 1) // 2
    while hole > top and less(h[parent], val):
        h[hole] = h[parent]
        hole = parent
        parent = (hole - 1) // 2
    h[hole] = val


def _pop_heap(h, less):  # std::pop_heap: the top goes to h[-1]
    if len(h) > 1:
        val = h[-1]
        h[-1] = h[0]
        _adjust_heap(h, 0, len(h) - 1, val, less)


def _make_heap(h, less):  # std::make_heap
    n = len(h)
    if n < 2:
        return
    parent = (n - 2) // 2
    while True:
        val = h[parent]
        _adjust_heap(h, parent, n, val, less)
        if parent == 0:
            return
        parent -= 1


# ---- what the pop tracer asks of seqset + readmap ----------------------------------------------------------------------------
class PopGraph:
    """The push tracer's Graph plus seqset_range::pop_front / push_front / front and the readmap's read objects."""

    def __init__(self, graph, entry_strings):
        self.g = graph
        self.ss, self.rm = graph.ss, graph.rm
        self.entries = entry_strings  # sequences of the seqset's entries, in order
        self.read_len = [int(x) for x in graph.rm.get("read_len")]

    def ctx_begin(self):
        return self.g.ctx_begin()

    def sequence(self, r):
        return self.entries[r[0]][:r[2]] if r[2] else ""

    def push_front_drop(self, r, base):
        return self.g.push_front_drop(r, base, 0)

    def push_front(self, r, base):  # seqset_range::push_front: invalid (None) when nothing is prefixed by base + sequence
        ob, oe, osz = self.ss.push_front([r[0]], [r[1]], [r[2]], [CODE[base]])
        return None if ob[0] >= oe[0] else (int(ob[0]), int(oe[0]), int(osz[0]))

    def pop_front(self, r):  # seqset.cpp:609-628 = the range of the sequence without its first base
        b, e, s = self.ss.find(self.sequence(r)[1:])
        assert e > b and s == r[2] - 1
        return (int(b), int(e), int(s))

    def front(self, r):  # seqset.cpp:663
        return self.entries[r[0]][0]

    def read_entry(self, rid):  # readmap::read::get_seqset_entry (readmap.h:249-251)
        return self.rm.read_info(self.ss, rid)[3]

    def prefix_reads(self, r, read_len_limit):  # readmap::get_prefix_reads (readmap.cpp:434-447), iteration order
        return self.rm.prefix_reads(self.ss, r, read_len_limit)


class PopOptions:
    """assemble_options fields of the pop tracer (assemble.h:490,523,527,585-589)."""

    def __init__(self, o, **kw):
        self.min_overlap = o.min_overlap
        self.max_pairs_per_read = o.max_pairs_per_read
        self.forward_pairs_face_inward = o.forward_pairs_face_inward
        self.max_pair_distance = o.max_pair_distance
        self.min_pair_distance = 100
        self.min_pop_overlap = 15
        self.pop_tracer_offset_slop = 1000
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("no pop tracer option %r" % k)
            setattr(self, k, v)


class _Entry:  # pop_tracer::entry (pop_tracer.h:58-84)
    __slots__ = ("left_offset", "right_offset", "start_limit", "end_limit", "orig_r", "popped_r", "seq", "seen_read_ids",
                 "matches_reference", "serial")

    def __init__(self):
        self.left_offset = None
        self.right_offset = None
        self.start_limit = self.end_limit = 0
        self.orig_r = self.popped_r = None
        self.seq = ""
        self.seen_read_ids = []
        self.matches_reference = False


def _popper_less(a, b):  # popper_sorter (:285-290): the heap's top is the longest popped range
    return a.popped_r[2] < b.popped_r[2]


class PopTracer:
    def __init__(self, pg, scaffold, options):
        self.pg, self.sc, self.o = pg, scaffold, options
        self.fronts = []   # m_fronts: a multimap ordered by (begin, size) of the key range; equal keys in insertion order
        self.poppers = []  # m_poppers (a heap by popper_sorter while entries are being added)

    # ---- m_fronts
    def _fronts_emplace(self, r, e):
        key = (r[0], r[2])
        i = len(self.fronts)
        while i > 0 and self.fronts[i - 1][0] > key:
            i -= 1
        self.fronts.insert(i, (key, e))

    def _fronts_lower_bound(self, r):
        key = (r[0], r[2])
        lo, hi = 0, len(self.fronts)
        while lo < hi:
            mid = (lo + hi) // 2
            if self.fronts[mid][0] < key:
                lo = mid + 1
            else:
                hi = mid
        return lo

    # ---- adding
    def add_entry_reads(self, p, r):  # :458-467
        count = 0
        for rid in self.pg.prefix_reads(r, self.o.min_overlap):
            count += 1
            if count > self.o.max_pairs_per_read:
                break
            p.seen_read_ids.append(self.pg.g.get_rev_comp(rid))

    def add_read(self, read_id, start_offset, limit_offset):  # :76-95
        r = self.pg.read_entry(read_id)
        e = _Entry()
        e.start_limit, e.end_limit = start_offset, limit_offset
        e.orig_r = e.popped_r = r
        self.add_entry_reads(e, r)
        self._fronts_emplace(r, e)

    def add_reference_at(self, offset, r):  # :97-160
        o, g = self.o, self.pg.g
        if r[2] < o.min_pop_overlap:
            return
        e = _Entry()
        e.left_offset = offset
        e.start_limit = e.end_limit = offset
        e.orig_r = e.popped_r = r
        e.matches_reference = True
        self.add_entry_reads(e, r)
        self._fronts_emplace(r, e)
        read_count = 0
        for rid in self.pg.prefix_reads(r, o.min_overlap):
            if not g.has_mate(rid):
                continue
            read_count += 1
            if read_count > o.max_pairs_per_read:
                break
            mate_rc = g.get_rev_comp(g.get_mate(rid))
            size = self.pg.read_len[rid]
            if size < o.min_pop_overlap:
                continue
            start_offset = limit_offset = offset
            if bool(g.is_forward[rid]) == bool(o.forward_pairs_face_inward):  # faces inward
                start_offset += size + o.min_pair_distance
                limit_offset += size + o.max_pair_distance
            else:  # faces outwards
                start_offset -= size + o.max_pair_distance
                limit_offset -= size + o.min_pair_distance
            self.add_read(mate_rc, start_offset, limit_offset)

    def add_reference(self, start, limit):  # :162-206
        rc = self.sc.rev_comp()
        scaffold_end = rc.end_pos
        it = _ScaffoldIt(rc)
        r = self.pg.ctx_begin()
        skip_pos = scaffold_end - limit - self.pg.rm.max_read_len
        if skip_pos > it.offset:
            it.skip_to(skip_pos)
        offset = scaffold_end - it.offset
        while not it.at_end() and offset >= start:
            if it.first_in_extent():
                self.add_reference_at(offset, r)
                r = self.pg.ctx_begin()
            nxt = self.pg.push_front_drop(r, COMP[it.base()])
            assert nxt is not None  # the four single-base scaffolds of a reference make every base present
            if nxt[2] <= r[2]:
                self.add_reference_at(offset, r)
            r = nxt
            if it.first_in_extent():
                offset = scaffold_end - it.offset - 1
            else:
                assert offset == scaffold_end - it.offset
                offset -= 1
            it.advance()
        self.add_reference_at(offset, r)

    def add_anchor_drop(self, a, right_anchor):  # :208-283
        o, g, pg = self.o, self.pg.g, self.pg
        offset = a.right_offset if right_anchor else a.left_offset
        n = len(a.seq)
        for rc_read_id in sorted(a.rc_read_ids):  # read_id_set iterates ascending
            if not g.has_mate(rc_read_id):
                continue
            start_offset = offset - n
            limit_offset = offset + n
            if right_anchor:
                start_offset -= n
            else:
                limit_offset += n
            mate = g.get_mate(rc_read_id)
            size = pg.read_len[rc_read_id]
            if bool(g.is_forward[rc_read_id]) == bool(o.forward_pairs_face_inward):
                start_offset -= size + o.max_pair_distance
            else:
                limit_offset += size + o.max_pair_distance
            self.add_read(mate, start_offset, limit_offset)
        e = _Entry()
        e.start_limit = offset - (n + o.max_pair_distance)
        e.end_limit = offset + (n + o.max_pair_distance)
        e.orig_r = pg.ctx_begin()
        for b in reversed(a.seq):  # `for b in a.seq.rev_comp(): push_front_drop(b.complement())` = the bases backwards
            e.orig_r = pg.push_front_drop(e.orig_r, b)
            assert e.orig_r is not None
        e.popped_r = pg.ctx_begin()
        if right_anchor:
            e.right_offset = a.right_offset
            e.seq = a.seq
        else:
            e.left_offset = a.left_offset
            for b in reversed(a.seq):
                pushed = pg.push_front(e.popped_r, b)
                if pushed is None:
                    break
                e.popped_r = pushed
            assert n >= e.popped_r[2]
            e.seq = a.seq[:n - e.popped_r[2]]
            self.add_popper(e)
        self._fronts_emplace(e.orig_r, e)

    def add_popper(self, p):  # :447-456 (the CHECKs on "NN" aside)
        self.poppers.append(p)
        _push_heap(self.poppers, _popper_less)

    # ---- assembling
    def assemble(self, emit):  # :292-311
        _make_heap(self.poppers, _popper_less)
        self.match_and_output_pass(emit)
        while self.poppers:
            self.pop_pass()
            self.match_and_output_pass(emit)

    def pop_pass(self):  # :364-445
        old = self.poppers
        self.poppers = []
        while old:
            _pop_heap(old, _popper_less)
            p = old.pop()
            assert p.right_offset is None and p.popped_r[2] >= self.o.min_pop_overlap
            popped = self.pg.pop_front(p.popped_r)
            if popped[2] < self.o.min_pop_overlap:
                continue
            p.seq += self.pg.front(p.popped_r)
            p.popped_r = popped
            self.add_entry_reads(p, popped)
            p.end_limit += 1
            self.add_popper(p)

    def _match_less(self, p):  # match_sorter (:313-362)
        p_middle = _div2(p.start_limit + p.end_limit) + len(p.seq)

        def less(li, ri):
            lhs, rhs = li[1], ri[1]
            if (lhs.left_offset is not None) != (rhs.left_offset is not None):
                return lhs.left_offset is not None  # reanchor to reference if possible
            if lhs.orig_r[2] != rhs.orig_r[2]:
                return lhs.orig_r[2] > rhs.orig_r[2]  # more specific is better
            ld = abs(p_middle - _div2(lhs.start_limit + lhs.end_limit))
            rd = abs(p_middle - _div2(rhs.start_limit + rhs.end_limit))
            if ld != rd:
                return ld < rd
            # (:347-349 compares rhs.seq.size() with itself: never decides)
            ls, rs = lhs.end_limit - lhs.start_limit, rhs.end_limit - rhs.start_limit
            if ls != rs:
                return ls < rs
            return False
        return less

    def match_and_output_pass(self, emit):  # :469-613
        o = self.o
        old = self.poppers
        self.poppers = []
        for p in old:
            matches = []
            i = self._fronts_lower_bound(p.popped_r)
            while i < len(self.fronts) and self.fronts[i][1].orig_r[0] < p.popped_r[1]:
                item = self.fronts[i]
                front = item[1]
                i += 1
                if front is p:
                    continue  # no loops
                if front.orig_r[1] > p.popped_r[1]:
                    continue  # not a prefix
                if p.end_limit + (len(p.seq) + o.pop_tracer_offset_slop) < front.start_limit:
                    continue  # out of range (1)
                if front.end_limit + (len(p.seq) + o.pop_tracer_offset_slop) < p.start_limit:
                    continue  # out of range (2)
                if front.right_offset is not None and p.left_offset is not None and front.right_offset < p.left_offset:
                    continue  # misordered
                if front.left_offset is not None and p.left_offset is not None and front.left_offset < p.left_offset:
                    continue  # misordered left offset
                matches.append(item)
            std_sort(matches, self._match_less(p))
            found = False
            for item in matches:
                front = item[1]
                found = True
                if front.left_offset is not None:  # the front is aligned to reference
                    p.right_offset = front.left_offset
                    if p.left_offset is not None:
                        self.output_assembly(emit, p)
                else:  # merge in
                    p.start_limit = max(p.start_limit, front.start_limit - len(p.seq))
                    p.end_limit = min(p.end_limit, front.end_limit - len(p.seq))
                    p.popped_r = front.popped_r
                    p.seq += front.seq
                    p.seen_read_ids.extend(front.seen_read_ids)
                    if front.right_offset is not None:
                        p.right_offset = front.right_offset
                        if p.left_offset is not None:
                            self.output_assembly(emit, p)
                    else:
                        self.poppers.append(p)  # keeps searching for something to align to on the right
                if front.left_offset is None:
                    for j, it in enumerate(self.fronts):  # m_fronts.erase(it): consumed
                        if it is item:
                            del self.fronts[j]
                            break
                break
            if not found:
                self.poppers.append(p)  # try again after popping more

    def output_assembly(self, emit, p):  # :615-651
        a = Assembly()
        a.tags = {"POP"}
        a.assembly_id = allocate_assembly_id()
        a.left_offset, a.right_offset = p.left_offset, p.right_offset
        a.seq = p.seq
        a.rc_read_ids = set(p.seen_read_ids)
        if a.left_offset == a.right_offset and not a.seq:
            return
        # (:631-636 sets matches_reference only inside `if (k_trace_dbg)`: never in a build)
        emit(a)


def _div2(x):  # C++ integer division truncates towards zero
    return int(x / 2) if x < 0 else x // 2


def reversed_copy(a, graph, ref_end):
    """`assembly rc_a = a; reverse_assembly_in_place(&rc_a, readmap, ref_end)` (assemble.cpp:933-984): unlike the assemblies
    that leave pair_counter, an anchor drop still carries its rc_read_ids, and they turn into their reverse complements."""
    from .align import copy_assembly
    b = copy_assembly(a)
    reverse_assembly_in_place(b, graph, ref_end)
    b.rc_read_ids = {graph.get_rev_comp(r) for r in a.rc_read_ids}
    return b


# ---- reversable_tracer with use_pop_tracer (reversable_tracer.cpp:11-47,49-74,118-126) -------------------------------------
class ReversablePopTracer:
    def __init__(self, pg, scaffold, options, pop_options, rev_comp):
        self.pg, self.o, self.rev_comp = pg, options, rev_comp
        self.ref_end = scaffold.end_pos
        self.run_sc = scaffold.rev_comp() if rev_comp else scaffold
        self.pop = PopTracer(pg, self.run_sc, pop_options)

    def report_anchor_drop(self, a, right_anchor):  # wrap_report_anchor_drop_for_pop_tracer: `a` in forward coordinates
        if self.rev_comp:
            self.pop.add_anchor_drop(reversed_copy(a, self.pg.g, self.ref_end), not right_anchor)
        else:
            self.pop.add_anchor_drop(a, right_anchor)

    def assemble(self, start_offset, limit_offset):
        """-> the pop tracer's assemblies as they reach the scaffold's pipeline: sorter(left_offset_less_than) ->
        pair_counter -> reversed back when rev_comp."""
        limit_offset = min(limit_offset, self.ref_end)
        out = []

        def final(a):
            if self.rev_comp:
                reverse_assembly_in_place(a, self.pg.g, self.ref_end)
            out.append(a)

        pc = PairCounter(self.pg.g, self.o, final)
        queued = []
        if self.rev_comp:
            self.pop.add_reference(self.ref_end - limit_offset, self.ref_end - start_offset)
        else:
            self.pop.add_reference(start_offset, limit_offset)
        self.pop.assemble(queued.append)
        std_sort(queued, lambda a, b: a.left_offset < b.left_offset)  # sorter::flush (sort.cpp:6-13)
        for a in queued:
            pc.add(a)
        pc.flush()
        return out


def execute_work(graph, entry_strings, scaffold, options, pop_options, start=0, limit=None, anchor_drop=None, skip_push_fwd=False,
                 skip_push_rev=False, skip_pop_fwd=False, skip_pop_rev=False):
    """trace_ref::execute_work (trace_ref.cpp:141-221) with pop_trace_anchor_drop: -> the assemblies of one work item in the
    order they reach the scaffold's pipeline (push forward, push reverse, pop forward, pop reverse)."""
    from .tracer import reversable_trace
    limit = scaffold.end_pos if limit is None else limit
    pg = PopGraph(graph, entry_strings)
    pops = []
    if not skip_pop_fwd:
        pops.append(ReversablePopTracer(pg, scaffold, options, pop_options, False))
    if not skip_pop_rev:
        pops.append(ReversablePopTracer(pg, scaffold, options, pop_options, True))
    out = []
    for rev in (False, True):
        if (skip_push_rev if rev else skip_push_fwd):
            continue

        def report(a, right_anchor, rev=rev):  # tracer.cpp:1233-1235 through reversable_tracer.cpp:28-35
            if rev:
                a, right_anchor = reversed_copy(a, graph, scaffold.end_pos), not right_anchor
            for p in reversed(pops):  # the later wrapper runs first (trace_ref.cpp:164-172)
                p.report_anchor_drop(a, right_anchor)

        out.extend(reversable_trace(graph, scaffold, options, rev_comp=rev, start_offset=start, limit_offset=limit,
                                    anchor_drop=anchor_drop, report_anchor_drop=report))
    for p in pops:
        out.extend(p.assemble(start, limit))
    return out
