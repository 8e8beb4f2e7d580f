"""ORACLE (test infrastructure only; never imported by the product path): a literal CPU restatement of the
reference's production "push" tracer -- modules/variants/tracer.cpp (assemble :195-318, read-ahead :327-445,
trace :477-526, expand_next_path :528-673, add_rejoins :675-724, add_pairs_to_next_path :737-845,
add_base_to_next_path :847-877, the queues :879-941, trace_next_paths :943-1021, output :1023-1250) with the
direction wrapper of reversable_tracer.cpp:11-145 and the pair_counter step that follows it (pair_counter.cpp).
The anchor_dropper step between them (anchor_drop.cpp) is restated in oracle/align.py and plugged in through
reversable_trace(anchor_drop=...); without it, assemblies that rejoin the reference are unchanged
(anchor_drop.cpp:196-202) and dead-end assemblies come out as the tracer emits them (right_anchor_len == 0).

Pinned on the vectors of modules/variants/tracer_test.cpp:167-304 (tests/test_oracle_tracer.py).

The seqset / readmap / ref_map are reached through a small adapter (`Graph`) so that the same state machine can be
driven by the oracle's structures here and compared with the GPU-backed driver in the -m gpu tests."""
import bisect

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
BASES = "ACGT"
INT_MAX = 2 ** 31 - 1
INT_MIN = -2 ** 31
U64_MAX = 2 ** 64 - 1


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


class Options:
    """assemble_options (assemble.h:475-720), the fields the push tracer and pair_counter read."""

    def __init__(self, **kw):
        self.ambiguous_branch_cost = 1
        self.rejoin_local_cost = 1
        self.max_pairs_per_read = 20
        self.base_cost = 1
        self.increase_max_between_pair_cost = 1
        self.max_cost = 1000 * 1000
        self.max_next_paths = 1024
        self.pairs_used_cost = -100
        self.avg_coverage_score = 300
        self.pair_match_score = 130
        self.min_overlap_score = 100
        self.min_coverage_score = 100
        self.anchor_drop_score = -10000
        self.max_ambiguous_bases = 300
        self.traverse_ref_cost = 100 * 1000
        self.only_trace_forward = False
        self.trace_dead_ends = True
        self.min_overlap = 100
        self.read_ahead_distance = 100000
        self.scaffold_split_size = 1000 * 1000
        self.decrease_overlap_cost = 1000
        self.size_change_cost = 5
        self.dead_end_cost = 500 * 1000
        self.max_branches_between_pairs = 2
        self.max_rejoins = 5
        self.max_bases_between_pairs = 300
        self.max_search_steps = 1000
        self.max_ambiguous_search_steps = 100
        self.initial_search_steps = 100
        self.max_search_steps_per_read = 3
        self.forward_pairs_face_inward = True
        self.max_pair_distance = 1000
        self.ignore_ambiguous_ref_pairs = True
        self.trace_ambiguous_ref = False
        self.anchor_drop_size_multiplier = 1.5
        self.trace_reference_assemblies = False
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("no assemble option %r" % k)
            setattr(self, k, v)


class Scaffold:
    """variants::scaffold: sorted N-free extents [(offset, bases)] and the end position."""

    def __init__(self, extents, end_pos=None):
        self.extents = [(int(o), s) for o, s in extents]
        calc = max([o + len(s) for o, s in self.extents] + [0])
        self.end_pos = calc if end_pos is None else end_pos

    def rev_comp(self):  # scaffold.cpp:163-175
        return Scaffold([(self.end_pos - o - len(s), revcomp(s)) for o, s in reversed(self.extents)], self.end_pos)

    def sub(self, offset, length):
        """Bases [offset, offset+length) when they lie inside one extent (get_ref_part_seq of the tests), else None."""
        for o, s in self.extents:
            if o <= offset and offset + length <= o + len(s):
                return s[offset - o: offset - o + length]
        return None


class _ScaffoldIt:
    """scaffold::iterator (scaffold.h:14-60, skip_to scaffold.cpp:205-247)."""

    def __init__(self, sc):
        self.sc = sc
        self.ext = 0
        self.pos = 0
        self.offset = sc.extents[0][0] if sc.extents else sc.end_pos

    def at_end(self):
        return self.ext >= len(self.sc.extents)

    def base(self):
        return self.sc.extents[self.ext][1][self.pos]

    def first_in_extent(self):
        return self.pos == 0

    def _next_extent(self):
        self.ext += 1
        self.pos = 0
        self.offset = self.sc.end_pos if self.at_end() else self.sc.extents[self.ext][0]

    def advance(self):
        self.pos += 1
        self.offset += 1
        if self.pos == len(self.sc.extents[self.ext][1]):
            self._next_extent()

    def skip_to(self, target):
        assert target >= self.offset
        while self.offset < target:
            left = len(self.sc.extents[self.ext][1]) - self.pos
            adv = min(target - self.offset, left)
            self.pos += adv
            self.offset += adv
            if self.pos == len(self.sc.extents[self.ext][1]):
                self._next_extent()
                if self.at_end():
                    return


class Graph:
    """What the tracer asks of seqset + readmap + ref_map.  Ranges are (begin, end, size) tuples."""

    def __init__(self, seqset, readmap, ref_map):
        import numpy as np
        self.ss = seqset
        self.rm = readmap
        self.n = len(seqset)
        sizes = seqset.arrays()[0]
        self.entry_sizes = [int(x) for x in sizes]
        self.max_read_len = seqset.max_len
        self.entry_id = [int(x) for x in readmap.get("entry_id")]      # sorted: read ids are grouped by entry
        self.mate_loop = [int(x) for x in readmap.get("mate_loop_ptr")]
        self.is_forward = [int(x) for x in readmap.get("is_forward")]
        self.ref_map = [int(x) & 63 for x in np.asarray(ref_map)]       # ref_map::entry::match_count

    def ctx_begin(self):
        return (0, self.n, 0)

    def ctx_entry(self, e):
        return (e, e + 1, self.entry_sizes[e])

    def push_front_drop(self, r, base, min_ctx=0):
        """seqset_range::push_front_drop(base, min_ctx) -> range or None (invalid)."""
        ob, oe, osz = self.ss.push_front_drop([r[0]], [r[1]], [r[2]], [CODE[base]], [min_ctx])
        if ob[0] >= oe[0]:
            return None
        return (int(ob[0]), int(oe[0]), int(osz[0]))

    def push_front_drop4(self, r, min_ctx):
        """The four pushes of tracer.cpp:553-554: {b: range after push_front_drop(complement of b), or None}."""
        ob, oe, osz = self.ss.push_front_drop([r[0]] * 4, [r[1]] * 4, [r[2]] * 4, [3, 2, 1, 0], [min_ctx] * 4)
        return {b: ((int(ob[i]), int(oe[i]), int(osz[i])) if ob[i] < oe[i] else None) for i, b in enumerate(BASES)}

    def is_entry(self, r):  # seqset.h:298-300
        return r[0] + 1 == r[1] and r[2] == self.entry_sizes[r[0]]

    def entry_to_index(self, e):  # readmap.h:114-116
        return bisect.bisect_left(self.entry_id, e), bisect.bisect_right(self.entry_id, e)

    def get_bit(self, e):  # readmap.cpp:197-200
        a, b = self.entry_to_index(e)
        return a != b

    def has_mate(self, rid):  # readmap.cpp:207-218
        return self.mate_loop[self.mate_loop[rid]] != rid

    def get_mate(self, rid):  # readmap.cpp:220-239
        return self.mate_loop[self.mate_loop[rid]]

    def get_rev_comp(self, rid):  # readmap.cpp:248-268
        for _ in range(1 if self.is_forward[rid] else 3):
            rid = self.mate_loop[rid]
        return rid

    def match_count(self, e):
        return self.ref_map[e]


class Path:
    __slots__ = ("prev", "cost", "min_overlap", "range", "seq", "part_of_assembly", "seen_pairs", "seen_read_ids", "serial")

    def __init__(self, serial):
        self.prev = None
        self.cost = 0
        self.min_overlap = 0
        self.range = None
        self.seq = ""
        self.part_of_assembly = False
        self.seen_pairs = []       # (read_id, offset)
        self.seen_read_ids = []
        self.serial = serial       # allocation order (stands in for the address order of std::set<const path*>)


class NextPath:
    __slots__ = ("new_path", "path_bases", "pushed_since_read", "pushed_since_pair", "ambiguous_bases",
                 "branch_count_since_pair", "max_between_pairs", "pairs_used", "num_reads", "loop_check_seqset_id",
                 "reads_until_loop_check")

    def __init__(self):
        self.new_path = None
        self.path_bases = 0
        self.pushed_since_read = 0
        self.pushed_since_pair = 0
        self.ambiguous_bases = 0
        self.branch_count_since_pair = 0
        self.max_between_pairs = 0
        self.pairs_used = 0
        self.num_reads = 0
        self.loop_check_seqset_id = U64_MAX
        self.reads_until_loop_check = 0

    def copy(self):
        c = NextPath()
        for k in NextPath.__slots__:
            setattr(c, k, getattr(self, k))
        return c


class Rejoin:
    __slots__ = ("p", "rejoin_cost", "right_offset", "right_anchor_len", "anchor_drop")

    def __init__(self):
        self.p = None
        self.rejoin_cost = 0
        self.right_offset = INT_MAX
        self.right_anchor_len = 0
        self.anchor_drop = False


_next_assembly_id = [1]


def allocate_assembly_id():  # assemble.cpp:873-877
    _next_assembly_id[0] += 1
    return _next_assembly_id[0] - 1


class Assembly:
    def __init__(self):
        self.assembly_id = allocate_assembly_id()  # assemble.cpp:1248; copies and split parts keep it
        self.left_offset = 0
        self.right_offset = 0
        self.left_anchor_len = 0
        self.right_anchor_len = 0
        self.seq = ""
        self.min_overlap = 0
        self.trace_steps = 0
        self.left_anchor_ambiguous_bases = 0
        self.matches_reference = False
        self.rc_read_ids = set()
        self.score = 0
        self.left_pair_matches = []
        self.right_pair_matches = []
        self.tags = set()
        self.strand_count = 0          # assemble.h:360-370: what the genotype step fills; 0 / 0.0 when it does not run
        self.genotype_quality = 0.0

    def key(self):
        return (self.left_offset, self.right_offset, self.left_anchor_len, self.right_anchor_len, self.seq,
                self.matches_reference, self.min_overlap)

    def __repr__(self):
        return "Assembly[%d:%d) anchors %d/%d %s%s" % (self.left_offset, self.right_offset, self.left_anchor_len,
                                                       self.right_anchor_len, self.seq, " REF" if self.matches_reference else "")


# ---- std::push_heap / std::pop_heap of libstdc++ (bits/stl_heap.h) with next_path_comparer ------------------
def _worse(a, b):  # next_path_comparer: true when b is better than a
    return a.new_path.cost > b.new_path.cost


def _push_heap(h):
    v = h[-1]
    hole = len(h) - 1
    parent = (hole - 1) // 2
    while hole > 0 and _worse(h[parent], v):
        h[hole] = h[parent]
        hole = parent
        parent = (hole - 1) // 2
    h[hole] = v


def _pop_heap(h):
    """Moves the best element to h[-1] (the caller pops it)."""
    n = len(h) - 1
    if n <= 0:
        return
    v = h[n]
    h[n] = h[0]
    hole, child = 0, 0
    while child < (n - 1) // 2:
        child = 2 * (child + 1)
        if _worse(h[child], h[child - 1]):
            child -= 1
        h[hole] = h[child]
        hole = child
    if (n & 1) == 0 and child == (n - 2) // 2:
        child = 2 * (child + 1)
        h[hole] = h[child - 1]
        hole = child - 1
    parent = (hole - 1) // 2
    while hole > 0 and _worse(h[parent], v):
        h[hole] = h[parent]
        hole = parent
        parent = (hole - 1) // 2
    h[hole] = v


class Tracer:
    """tracer (one direction).  `graph`: Graph, `scaffold`: Scaffold already in the direction traced."""

    def __init__(self, graph, scaffold, options):
        self.g = graph
        self.sc = scaffold
        self.o = options
        self.cur_offset = 0
        self.stats = {}
        self.report_anchor_drop = None

    def _stat(self, k, n=1):
        self.stats[k] = self.stats.get(k, 0) + n

    # ---- assemble (tracer.cpp:195-318)
    def assemble(self, start_offset=0, limit_offset=INT_MAX, emit=None):
        o, g = self.o, self.g
        assert limit_offset <= start_offset + o.scaffold_split_size or limit_offset == INT_MAX and start_offset == 0
        out = []
        self.emit = emit if emit is not None else out.append
        self.start_offset, self.limit_offset = start_offset, limit_offset
        self.it = _ScaffoldIt(self.sc)
        self.read_ahead_offset = self.it.offset
        self.position_entry_index = 0
        self.position_entries = []      # deque of (offset, seqset id); popped from the front
        self.entry_positions = {}       # seqset id -> [offsets]
        self.rc_mate_read_positions = {}
        self.read_ahead_range = g.ctx_begin()
        self._skip_ahead_to(start_offset - o.max_pair_distance - g.max_read_len)
        for part_offset, part_seq in self.sc.extents:
            if part_offset < self.cur_offset:
                raise ValueError("Ref parts must be sorted")
            self.cur_offset = part_offset
            if part_offset + len(part_seq) <= start_offset:
                continue
            if part_offset > limit_offset:
                break
            to_process = part_seq
            if part_offset + g.max_read_len < start_offset:
                skip_len = start_offset - part_offset - g.max_read_len
                to_process = to_process[skip_len:]
                self.cur_offset += skip_len
            self.cur_left_anchor = ""
            self.cur_ref_ambiguous_bases = 0
            for b in to_process:
                self._advance_read_ahead_to(self.cur_offset + o.read_ahead_distance)
                self._advance_trail_behind_to(self.cur_offset - o.max_pair_distance)
                self.cur_left_anchor += b
                self.cur_ref_ambiguous_bases += 1
                if self.cur_offset >= limit_offset:
                    return out
                self.cur_offset += 1
                if self.cur_offset < start_offset:
                    continue
                pe = self.position_entries
                while self.position_entry_index != len(pe) and pe[self.position_entry_index][0] < self.cur_offset:
                    self.position_entry_index += 1
                if self.position_entry_index < len(pe) and pe[self.position_entry_index][0] == self.cur_offset:
                    self.cur_range = g.ctx_entry(pe[self.position_entry_index][1])
                    if self._is_read(self.cur_range):
                        if len(self.cur_left_anchor) < self.cur_range[2]:
                            continue
                        self.cur_left_anchor = self.cur_left_anchor[len(self.cur_left_anchor) - self.cur_range[2]:]
                        self._trace()
                        self._output_rejoins()
        return out

    def _is_read(self, r):  # :521-526
        return self.g.is_entry(r) and self.g.get_bit(r[0])

    # ---- read ahead / trail behind (:327-445)
    def _skip_ahead_to(self, offset):
        if offset <= self.read_ahead_offset or self.it.at_end():
            return
        self.it.skip_to(offset)
        self.read_ahead_offset = self.it.offset
        # (skipping past the end of the scaffold leaves nothing to read ahead)
        if not self.it.at_end():
            r = self.g.push_front_drop(self.g.ctx_begin(), COMP[self.it.base()])
            self.read_ahead_range = r if r is not None else (0, 0, 0)

    def _pairing_skipped(self, seqset_id):
        return self.o.ignore_ambiguous_ref_pairs and self.g.match_count(seqset_id) > 1

    def _rc_mates_of(self, seqset_id):
        g = self.g
        a, b = g.entry_to_index(seqset_id)
        for rid in range(a, b):
            if g.has_mate(rid):
                yield g.get_rev_comp(g.get_mate(rid))

    def _advance_read_ahead_to(self, offset):
        g, it = self.g, self.it
        while self.read_ahead_offset < offset:
            if it.at_end():
                self.read_ahead_offset = offset
                return
            if offset <= it.offset:
                return
            if it.first_in_extent():
                self.read_ahead_range = g.ctx_begin()
                assert it.offset >= self.read_ahead_offset
                self.read_ahead_offset = it.offset
            else:
                assert self.read_ahead_offset == it.offset
            r = g.push_front_drop(self.read_ahead_range, COMP[it.base()])
            assert r is not None  # min_ctx 0: falls back to the single base, which every seqset of real reads has
            self.read_ahead_range = r
            self.read_ahead_offset += 1
            it.advance()
            if self._is_read(self.read_ahead_range):
                sid = self.read_ahead_range[0]
                self.position_entries.append((self.read_ahead_offset, sid))
                if not self._pairing_skipped(sid):
                    for rc_mate in self._rc_mates_of(sid):
                        self.rc_mate_read_positions.setdefault(rc_mate, []).append(self.read_ahead_offset)
                self.entry_positions.setdefault(sid, []).append(self.read_ahead_offset)
            if it.at_end():
                return

    def _advance_trail_behind_to(self, offset):
        while self.position_entries:
            old_offset, old_sid = self.position_entries[0]
            if old_offset >= offset:
                return
            self.position_entries.pop(0)
            if self.position_entry_index:
                self.position_entry_index -= 1
            if not self._pairing_skipped(old_sid):
                for rc_mate in self._rc_mates_of(old_sid):
                    self.rc_mate_read_positions[rc_mate].remove(old_offset)
            self.entry_positions[old_sid].remove(old_offset)

    def _has_rc_mate_in_range(self, read_id, start, limit):  # :466-475
        return any(start <= x < limit for x in self.rc_mate_read_positions.get(read_id, ()))

    # ---- one trace (:477-519)
    def _alloc(self):
        self.n_paths += 1
        return Path(self.n_paths)

    def _trace(self):
        o = self.o
        self.n_paths = 0
        self.next_paths = []
        self.rejoin_paths = []      # ascending std::multiset order: worst first, best last
        self.dead_end_rejoins = []  # paths, kept in allocation order
        self.trace_outputs = 0
        self.search_step_count = 0
        self.ambiguous_search_step_count = 0
        if self.cur_offset < self.start_offset or self.cur_offset >= self.limit_offset:
            return
        if self.g.match_count(self.cur_range[0]) > 1:
            self._stat("ambiguous_ref_reads")
            if not o.trace_ambiguous_ref:
                return
        else:
            self._stat("ref_reads")
            self.cur_ref_ambiguous_bases = 0
        np_ = NextPath()
        np_.new_path = self._alloc()
        np_.new_path.range = self.cur_range
        np_.new_path.seq = self.cur_left_anchor
        np_.ambiguous_bases = self.cur_ref_ambiguous_bases
        np_.new_path.min_overlap = len(self.cur_left_anchor) & 0xFF  # path::min_overlap is a uint8_t (tracer.h:135)
        self._push_next_path(np_)
        self._trace_next_paths()

    # ---- expand (:528-673)
    def _expand_next_path(self, np_):
        o, g = self.o, self.g
        new_path = np_.new_path
        pushed_any = False
        while True:
            push_lengths = []
            valid_count = 0
            next_base = None
            pushed = g.push_front_drop4(new_path.range, o.min_overlap + np_.pushed_since_read)
            for b in BASES:
                r = pushed[b]
                if r is None:
                    continue
                if g.is_entry(r) and np_.loop_check_seqset_id == r[0]:
                    self._stat("loops")
                    pushed[b] = None
                    continue
                valid_count += 1
                next_base = b
                push_lengths.append(r[2])
            if valid_count != 1:
                self.ambiguous_search_step_count += 1
                break
            if pushed_any and self._is_read(new_path.range):
                break
            new_path.range = pushed[next_base]
            self._add_base_to_next_path(np_, next_base)
            pushed_any = True
            if np_.pushed_since_pair > o.max_bases_between_pairs:
                self._stat("too_far_without_pair")
                return
            if self._add_rejoins(np_):
                return
        if new_path.prev is not None:
            self._add_dead_end_rejoin(np_)
        if not push_lengths:
            return
        push_lengths.sort()
        unambiguous_len = push_lengths[-1]
        if len(push_lengths) > 1 and push_lengths[-2] == unambiguous_len:
            unambiguous_len += 1  # two with the same length: every branch is ambiguous
        for b in BASES:
            if pushed[b] is None:
                continue
            new_np = np_.copy()
            new_np.new_path = self._alloc()
            new_np.new_path.prev = new_path
            new_np.new_path.min_overlap = new_path.min_overlap
            new_np.new_path.cost = new_path.cost
            new_np.new_path.range = pushed[b]
            self._add_base_to_next_path(new_np, b)
            if pushed[b][2] < unambiguous_len:
                new_np.new_path.cost += o.ambiguous_branch_cost
                new_np.branch_count_since_pair += 1
                if new_np.branch_count_since_pair > o.max_branches_between_pairs:
                    self._stat("exceeded_branch_limit")
                    continue
            if not self._add_rejoins(new_np):
                self._push_next_path(new_np)

    def _add_rejoins(self, np_):  # :675-724
        o = self.o
        p = np_.new_path
        if not self._is_read(p.range):
            return False
        sid = p.range[0]
        best_distance, best_offset = None, -1
        for cand in self.entry_positions.get(sid, ()):
            if cand <= self.cur_offset:
                continue
            distance = abs(self.cur_offset + np_.path_bases - cand)
            if best_distance is None or distance < best_distance:
                best_distance, best_offset = distance, cand
        if best_distance is None:
            return False
        if np_.ambiguous_bases and best_distance > np_.ambiguous_bases:
            self._stat("extend_ambiguous_rejoin")
            return False
        r = Rejoin()
        r.p = p
        r.rejoin_cost = o.rejoin_local_cost + best_distance * o.size_change_cost - o.traverse_ref_cost
        r.right_offset = best_offset
        r.right_anchor_len = p.range[2]
        self._push_rejoin(r)
        return True

    def _add_dead_end_rejoin(self, np_):  # :726-735
        if np_.num_reads <= 1 or not self.o.trace_dead_ends:
            return
        p = np_.new_path
        if p not in self.dead_end_rejoins:
            self.dead_end_rejoins.append(p)
        if p.prev is not None and p.prev.min_overlap == p.min_overlap and p.prev in self.dead_end_rejoins:
            self.dead_end_rejoins.remove(p.prev)

    def _add_pairs_to_next_path(self, np_):  # :737-845
        o, g = self.o, self.g
        p = np_.new_path
        if not self._is_read(p.range):
            return
        sid = p.range[0]
        np_.num_reads += 1
        if np_.reads_until_loop_check:
            np_.reads_until_loop_check -= 1
        else:
            np_.loop_check_seqset_id = sid
            np_.reads_until_loop_check = np_.num_reads // 2
        ra, rb = g.entry_to_index(sid)
        if ra != rb:
            np_.pushed_since_read = 0
            ref_count = g.match_count(sid)
            if ref_count > 1:
                np_.ambiguous_bases = max(np_.ambiguous_bases, p.range[2])
                if not o.trace_ambiguous_ref:
                    p.cost += o.max_cost
                    self._stat("prune_ambiguous_ref")
            elif ref_count == 1:
                p.cost += o.traverse_ref_cost
        matched_pair = False
        added_pairs = added_reads = 0
        for rid in range(ra, rb):
            if not g.has_mate(rid):
                continue
            if added_reads < o.max_pairs_per_read:
                p.seen_read_ids.append(rid)
                added_reads += 1
            if bool(g.is_forward[rid]) == o.forward_pairs_face_inward:
                if matched_pair:
                    continue
                ambiguous_in_ref = np_.ambiguous_bases - np_.path_bases if np_.ambiguous_bases > np_.path_bases else 0
                if self._has_rc_mate_in_range(rid, self.cur_offset + np_.path_bases - o.max_pair_distance,
                                              self.cur_offset - max(0, ambiguous_in_ref)):
                    matched_pair = True
                    continue
                if self._path_has_read_in_range(p, rid, np_.ambiguous_bases, o.max_pair_distance):
                    matched_pair = True
                    continue
            else:
                if added_pairs < o.max_pairs_per_read:
                    p.seen_pairs.append((g.get_rev_comp(g.get_mate(rid)), len(p.seq)))
                    added_pairs += 1
                    self._stat("found_pairs")
                    if added_pairs >= o.max_pairs_per_read:
                        self._stat("too_many_pairs")
                if matched_pair:
                    continue
                if self._has_rc_mate_in_range(rid, self.cur_offset, self.cur_offset + np_.path_bases + o.max_pair_distance):
                    matched_pair = True
                    continue
        if matched_pair:
            np_.pushed_since_pair = 0
            np_.branch_count_since_pair = 0
            np_.ambiguous_bases = 0
            np_.pairs_used += 1
            p.cost += o.pairs_used_cost
            self._stat("matched_pairs")

    def _path_has_read_in_range(self, start_path, read_id, start, limit):  # :1129-1160, as written there
        cur_distance = 0
        cur_path = start_path
        while cur_path is not None:
            last_offset = len(start_path.seq)
            for s_read, s_off in reversed(start_path.seen_pairs):
                assert s_off <= last_offset
                cur_distance += last_offset - s_off
                if cur_distance >= limit:
                    return False
                if cur_distance < start:
                    continue
                if s_read == read_id and cur_distance > 0:
                    return True
            cur_distance += last_offset
            if cur_distance >= limit:
                return False
            cur_path = cur_path.prev
        return False

    def _add_base_to_next_path(self, np_, b):  # :847-877
        o = self.o
        p = np_.new_path
        p.seq += b
        np_.pushed_since_read += 1
        np_.path_bases += 1
        if np_.ambiguous_bases:
            np_.ambiguous_bases += 1
        p.cost += o.base_cost
        assert p.range[2] > np_.pushed_since_read
        overlap = p.range[2] - np_.pushed_since_read
        if overlap < p.min_overlap:
            decrease = p.min_overlap - overlap
            p.cost += o.decrease_overlap_cost * decrease
            p.min_overlap = overlap
            np_.max_between_pairs += decrease * 2
        np_.pushed_since_pair += 1
        if np_.pushed_since_pair > np_.max_between_pairs:
            np_.max_between_pairs += 1
            p.cost += o.increase_max_between_pair_cost
            assert np_.max_between_pairs == np_.pushed_since_pair
        self._add_pairs_to_next_path(np_)

    # ---- queues (:879-941)
    def _worst_rejoin_total(self):
        w = self.rejoin_paths[0]
        return w.p.cost + w.rejoin_cost

    def _push_next_path(self, np_):
        o = self.o
        if np_.new_path.cost > o.max_cost:
            self._stat("max_branch_cost")
            return
        if self.rejoin_paths and len(self.rejoin_paths) >= o.max_rejoins:
            worst = self._worst_rejoin_total() & 0xFFFFFFFF  # "unsigned worst_rejoin_cost" (:888)
            if np_.new_path.cost > worst:
                self._stat("suboptimal_path_prune")
                return
        self.next_paths.append(np_)
        _push_heap(self.next_paths)
        if len(self.next_paths) > o.max_next_paths:
            self.next_paths.pop()
            self._stat("next_paths_too_big")

    def _pop_next_path(self):
        _pop_heap(self.next_paths)
        return self.next_paths.pop()

    def _push_rejoin(self, r):
        o = self.o
        total = r.p.cost + r.rejoin_cost
        if total > o.max_cost:
            self._stat("max_branch_cost")
            return
        # std::multiset with rejoin_comparer (a before b when a's total is larger); equal elements go after their equals
        i = 0
        while i < len(self.rejoin_paths) and self.rejoin_paths[i].p.cost + self.rejoin_paths[i].rejoin_cost >= total:
            i += 1
        self.rejoin_paths.insert(i, r)
        if len(self.rejoin_paths) > o.max_rejoins:
            self.rejoin_paths.pop(0)

    def _trace_next_paths(self):  # :943-1021
        o = self.o
        while self.next_paths:
            np_ = self._pop_next_path()
            if (self.rejoin_paths and len(self.rejoin_paths) >= o.max_rejoins
                    and np_.new_path.cost > self._worst_rejoin_total()):
                self._stat("suboptimal_path_prune")
                continue
            if self.search_step_count > o.max_search_steps_per_read * np_.num_reads + o.initial_search_steps:
                self._stat("search_not_fast_enough")
                continue
            if np_.ambiguous_bases > o.max_ambiguous_bases:
                self._stat("too_many_ambiguous_bases")
                continue
            self._stat("step_count")
            self.search_step_count += 1
            if self.search_step_count > o.max_search_steps:
                self._stat("too_many_steps")
                return
            if self.ambiguous_search_step_count > o.max_ambiguous_search_steps:
                self._stat("too_many_ambiguous_steps")
                return
            self._expand_next_path(np_)

    # ---- output (:1023-1250)
    def _output_rejoins(self):
        o = self.o
        output_count = 0
        while output_count < o.max_rejoins and self.rejoin_paths:
            r = self.rejoin_paths.pop()  # the best one
            if r.p.part_of_assembly:
                continue
            self._output_assembly(r)
            output_count += 1
        if output_count < o.max_rejoins:
            self._output_dead_ends(o.max_rejoins - output_count)

    @staticmethod
    def _dead_end_key(p):
        """dead_end_sorter (:1062-1097) as a sort key: larger min_overlap first, then smaller cost, longer range,
        smaller range begin, part_of_assembly last, seen_pairs, seen_read_ids, seq, then the previous paths."""
        key = []
        while p is not None:
            key.append((-p.min_overlap, p.cost, -p.range[2], p.range[0], p.range[1], 1 if p.part_of_assembly else 0,
                        tuple(p.seen_pairs), tuple(p.seen_read_ids), tuple(CODE[c] for c in p.seq)))
            p = p.prev
        return tuple(key)

    def _output_dead_ends(self, max_to_output):
        dead_ends = sorted(self.dead_end_rejoins, key=self._dead_end_key)
        self.dead_end_rejoins = []
        output_count = 0
        for p in dead_ends:
            if output_count >= max_to_output:
                return
            if p.part_of_assembly:
                continue
            r = Rejoin()
            r.p = p
            r.rejoin_cost = self.o.dead_end_cost + p.cost
            r.anchor_drop = True
            self._output_assembly(r)
            output_count += 1

    def _output_assembly(self, r):
        o = self.o
        out = Assembly()
        out.tags.add("PUSH")
        seq = ""
        cur = r.p
        while cur is not None:
            seq += revcomp(cur.seq)
            cur.part_of_assembly = True
            out.rc_read_ids.update(cur.seen_read_ids)
            cur.seen_read_ids = []
            cur = cur.prev
        out.left_offset = self.cur_offset - len(self.cur_left_anchor)
        out.left_anchor_len = len(self.cur_left_anchor)
        if r.anchor_drop:
            out.right_offset = int(self.cur_offset + len(seq) * o.anchor_drop_size_multiplier)  # aoffset_t conversion
            out.right_anchor_len = 0
            out.score += o.anchor_drop_score
        else:
            out.right_offset = r.right_offset
            out.right_anchor_len = r.right_anchor_len
            assert out.right_offset <= self.sc.end_pos
        out.min_overlap = r.p.min_overlap
        out.trace_steps = self.search_step_count
        out.left_anchor_ambiguous_bases = self.cur_ref_ambiguous_bases
        out.seq = revcomp(seq)
        out.matches_reference = ((out.left_anchor_len + out.right_anchor_len) >= (out.right_offset - out.left_offset)
                                 and (out.right_offset - out.left_offset) == len(out.seq))
        if out.matches_reference:
            out.left_anchor_len = out.right_anchor_len = 0
        max_tot_anchor = min(len(out.seq), out.right_offset - out.left_offset)
        if out.right_anchor_len + out.left_anchor_len > max_tot_anchor:
            assert out.right_anchor_len > 0
            half = int(max_tot_anchor / 2)  # C++ integer division truncates toward zero
            out.right_anchor_len = min(out.right_anchor_len, half)
            out.left_anchor_len = min(out.left_anchor_len, half)
        if r.anchor_drop and self.report_anchor_drop is not None:  # tracer.cpp:1233-1235: anchored on the left only
            self.report_anchor_drop(out, False)
        self._stat("output_count")
        self.emit(out)


def reverse_assembly_in_place(a, graph, ref_end_pos):  # assemble.cpp:933-960
    a.seq = revcomp(a.seq)
    a.left_anchor_len, a.right_anchor_len = a.right_anchor_len, a.left_anchor_len
    a.left_offset, a.right_offset = ref_end_pos - a.right_offset, ref_end_pos - a.left_offset
    a.left_pair_matches, a.right_pair_matches = ([graph.get_rev_comp(r) for r in a.right_pair_matches],
                                                 [graph.get_rev_comp(r) for r in a.left_pair_matches])


class PairCounter:
    """pair_counter.cpp (without calc_score's coverage terms: the tracer computes no coverage here)."""

    def __init__(self, graph, options, emit):
        self.g, self.o, self.emit = graph, options, emit
        self.cur_offset = 0
        self.left = []    # (right_offset, read ids), ascending
        self.active = []  # (right_offset, assembly)
        self.right = []   # (left_offset, assembly)
        self.left_ids = {}
        self.right_ids = {}

    @staticmethod
    def _inc(d, k):
        d[k] = d.get(k, 0) + 1

    @staticmethod
    def _dec(d, k):
        d[k] -= 1
        if not d[k]:
            del d[k]

    def add(self, a):
        while self._advance_some(a.left_offset - self.o.max_pair_distance):
            pass
        for rid in a.rc_read_ids:
            self._inc(self.right_ids, rid)
        bisect.insort(self.right, (a.left_offset, a), key=lambda x: x[0])

    def flush(self):
        while self._advance_some(INT_MAX):
            pass

    def _advance_some(self, target):
        if self.cur_offset >= target:
            return False
        amount = target - self.cur_offset
        if self.left:
            amount = min(amount, self.left[0][0] + self.o.max_pair_distance - self.cur_offset + 1)
        if self.active:
            amount = min(amount, self.active[0][0] - self.cur_offset + 1)
        if self.right:
            amount = min(amount, self.right[0][0] - self.cur_offset + 1)
        assert amount > 0
        self.cur_offset += amount
        self._advance_right()
        self._advance_active()
        self._advance_left()
        return True

    def _matches(self, a, ids, forward):
        g = self.g
        out = []
        for rid in sorted(a.rc_read_ids):
            if not g.has_mate(rid) or bool(g.is_forward[rid]) != forward:
                continue
            if g.get_rev_comp(g.get_mate(rid)) in ids:
                out.append(rid)
        return out

    def _advance_active(self):
        while self.active and self.active[0][0] < self.cur_offset:
            _, a = self.active.pop(0)
            a.right_pair_matches = self._matches(a, self.right_ids, not self.o.forward_pairs_face_inward)
            ids = a.rc_read_ids
            a.rc_read_ids = set()
            for rid in ids:
                self._inc(self.left_ids, rid)
            bisect.insort(self.left, (a.right_offset, ids), key=lambda x: x[0])
            if a.matches_reference and not self.o.trace_reference_assemblies:
                continue
            a.score += a.min_overlap * self.o.min_overlap_score
            a.score += (len(a.left_pair_matches) + len(a.right_pair_matches)) * self.o.pair_match_score
            self.emit(a)

    def _advance_right(self):
        while self.right and self.right[0][0] < self.cur_offset:
            _, a = self.right.pop(0)
            for rid in a.rc_read_ids:
                self._dec(self.right_ids, rid)
            a.left_pair_matches = self._matches(a, self.left_ids, self.o.forward_pairs_face_inward)
            bisect.insort(self.active, (a.right_offset, a), key=lambda x: x[0])

    def _advance_left(self):
        while self.left and self.left[0][0] < self.cur_offset - self.o.max_pair_distance:
            _, ids = self.left.pop(0)
            for rid in ids:
                self._dec(self.left_ids, rid)


def reversable_trace(graph, scaffold, options, rev_comp=False, start_offset=0, limit_offset=None, pair_count=True,
                     anchor_drop=None, report_anchor_drop=None):
    """reversable_tracer::assemble (reversable_tracer.cpp:76-145) for the push tracer:
    tracer -> [anchor_dropper] -> pair_counter -> (reversed back when rev_comp) -> list of assemblies in output order.
    `anchor_drop(scaffold, emit)` builds the anchor_dropper step (oracle/align.py's AnchorDropper; an object with
    add / flush) over the scaffold the tracer runs on; None leaves the step out."""
    ref_end = scaffold.end_pos
    limit_offset = ref_end if limit_offset is None else min(limit_offset, ref_end)
    out = []
    if rev_comp and options.only_trace_forward:
        return out

    def final(a):
        if rev_comp:
            reverse_assembly_in_place(a, graph, ref_end)
        out.append(a)

    pc = PairCounter(graph, options, final) if pair_count else None
    sink = pc.add if pc is not None else final
    run_sc = scaffold.rev_comp() if rev_comp else scaffold
    ad = anchor_drop(run_sc, sink) if anchor_drop is not None else None
    if ad is not None:
        sink = ad.add
    t = Tracer(graph, run_sc, options)
    t.report_anchor_drop = report_anchor_drop  # assemble_options::report_anchor_drop_func, in the TRACER's coordinates
    if rev_comp:
        t.assemble(ref_end - limit_offset, ref_end - start_offset, emit=sink)
    else:
        t.assemble(start_offset, limit_offset, emit=sink)
    if ad is not None:
        ad.flush()
    if pc is not None:
        pc.flush()
    return out
