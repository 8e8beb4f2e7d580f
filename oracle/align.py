"""ORACLE (test infrastructure only; never imported by the product path): a literal CPU restatement of the steps
that follow the push tracer in the reference's assembly pipeline --

  anchor_dropper   modules/variants/anchor_drop.cpp:11-343  (between tracer and pair_counter, reversable_tracer.cpp:124-125)
  aligner          modules/variants/align.cpp:11-385        (first step of add_standard_variants_pipeline, pipeline.cpp:92-95)
  align_splitter   modules/variants/align.cpp:387-460
  split_assembly   modules/variants/assemble.cpp:128-231
  sorter           modules/variants/sort.cpp:6-13           (head of the serial pipeline, pipeline.cpp:48)
  ref_trimmer      modules/variants/trim_ref.cpp:10-113     (pipeline.cpp:52; fully anchored assemblies only)
  deduper          modules/variants/dedup.cpp:7-144         (pipeline.cpp:53)
  normalizer       modules/variants/normalize.cpp:9-118     (pipeline.cpp:96)  } no reference test holds vectors for
  exact_deduper    modules/variants/dedup.cpp:146-216       (pipeline.cpp:97)  } these three: PARITY UNPINNED for them
  vcf_padder       modules/variants/normalize.cpp:128-162   (pipeline.cpp:98)  } (product == this restatement by fuzz)
  the sorted output queue every one of them emits through (assemble.cpp:291-346, canon order :1029-1085)
  scaffold helpers modules/variants/scaffold.cpp:39-203

with the container semantics the results depend on written out: std::upper_bound's probe sequence over the match
sizes (align.cpp:262-278), and libstdc++'s unordered_multimap::equal_range order -- a key's values come back newest
insertion first (bits/hashtable.h _M_insert_multi_node puts the new node in front of its equals; checked with g++ 11
here), which decides ties in find_match (`distance > best` keeps the LATER candidate) and in try_long_rejoin
(`shared <= best` keeps the EARLIER one).

As written in the reference, advance_trail_behind_to advances the READ-AHEAD iterator (anchor_drop.cpp:86), which is
already past the target, so k-mers are never removed from the window table; candidates left of the assembly are
filtered at use (:131).  The restatement keeps that.

The sorter is std::sort there (order among assemblies that compare equal in canon order is unspecified); here and in
the product it is a stable sort.  Half-anchored assemblies (one offset absent) do not occur behind the push tracer --
it reports those through report_half_aligned_func (anchor_drop.cpp:301-329): AnchorDropper's `half_aligned` hook, which the
product's bg_trace_result_half_aligned is compared with (tests/test_gpu_tracer.py); as assemblies they are not modelled.

The ML features (add_ml_features, biograph_variants.cpp:746-779) and the merge counts the dedupers keep are restated
too (ml_features_of, merge_bookkeeping); no reference test or golden holds them: PARITY UNPINNED for those.

Pinned on the vectors of modules/variants/align_test.cpp:59-375, trim_ref_test.cpp:24-112, dedup_test.cpp:24-131
(tests/test_oracle_align.py) and, through the tracer, on the dead-end expectations of tracer_test.cpp:246-316."""
from .tracer import Assembly, Scaffold, _ScaffoldIt, revcomp, INT_MAX, INT_MIN

I64_MAX = 2 ** 63 - 1
ANCHORED_TO_BOTH, ANCHORED_TO_LEFT, ANCHORED_TO_RIGHT = 0, 1, 2


class AlignOptions:
    """The assemble_options fields these steps read (assemble.h:504,520,532,677-682,696)."""

    def __init__(self, **kw):
        self.min_overlap = 100
        self.read_ahead_distance = 100000
        self.min_anchor_drop_overlap = 15
        self.ref_align_factor = 4
        self.max_ref_align_bases = 30
        self.trace_reference_assemblies = False
        self.max_kmer_size = 31  # anchor_dropper::g_max_kmer_size, anchor_drop.cpp:9
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("no align option %r" % k)
            setattr(self, k, v)


def check(cond, what=""):
    """CHECK(...) of the reference: the process aborts there; here the test sees it."""
    if not cond:
        raise AssertionError("CHECK failed: " + what)


# ---- scaffold.cpp ---------------------------------------------------------------------------------------------
def subscaffold(sc, start, length):  # :46-80
    limit = start + length
    check(start >= 0 and length >= 0 and limit >= 0, "subscaffold range")
    ext = []
    for off, s in sc.extents:
        if off >= limit or off + len(s) <= start:
            continue
        rs = max(start - off, 0)
        rl = min(limit - off, len(s))
        ext.append((off + rs - start, s[rs:rl]))
    out = Scaffold(ext, length)
    check(length >= max([o + len(s) for o, s in ext] + [0]), "subscaffold end")
    return out


def is_simple(sc):  # :39
    return len(sc.extents) == 1 and sc.extents[0][0] == 0


def _shared_prefix(a, b):
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    return i


def shared_prefix_length(sc, seq):  # :150-161
    if not sc.extents or sc.extents[0][0] != 0:
        return 0
    return _shared_prefix(seq, sc.extents[0][1])


def split_extent_at(sc, start):  # :189-203
    for off, s in sc.extents:
        if off > start:
            break
        if off + len(s) < start:
            continue
        return s[:start - off], s[start - off:]
    return "", ""


# ---- assembly plumbing ------------------------------------------------------------------------------------------
def copy_assembly(a):
    b = Assembly()
    b.__dict__.update(a.__dict__)
    b.rc_read_ids = set(a.rc_read_ids)
    b.left_pair_matches = list(a.left_pair_matches)
    b.right_pair_matches = list(a.right_pair_matches)
    b.tags = set(a.tags)
    b.aligned_variants = list(getattr(a, "aligned_variants", []))
    return b


def split_assembly(a, split_pos, offset_split_pos):  # assemble.cpp:128-231
    check(0 <= offset_split_pos <= a.right_offset - a.left_offset, "split offset")
    check(0 <= split_pos <= len(a.seq), "split pos")
    left, right = copy_assembly(a), copy_assembly(a)
    left.right_offset = right.left_offset = a.left_offset + offset_split_pos
    left.seq, right.seq = a.seq[:split_pos], a.seq[split_pos:]
    if a.left_anchor_len > split_pos:
        right.left_anchor_len = a.left_anchor_len - split_pos
        left.left_anchor_len = split_pos
    else:
        right.left_anchor_len = 0
    rest = len(a.seq) - split_pos
    if a.right_anchor_len > rest:
        left.right_anchor_len = a.right_anchor_len - rest
        right.right_anchor_len = rest
    else:
        left.right_anchor_len = 0
    for part in (left, right):
        check(part.left_anchor_len + part.right_anchor_len <= len(part.seq), "anchors over seq after split")
        check(part.left_anchor_len + part.right_anchor_len <= part.right_offset - part.left_offset,
              "anchors over ref after split")
    left.aligned_variants, right.aligned_variants = [], []
    seq_offset, ref_offset = 0, a.left_offset
    for v in getattr(a, "aligned_variants", []):
        vl, vr, vs = v
        seq_offset += vl - ref_offset
        ref_offset = vl
        if seq_offset >= len(left.seq) and vl >= right.left_offset:
            right.aligned_variants.append(v)
        else:
            check(seq_offset + len(vs) <= len(left.seq), "var crosses split (seq)")
            check(vr <= left.right_offset, "var crosses split (ref)")
            left.aligned_variants.append(v)
        seq_offset += len(vs)
        ref_offset = vr
    return left, right


def canon_key(a):
    """canon_assembly_order, GRAPH_DISCOVER flavour (assemble.cpp:1046-1082), fully anchored assemblies."""
    return (min(a.left_offset, a.right_offset), a.matches_reference, -max(a.left_offset, a.right_offset), a.seq,
            sorted(a.tags), a.left_anchor_len, a.right_anchor_len, a.score, len(a.left_pair_matches),
            len(a.right_pair_matches), len(a.rc_read_ids))


class SortedOutput:
    """sorted_output_pipeline_step (assemble.cpp:291-346): a multimap in canon order (equal keys keep arrival
    order) drained up to the left offset of the assembly being worked on."""

    def __init__(self, emit):
        self.emit = emit
        self.queue = []
        self.flush_point = INT_MIN  # assemble.h:902
        self.left_offsets = []  # m_left_offsets (a multiset): offsets of assemblies a step still holds back

    def track_left_offset(self, offset):  # :303-306
        check(offset >= self.flush_point, "tracked offset behind the flush point")
        self.left_offsets.append(offset)

    def untrack_left_offset(self, offset):  # :308-312
        check(offset in self.left_offsets, "offset is not tracked")
        self.left_offsets.remove(offset)

    def flush_sorted_to(self, offset):
        check(offset >= self.flush_point, "flush goes backwards")
        if self.left_offsets and offset > min(self.left_offsets):  # :316-321
            offset = min(self.left_offsets)
        while self.queue:
            a = self.queue[0][1]
            if min(a.left_offset, a.right_offset) >= offset:
                return
            self.queue.pop(0)
            self.emit(a)
        self.flush_point = offset

    def sort_and_output(self, a):
        check(min(a.left_offset, a.right_offset) >= self.flush_point, "output behind the flush point")
        k = canon_key(a)
        i = len(self.queue)
        while i > 0 and self.queue[i - 1][0] > k:
            i -= 1
        self.queue.insert(i, (k, a))

    def flush(self):
        self.flush_sorted_to(INT_MAX)


# ---- aligner (align.cpp) ------------------------------------------------------------------------------------------
def find_match(seq, sc, match_len, anchor):  # :290-385
    """-> (seq_match_start, scaffold_match_start) or None."""
    if match_len > sc.end_pos or match_len > len(seq):
        return None
    if anchor == ANCHORED_TO_BOTH:
        seq_anchor, sc_anchor = (len(seq) - match_len) // 2, (sc.end_pos - match_len) // 2
    elif anchor == ANCHORED_TO_LEFT:
        seq_anchor, sc_anchor = 0, 0
    else:
        seq_anchor, sc_anchor = len(seq), sc.end_pos
    where = {}
    for so in range(len(seq) - match_len + 1):
        where.setdefault(seq[so:so + match_len], []).append(so)
    best, found = I64_MAX, None
    for off, s in sc.extents:
        for i in range(len(s) - match_len + 1):
            so_list = where.get(s[i:i + match_len])
            if not so_list:
                continue
            sc_dist = off + i - sc_anchor
            for so in reversed(so_list):  # equal_range: newest insertion first
                d = abs((so - seq_anchor) - sc_dist)
                if d > best:
                    continue
                best, found = d, (so, off + i)
    return found


def find_biggest_match(o, seq, sc, anchor):  # :232-288
    """-> (match_len, seq_match_start, scaffold_match_start) or None, and min_match_size."""
    # C++ integer division of non-negative ints
    min_match = max(len(seq), sc.end_pos) // o.ref_align_factor
    min_match = max(min_match, 1)
    min_match = min(min_match, o.max_ref_align_bases)
    if anchor != ANCHORED_TO_BOTH and min_match < int(o.min_anchor_drop_overlap):
        min_match = int(o.min_anchor_drop_overlap)
    max_match = min(len(seq), sc.end_pos)
    if max_match < min_match:
        return None, min_match
    got = [0, 0, 0]

    def no_match(size):  # the comparator with value = false: true means "go left"
        m = find_match(seq, sc, size, anchor)
        if m is not None:
            if size > got[0]:
                got[:] = [size, m[0], m[1]]
            return False
        return True

    first, n = min_match, max_match + 1 - min_match  # std::upper_bound, bits/stl_algo.h
    while n > 0:
        half = n >> 1
        if no_match(first + half):
            n = half
        else:
            first = first + half + 1
            n = n - half - 1
    if not got[0]:
        check(first == min_match, "upper_bound without a match moved")
        return None, min_match
    check(first == got[0] + 1, "match sizes not monotone")
    return tuple(got), min_match


def find_end_matches(o, seq, sc, max_match_size):  # :178-230
    shortest = min(len(seq), sc.end_pos)
    max_match_size = min(shortest, max_match_size)
    for match_len in range(max_match_size, 0, -1):
        search_len = {1: 2, 2: 3, 3: 5, 4: 7}.get(match_len, match_len * o.ref_align_factor)
        search_len = min(shortest, search_len)
        m = find_match(seq[:search_len], subscaffold(sc, 0, search_len), match_len, ANCHORED_TO_LEFT)
        if m is not None:
            return match_len, m[0], m[1]
        start = len(seq) - search_len  # subseq(size - search_len, match_len), as written
        m = find_match(seq[start:start + match_len], subscaffold(sc, sc.end_pos - search_len, match_len), match_len,
                       ANCHORED_TO_RIGHT)
        if m is not None:
            return match_len, m[0] + len(seq) - search_len, m[1] + sc.end_pos - search_len
    return None


def find_biggest_match_with_ends(o, seq, sc):  # :156-176
    m, min_match = find_biggest_match(o, seq, sc, ANCHORED_TO_BOTH)
    if m is not None:
        return m
    return find_end_matches(o, seq, sc, min_match - 1)


class Aligner(SortedOutput):
    def __init__(self, options, scaffold, emit):
        SortedOutput.__init__(self, emit)
        self.o, self.sc = options, scaffold

    def add(self, a):  # on_assembly :11-98
        self.flush_sorted_to(a.left_offset)
        if a.matches_reference:
            self.sort_and_output(a)
            return
        check(a.left_anchor_len >= 0 and a.right_anchor_len >= 0, "negative anchor")
        s = subscaffold(self.sc, a.left_offset, a.right_offset - a.left_offset)
        max_len = min(len(a.seq), a.right_offset - a.left_offset)
        shared_left = shared_prefix_length(s, a.seq)
        check(shared_left >= a.left_anchor_len, "left anchor does not match reference")
        shared_right = shared_prefix_length(s.rev_comp(), revcomp(a.seq))
        check(shared_right >= a.right_anchor_len, "right anchor does not match reference")
        if shared_left + shared_right > max_len:
            shared_left = max_len - shared_right
        a.left_anchor_len, a.right_anchor_len = shared_left, shared_right
        a.aligned_variants = []
        vl, vr = a.left_offset + shared_left, a.right_offset - shared_right
        work = [(vl, vr, a.seq[shared_left:len(a.seq) - shared_right],
                 subscaffold(s, shared_left, a.right_offset - shared_right - shared_left - a.left_offset))]
        while work:
            self._process(a, work, *work.pop())
        a.aligned_variants.sort()
        if not a.aligned_variants:
            check(len(a.seq) == a.right_offset - a.left_offset, "no variants but lengths differ")
            if not self.o.trace_reference_assemblies:
                return
            a.matches_reference = True
            a.left_anchor_len = a.right_anchor_len = 0
        self.sort_and_output(a)

    def _process(self, a, work, vl, vr, vseq, vs):  # :100-154
        m = find_biggest_match_with_ends(self.o, vseq, vs)
        if m is None:
            if not (vl == vr and not vseq):
                a.aligned_variants.append((vl, vr, vseq))
            return
        match_len, seq_start, sc_start = m
        check(match_len >= 0 and sc_start >= 0, "negative match")
        left = (vl, vl + sc_start, vseq[:seq_start], subscaffold(vs, 0, sc_start))
        check(sc_start + match_len <= vs.end_pos, "match past the scaffold")
        right = (vl + sc_start + match_len, vr, vseq[seq_start + match_len:],
                 subscaffold(vs, sc_start + match_len, vs.end_pos - (sc_start + match_len)))
        for w in (left, right):
            if not (w[0] == w[1] and not w[2]):
                work.append(w)


class AlignSplitter(SortedOutput):  # :387-460
    def add(self, a):
        self.flush_sorted_to(a.left_offset)
        ref_offset, orig_left, orig_right = a.left_offset, a.left_offset, a.right_offset
        for v in list(getattr(a, "aligned_variants", [])):
            vl, vr, vs = v
            check(vl >= orig_left and vr <= orig_right and vl >= ref_offset, "variant outside assembly")
            if vl != ref_offset:
                chunk = vl - ref_offset
                check(len(a.seq) >= chunk, "ref chunk longer than seq")
                left, a = split_assembly(a, chunk, chunk)
                self._set_matches_reference(left)
                self.sort_and_output(left)
                ref_offset = vl
            left, a = split_assembly(a, len(vs), vr - vl)
            check(left.seq == vs, "split variant differs")
            self.sort_and_output(left)
            ref_offset = a.left_offset
        check(ref_offset == a.left_offset, "split lost its place")
        if a.left_offset == a.right_offset and not a.seq:
            return
        self._set_matches_reference(a)
        self.sort_and_output(a)

    @staticmethod
    def _set_matches_reference(a):
        check(len(a.seq) == a.right_offset - a.left_offset, "reference part with a length change")
        a.matches_reference = True
        a.left_anchor_len = a.right_anchor_len = 0


# ---- anchor_dropper (anchor_drop.cpp) --------------------------------------------------------------------------
class AnchorDropper(SortedOutput):
    def __init__(self, options, scaffold, emit, half_aligned=None):
        SortedOutput.__init__(self, emit)
        self.o, self.sc = options, scaffold
        self.half_aligned = half_aligned
        self.kmer_size = min(int(options.min_overlap), options.max_kmer_size)  # :21
        self.kmer_skip_len = int(options.min_overlap) - self.kmer_size  # :24
        self.it = _ScaffoldIt(scaffold)  # m_read_ahead
        self.kmer_needed = 0
        self.window = ""
        self.kmers = {}  # kmer -> offsets in insertion order; equal_range walks them backwards

    def _advance_read_ahead_to(self, offset):  # advance_kmer :30-56 with the inserting callback :65-74
        it = self.it
        while it.offset < offset:
            if it.at_end():
                return
            if it.first_in_extent():
                self.kmer_needed = self.kmer_size - 1
            self.window = (self.window + it.base())[-self.kmer_size:]
            if self.kmer_needed:
                self.kmer_needed -= 1
            else:
                self.kmers.setdefault(self.window, []).append(it.offset + 1 - self.kmer_size)
                self.kmer_needed = self.kmer_skip_len
            it.advance()

    def _advance_range_to(self, trail_behind, read_ahead):  # :76-84
        it = self.it
        target = trail_behind - self.kmer_size
        if it.offset < target and not it.at_end():  # skip_to :58-63
            it.skip_to(target)
            self.kmer_needed = self.kmer_size - 1
        self._advance_read_ahead_to(read_ahead)
        # advance_trail_behind_to(trail_behind) walks m_read_ahead, which is already past it: nothing is removed.

    def _try_long_rejoin(self, a):  # :107-192
        o = self.o
        if len(a.seq) < int(o.min_overlap) + a.left_anchor_len:
            return False
        best_len = best_seq = best_ref = 0
        k = self.kmer_size
        for seq_offset in range(len(a.seq) - k + 1):
            if seq_offset < a.left_anchor_len:
                continue
            for ref_at in reversed(self.kmers.get(a.seq[seq_offset:seq_offset + k], ())):
                if ref_at < a.left_offset + a.left_anchor_len:
                    continue
                max_before = min(seq_offset - a.left_anchor_len, ref_at - a.left_offset - a.left_anchor_len)
                ref_before, ref_after = split_extent_at(self.sc, ref_at)
                seq_before, seq_after = a.seq[a.left_anchor_len:seq_offset], a.seq[seq_offset:]
                before = min(_shared_prefix(ref_before[::-1], seq_before[::-1]), max_before)
                shared = before + _shared_prefix(ref_after, seq_after)
                if shared <= best_len or shared < int(o.min_overlap):
                    continue
                best_len, best_seq, best_ref = shared, seq_offset - before, ref_at - before
        if best_len == 0:
            return False
        check(best_ref > a.left_offset, "long rejoin at the left anchor")
        a.right_offset = best_ref + best_len
        anchored, _ = split_assembly(a, best_seq + best_len, a.right_offset - a.left_offset)
        anchored.right_anchor_len = best_len
        if anchored.right_anchor_len + anchored.left_anchor_len > anchored.right_offset - anchored.left_offset:
            anchored.right_anchor_len = anchored.right_offset - anchored.left_offset - anchored.left_anchor_len
        self.sort_and_output(anchored)
        return True

    def add(self, a):  # on_assembly :194-341
        self.flush_sorted_to(a.left_offset)
        if a.right_anchor_len or a.matches_reference:
            if not a.matches_reference:
                check(a.left_anchor_len, "right anchor without a left one")
            self.sort_and_output(a)
            return
        check(a.left_anchor_len, "dead end without a left anchor")
        self._advance_range_to(a.left_offset, a.left_offset + int(self.o.read_ahead_distance))
        if a.left_anchor_len >= len(a.seq):
            return
        if self._try_long_rejoin(a):
            return
        new = None
        seq_start, ref_start = 0, a.left_offset
        while True:
            vseq = a.seq[seq_start:]
            shared = shared_prefix_length(subscaffold(self.sc, ref_start, len(vseq)), vseq)
            seq_start += shared
            check(seq_start >= a.left_anchor_len, "left anchor does not match reference")
            ref_start += shared
            vseq = a.seq[seq_start:]
            if not vseq:
                break
            vs = subscaffold(self.sc, ref_start, len(vseq) * 5 // 4 + 3)
            m, _ = find_biggest_match(self.o, vseq, vs, ANCHORED_TO_LEFT)
            if m is None:
                break
            match_len, sm, rm = m
            seq_start += sm + match_len
            ref_start += rm + match_len
            new = (seq_start, ref_start - a.left_offset, match_len)
        if new is not None:
            new_seq_len, new_ref_len, new_match_len = new
            a.right_offset = a.left_offset + new_ref_len
            anchored, discard = split_assembly(a, new_seq_len, new_ref_len)
            anchored.right_anchor_len = new_match_len
            self.sort_and_output(anchored)
            if self.half_aligned and discard.seq:
                check(discard.left_anchor_len == 0, "discarded part keeps an anchor")
                self.half_aligned(discard.left_offset, discard.seq)
        elif self.half_aligned:
            self.half_aligned(a.left_offset + a.left_anchor_len, a.seq[a.left_anchor_len:])


# ---- sorter / ref_trimmer / deduper: the head of the serial pipeline (pipeline.cpp:40-57) -------------------------
K_MAX_BACKTRACK_LEN = 1000  # trim_ref.h:31


class RefTrimmer(SortedOutput):  # trim_ref.cpp:10-113
    def __init__(self, scaffold, emit):
        SortedOutput.__init__(self, emit)
        self.sc = scaffold

    def add(self, a):
        self.flush_sorted_to(min(a.left_offset, a.right_offset) - K_MAX_BACKTRACK_LEN)
        if a.matches_reference:
            return
        s = subscaffold(self.sc, a.left_offset, a.right_offset - a.left_offset)
        if not s.extents:
            self.sort_and_output(a)
            return
        shared_left = shared_prefix_length(s, a.seq)
        check(shared_left >= a.left_anchor_len, "left anchor does not match reference")
        shared_right = shared_prefix_length(s.rev_comp(), revcomp(a.seq))
        check(shared_right >= a.right_anchor_len, "right anchor does not match reference")
        max_anchor = min(len(a.seq), a.right_offset - a.left_offset)
        if shared_right + shared_left > max_anchor and shared_right + a.left_anchor_len <= max_anchor:
            shared_left = max_anchor - shared_right
        if shared_right + shared_left > max_anchor and a.right_anchor_len + shared_left <= max_anchor:
            shared_right = max_anchor - shared_left
        if shared_right + shared_left <= max_anchor:
            a.left_anchor_len, a.right_anchor_len = shared_left, shared_right
        new_size = len(a.seq) - a.left_anchor_len - a.right_anchor_len
        check(new_size >= 0, "anchors longer than the sequence")
        a.seq = a.seq[a.left_anchor_len:a.left_anchor_len + new_size]
        a.left_offset += a.left_anchor_len
        a.right_offset -= a.right_anchor_len
        a.left_anchor_len = a.right_anchor_len = 0
        check(a.right_offset >= a.left_offset, "trimmed past itself")
        if new_size == 0 and a.left_offset == a.right_offset:
            return
        self.sort_and_output(a)


class Deduper:  # dedup.cpp:7-144
    def __init__(self, emit):
        self.emit = emit
        self.queued = []  # multimap<left_offset, assembly>: by key, equal keys in arrival order

    @staticmethod
    def _combine(a, target):  # :7-80
        if a.left_offset < target.left_offset:
            a, target = target, a
        add = a.right_offset - target.right_offset
        if add > 0:
            check(add < len(a.seq), "combine adds the whole sequence")
            target.seq += a.seq[len(a.seq) - add:]
            target.right_anchor_len += add
            target.right_offset += add
        target.left_pair_matches = sorted(set(target.left_pair_matches) | set(a.left_pair_matches))
        target.right_pair_matches = sorted(set(target.right_pair_matches) | set(a.right_pair_matches))
        check(target.left_anchor_len >= a.left_anchor_len and target.right_anchor_len >= a.right_anchor_len,
              "combined anchors shrink")
        for f in ("trace_steps", "min_overlap", "left_anchor_ambiguous_bases"):
            setattr(target, f, max(getattr(target, f), getattr(a, f)))
        merge_bookkeeping(target, a)
        return target

    def _advance_to(self, offset):  # :131-141
        while self.queued:
            q = self.queued[0][1]
            if q.left_offset + q.left_anchor_len >= offset:
                break
            self.queued.pop(0)
            self.emit(q)

    def add(self, a):  # :82-129
        self._advance_to(a.left_offset)
        for i, (_, q) in enumerate(self.queued):
            if q.left_offset > a.left_offset + a.left_anchor_len:
                break
            if a.matches_reference or q.matches_reference:
                continue
            if a.left_offset + a.left_anchor_len != q.left_offset + q.left_anchor_len:
                continue
            if a.right_offset - a.right_anchor_len != q.right_offset - q.right_anchor_len:
                continue
            check(len(a.seq) >= a.left_anchor_len + a.right_anchor_len and
                  len(q.seq) >= q.left_anchor_len + q.right_anchor_len, "anchors longer than the sequence")
            if a.seq[a.left_anchor_len:len(a.seq) - a.right_anchor_len] != q.seq[q.left_anchor_len:len(q.seq) - q.right_anchor_len]:
                continue
            self.queued.pop(i)
            a = self._combine(q, a)
            break
        i = len(self.queued)
        while i > 0 and self.queued[i - 1][0] > a.left_offset:
            i -= 1
        self.queued.insert(i, (a.left_offset, a))

    def flush(self):
        self._advance_to(INT_MAX)
        check(not self.queued, "deduper keeps assemblies after its flush")


class Normalizer(SortedOutput):  # normalize.cpp:7-126: shift a variant left while the base it drops equals the one it gains
    def __init__(self, scaffold, emit):
        SortedOutput.__init__(self, emit)
        self.sc = scaffold
        self.cur_offset = 0  # normalize.h:24

    def _advance_to(self, new_offset):  # :120-126
        if new_offset < self.cur_offset:
            return
        self.cur_offset = new_offset
        self.flush_sorted_to(new_offset)

    def add(self, a):
        if a.matches_reference:
            self.sort_and_output(a)
            return
        a.aligned_variants = []
        while True:
            left_ref, after = split_extent_at(self.sc, a.left_offset)
            self._advance_to(a.left_offset - len(left_ref) - 1)
            if not left_ref or len(after) < a.right_offset - a.left_offset:
                break
            asm_ref = after[:a.right_offset - a.left_offset]
            left_base = left_ref[-1]
            check(asm_ref or a.seq, "normalizer: empty variant")
            if asm_ref and a.seq and asm_ref[-1] != a.seq[-1]:
                break
            right_base = asm_ref[-1] if asm_ref else a.seq[-1]
            if left_base != right_base:
                break
            a.left_offset -= 1
            a.right_offset -= 1
            if a.seq:
                a.seq = left_base + a.seq[:-1]
        self.sort_and_output(a)


class ExactDeduper:  # dedup.cpp:146-216
    def __init__(self, emit):
        self.emit = emit
        self.queued = []  # multimap<left_offset, assembly>

    def _advance_to(self, offset):
        while self.queued and self.queued[0][0] < offset:
            self.emit(self.queued.pop(0)[1])

    def add(self, a):
        self._advance_to(a.left_offset)
        for _, q in self.queued:
            if (a.left_offset, a.right_offset, a.seq, a.matches_reference) == (q.left_offset, q.right_offset, q.seq, q.matches_reference):
                q.left_pair_matches = sorted(set(q.left_pair_matches) | set(a.left_pair_matches))
                q.right_pair_matches = sorted(set(q.right_pair_matches) | set(a.right_pair_matches))
                merge_bookkeeping(q, a)
                return
        i = len(self.queued)
        while i > 0 and self.queued[i - 1][0] > a.left_offset:
            i -= 1
        self.queued.insert(i, (a.left_offset, a))

    def flush(self):
        self._advance_to(INT_MAX)
        check(not self.queued, "exact_deduper keeps assemblies after its flush")


class VcfPadder(SortedOutput):  # normalize.cpp:128-162: empty seq or empty ref span gets the reference base to its left
    def __init__(self, scaffold, emit):
        SortedOutput.__init__(self, emit)
        self.sc = scaffold

    def add(self, a):
        if a.left_offset > 3:
            self.flush_sorted_to(a.left_offset - 2)
        if (a.left_offset < a.right_offset and a.seq) or a.left_offset <= 0:
            self.sort_and_output(a)
            return
        s = subscaffold(self.sc, a.left_offset - 1, 1)
        if not is_simple(s):
            self.sort_and_output(a)
            return
        a.left_offset -= 1
        a.seq = s.extents[0][1] + a.seq
        self.sort_and_output(a)


def merge_bookkeeping(target, a):
    """What deduper::combine and exact_deduper keep of the assembly they swallow (dedup.cpp:51-53,71-78,171-183):
    its id and the ids it had merged (here: their number -- NUMASM of the VCF is that number + 1), and the ML
    features of whichever has the longer original sequence."""
    target.merged = getattr(target, "merged", 0) + 1 + getattr(a, "merged", 0)
    tm, am = getattr(target, "ml", None), getattr(a, "ml", None)
    if tm is not None or am is not None:
        check(tm is not None and am is not None, "ML features on one side of a merge only")
        if tm[6] < am[6]:
            target.ml = am


def ml_features_of(a, scaffold):
    """add_ml_features of biograph_variants.cpp:746-779, run on every discovered assembly right behind the sorter
    (report_discovered_assemblies, pipeline.cpp:49-51): (score, refspan, lanch, ranch, refgc, altgc, alt_seq length)
    with the GC fractions in single precision as there."""
    import numpy as np
    refspan = a.right_offset - a.left_offset
    refgc = np.float32(0)
    if refspan:
        gc = sum(s.count("G") + s.count("C") for _, s in subscaffold(scaffold, a.left_offset, refspan).extents)
        refgc = np.float32(gc) / np.float32(refspan)
    altgc = np.float32(0)
    if a.seq:
        altgc = np.float32(a.seq.count("G") + a.seq.count("C")) / np.float32(len(a.seq))
    return (int(a.score), int(refspan), int(a.left_anchor_len), int(a.right_anchor_len), float(refgc), float(altgc), len(a.seq))


def variants_pipeline(options, scaffold, assemblies, through_vcf_pad=False, ml_features=False):
    """The serial pipeline of pipeline.cpp:40-57 + :92-98, fed with everything the tracers of one scaffold produced (both
    directions): sorter(canon) -> ref_trimmer -> deduper -> aligner -> align_splitter [-> normalizer -> exact_deduper ->
    vcf_padder].  -> (deduped, out): the assemblies entering the aligner, and leaving the last step."""
    deduped, out = [], []
    tail = []
    sink = out.append
    if through_vcf_pad:  # pipeline.cpp:96-98: normalizer -> exact_deduper -> vcf_padder
        pad = VcfPadder(scaffold, sink)
        xd = ExactDeduper(pad.add)
        norm = Normalizer(scaffold, xd.add)
        sink, tail = norm.add, [norm.flush, xd.flush, pad.flush]
    split = AlignSplitter(sink)
    al = Aligner(options, scaffold, split.add)

    def save(a):
        deduped.append(copy_assembly(a))
        al.add(a)

    dd = Deduper(save)
    tr = RefTrimmer(scaffold, dd.add)
    for a in sorted((copy_assembly(a) for a in assemblies), key=canon_key):
        if ml_features:
            a.ml, a.merged = ml_features_of(a, scaffold), 0
        tr.add(a)
    tr.flush()
    dd.flush()
    al.flush()
    split.flush()
    for f in tail:
        f()
    return deduped, out


def align_pipeline(options, scaffold, assemblies, half_aligned=None):
    """The chain of align_test.cpp:31-52: anchor_dropper -> aligner -> (copy) -> align_splitter -> output.
    -> (aligned, split): the assemblies as they leave the aligner, and as they leave the splitter."""
    aligned, out = [], []
    split = AlignSplitter(out.append)

    def save(a):
        aligned.append(copy_assembly(a))
        split.add(a)

    al = Aligner(options, scaffold, save)
    ad = AnchorDropper(options, scaffold, al.add, half_aligned)
    for a in assemblies:
        ad.add(copy_assembly(a))
    ad.flush()  # destruction order of the unique_ptr chain: each step flushes into the next
    al.flush()
    split.flush()
    return aligned, out
