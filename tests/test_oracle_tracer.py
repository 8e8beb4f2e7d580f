"""The oracle's restatement of the production push tracer (oracle/tracer.py) against the vectors of the reference's
own test, modules/variants/tracer_test.cpp:167-304 (fwd_tracer_test and rev_tracer_test instantiations), transcribed:
reference = create_reference({tseq(a..z), tseq(A..M_12345_N..Z), tseq(0..9)}) plus the four single-base scaffolds of
reference_testutil.cpp:67-75, min_overlap = 2 tseq letters, trace_reference_assemblies on.  CPU only.

This file runs the chain WITHOUT the anchor_dropper between tracer and pair_counter: assemblies that rejoin the
reference pass through it unchanged (anchor_drop.cpp:196-202) -- every expectation of the first five cases; for the
three dead_end cases it checks the raw dead-end assembly the tracer hands over.  Their full expectations, with the
anchor_dropper of oracle/align.py plugged in, are in tests/test_oracle_align.py."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import tracer as T

tseq, tseq_rc = O.tseq, O.tseq_rc
REF = [tseq("abcdefghijklmnopqrstuvwxyz"), tseq("ABCDEFGHIJKLM_12345_NOPQRSTUVWXYZ"), tseq("0123456789"), "A", "C", "G", "T"]


def world(pairs, reads=()):
    """use_reads of tracer_test.cpp:60-82: seqset_for_reads(all reads), readmap_for_reads(pairs, then unpaired), ref_map."""
    all_reads = [r for p in pairs for r in p] + list(reads)
    seq, offs = O.pack_strings(all_reads)
    ent = O.Entries.build(seq, offs, "simple")
    ss = O.Seqset.from_entries(ent)
    rm = O.Readmap(ss, ent, [tuple(p) for p in pairs] + [(r,) for r in reads], True)
    codes = np.concatenate([np.frombuffer(s.encode(), np.uint8) for s in REF])
    codes = np.array([b"ACGT".index(bytes([c])) for c in codes], np.uint8)
    starts = np.cumsum([0] + [len(s) for s in REF[:-1]]).astype(np.uint64)
    sizes = np.array([len(s) for s in REF], np.uint64)
    return T.Graph(ss, rm, ss.ref_map_build(codes, starts, sizes))


def dropper(o, **align_opts):
    """The anchor_dropper of oracle/align.py with the tracer's options, as reversable_trace's anchor_drop argument."""
    from oracle import align as A
    ao = A.AlignOptions(min_overlap=o.min_overlap, read_ahead_distance=o.read_ahead_distance,
                        trace_reference_assemblies=bool(o.trace_reference_assemblies), **align_opts)
    return lambda sc, emit: A.AnchorDropper(ao, sc, emit)


def run(parts, pairs, reads=(), rev_comp=False, drop=None, **opts):
    """drop: None = no anchor_dropper (the raw tracer output); a dict = AlignOptions overrides for the step."""
    g = world(pairs, reads)
    o = T.Options(**dict(dict(min_overlap=20, trace_reference_assemblies=True), **opts))
    sc = T.Scaffold(parts)
    out = T.reversable_trace(g, sc, o, rev_comp=rev_comp, anchor_drop=None if drop is None else dropper(o, **drop))
    if rev_comp:  # tracer_test.cpp:128-133
        out.reverse()
    for a in out:  # the lambda of tracer_test.cpp:94-113
        assert len(a.seq) > a.left_anchor_len and len(a.seq) > a.right_anchor_len
        if a.left_anchor_len:
            assert a.seq[:a.left_anchor_len] == sc.sub(a.left_offset, a.left_anchor_len)
        if a.right_anchor_len:
            assert a.seq[len(a.seq) - a.right_anchor_len:] == sc.sub(a.right_offset - a.right_anchor_len, a.right_anchor_len)
    return out


def is_ref(a, s):  # RefTrAssemblyIs
    return (a.right_offset - a.left_offset == len(s) and a.left_anchor_len == 0 and a.right_anchor_len == 0
            and a.matches_reference and a.seq == s)


def is_var(a, left, mid, right):  # TrAssemblyIs
    return (len(a.seq) > a.left_anchor_len + a.right_anchor_len and a.left_anchor_len == len(left)
            and a.right_anchor_len == len(right) and a.seq == left + mid + right)


def expect(out, want, ordered=True):
    assert len(out) == len(want), out
    pool = list(out)
    for i, w in enumerate(want):
        cands = [out[i]] if ordered else pool
        hit = [a for a in cands if (is_ref(a, w) if isinstance(w, str) else is_var(a, *w))]
        assert hit, (i, w, out)
        if not ordered:
            pool.remove(hit[0])


DIRS = [False, True]


@pytest.mark.parametrize("rev", DIRS)
def test_only_reference(rev):
    out = run([(5, tseq("abcdefghijklmnopqrst"))],
              [(tseq("abcde"), tseq_rc("ijklm")), (tseq("cdefg"), tseq_rc("klmno")), (tseq("efghi"), tseq_rc("mnopq")),
               (tseq("ghijk"), tseq_rc("opqrs"))], rev_comp=rev)
    expect(out, [tseq(x) for x in ("abcdefg", "cdefghi", "efghijk", "ghijklm", "ijklmno", "klmnopq", "mnopqrs")])
    assert [len(a.right_pair_matches) for a in out] == [1, 1, 1, 0, 0, 0, 0]
    assert [len(a.left_pair_matches) for a in out] == [0, 0, 0, 0, 1, 1, 1]


@pytest.mark.parametrize("rev", DIRS)
def test_simple_variant(rev):
    out = run([(0, tseq("abcdefghijklmnopqrstuvw"))],
              [(tseq("bcdef"), tseq_rc("klmno")), (tseq("efgHi"), tseq_rc("nopqr")), (tseq("gHijk"), tseq_rc("qrstu")),
               (tseq("Hijkl"), tseq_rc("stuvw"))], rev_comp=rev)
    expect(out, [(tseq("bcdef"), tseq("gHij"), tseq("klmno")), tseq("klmnopqr"), tseq("nopqrstu"), tseq("qrstuvw")])
    assert [len(a.right_pair_matches) for a in out] == [1, 0, 0, 0]
    assert [len(a.left_pair_matches) for a in out] == [0, 0, 0, 1]


@pytest.mark.parametrize("rev", DIRS)
def test_compound_variant(rev):
    out = run([(0, tseq("abcdefghijklmnopqrstuvw"))],
              [(tseq("bcdef"), tseq_rc("ghijk")), (tseq("defgh"), tseq_rc("ijklm")), (tseq("cdefg"), tseq_rc("lmnop")),
               (tseq("efgHi"), tseq_rc("nopqr")), (tseq("gHijk"), tseq_rc("qrstu")), (tseq("Hijkl"), tseq_rc("stuvw"))],
              rev_comp=rev)
    expect(out, [tseq("bcdefg"), tseq("cdefgh"), tseq("defghijk"), tseq("ghijklm"), tseq("ijklmnop"), tseq("lmnopqr"),
                 tseq("nopqrstu"), tseq("qrstuvw"), (tseq("cdefg"), tseq("H"), tseq("ijklm"))], ordered=False)


@pytest.mark.parametrize("rev", DIRS)
def test_cross_ref_sections(rev):
    out = run([(0, tseq("abcdef")), (len(tseq("abcdef")) + len(tseq("gh")), tseq("ijklmnopqr"))],
              [(tseq("abcde"), tseq_rc("ghijk")), (tseq("cdefg"), tseq_rc("ijklm")), (tseq("efghi"), tseq_rc("klmno"))],
              rev_comp=rev)
    expect(out, [(tseq("abcde"), tseq("fgh"), tseq("ijklm")), tseq("ijklmno")])
    assert [len(a.right_pair_matches) for a in out] == [0, 0]  # "these pairs aren't far enough apart to get counted"
    assert [len(a.left_pair_matches) for a in out] == [0, 0]


@pytest.mark.parametrize("rev", DIRS)
def test_cross_ref_delete(rev):
    out = run([(0, tseq("abcdefg")), (100, tseq("hijklmno"))],
              [(tseq("abcde"), tseq_rc("cdefg")), (tseq("efghi"), tseq_rc("ghijk")), (tseq("ijklm"), tseq_rc("klmno"))],
              rev_comp=rev)
    expect(out, [tseq("abcdefg"), (tseq("cdefg"), tseq("h"), tseq("ijklm")), tseq("ijklmno")])
    assert [len(a.right_pair_matches) for a in out] == [0, 0, 0]
    assert [len(a.left_pair_matches) for a in out] == [0, 0, 0]


def reads_for_seq(seq, read_length, read_distance):  # assemble_testutil.cpp:73-88
    out, i = [], 0
    while i + read_length <= len(seq):
        out.append(seq[i:i + read_length])
        i += read_distance
    out.append(seq[len(seq) - read_length:])
    return out


DEAD = tseq("abcde") + "G" + tseq("f") + "G" + tseq("ghi") + "G" + tseq("j")


@pytest.mark.parametrize("rev", DIRS)
def test_dead_end(rev):
    """tracer_test.cpp:246-266 (dead_end) and :268-289 (dead_end2: differs only in the anchor_dropper's option)."""
    out = run([(0, tseq("abcdefghijklmnopqrstuvwxyz"))], [], reads_for_seq(DEAD, len(tseq("abcde")), len(tseq("a"))),
              rev_comp=rev, min_overlap=len(tseq("abcd")))
    if rev:
        assert out == []  # "this dead end is only reachable going forward"
        return
    assert len(out) == 1
    a = out[0]
    # what the tracer hands to the anchor_dropper: anchored on the left only, carrying the whole dead-end path
    assert a.right_anchor_len == 0 and a.left_anchor_len == len(tseq("abcde")) and a.left_offset == 0
    assert a.seq == DEAD and not a.matches_reference
    assert a.right_offset == int(len(tseq("abcde")) + len(DEAD) * 1.5)


@pytest.mark.parametrize("rev", DIRS)
def test_dead_end_big_drop(rev):
    out = run([(0, tseq("abcdefghijklmnopqrstuvwxyz"))],
              [(tseq("bcdef"), tseq("efgH") + "G" + tseq("y")), (tseq("defgH") + "G", tseq("fgH") + "G" + tseq("yz"))],
              rev_comp=rev)
    if rev:
        assert out == []
        return
    assert len(out) == 1
    a = out[0]
    assert a.right_anchor_len == 0 and a.seq.startswith(tseq("bcdef") + tseq("gH") + "G" + tseq("y"))
    assert a.left_anchor_len == len(tseq("bcdef")) and a.left_offset == len(tseq("a"))


def test_next_path_heap_is_libstdcxx():
    """The order in which equal-cost paths leave the queue is part of the result: _push_heap/_pop_heap must move
    elements exactly as bits/stl_heap.h does.  Checked against a direct transcription of the textbook algorithm's
    observable property (heap order) and against a recorded sequence from std::push_heap/pop_heap (g++ 11)."""
    class P:
        def __init__(self, c, tag):
            self.new_path = type("X", (), {"cost": c})()
            self.tag = tag
    # recorded with a 12-line C++ program (costs in push order; pops interleaved as marked)
    costs = [5, 3, 5, 1, 3, 5, 1, 4, 5, 3]
    h, order = [], []
    for i, c in enumerate(costs):
        h.append(P(c, i))
        T._push_heap(h)
    while h:
        T._pop_heap(h)
        order.append(h.pop().tag)
    assert order == [3, 6, 9, 1, 4, 7, 2, 5, 0, 8]
