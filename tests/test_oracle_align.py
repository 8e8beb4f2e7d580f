"""The oracle's restatement of anchor_dropper -> aligner -> align_splitter (oracle/align.py) against the vectors of the
reference's own test, modules/variants/align_test.cpp:59-375, transcribed (tseq letters are 10 bases each,
dna_testutil), plus the scaffold helpers against scaffold_test-style hand cases.  CPU only."""
import pytest

from oracle import align as A
from oracle import oracle as O
from oracle import tracer as T

tseq = O.tseq


def asm(left, right, seq, la, ra):
    a = T.Assembly()
    a.left_offset, a.right_offset, a.seq, a.left_anchor_len, a.right_anchor_len = left, right, seq, la, ra
    return a


def is_ref(a, left, right):  # RefAssemblyIs, assemble_testutil.h:208-212
    return a.matches_reference and (a.left_offset, a.right_offset, a.left_anchor_len, a.right_anchor_len) == (left, right, 0, 0)


def is_var(a, left, seq, right):  # AssemblyIs, assemble_testutil.h:200-206
    return (not a.matches_reference) and (a.left_offset, a.right_offset, a.seq) == (left, right, seq)


def run(parts, asms, **opts):
    sc = T.Scaffold(parts)
    aligned, out = A.align_pipeline(A.AlignOptions(**opts), sc, asms)
    for x, y in zip(out, out[1:]):  # expect_sorted(left_offset_less_than)
        assert x.left_offset <= y.left_offset
    for a in out:  # test_output of assemble_testutil.h:27-70
        assert a.right_offset >= a.left_offset and (a.seq or a.right_offset > a.left_offset)
        if a.matches_reference:
            sub = A.subscaffold(sc, a.left_offset, a.right_offset - a.left_offset)
            assert A.is_simple(sub) and sub.extents[0][1] == a.seq
    return aligned, out


def test_simple():
    orig, variant = "T", "AA"
    ref = tseq("abc") + orig + tseq("def")
    aligned, out = run([(0, ref)], [asm(0, len(ref), tseq("abc") + variant + tseq("def"), 20, 20)])
    assert [a.aligned_variants for a in aligned] == [[(30, 31, variant)]]
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, variant, 31) and is_ref(out[2], 31, 61)


@pytest.mark.parametrize("kmer", [1, 2, 3, 4, 20, 31])
def test_drop_right(kmer):
    orig, variant = "T", "AA"
    ref = tseq("abc") + orig + tseq("defdefdef")
    aligned, out = run([(0, ref)], [asm(0, len(ref), tseq("abc") + variant + tseq("d"), 20, 0)],
                       min_anchor_drop_overlap=10, max_kmer_size=kmer)
    assert [a.aligned_variants for a in aligned] == [[(30, 31, variant)]]
    assert (aligned[0].left_offset, aligned[0].right_offset) == (0, 41)
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, variant, 31) and is_ref(out[2], 31, 41)


@pytest.mark.parametrize("kmer", [1, 2, 3, 4, 20, 31])
def test_drop_right_short(kmer):
    ref = tseq("abcdefghijklmnopqrstuvwxyz")
    seq = tseq("abcd") + "G" + tseq("yz")
    aligned, out = run([(0, ref)], [asm(0, len(seq), seq, 40, 0)], min_overlap=20, max_kmer_size=kmer)
    assert [a.aligned_variants for a in aligned] == [[(40, 240, "G")]]
    assert (aligned[0].left_offset, aligned[0].right_offset, aligned[0].right_anchor_len) == (0, 260, 20)


def test_two_snps():
    orig, variant = "TAT", "GAG"
    ref = tseq("abc") + orig + tseq("def")
    aligned, out = run([(5, ref)], [asm(5, 5 + len(ref), tseq("abc") + variant + tseq("def"), 1, 1)])
    assert [a.aligned_variants for a in aligned] == [[(35, 36, "G"), (37, 38, "G")]]
    assert len(out) == 5
    assert is_ref(out[0], 5, 35) and is_var(out[1], 35, "G", 36) and is_ref(out[2], 36, 37)
    assert is_var(out[3], 37, "G", 38) and is_ref(out[4], 38, 68)


def test_multi_extent():
    aligned, out = run([(0, tseq("abc")), (100, tseq("ghi"))], [asm(0, 130, tseq("abcdefghi"), 1, 1)])
    assert [a.aligned_variants for a in aligned] == [[(30, 100, tseq("def"))]]
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, tseq("def"), 100) and is_ref(out[2], 100, 130)


def test_insert_base():
    aligned, out = run([(0, tseq("abcdef"))], [asm(0, 60, tseq("abc") + "T" + tseq("def"), 1, 1)])
    assert [a.aligned_variants for a in aligned] == [[(30, 30, "T")]]
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, "T", 30) and is_ref(out[2], 30, 60)


def test_delete_base():
    aligned, out = run([(0, tseq("abc") + "T" + tseq("def"))], [asm(0, 61, tseq("abcdef"), 1, 1)])
    assert [a.aligned_variants for a in aligned] == [[(30, 31, "")]]
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, "", 31) and is_ref(out[2], 31, 61)


def test_sorts_by_left():
    """align_test.cpp:287-312 builds two overlapping assemblies but, as written there, never runs them (and the
    second one's offsets do not fit its sequence).  The same idea with consistent offsets: the output of the chain
    stays sorted by left offset although the second assembly overlaps the tail of the first."""
    ref = tseq("abc") + "T" + tseq("def") + "T" + tseq("ghi") + "T" + tseq("jkl")
    a1 = asm(0, 30 + 1 + 30 + 1 + 30, tseq("abc") + "C" + tseq("def") + "C" + tseq("ghi"), 1, 1)
    a2 = asm(62, 123, tseq("ghi") + "C" + tseq("jkl"), 1, 1)
    aligned, out = run([(0, ref)], [a1, a2])
    assert [a.aligned_variants for a in aligned] == [[(30, 31, "C"), (61, 62, "C")], [(92, 93, "C")]]
    assert [(a.left_offset, a.right_offset, a.matches_reference) for a in out] == [
        (0, 30, True), (30, 31, False), (31, 61, True), (61, 62, False), (62, 92, True), (62, 92, True), (92, 93, False),
        (93, 123, True)]


def test_wild_aligns_smaller_near_ends():
    pre = ("TGTGATAGTAGATCAGAGGTGGAGACATCAACGTAAACTTATGTTTAGTTTAATATAGACACACACAGTTCT"
           "ACATAGAAAACTTTATAATTAGGTGTGT")
    post = ("ACCATTCTGAAAGGAATCAGGCTCTTTGAAGAAATGTCTGATACTAGAACTGGGACAGTAAATATAGGAGCC"
            "AGGATAATCTGGAAGTATCAGAAAGTAAGTACTAAAAAAATTAAAATATATCAAACAAAAATAAAAGCCAAT"
            "AAAA")
    parts = [("G", "A"), ("TAGGTAGGTTAGACAC",) * 2, ("G", "A"), ("CACATATACTTCCTAGCATTGC",) * 2, ("T", "C"),
             ("AATGAGGGA",) * 2, ("C", "G"), ("AAGATACAATGTGC",) * 2, ("", "TC"), ("ATTCAGCAGCCA",) * 2, ("C", "G"),
             ("ATGTAAGTTTTCC",) * 2, ("C", "T")]
    parts = [(pre, pre)] + parts + [(post, post)]
    ref = var = ""
    want = []
    for r, v in parts:
        if r != v:
            want.append((len(ref), len(ref) + len(r), v))
        ref += r
        var += v
    aligned, out = run([(0, ref)], [asm(0, len(ref), var, len(pre), len(post))])
    assert [a.aligned_variants for a in aligned] == [want]


# ---- helpers ----------------------------------------------------------------------------------------------------
def test_subscaffold_and_friends():
    sc = T.Scaffold([(2, "ACGT"), (10, "TTGCA")], 20)
    s = A.subscaffold(sc, 3, 9)
    assert s.extents == [(0, "CGT"), (7, "TT")] and s.end_pos == 9
    assert A.shared_prefix_length(s, "CGA") == 2 and A.shared_prefix_length(A.subscaffold(sc, 0, 5), "AC") == 0
    assert A.is_simple(A.subscaffold(sc, 10, 3)) and not A.is_simple(s) and not A.is_simple(A.subscaffold(sc, 1, 3))
    assert A.split_extent_at(sc, 12) == ("TT", "GCA") and A.split_extent_at(sc, 6) == ("ACGT", "")
    assert A.split_extent_at(sc, 8) == ("", "") and A.split_extent_at(sc, 10) == ("", "TTGCA")
    assert s.rev_comp().extents == [(0, "AA"), (6, "ACG")]


def test_find_match_ties_take_the_later_candidate():
    """`distance > best` skips only strictly worse candidates, so of equally distant ones the last visited wins:
    scaffold offsets ascending, and within one scaffold offset the seq offsets newest-first (descending)."""
    sc = T.Scaffold([(0, "ACACAC")])
    # seq "ACAC": "AC" at 0, 2; scaffold "AC" at 0, 2, 4.  Left anchored distance = |so - ro|: 0 for (0,0), (2,2).
    assert A.find_match("ACAC", sc, 2, A.ANCHORED_TO_LEFT) == (2, 2)
    # both-anchored: seq anchor (4-2)//2 = 1, scaffold anchor (6-2)//2 = 2: distance |so-1 - (ro-2)|, 1 at best;
    # the last visited with distance 1 is ro=4, so: so=2 gives |1-2| = 1 first, then so=0: |-1-2| = 3.
    assert A.find_match("ACAC", sc, 2, A.ANCHORED_TO_BOTH) == (2, 4)


def test_split_assembly_anchors_and_vars():
    a = asm(100, 130, "ACGTACGTAC" * 3, 12, 5)
    a.aligned_variants = [(105, 106, "C"), (120, 121, "A")]
    l, r = A.split_assembly(a, 10, 10)
    assert (l.left_offset, l.right_offset, r.left_offset, r.right_offset) == (100, 110, 110, 130)
    assert (l.left_anchor_len, l.right_anchor_len, r.left_anchor_len, r.right_anchor_len) == (10, 0, 2, 5)
    assert l.aligned_variants == [(105, 106, "C")] and r.aligned_variants == [(120, 121, "A")]
    with pytest.raises(AssertionError):
        A.split_assembly(a, 10, 31)


# ---- tracer -> anchor_dropper -> pair_counter: the dead-end expectations of tracer_test.cpp:246-316 -------------
from tests.test_oracle_tracer import DEAD, expect, reads_for_seq, world  # noqa: E402

ALPHABET = [(0, tseq("abcdefghijklmnopqrstuvwxyz"))]


def traced(parts, pairs, reads, rev, min_overlap, **align_opts):
    g = world(pairs, reads)
    o = T.Options(min_overlap=min_overlap, trace_reference_assemblies=True)
    ao = A.AlignOptions(min_overlap=min_overlap, read_ahead_distance=o.read_ahead_distance,
                        trace_reference_assemblies=True, **align_opts)
    out = T.reversable_trace(g, T.Scaffold(parts), o, rev_comp=rev,
                             anchor_drop=lambda sc, emit: A.AnchorDropper(ao, sc, emit))
    if rev:
        out.reverse()
    return out


@pytest.mark.parametrize("rev", [False, True])
def test_tracer_dead_end(rev):  # :246-266
    out = traced(ALPHABET, [], reads_for_seq(DEAD, 50, 10), rev, 40, min_anchor_drop_overlap=10)
    expect(out, [] if rev else [(tseq("abcde"), "G" + tseq("f") + "G" + tseq("ghi") + "G", tseq("j"))])


@pytest.mark.parametrize("rev", [False, True])
def test_tracer_dead_end2(rev):  # :268-289: min_anchor_drop_overlap too long to match the "j"
    out = traced(ALPHABET, [], reads_for_seq(DEAD, 50, 10), rev, 40, min_anchor_drop_overlap=11)
    expect(out, [] if rev else [(tseq("abcde"), "G" + tseq("f") + "G", tseq("ghi"))])


@pytest.mark.parametrize("rev", [False, True])
def test_tracer_dead_end_big_drop(rev):  # :293-316: the long rejoin, anchor of length min_overlap
    pairs = [(tseq("bcdef"), tseq("efgH") + "G" + tseq("y")), (tseq("defgH") + "G", tseq("fgH") + "G" + tseq("yz"))]
    out = traced(ALPHABET, pairs, [], rev, 20)
    expect(out, [] if rev else [(tseq("bcdef"), tseq("gH") + "G", tseq("yz"))])
    if not rev:
        assert (out[0].left_offset, out[0].right_offset) == (10, 260)


# ---- ref_trimmer (trim_ref_test.cpp:24-112) and deduper (dedup_test.cpp:24-131) ------------------------------------
def trim(parts, asms):
    out = []
    tr = A.RefTrimmer(T.Scaffold(parts), out.append)
    for a in asms:
        tr.add(a)
    tr.flush()
    return out


def dedup(asms):
    out = []
    dd = A.Deduper(out.append)
    for a in asms:
        dd.add(A.copy_assembly(a))
    dd.flush()
    for x, y in zip(out, out[1:]):
        assert x.left_offset <= y.left_offset
    return out


def test_trim_ref_ignores_reference_and_reference_equivalents():
    a = asm(0, 30, tseq("abc"), 0, 0)
    a.matches_reference = True
    assert trim(ALPHABET, [a]) == []
    assert trim(ALPHABET, [asm(0, 30, tseq("abc"), 0, 0)]) == []


def test_trim_ref_expands():
    out = trim(ALPHABET, [asm(30, 140, tseq("de") + "G" + tseq("FGHI") + "G" + tseq("jklmn"), 10, 20)])
    assert len(out) == 1 and is_var(out[0], 50, "G" + tseq("FGHI") + "G", 90)
    assert (out[0].left_anchor_len, out[0].right_anchor_len) == (0, 0)


def test_trim_ref_duplicate_insert_and_delete_call_leftmost():
    ref = tseq("abc") + "G" + tseq("defghijklmnopqrstuvwxyz")
    out = trim([(0, ref)], [asm(0, 91, tseq("abc") + "GG" + tseq("defghi"), 1, 1)])
    assert len(out) == 1 and is_var(out[0], 30, "G", 30) and (out[0].left_anchor_len, out[0].right_anchor_len) == (0, 0)
    ref = tseq("abc") + "GG" + tseq("defghijklmnopqrstuvwxyz")
    out = trim([(0, ref)], [asm(0, 92, tseq("abc") + "G" + tseq("defghi"), 1, 1)])
    assert len(out) == 1 and is_var(out[0], 30, "", 31) and (out[0].left_anchor_len, out[0].right_anchor_len) == (0, 0)


def test_dedup_duplicate_and_non_duplicate():
    a = asm(1, 5, tseq("abc"), 0, 0)
    out = dedup([a, a])
    assert len(out) == 1 and is_var(out[0], 1, tseq("abc"), 5)
    b = asm(1, 5, tseq("def"), 0, 0)
    c = asm(1, 6, tseq("def"), 0, 0)
    d = asm(2, 6, tseq("def"), 0, 0)
    out = dedup([a, b, b, c, d])
    assert sorted((x.left_offset, x.seq, x.right_offset) for x in out) == sorted(
        [(1, tseq("abc"), 5), (1, tseq("def"), 5), (1, tseq("def"), 6), (2, tseq("def"), 6)])
    assert out[-1].left_offset == 2


@pytest.mark.parametrize("order", ["overlapping", "encompassing"])
def test_dedup_overlapping_and_encompassing(order):
    if order == "overlapping":
        asms = [asm(0, 40, tseq("ab") + "G" + tseq("d"), 20, 10), asm(10, 50, tseq("b") + "G" + tseq("de"), 10, 20)]
    else:
        asms = [asm(0, 50, tseq("ab") + "G" + tseq("de"), 20, 20), asm(10, 40, tseq("b") + "G" + tseq("d"), 10, 10)]
    out = dedup(asms)
    assert len(out) == 1 and is_var(out[0], 0, tseq("ab") + "G" + tseq("de"), 50)
    assert (out[0].left_anchor_len, out[0].right_anchor_len) == (20, 20)


def test_variants_pipeline_merges_the_two_directions():
    """What the serial pipeline does with a forward and a reverse discovery of the same SNP plus the tracers'
    reference assemblies: the reference assembly is dropped by the ref_trimmer, the two discoveries trim to the same
    anchor-free variant and are merged (pair matches united), and nothing is left for the splitter to cut."""
    ref = tseq("abcdefgh")
    var = tseq("abc") + "T" + tseq("defgh")[1:]
    f = asm(10, 70, var[10:70], 10, 10)
    r = asm(0, 80, var, 25, 30)
    r.left_pair_matches = [7]
    refa = asm(0, 40, ref[:40], 0, 0)
    refa.matches_reference = True
    deduped, out = A.variants_pipeline(A.AlignOptions(), T.Scaffold([(0, ref)]), [r, refa, f])
    assert [(a.left_offset, a.right_offset, a.seq, a.left_pair_matches) for a in deduped] == [(30, 31, "T", [7])]
    assert [(a.left_offset, a.right_offset, a.matches_reference) for a in out] == [(30, 31, False)]


# ---- end to end: the vectors of modules/variants/pipeline_test.cpp:158-213 --------------------------------------
# trace_ref (one direction) -> sorter -> ref_trimmer -> deduper -> aligner -> [ploid_limiter] -> align_splitter.
# The ploid_limiter of that test (max_ploids = 2; not a step of the production pipeline, pipeline.cpp:92-105) is left
# out: no case here has more than two alleles at a site, and the expectations are about the variant sections only.
import numpy as np  # noqa: E402

PIPE_REF = [tseq("abc"), tseq("defg") + "G" + tseq("hijkl") + "GG" + tseq("mnopqrstuvwxyz") + tseq("ABCD") + "A" +
            tseq("EFGH") + "T" + tseq("EFGHIJKL")]
PIPE_PARTS = [(0, PIPE_REF[0]), (30 + 100, PIPE_REF[1])]
DEF = 130
PIPE_CASES = {
    "ref_only": ([tseq("mnopq"), tseq("opqrst"), tseq("rstuv"), tseq("uvwxy")], []),
    "homozygous_snp": ([tseq("defg"), tseq("fg") + "C" + tseq("hi"), tseq("hijk")], [(DEF + 40, "C", DEF + 41)]),
    "hetrozygous_snp": ([tseq("defg"), tseq("fg") + "A" + tseq("hi"), tseq("fg") + "G" + tseq("hi"), tseq("hijk")],
                        [(DEF + 40, "A", DEF + 41)]),
    "compound_hetrozygous_snp": ([tseq("defg"), tseq("fg") + "A" + tseq("hi"), tseq("fg") + "T" + tseq("hi"), tseq("hijk")],
                                 [(DEF + 40, "A", DEF + 41), (DEF + 40, "T", DEF + 41)]),
    "interscaffold_delete": ([tseq("abc"), tseq("bcd"), tseq("cde"), tseq("def"), tseq("efg")], [(30, "", DEF)]),
    "interscaffold_insert": ([tseq("abc"), tseq("bc") + "C" + tseq("d"), tseq("c") + "C" + tseq("de"), tseq("def"), tseq("efg")],
                             [(30, "C", DEF)]),
}


def pipe_world(reads, reference=PIPE_REF):
    seq, offs = O.pack_strings(list(reads))
    ent = O.Entries.build(seq, offs, "simple")
    ss = O.Seqset.from_entries(ent)
    rm = O.Readmap(ss, ent, [(r,) for r in reads], True)
    codes = np.array([b"ACGT".index(bytes([c])) for c in "".join(reference).encode()], np.uint8)
    starts = np.cumsum([0] + [len(s) for s in reference[:-1]]).astype(np.uint64)
    sizes = np.array([len(s) for s in reference], np.uint64)
    return T.Graph(ss, rm, ss.ref_map_build(codes, starts, sizes))


@pytest.mark.parametrize("rev", [False, True])
@pytest.mark.parametrize("name", sorted(PIPE_CASES))
def test_pipeline_test_vectors(name, rev):
    reads, want = PIPE_CASES[name]
    g = pipe_world(reads)
    o = T.Options(min_overlap=20, trace_reference_assemblies=True)
    ao = A.AlignOptions(min_overlap=20, trace_reference_assemblies=True)
    sc = T.Scaffold(PIPE_PARTS)
    traced_ = T.reversable_trace(g, sc, o, rev_comp=rev, anchor_drop=lambda s_, emit: A.AnchorDropper(ao, s_, emit))
    _, out = A.variants_pipeline(ao, sc, traced_)
    got = [(a.left_offset, a.seq, a.right_offset) for a in out if not a.matches_reference]
    assert sorted(got) == sorted(want)
    for a in out:
        if a.matches_reference:
            assert sc.sub(a.left_offset, a.right_offset - a.left_offset) == a.seq


# ---- normalizer / exact_deduper / vcf_padder: no reference vectors exist; hand cases from the code's own rules ----
def test_normalizer_shifts_left_in_repeats():
    ref = "ACGTTTTTGCA"
    out = []
    n = A.Normalizer(T.Scaffold([(0, ref)]), out.append)
    n.add(asm(8, 8, "T", 0, 0))      # insert of a T after the run of T's: slides to the run's start
    n.add(asm(7, 8, "", 0, 0))       # deletion of the last T of the run: slides to the first
    n.add(asm(9, 10, "A", 0, 0))     # SNP C->A: bases differ, stays
    n.flush()
    assert sorted((a.left_offset, a.right_offset, a.seq) for a in out) == [(3, 3, "T"), (3, 4, ""), (9, 10, "A")]


def test_vcf_padder_and_exact_deduper():
    ref = "ACGTTTTTGCA"
    out = []
    p = A.VcfPadder(T.Scaffold([(5, ref)]), out.append)
    for a in (asm(5, 5, "G", 0, 0), asm(8, 8, "T", 0, 0), asm(8, 9, "", 0, 0), asm(9, 10, "A", 0, 0)):
        p.add(a)
    p.flush()
    # nothing to the left of the extent's first base to pad with; the others take the base at left - 1 (scaffold
    # position 7 is ref[2] = G); equal left offsets leave in canon order: the longer reference span first
    assert [(a.left_offset, a.right_offset, a.seq) for a in out] == [(5, 5, "G"), (7, 9, "G"), (7, 8, "GT"), (9, 10, "A")]
    out = []
    d = A.ExactDeduper(out.append)
    a, b = asm(3, 4, "G", 0, 0), asm(3, 4, "G", 0, 0)
    a.left_pair_matches, b.left_pair_matches = [1, 5], [2, 5]
    for x in (a, b, asm(3, 4, "T", 0, 0), asm(4, 5, "G", 0, 0)):
        d.add(x)
    d.flush()
    assert [(x.left_offset, x.seq, x.left_pair_matches) for x in out] == [(3, "G", [1, 2, 5]), (3, "T", []), (4, "G", [])]


# ---- the whole chain on the reference's own sample: lambdaToyData/benchmark/father.vcf -----------------------------
def father_vcf_records(golden_dir):
    """(POS, REF, ALT, OV) of the 8 records of datasets/lambdaToyData/benchmark/father.vcf (a fixture: the output of
    the reference's discovery on father_lambda.bg; columns of ploidless_vcf_export.cpp:77-85,150)."""
    out = []
    for line in open(golden_dir + "/father_lambda/father.vcf"):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        assert f[8] == "GT:PG:GQ:PI:OV:DP:AD:PDP:PAD"
        out.append((int(f[1]), f[3], f[4], int(f[9].split(":")[4])))
    assert len(out) == 8
    return out


def vcf_columns(out, ref):
    """POS / REF / ALT / OV as ploidless_vcf_export.cpp:12-43,77-85,150 derives them from a padded variant section."""
    return [(a.left_offset + 1, ref[a.left_offset:a.right_offset], a.seq, a.min_overlap) for a in out if not a.matches_reference]


def test_father_vcf_records_from_the_golden_biograph(golden_dir):
    """tracer -> anchor_dropper -> pair_counter (both directions) -> sorter -> ref_trimmer -> deduper -> aligner ->
    align_splitter -> normalizer -> exact_deduper -> vcf_padder over the golden father_lambda seqset + readmap, with the
    discovery command's option (min_overlap = 0.7 x read length, biograph_variants.cpp:674-678): position, REF, ALT and
    the OV (minimum overlap) column of every record of father.vcf, and nothing else."""
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    g = O.decode_readmap(golden_dir + "/father_lambda/readmap", d)
    oss = O.Seqset.from_arrays(d["sizes"], d["shared"], d["bits"])
    orm = O.Readmap.from_arrays(d["n"], g["entry"], g["read_len"], g["mate_loop_ptr"], g["is_forward"], 150)
    ref = "".join(x.strip() for x in open(golden_dir + "/father_lambda/reference.fasta") if not x.startswith(">")).upper()
    codes = np.array([b"ACGT".index(bytes([c])) for c in ref.encode()], np.uint8)
    graph = T.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([len(ref)], np.uint64)))
    mo = int(0.7 * 150)
    o, ao, sc = T.Options(min_overlap=mo), A.AlignOptions(min_overlap=mo), T.Scaffold([(0, ref)])
    traced_ = []
    for rev in (False, True):
        traced_ += T.reversable_trace(graph, sc, o, rev_comp=rev, anchor_drop=lambda s_, emit: A.AnchorDropper(ao, s_, emit))
    _, out = A.variants_pipeline(ao, sc, traced_, through_vcf_pad=True)
    assert vcf_columns(out, ref) == father_vcf_records(golden_dir)


# ---- scaffold_test.cpp:24-140 (subscaffold, is_simple, the iterator) ------------------------------------------------
def test_scaffold_test_vectors():
    two = T.Scaffold([(5, tseq("abcde")), (100, tseq("fghi"))])
    assert two.end_pos == 140 and not A.is_simple(two)
    assert A.is_simple(T.Scaffold([(0, tseq("abcde"))])) and not A.is_simple(T.Scaffold([]))
    sub = A.subscaffold(two, 80, 1)  # empty_subscaffold
    assert sub.extents == [] and not A.is_simple(sub)
    sub = A.subscaffold(two, 1, 139)  # offset_subscaffold
    assert sub.end_pos == 139 and sub.extents == [(4, tseq("abcde")), (99, tseq("fghi"))]
    sub = A.subscaffold(two, 10, 98)  # partial_subscaffold
    assert sub.end_pos == 98 and sub.extents == [(0, tseq("abcde")[5:]), (90, tseq("fghi")[:8])]
    sub = A.subscaffold(two, 55, 45)  # subscaffold_inside: the gap between the regions
    assert sub.extents == [] and sub.end_pos == 45 and not A.is_simple(sub)
    sub = A.subscaffold(two, 5, 135)  # subscaffold_outside
    assert sub.end_pos == 135 and sub.extents == [(0, tseq("abcde")), (95, tseq("fghi"))] and not A.is_simple(sub)
    it = T._ScaffoldIt(T.Scaffold([(5, "GA"), (10, "T")]))  # iterator
    seen = []
    while not it.at_end():
        seen.append((it.offset, it.base()))
        it.advance()
    assert seen == [(5, "G"), (6, "A"), (10, "T")] and it.offset == 11


# ---- trace_ref_test.cpp:210-228 (trace_ref_push_test: a two-scaffold reference, one of them with an N gap) ----------
TR_ALPHA1, TR_ALPHA3, TR_NUM = tseq("abcdefg"), tseq("nopqrstuvw"), tseq("0123456789")
TR_REF = [TR_ALPHA1, TR_ALPHA3, TR_NUM]
TR_SCAFFOLDS = {"alpha": [(0, TR_ALPHA1), (len(TR_ALPHA1) + len(tseq("hijklm")), TR_ALPHA3)], "num": [(0, TR_NUM)]}
TR_CASES = {
    "all_reference": ("num", [tseq("012345"), tseq("34567")], [tseq("01234567")]),
    "spans_extent": ("alpha", [tseq("bcdef"), tseq("efghijklmnop"), tseq("nopqrstuvw")],
                     [(tseq("bcdef"), tseq("ghijklm"), tseq("nopqrstuvw"))]),
}


@pytest.mark.parametrize("rev", [False, True])
@pytest.mark.parametrize("name", sorted(TR_CASES))
def test_trace_ref_push_vectors(name, rev):
    scaffold, reads, want = TR_CASES[name]
    g = pipe_world(reads, TR_REF)
    o = T.Options(min_overlap=20, trace_reference_assemblies=True)
    ao = A.AlignOptions(min_overlap=20, trace_reference_assemblies=True)
    out = T.reversable_trace(g, T.Scaffold(TR_SCAFFOLDS[scaffold]), o, rev_comp=rev,
                             anchor_drop=lambda s_, emit: A.AnchorDropper(ao, s_, emit))
    expect(out, want, ordered=False)
