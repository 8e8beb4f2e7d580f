"""csrc/align.cpp (bg_assemblies_process: anchor_dropper, sorter, ref_trimmer, deduper, aligner, align_splitter,
normalizer, exact_deduper, vcf_padder)
against the reference's own vectors -- modules/variants/align_test.cpp:59-375, trim_ref_test.cpp:24-112,
dedup_test.cpp:24-131 -- and, on random assemblies, against the oracle's restatement (oracle/align.py).
Host-only entry points: runs without a GPU."""
import random

import pytest

from biograph_amd import tracer as BT
from oracle import align as A
from oracle import oracle as O
from oracle import tracer as T

tseq = O.tseq
ALPHABET = [(0, tseq("abcdefghijklmnopqrstuvwxyz"))]


def asm(left, right, seq, la, ra, **kw):
    return BT.TraceAssembly(left_offset=left, right_offset=right, seq=seq, left_anchor_len=la, right_anchor_len=ra, **kw)


def opts(**kw):
    return BT.TraceOptions(**kw)


def run(parts, asms, steps, **kw):
    return BT.process_assemblies(asms, parts, options=opts(**kw), steps=steps)


def is_ref(a, left, right):
    return a.matches_reference and (a.left_offset, a.right_offset, a.left_anchor_len, a.right_anchor_len) == (left, right, 0, 0)


def is_var(a, left, seq, right):
    return (not a.matches_reference) and (a.left_offset, a.right_offset, a.seq) == (left, right, seq)


DROP_ALIGN = BT.STEP_ANCHOR_DROP | BT.STEP_ALIGN
CHAIN = DROP_ALIGN | BT.STEP_SPLIT  # align_test.cpp:31-52


def test_simple():
    ref = tseq("abc") + "T" + tseq("def")
    a = [asm(0, len(ref), tseq("abc") + "AA" + tseq("def"), 20, 20)]
    assert [x.aligned_variants for x in run([(0, ref)], a, DROP_ALIGN)] == [[(30, 31, "AA")]]
    out = run([(0, ref)], a, CHAIN)
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, "AA", 31) and is_ref(out[2], 31, 61)


@pytest.mark.parametrize("kmer", [1, 2, 3, 4, 20, 31])
def test_drop_right(kmer):
    ref = tseq("abc") + "T" + tseq("defdefdef")
    a = [asm(0, len(ref), tseq("abc") + "AA" + tseq("d"), 20, 0)]
    kw = dict(min_anchor_drop_overlap=10, max_anchor_kmer_size=kmer)
    al = run([(0, ref)], a, DROP_ALIGN, **kw)
    assert [x.aligned_variants for x in al] == [[(30, 31, "AA")]] and (al[0].left_offset, al[0].right_offset) == (0, 41)
    out = run([(0, ref)], a, CHAIN, **kw)
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, "AA", 31) and is_ref(out[2], 31, 41)


@pytest.mark.parametrize("kmer", [1, 2, 3, 4, 20, 31])
def test_drop_right_short(kmer):
    seq = tseq("abcd") + "G" + tseq("yz")
    al = run(ALPHABET, [asm(0, len(seq), seq, 40, 0)], DROP_ALIGN, min_overlap=20, max_anchor_kmer_size=kmer)
    assert [x.aligned_variants for x in al] == [[(40, 240, "G")]]
    assert (al[0].left_offset, al[0].right_offset, al[0].right_anchor_len) == (0, 260, 20)


def test_two_snps_multi_extent_insert_delete():
    ref = tseq("abc") + "TAT" + tseq("def")
    a = [asm(5, 5 + len(ref), tseq("abc") + "GAG" + tseq("def"), 1, 1)]
    assert [x.aligned_variants for x in run([(5, ref)], a, DROP_ALIGN)] == [[(35, 36, "G"), (37, 38, "G")]]
    out = run([(5, ref)], a, CHAIN)
    assert [(x.left_offset, x.right_offset, x.matches_reference) for x in out] == [
        (5, 35, True), (35, 36, False), (36, 37, True), (37, 38, False), (38, 68, True)]
    parts = [(0, tseq("abc")), (100, tseq("ghi"))]
    a = [asm(0, 130, tseq("abcdefghi"), 1, 1)]
    assert [x.aligned_variants for x in run(parts, a, DROP_ALIGN)] == [[(30, 100, tseq("def"))]]
    out = run(parts, a, CHAIN)
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, tseq("def"), 100) and is_ref(out[2], 100, 130)
    out = run([(0, tseq("abcdef"))], [asm(0, 60, tseq("abc") + "T" + tseq("def"), 1, 1)], CHAIN)
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, "T", 30) and is_ref(out[2], 30, 60)
    out = run([(0, tseq("abc") + "T" + tseq("def"))], [asm(0, 61, tseq("abcdef"), 1, 1)], CHAIN)
    assert len(out) == 3 and is_ref(out[0], 0, 30) and is_var(out[1], 30, "", 31) and is_ref(out[2], 31, 61)


def test_wild_aligns_smaller_near_ends():
    pre = ("TGTGATAGTAGATCAGAGGTGGAGACATCAACGTAAACTTATGTTTAGTTTAATATAGACACACACAGTTCT"
           "ACATAGAAAACTTTATAATTAGGTGTGT")
    post = ("ACCATTCTGAAAGGAATCAGGCTCTTTGAAGAAATGTCTGATACTAGAACTGGGACAGTAAATATAGGAGCC"
            "AGGATAATCTGGAAGTATCAGAAAGTAAGTACTAAAAAAATTAAAATATATCAAACAAAAATAAAAGCCAAT"
            "AAAA")
    parts = [(pre, pre), ("G", "A"), ("TAGGTAGGTTAGACAC",) * 2, ("G", "A"), ("CACATATACTTCCTAGCATTGC",) * 2, ("T", "C"),
             ("AATGAGGGA",) * 2, ("C", "G"), ("AAGATACAATGTGC",) * 2, ("", "TC"), ("ATTCAGCAGCCA",) * 2, ("C", "G"),
             ("ATGTAAGTTTTCC",) * 2, ("C", "T"), (post, post)]
    ref = var = ""
    want = []
    for r, v in parts:
        if r != v:
            want.append((len(ref), len(ref) + len(r), v))
        ref += r
        var += v
    al = run([(0, ref)], [asm(0, len(ref), var, len(pre), len(post))], DROP_ALIGN)
    assert [x.aligned_variants for x in al] == [want]


def test_trim_ref_vectors():
    a = asm(0, 30, tseq("abc"), 0, 0, matches_reference=True)
    assert run(ALPHABET, [a], BT.STEP_TRIM_REF) == []
    assert run(ALPHABET, [asm(0, 30, tseq("abc"), 0, 0)], BT.STEP_TRIM_REF) == []
    out = run(ALPHABET, [asm(30, 140, tseq("de") + "G" + tseq("FGHI") + "G" + tseq("jklmn"), 10, 20)], BT.STEP_TRIM_REF)
    assert len(out) == 1 and is_var(out[0], 50, "G" + tseq("FGHI") + "G", 90)
    assert (out[0].left_anchor_len, out[0].right_anchor_len) == (0, 0)
    ref = tseq("abc") + "G" + tseq("defghijklmnopqrstuvwxyz")
    out = run([(0, ref)], [asm(0, 91, tseq("abc") + "GG" + tseq("defghi"), 1, 1)], BT.STEP_TRIM_REF)
    assert len(out) == 1 and is_var(out[0], 30, "G", 30)
    ref = tseq("abc") + "GG" + tseq("defghijklmnopqrstuvwxyz")
    out = run([(0, ref)], [asm(0, 92, tseq("abc") + "G" + tseq("defghi"), 1, 1)], BT.STEP_TRIM_REF)
    assert len(out) == 1 and is_var(out[0], 30, "", 31)


def test_dedup_vectors():
    a = asm(1, 5, tseq("abc"), 0, 0)
    out = run([], [a, a], BT.STEP_DEDUP, ) if False else BT.process_assemblies([a, a], [], end_pos=10, steps=BT.STEP_DEDUP)
    assert len(out) == 1 and is_var(out[0], 1, tseq("abc"), 5)
    b, c, d = asm(1, 5, tseq("def"), 0, 0), asm(1, 6, tseq("def"), 0, 0), asm(2, 6, tseq("def"), 0, 0)
    out = BT.process_assemblies([a, b, b, c, d], [], end_pos=10, steps=BT.STEP_DEDUP)
    assert sorted((x.left_offset, x.seq, x.right_offset) for x in out) == sorted(
        [(1, tseq("abc"), 5), (1, tseq("def"), 5), (1, tseq("def"), 6), (2, tseq("def"), 6)])
    assert out[-1].left_offset == 2
    for asms in ([asm(0, 40, tseq("ab") + "G" + tseq("d"), 20, 10), asm(10, 50, tseq("b") + "G" + tseq("de"), 10, 20)],
                 [asm(0, 50, tseq("ab") + "G" + tseq("de"), 20, 20), asm(10, 40, tseq("b") + "G" + tseq("d"), 10, 10)]):
        out = run(ALPHABET, asms, BT.STEP_DEDUP)
        assert len(out) == 1 and is_var(out[0], 0, tseq("ab") + "G" + tseq("de"), 50)
        assert (out[0].left_anchor_len, out[0].right_anchor_len) == (20, 20)


def test_a_reference_abort_is_an_error():
    """aligner CHECKs that the anchors match the reference (align.cpp:31,36); the library fails the call."""
    with pytest.raises(Exception, match="anchor does not match"):
        run(ALPHABET, [asm(0, 30, "T" + tseq("abc")[1:], 5, 5)], BT.STEP_ALIGN)
    with pytest.raises(Exception, match="not sorted|backwards"):
        run(ALPHABET, [asm(100, 130, tseq("klm"), 1, 1), asm(0, 30, tseq("abc"), 1, 1)], BT.STEP_ALIGN)


# ---- random assemblies: product == oracle, step by step ---------------------------------------------------------
def to_oracle(t):
    a = T.Assembly()
    for f in ("left_offset", "right_offset", "left_anchor_len", "right_anchor_len", "seq", "min_overlap", "trace_steps",
              "left_anchor_ambiguous_bases", "matches_reference", "score"):
        setattr(a, f, getattr(t, f))
    a.rc_read_ids = set(t.rc_read_ids)
    a.left_pair_matches, a.right_pair_matches = list(t.left_pair_matches), list(t.right_pair_matches)
    return a


def full_key(a):
    ml = getattr(a, "ml", None)
    return (a.left_offset, a.right_offset, a.left_anchor_len, a.right_anchor_len, a.seq, bool(a.matches_reference),
            a.min_overlap, a.trace_steps, a.left_anchor_ambiguous_bases, a.score, list(a.left_pair_matches),
            list(a.right_pair_matches), list(getattr(a, "aligned_variants", [])), getattr(a, "merged", 0),
            None if ml is None else tuple(ml[:4]) + (round(ml[4], 6), round(ml[5], 6), ml[6]))


def mutate(rng, s):
    out = []
    i = 0
    while i < len(s):
        r = rng.random()
        if r < 0.03:
            out.append(rng.choice("ACGT"))
            i += 1
        elif r < 0.045:
            out.append("".join(rng.choice("ACGT") for _ in range(rng.randint(1, 12))))
        elif r < 0.06:
            i += rng.randint(1, 12)
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def random_case(rng):
    # a scaffold with repeats (so that ties and long matches happen) in 1-3 extents
    unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(3, 40)))
    parts, pos = [], rng.randint(0, 20)
    for _ in range(rng.randint(1, 3)):
        s = ""
        while len(s) < rng.randint(150, 500):
            s += unit if rng.random() < 0.3 else "".join(rng.choice("ACGT") for _ in range(rng.randint(5, 60)))
        parts.append((pos, s))
        pos += len(s) + rng.randint(1, 30)
    sc = T.Scaffold(parts)
    asms = []
    for _ in range(rng.randint(1, 12)):
        off, s = rng.choice(parts)
        la = rng.randint(1, 30)
        left = off + rng.randint(0, len(s) - la - 1)
        span = rng.randint(la, min(250, off + len(s) - left))
        dead_end = rng.random() < 0.4
        ra = 0 if dead_end else rng.randint(1, max(1, min(30, span - la)))
        if la + ra > span:
            continue
        ref_span = sc.sub(left, span)
        if ref_span is None:
            continue
        mid = mutate(rng, ref_span[la:span - ra])
        if dead_end:
            seq = ref_span[:la] + mid + "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 40)))
            right = left + int(len(seq) * 1.5)
        else:
            seq = ref_span[:la] + mid + ref_span[span - ra:]
            right = left + span
        if len(seq) <= la + ra:
            continue
        a = asm(left, right, seq, la, ra, score=rng.randint(0, 3), min_overlap=rng.randint(10, 60), trace_steps=rng.randint(0, 9),
                left_pair_matches=sorted(rng.sample(range(50), rng.randint(0, 3))),
                right_pair_matches=sorted(rng.sample(range(50), rng.randint(0, 3))))
        asms.append(a)
        if rng.random() < 0.3:
            asms.append(a)
    if rng.random() < 0.5:
        off, s = parts[0]
        asms.append(asm(off, off + 20, s[:20], 0, 0, matches_reference=True))
    asms.sort(key=lambda a: a.left_offset)
    kw = dict(min_overlap=rng.choice([8, 12, 20, 35, 70]), min_anchor_drop_overlap=rng.choice([4, 8, 15]),
              max_anchor_kmer_size=rng.choice([1, 3, 8, 31]), read_ahead_distance=rng.choice([50, 300, 100000]),
              trace_reference_assemblies=rng.choice([0, 1]), ref_align_factor=rng.choice([2, 4]),
              max_ref_align_bases=rng.choice([5, 30]), output_ml_features=rng.choice([0, 1]))
    return parts, sc, asms, kw


def oracle_steps(sc, asms, kw, steps):
    ao = A.AlignOptions(min_overlap=kw["min_overlap"], min_anchor_drop_overlap=kw["min_anchor_drop_overlap"],  # (ML switch: not an align option)
                        max_kmer_size=kw["max_anchor_kmer_size"], read_ahead_distance=kw["read_ahead_distance"],
                        trace_reference_assemblies=bool(kw["trace_reference_assemblies"]),
                        ref_align_factor=kw["ref_align_factor"], max_ref_align_bases=kw["max_ref_align_bases"])
    cur = [to_oracle(a) for a in asms]
    if steps & BT.STEP_ANCHOR_DROP:
        nxt = []
        ad = A.AnchorDropper(ao, sc, nxt.append)
        for a in cur:
            ad.add(a)
        ad.flush()
        cur = nxt
    if steps & BT.STEP_SORT:
        cur = sorted(cur, key=A.canon_key)
    if kw.get("output_ml_features"):
        for a in cur:
            a.ml, a.merged = A.ml_features_of(a, sc), 0
    out = []
    chain = out.append
    flush = []
    if steps & BT.STEP_VCF_PAD:
        pad = A.VcfPadder(sc, chain)
        chain, flush = pad.add, [pad.flush] + flush
    if steps & BT.STEP_EXACT_DEDUP:
        xd = A.ExactDeduper(chain)
        chain, flush = xd.add, [xd.flush] + flush
    if steps & BT.STEP_NORMALIZE:
        nm = A.Normalizer(sc, chain)
        chain, flush = nm.add, [nm.flush] + flush
    if steps & BT.STEP_SPLIT:
        sp = A.AlignSplitter(chain)
        chain, flush = sp.add, [sp.flush] + flush
    if steps & BT.STEP_ALIGN:
        al = A.Aligner(ao, sc, chain)
        chain, flush = al.add, [al.flush] + flush
    if steps & BT.STEP_DEDUP:
        dd = A.Deduper(chain)
        chain, flush = dd.add, [dd.flush] + flush
    if steps & BT.STEP_TRIM_REF:
        tr = A.RefTrimmer(sc, chain)
        chain, flush = tr.add, [tr.flush] + flush
    for a in cur:
        chain(a)
    for f in flush:
        f()
    return out


def test_normalizer_padder_exact_deduper_hand_cases():
    ref = "ACGTTTTTGCA"
    out = run([(0, ref)], [asm(7, 8, "", 0, 0), asm(8, 8, "T", 0, 0), asm(9, 10, "A", 0, 0)], BT.STEP_NORMALIZE)
    assert sorted((a.left_offset, a.right_offset, a.seq) for a in out) == [(3, 3, "T"), (3, 4, ""), (9, 10, "A")]
    out = run([(5, ref)], [asm(5, 5, "G", 0, 0), asm(8, 8, "T", 0, 0), asm(8, 9, "", 0, 0), asm(9, 10, "A", 0, 0)], BT.STEP_VCF_PAD)
    assert [(a.left_offset, a.right_offset, a.seq) for a in out] == [(5, 5, "G"), (7, 9, "G"), (7, 8, "GT"), (9, 10, "A")]
    a, b = asm(3, 4, "G", 0, 0, left_pair_matches=[1, 5]), asm(3, 4, "G", 0, 0, left_pair_matches=[2, 5])
    out = run([(0, ref)], [a, b, asm(3, 4, "T", 0, 0), asm(4, 5, "G", 0, 0)], BT.STEP_EXACT_DEDUP)
    assert [(x.left_offset, x.seq, x.left_pair_matches) for x in out] == [(3, "G", [1, 2, 5]), (3, "T", []), (4, "G", [])]


STEP_SETS = [BT.STEP_ANCHOR_DROP, DROP_ALIGN, CHAIN, BT.STEP_ANCHOR_DROP | BT.STEP_PIPELINE, BT.STEP_ANCHOR_DROP | BT.STEP_THROUGH_SPLIT,
             CHAIN | BT.STEP_NORMALIZE, CHAIN | BT.STEP_NORMALIZE | BT.STEP_EXACT_DEDUP,
             BT.STEP_ANCHOR_DROP | BT.STEP_SORT | BT.STEP_TRIM_REF | BT.STEP_DEDUP,
             BT.STEP_ANCHOR_DROP | BT.STEP_SORT | BT.STEP_TRIM_REF | BT.STEP_DEDUP | BT.STEP_ALIGN]


@pytest.mark.parametrize("seed", range(6))
def test_random_assemblies_equal_the_oracle(seed):
    rng = random.Random(1234 + seed)
    n_out = 0
    for _ in range(60):
        parts, sc, asms, kw = random_case(rng)
        for steps in STEP_SETS:
            try:
                want = [full_key(a) for a in oracle_steps(sc, asms, kw, steps)]
            except AssertionError as e:  # a CHECK of the reference: the library must refuse as well
                with pytest.raises(Exception):
                    run(parts, asms, steps, **kw)
                continue
            got = [full_key(a) for a in run(parts, asms, steps, **kw)]
            assert got == want, (seed, steps, kw)
            n_out += len(got)
    assert n_out > 200


@pytest.mark.parametrize("seed", range(4))
def test_pipeline_fed_piece_by_piece_equals_the_whole(seed):
    """bg_pipeline_open / feed / finish (the steps behind a scaffold's tracers, piece by piece with marks) against
    bg_assemblies_process over the same assemblies in the same order; a mark that a later piece breaks is reported."""
    rng = random.Random(777 + seed)
    sets = [BT.STEP_PIPELINE, BT.STEP_SORT | BT.STEP_TRIM_REF | BT.STEP_DEDUP, BT.STEP_SORT | BT.STEP_TRIM_REF | BT.STEP_DEDUP | BT.STEP_ALIGN,
            BT.STEP_SORT]
    n_out = n_broken = n_early = n_early_broken = 0
    for _ in range(60):
        parts, sc, asms, kw = random_case(rng)
        kw = dict(kw, output_ml_features=rng.choice([0, 1]))
        try:
            anchored = run(parts, asms * rng.randint(1, 3), BT.STEP_ANCHOR_DROP, **kw)
        except Exception:
            continue
        if not anchored:
            continue
        first = lambda a: min(a.left_offset, a.right_offset)  # noqa: E731
        lo, hi = min(map(first, anchored)), max(map(first, anchored)) + 1
        cuts = sorted({lo} | {rng.randint(lo, hi) for _ in range(rng.randint(0, 5))})
        reach = rng.choice([0, 5, 40, 1000])  # how far before its piece an assembly may lie
        batches = [[] for _ in cuts]
        for a in anchored:
            f = first(a) + rng.randint(0, reach)
            batches[max(i for i, c in enumerate(cuts) if c <= f)].append(a)
        for b in batches:
            rng.shuffle(b)
        order = [a for b in batches for a in b]
        marks = [c - reach for c in cuts[1:]] + [hi + reach]
        for steps in sets:
            try:
                want = [full_key(a) for a in run(parts, order, steps, **kw)]
            except Exception:  # a CHECK of the reference: piece by piece it must be refused as well
                with pytest.raises(Exception):
                    BT.process_assemblies_streamed(list(zip(batches, marks)), parts, options=opts(**kw), steps=steps)
                continue
            got, broken = BT.process_assemblies_streamed(list(zip(batches, marks)), parts, options=opts(**kw), steps=steps)
            assert broken is None and [full_key(a) for a in got] == want, (seed, steps, kw)
            n_out += len(got)
            # the normalizer letting go early, what has left taken after every piece: the same assemblies, or a break that is said
            for margin in (0, 3, 25, 400):
                got, broken, before = BT.process_assemblies_streamed(list(zip(batches, marks)), parts, options=opts(**kw), steps=steps,
                                                                     release_early=margin, take=True)
                if broken is None:
                    assert [full_key(a) for a in got] == want, (seed, steps, kw, margin)
                    n_early += before
                else:
                    assert got is None and steps & BT.STEP_NORMALIZE
                    n_early_broken += 1
        # marks that promise too much: the first piece that lies behind one says so
        if len(batches) > 1 and reach:
            greedy = [c for c in cuts[1:]] + [hi + reach]
            late = next((k for k in range(1, len(batches)) if any(first(a) < greedy[k - 1] for a in batches[k])), None)
            got, broken = BT.process_assemblies_streamed(list(zip(batches, greedy)), parts, options=opts(**kw), steps=BT.STEP_PIPELINE)
            if late is None:
                assert broken is None
            else:
                assert got is None and broken == late
                n_broken += 1
    assert n_out > 300 and n_broken > 0 and n_early > 100, (n_out, n_broken, n_early, n_early_broken)
    with pytest.raises(Exception):  # the anchor_dropper works on the tracers' own order: not a step of the fed pipeline
        BT.process_assemblies_streamed([([], 0)], [(0, tseq("abc"))], steps=BT.STEP_ANCHOR_DROP | BT.STEP_SORT)


def test_ml_features_and_merge_counts():
    """add_ml_features (biograph_variants.cpp:746-779) behind the sorter, and what the dedupers keep of them."""
    ref = tseq("abc") + "T" + tseq("defgh")
    a1 = asm(10, 60, ref[10:30] + "C" + ref[31:60], 15, 20, score=7)
    a2 = asm(20, 71, ref[20:30] + "C" + ref[31:71], 5, 30, score=9)
    out = run([(0, ref)], [a1, a2], BT.STEP_SORT | BT.STEP_TRIM_REF | BT.STEP_DEDUP, output_ml_features=1)
    assert len(out) == 1 and (out[0].left_offset, out[0].right_offset, out[0].seq, out[0].merged) == (30, 31, "C", 1)
    gc = lambda x: (x.count("G") + x.count("C")) / len(x)  # noqa: E731
    want = (9, 51, 5, 30, gc(ref[20:71]), gc(a2.seq), 51)   # the longer original sequence wins
    assert out[0].ml[:4] == want[:4] and out[0].ml[6] == 51
    assert abs(out[0].ml[4] - want[4]) < 1e-6 and abs(out[0].ml[5] - want[5]) < 1e-6
    plain = run([(0, ref)], [a1, a2], BT.STEP_SORT | BT.STEP_TRIM_REF | BT.STEP_DEDUP)
    assert plain[0].ml is None and plain[0].merged == 1
