"""Row f.2 on the GPU: bg_push_trace (host state machine csrc/tracer.cpp over the device walks bg_ref_walk_entries /
bg_tracer_extend_batch, anchor_dropper of csrc/align.cpp, pair_counter) against (1) the vectors of the reference's
tracer_test.cpp:167-316 and (2) the oracle's literal restatement (oracle/tracer.py + oracle/align.py, themselves pinned
on those vectors) on synthetic reads with planted variants, errors and repeats -- assembly for assembly, forward and
reverse, whole scaffold and chunked, with the anchor_dropper (the reference's chain) and without it (the raw tracer)."""
import numpy as np
import pytest

import biograph_amd as bg
from biograph_amd import synth, tracer
from oracle import oracle as O
from oracle import tracer as T
from tests import test_oracle_align as VA
from tests import test_oracle_tracer as V

pytestmark = pytest.mark.gpu
tseq, tseq_rc = O.tseq, O.tseq_rc


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def gpu_world(ctx, pairs, reads, reference):
    """The same seqset / readmap / ref_map as V.world, built by the library from the reads as given (no correction)."""
    all_reads = [r for p in pairs for r in p] + list(reads)
    packed = ctx.pack_reads(all_reads)
    ss, _ = ctx.build_seqset(packed, 30)
    rm = build_readmap(ctx, ss, pairs, reads)
    return ss, rm, tracer.reference_ref_map(ss, reference)


def build_readmap(ctx, ss, pairs, reads):
    """readmap_for_reads (seqset_testutil.cpp:73-125): pairs first, then the unpaired reads, is_paired = true.  The
    library's paired build takes records of two reads; a single read is a record whose second read is empty."""
    seqs = []
    for a, b in pairs:
        seqs += [a, b]
    for r in reads:
        seqs += [r, ""]
    packed = ctx.pack_reads(seqs)
    return ctx.build_readmap(ss, packed, True)


def run_gpu(ctx, parts, pairs, reads=(), rev_comp=False, reference=V.REF, **opts):
    ss, rm, rmap = gpu_world(ctx, pairs, reads, reference)
    o = tracer.TraceOptions(**dict(dict(min_overlap=20, trace_reference_assemblies=1), **opts))
    try:
        out = tracer.push_trace(ss, rm, rmap, parts, rev_comp=rev_comp, options=o)
    finally:
        rmap.free()
    if rev_comp:
        out.reverse()
    return out


CASES = {
    "only_reference": ([(5, tseq("abcdefghijklmnopqrst"))],
                       [(tseq("abcde"), tseq_rc("ijklm")), (tseq("cdefg"), tseq_rc("klmno")), (tseq("efghi"), tseq_rc("mnopq")),
                        (tseq("ghijk"), tseq_rc("opqrs"))], [], {}),
    "simple_variant": ([(0, tseq("abcdefghijklmnopqrstuvw"))],
                       [(tseq("bcdef"), tseq_rc("klmno")), (tseq("efgHi"), tseq_rc("nopqr")), (tseq("gHijk"), tseq_rc("qrstu")),
                        (tseq("Hijkl"), tseq_rc("stuvw"))], [], {}),
    "compound_variant": ([(0, tseq("abcdefghijklmnopqrstuvw"))],
                         [(tseq("bcdef"), tseq_rc("ghijk")), (tseq("defgh"), tseq_rc("ijklm")), (tseq("cdefg"), tseq_rc("lmnop")),
                          (tseq("efgHi"), tseq_rc("nopqr")), (tseq("gHijk"), tseq_rc("qrstu")), (tseq("Hijkl"), tseq_rc("stuvw"))],
                         [], {}),
    "cross_ref_sections": ([(0, tseq("abcdef")), (len(tseq("abcdef")) + len(tseq("gh")), tseq("ijklmnopqr"))],
                           [(tseq("abcde"), tseq_rc("ghijk")), (tseq("cdefg"), tseq_rc("ijklm")), (tseq("efghi"), tseq_rc("klmno"))],
                           [], {}),
    "cross_ref_delete": ([(0, tseq("abcdefg")), (100, tseq("hijklmno"))],
                         [(tseq("abcde"), tseq_rc("cdefg")), (tseq("efghi"), tseq_rc("ghijk")), (tseq("ijklm"), tseq_rc("klmno"))],
                         [], {}),
    "dead_end": ([(0, tseq("abcdefghijklmnopqrstuvwxyz"))], [], V.reads_for_seq(V.DEAD, len(tseq("abcde")), len(tseq("a"))),
                 {"min_overlap": len(tseq("abcd")), "min_anchor_drop_overlap": 10}),
    "dead_end2": ([(0, tseq("abcdefghijklmnopqrstuvwxyz"))], [], V.reads_for_seq(V.DEAD, len(tseq("abcde")), len(tseq("a"))),
                  {"min_overlap": len(tseq("abcd")), "min_anchor_drop_overlap": 11}),
    "dead_end_big_drop": ([(0, tseq("abcdefghijklmnopqrstuvwxyz"))],
                          [(tseq("bcdef"), tseq("efgH") + "G" + tseq("y")), (tseq("defgH") + "G", tseq("fgH") + "G" + tseq("yz"))],
                          [], {}),
}

WANT = {
    "only_reference": [tseq(x) for x in ("abcdefg", "cdefghi", "efghijk", "ghijklm", "ijklmno", "klmnopq", "mnopqrs")],
    "simple_variant": [(tseq("bcdef"), tseq("gHij"), tseq("klmno")), tseq("klmnopqr"), tseq("nopqrstu"), tseq("qrstuvw")],
    "cross_ref_sections": [(tseq("abcde"), tseq("fgh"), tseq("ijklm")), tseq("ijklmno")],
    "cross_ref_delete": [tseq("abcdefg"), (tseq("cdefg"), tseq("h"), tseq("ijklm")), tseq("ijklmno")],
    # tracer_test.cpp:246-316, forward; nothing in the reverse direction
    "dead_end": [(tseq("abcde"), "G" + tseq("f") + "G" + tseq("ghi") + "G", tseq("j"))],
    "dead_end2": [(tseq("abcde"), "G" + tseq("f") + "G", tseq("ghi"))],
    "dead_end_big_drop": [(tseq("bcdef"), tseq("gH") + "G", tseq("yz"))],
}


def full_key(a):
    return a.key() + (sorted(a.left_pair_matches), sorted(a.right_pair_matches), a.score, a.trace_steps,
                      a.left_anchor_ambiguous_bases)


@pytest.mark.parametrize("drop", [True, False])
@pytest.mark.parametrize("rev", [False, True])
@pytest.mark.parametrize("name", sorted(CASES))
def test_tracer_test_vectors(ctx, name, rev, drop):
    parts, pairs, reads, opts = CASES[name]
    out = run_gpu(ctx, parts, pairs, reads, rev_comp=rev, anchor_drop=int(drop), **opts)
    # (1) the reference's own expectations (the dead-end ones are what the anchor_dropper makes of them)
    if name in WANT and (drop or not name.startswith("dead_end")):
        V.expect(out, [] if (rev and name.startswith("dead_end")) else WANT[name])
    if name.startswith("dead_end") and not drop and not rev:
        assert len(out) == 1 and out[0].right_anchor_len == 0  # the raw dead end of the tracer
    if name == "only_reference":
        assert [len(a.right_pair_matches) for a in out] == [1, 1, 1, 0, 0, 0, 0]
        assert [len(a.left_pair_matches) for a in out] == [0, 0, 0, 0, 1, 1, 1]
    if name == "simple_variant":
        assert [len(a.right_pair_matches) for a in out] == [1, 0, 0, 0]
        assert [len(a.left_pair_matches) for a in out] == [0, 0, 0, 1]
    if name.startswith("dead_end") and rev:
        assert out == []
    # (2) the oracle, field for field
    align_keys = ("min_anchor_drop_overlap",)
    want = V.run(parts, pairs, reads, rev_comp=rev, drop={k: opts[k] for k in align_keys if k in opts} if drop else None,
                 **{k: v for k, v in opts.items() if k not in align_keys})
    assert [full_key(a) for a in out] == [full_key(a) for a in want]


def synthetic_world(ctx, G, seed, read_len=100, coverage=24, repeats=None):
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=coverage, read_len=read_len, seed=seed, device="cpu",
                                            return_genome=True, err=0.001, repeats=repeats)
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    reads = [r for r in reads]
    keep = [(a, b) for a, b in zip(reads[0::2], reads[1::2]) if "N" not in a and "N" not in b]
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.numpy()].tobytes().decode()
    return keep, ref


@pytest.mark.parametrize("seed,G,repeats", [(3, 6000, None), (4, 9000, (300, 3, 0.01))])
def test_synthetic_equals_oracle(ctx, seed, G, repeats):
    """Planted SNPs / indels, read errors, a three-copy repeat: every assembly the oracle's tracer emits, in order, with
    its pair matches and scores -- both directions, whole scaffold and in 2000-base chunks (trace_ref.cpp:44-63), with a
    gap in the scaffold."""
    pairs, ref = synthetic_world(ctx, G, seed, repeats=repeats)
    gap = (G // 2, G // 2 + 40)
    parts = [(0, ref[:gap[0]]), (gap[1], ref[gap[1]:])]
    reference = [ref]
    ss, rm, rmap = gpu_world(ctx, pairs, [], reference)
    # the oracle's world from the same reads
    all_reads = [r for p in pairs for r in p]
    seq, offs = O.pack_strings(all_reads)
    ent = O.Entries.build(seq, offs, "simple")
    oss = O.Seqset.from_entries(ent)
    orm = O.Readmap(oss, ent, [tuple(p) for p in pairs], True)
    codes = np.array([b"ACGT".index(bytes([c])) for c in ref.encode()], np.uint8)
    g = T.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([len(ref)], np.uint64)))
    op, gp = oss.parts(), ss.parts()
    assert all(op[n] == gp[n] for n in bg.PART_NAMES)
    kw = dict(min_overlap=70, max_pair_distance=600, read_ahead_distance=3000, trace_reference_assemblies=1)
    o_cpu = T.Options(**dict(kw, trace_reference_assemblies=True))
    sc = T.Scaffold(parts)
    try:
        for rev, drop in ((False, True), (True, True), (False, False), (True, False)):
            o_gpu = tracer.TraceOptions(anchor_drop=int(drop), **kw)
            ad = V.dropper(o_cpu) if drop else None
            # what the dropper cannot anchor on the right (report_half_aligned_func), in forward coordinates
            # (reverse_half_aligned, assemble.cpp:1002-1013)
            want_half, got_half = [], []
            ad_half = None
            if drop:
                from oracle import align as OA
                ao = OA.AlignOptions(min_overlap=o_cpu.min_overlap, read_ahead_distance=o_cpu.read_ahead_distance, trace_reference_assemblies=True)
                comp = str.maketrans("ACGT", "TGCA")
                end = sc.end_pos

                def report(off, seq, rev=rev):
                    want_half.append((end - off, True, seq.translate(comp)[::-1]) if rev else (off, False, seq))
                ad_half = lambda s_, emit: OA.AnchorDropper(ao, s_, emit, half_aligned=report)  # noqa: E731
            want = T.reversable_trace(g, sc, o_cpu, rev_comp=rev, anchor_drop=ad_half if drop else ad)
            got, st = tracer.push_trace(ss, rm, rmap, parts, rev_comp=rev, options=o_gpu, with_stats=True, half_aligned=got_half)
            assert [full_key(a) for a in got] == [full_key(a) for a in want]
            assert [h[:3] for h in got_half] == want_half, (rev, drop)
            if drop:
                assert len(want_half) >= 1
            assert len([a for a in got if not a.matches_reference]) > 3 and st["traces"] > G // 10
            assert st["device_rounds"] < 40 * 50  # lock step: rounds follow the longest trace, not the number of traces
            for start in range(0, G, 2000):
                wc = T.reversable_trace(g, sc, o_cpu, rev_comp=rev, start_offset=start, limit_offset=start + 2000, anchor_drop=ad)
                gc = tracer.push_trace(ss, rm, rmap, parts, start=start, limit=start + 2000, rev_comp=rev, options=o_gpu)
                assert [full_key(a) for a in gc] == [full_key(a) for a in wc], (rev, drop, start)
            # one session, the chunks one after the other with the host steps of each left to the helper thread
            # (bg_tracer_trace_async): the same assemblies
            t = tracer.PushTracer(ss, rm, rmap, o_gpu)
            try:
                for start in range(0, G, 2000):
                    ga = t.trace(parts, start, start + 2000, None, rev, True, asynchronous=True)
                    gs = t.trace(parts, start, start + 2000, None, rev, True)
                    assert [full_key(a) for a in ga] == [full_key(a) for a in gs], (rev, drop, start)
            finally:
                t.close()
    finally:
        rmap.free()


@pytest.mark.parametrize("rev", [False, True])
@pytest.mark.parametrize("name", sorted(VA.PIPE_CASES))
def test_pipeline_test_vectors(ctx, name, rev):
    """modules/variants/pipeline_test.cpp:158-213 end to end through the library: bg_push_trace (one direction, as the
    test's skip_push_trace_* does) -> bg_assemblies_process(sort, ref trim, dedup, align, split); the expectations are
    the variant sections.  (The test's ploid_limiter, not a production step, is left out: see tests/test_oracle_align.py.)"""
    reads, want = VA.PIPE_CASES[name]
    ss, rm, rmap = gpu_world(ctx, [], reads, VA.PIPE_REF)
    o = tracer.TraceOptions(min_overlap=20, trace_reference_assemblies=1)
    try:
        traced = tracer.push_trace(ss, rm, rmap, VA.PIPE_PARTS, rev_comp=rev, options=o)
    finally:
        rmap.free()
    out = tracer.process_assemblies(traced, VA.PIPE_PARTS, options=o, steps=tracer.STEP_THROUGH_SPLIT)
    got = [(a.left_offset, a.seq, a.right_offset) for a in out if not a.matches_reference]
    assert sorted(got) == sorted(want)
    sc = T.Scaffold(VA.PIPE_PARTS)
    for a in out:
        if a.matches_reference:
            assert sc.sub(a.left_offset, a.right_offset - a.left_offset) == a.seq


@pytest.mark.parametrize("rev", [False, True])
@pytest.mark.parametrize("name", sorted(VA.TR_CASES))
def test_trace_ref_push_vectors(ctx, name, rev):
    """trace_ref_test.cpp:210-228: a reference of two scaffolds, one with an N gap, one direction at a time."""
    scaffold, reads, want = VA.TR_CASES[name]
    ss, rm, rmap = gpu_world(ctx, [], reads, VA.TR_REF)
    try:
        out = tracer.push_trace(ss, rm, rmap, VA.TR_SCAFFOLDS[scaffold], rev_comp=rev,
                                options=tracer.TraceOptions(min_overlap=20, trace_reference_assemblies=1))
    finally:
        rmap.free()
    V.expect(out, want, ordered=False)


def test_father_vcf_records_from_the_golden_biograph(ctx, golden_dir):
    """The library's whole chain on the reference's own sample (golden father_lambda seqset + readmap as written by the
    reference): bg_push_trace both directions (tracer, anchor_dropper, pair_counter) -> bg_assemblies_process(sort ...
    vcf_padder), min_overlap = 0.7 x read length as the discovery command sets it: POS, REF, ALT and OV of every
    record of lambdaToyData/benchmark/father.vcf, and nothing else."""
    ss = ctx.open_seqset(golden_dir + "/father_lambda/seqset")
    rm = ctx.open_readmap(golden_dir + "/father_lambda/readmap")
    ref = "".join(x.strip() for x in open(golden_dir + "/father_lambda/reference.fasta") if not x.startswith(">")).upper()
    rmap = tracer.reference_ref_map(ss, [ref])
    o = tracer.TraceOptions(min_overlap=int(0.7 * 150))
    try:
        traced = []
        for rev in (False, True):
            traced += tracer.push_trace(ss, rm, rmap, [(0, ref)], rev_comp=rev, options=o)
    finally:
        rmap.free()
    out = tracer.process_assemblies(traced, [(0, ref)], options=o)
    assert VA.vcf_columns(out, ref) == VA.father_vcf_records(golden_dir)


def test_pipeline_behind_the_tracers_equals_oracle(ctx):
    """Both directions of a synthetic scaffold through bg_assemblies_process (sort, ref trim, dedup, align, split)
    against the oracle's variants_pipeline over the oracle's traces; the planted SNPs come out as 1-base variant
    sections between reference sections."""
    G = 8000
    pairs, ref = synthetic_world(ctx, G, 11)
    parts = [(0, ref)]
    ss, rm, rmap = gpu_world(ctx, pairs, [], [ref])
    all_reads = [r for p in pairs for r in p]
    seq, offs = O.pack_strings(all_reads)
    ent = O.Entries.build(seq, offs, "simple")
    oss = O.Seqset.from_entries(ent)
    orm = O.Readmap(oss, ent, [tuple(p) for p in pairs], True)
    codes = np.array([b"ACGT".index(bytes([c])) for c in ref.encode()], np.uint8)
    g = T.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([len(ref)], np.uint64)))
    from oracle import align as A
    for tra in (0, 1):
        kw = dict(min_overlap=70, max_pair_distance=600, read_ahead_distance=3000, trace_reference_assemblies=tra)
        o_gpu = tracer.TraceOptions(**kw)
        o_cpu = T.Options(**dict(kw, trace_reference_assemblies=bool(tra)))
        sc = T.Scaffold(parts)
        want_in, got_in = [], []
        for rev in (False, True):
            want_in += T.reversable_trace(g, sc, o_cpu, rev_comp=rev, anchor_drop=V.dropper(o_cpu))
            got_in += tracer.push_trace(ss, rm, rmap, parts, rev_comp=rev, options=o_gpu)
        ao = A.AlignOptions(min_overlap=70, read_ahead_distance=3000, trace_reference_assemblies=bool(tra))
        _, want = A.variants_pipeline(ao, sc, want_in, through_vcf_pad=True)
        got = tracer.process_assemblies(got_in, parts, options=o_gpu)
        _, want_split = A.variants_pipeline(ao, sc, want_in)
        got_split = tracer.process_assemblies(got_in, parts, options=o_gpu, steps=tracer.STEP_THROUGH_SPLIT)

        def key(a):
            return (a.left_offset, a.right_offset, a.seq, bool(a.matches_reference), a.left_anchor_len, a.right_anchor_len,
                    sorted(a.left_pair_matches), sorted(a.right_pair_matches), a.score, a.min_overlap, a.trace_steps)
        assert [key(a) for a in got] == [key(a) for a in want]
        assert [key(a) for a in got_split] == [key(a) for a in want_split]
        variants = [a for a in got if not a.matches_reference]
        assert len(variants) >= 3
        for a in variants:  # a variant differs from the reference it replaces, and (VCF) neither side is empty
            assert a.seq != ref[a.left_offset:a.right_offset] and a.seq and a.right_offset > a.left_offset
        for x, y in zip(got, got[1:]):
            assert x.left_offset <= y.left_offset
    rmap.free()


def test_ref_walk_entries_equals_sequential_walk(ctx):
    """bg_ref_walk_entries (1024-base chains with a warm-up) against one sequential push_front_drop chain of the oracle,
    with and without the doubled first base of tracer::skip_ahead_to."""
    import ctypes as C
    from biograph_amd import capi
    pairs, ref = synthetic_world(ctx, 5000, 9)
    ss, rm, rmap = gpu_world(ctx, pairs, [], [ref])
    rmap.free()
    all_reads = [r for p in pairs for r in p]
    seq, offs = O.pack_strings(all_reads)
    ent = O.Entries.build(seq, offs, "simple")
    oss = O.Seqset.from_entries(ent)
    sizes = oss.arrays()[0]
    codes = np.array([b"ACGT".index(bytes([c])) for c in ref.encode()], np.uint8)
    for dup in (0, 1):
        b, e, s = 0, len(oss), 0
        want = []
        walk = ([codes[0]] if dup else []) + list(codes)
        for i, c in enumerate(walk):
            ob, oe, osz = oss.push_front_drop([b], [e], [s], [3 - c], [0])
            b, e, s = int(ob[0]), int(oe[0]), int(osz[0])
            if dup and i == 0:
                continue
            want.append(b if (b + 1 == e and s == sizes[b]) else 2 ** 64 - 1)
        dcodes = bg.DeviceArray.from_numpy(ctx, codes)
        dout = bg.DeviceArray(ctx, len(codes) * 8)
        capi.check(capi.lib().bg_ref_walk_entries(ctx.h, ss.h, C.c_void_p(dcodes.ptr), len(codes), dup, C.c_void_p(dout.ptr)))
        got = dout.to_numpy(np.uint64, len(codes))
        dcodes.free()
        dout.free()
        assert [int(x) for x in got] == want
