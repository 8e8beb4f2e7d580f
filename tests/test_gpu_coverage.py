"""Row f.2, third slice, on the GPU: bg_assemblies_coverage (csrc/coverage.cpp: path_group + calc_coverage over the
device walk bg_walk_chains, simple_genotype_filter) and bg_vcf_record against (1) the vectors of the reference's
calc_coverage_test.cpp, (2) every record line of its own father.vcf from the golden father_lambda biograph, and
(3) the oracle's restatement (oracle/coverage.py, itself pinned on both) on a synthetic sample with SNPs and indels."""
import numpy as np
import pytest

import biograph_amd as bg
from biograph_amd import tracer
from oracle import align as A
from oracle import coverage as C
from oracle import oracle as O
from oracle import tracer as T
from tests import test_oracle_coverage as VC
from tests.test_gpu_tracer import build_readmap, gpu_world, synthetic_world

pytestmark = pytest.mark.gpu
tseq, tseq_rc = O.tseq, O.tseq_rc
over = VC.over


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def world(ctx, reads):
    packed = ctx.pack_reads(list(reads))
    ss, _ = ctx.build_seqset(packed, 30)
    return ss, build_readmap(ctx, ss, [], reads)


def asm(left, right, seq, **kw):
    return tracer.TraceAssembly(left_offset=left, right_offset=right, seq=seq, **kw)


def run(ctx, reads, asms, parts=(), **opts):
    ss, rm = world(ctx, reads)
    o = tracer.CoverageOptions(**dict(dict(penalize_directional_coverage=0), **opts))
    return tracer.assemblies_coverage(ss, rm, asms, list(parts), end_pos=max([o_ + len(s) for o_, s in parts] + [2000]),
                                      options=o, genotype_filter=False)


def test_walk_chains_equals_single_steps(ctx):
    """bg_walk_chains against chained bg_push_front_drop_batch calls (one base at a time)."""
    import ctypes as CT
    from biograph_amd import capi
    pairs, ref = synthetic_world(ctx, 3000, 21)
    ss, rm, rmap = gpu_world(ctx, pairs, [], [ref])
    rmap.free()
    rng = np.random.default_rng(5)
    n, length = 5, 400
    seq = ref[700:700 + length]
    codes = np.array([b"ACGT".index(bytes([c])) for c in seq.encode()], np.uint8)
    start = np.zeros((n, 3), np.uint64)
    start[:, 1] = ss.size()
    # different histories: walk chain i over i * 37 random bases first
    for i in range(1, n):
        b, e, s = np.array([0], np.uint64), np.array([ss.size()], np.uint64), np.array([0], np.uint32)
        for c in rng.integers(0, 4, i * 37):
            b, e, s = ss.push_front_drop_batch(b, e, s, np.array([c], np.uint8), np.array([0], np.uint32))
        start[i] = (b[0], e[0], s[0])
    out = np.zeros((length, n, 3), np.uint64)
    capi.check(capi.lib().bg_walk_chains(ctx.h, ss.h, n, capi.np_ptr(start), capi.np_ptr(codes), length, capi.np_ptr(out)))
    b, e, s = start[:, 0].copy(), start[:, 1].copy(), start[:, 2].astype(np.uint32)
    for t in range(length):
        b, e, s = ss.push_front_drop_batch(b, e, s, np.full(n, 3 - codes[t], np.uint8), np.zeros(n, np.uint32))
        assert (out[t, :, 0] == b).all() and (out[t, :, 1] == e).all() and (out[t, :, 2] == s).all(), t
    assert len({tuple(x) for x in out[-1]}) == 1  # all histories have merged after max_read_len bases


def test_calc_coverage_test_vectors(ctx):
    out = run(ctx, [tseq("bcdefg")], [asm(0, 1, tseq("abcdefgh"))])
    assert [a.coverage for a in out] == [[0] + over("a", 0) + [0] + over("bcdefg", 1) + [0] + over("h", 0) + [0]]
    assert out[0].pair_coverage == [0] + over("abcdefgh", 0) + [0]
    out = run(ctx, [tseq("abc")], [asm(0, 1, tseq("abcdefg"))])
    assert out[0].coverage == [0] + over("abc", 1) + [0] + over("defg", 0) + [0]
    out = run(ctx, [tseq("efg")], [asm(0, 1, tseq("abcdefg"))])
    assert out[0].coverage == [0] + over("abcd", 0) + [0] + over("efg", 1) + [0]
    out = run(ctx, [tseq("abcdef"), tseq("cdefg")], [asm(0, 1000, tseq("abcdefg"))])
    assert out[0].coverage == [0] + over("ab", 1) + [1] + over("cdef", 2) + [1] + over("g", 1) + [0]


def test_calc_coverage_pair_matches(ctx):
    reads = [tseq("bcdefg"), tseq("defgh")]
    ss, rm = world(ctx, reads)
    b, e = ss.find_batch([tseq_rc("defgh")])
    assert int(e[0]) == int(b[0]) + 1
    entry_id, read_len, _, _ = rm.arrays()
    rid = [i for i in range(len(entry_id)) if int(entry_id[i]) == int(b[0]) and int(read_len[i]) == 50][0]  # read_id_for_seq
    a = asm(0, 1, tseq("abcdefghi"), right_pair_matches=[rid])
    out = tracer.assemblies_coverage(ss, rm, [a], [], end_pos=2000, options=tracer.CoverageOptions(penalize_directional_coverage=0),
                                     genotype_filter=False)
    assert out[0].coverage == [0] + over("a", 0) + [0] + over("bc", 1) + [1] + over("defg", 2) + [1] + over("h", 1) + [0] + over("i", 0) + [0]
    assert out[0].pair_coverage == [0] + over("abc", 0) + [0] + over("defgh", 1) + [0] + over("i", 0) + [0]


@pytest.mark.parametrize("reads,cov,ref_depth", [
    ([tseq("abcd") + "A" + tseq("efg")], [1, 1], None),
    ([tseq("abcd") + "A", tseq("cd") + "A" + tseq("efg")], [2, 1], None),
    ([tseq("abcd") + "A" + tseq("efg"), "A" + tseq("efgh")], [1, 2], 0),
    ([tseq("abcd") + "T" + tseq("efg")], [0, 0], 1),
])
def test_basic_with_ref(ctx, reads, cov, ref_depth):
    out = run(ctx, reads, [asm(40, 41, "A")], parts=[(0, VC.REF_T)])
    assert len(out) == 1 and out[0].coverage == cov and out[0].other_depth == 0
    if ref_depth is not None:
        assert out[0].ref_depth == ref_depth


def test_insert_between_scaffolds_and_long_overlap(ctx):
    out = run(ctx, [tseq("abcd"), tseq("cdef"), tseq("efgh"), tseq("ghij"), tseq("ijkl")], [asm(40, 1000, tseq("efgh"))],
              parts=[(0, tseq("abcd")), (1000, tseq("ijkl"))])
    assert out[0].coverage == [1] + over("ef", 2) + [1] + over("gh", 2) + [1]
    assert (out[0].other_depth, out[0].ref_depth) == (0, 0)
    seq = tseq("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ")
    reads_seq = tseq("XXXX") + seq + tseq("XXXX")
    reads = [reads_seq[i:i + 41] for i in range(1, len(reads_seq) - 40, 10)]
    out = run(ctx, reads, [asm(40, len(reads_seq) - 40, seq)], parts=[(0, reads_seq)])
    assert out[0].coverage == [4] * (len(seq) + 1)


@pytest.mark.parametrize("second", [False, True])
@pytest.mark.parametrize("edge", range(38, 44))
def test_part_overlaps_extent_edge(ctx, edge, second):
    nxt = tseq("ghijklmnopqrstuvwxyz")
    main = tseq("abcdef") + nxt[:edge]
    last_pos = len(main) - 40
    reads = [tseq("abcd"), tseq("cdef"), tseq("efgh"), main[last_pos:last_pos + 40]]
    asms = [asm(60, 61, "T")] + ([asm(1020, 1021, "T")] if second else [])
    out = run(ctx, reads, asms, parts=[(0, main), (1000, tseq("ijkl"))])
    assert (out[0].coverage, out[0].other_depth, out[0].ref_depth) == ([0, 0], 0, 2 if last_pos < 60 else 1)
    assert len(out) == (2 if second else 1)
    if second:
        assert (out[1].coverage, out[1].other_depth, out[1].ref_depth) == ([0, 0], 0, 0)


@pytest.mark.parametrize("sample,n_records", [("father", 8), ("mother", 8), ("proband", 17)])
def test_father_vcf_lines_from_the_golden_biograph(ctx, golden_dir, sample, n_records):
    """The whole engine through the library on the reference's own samples (datasets/lambdaToyData/benchmark/
    {father,mother,proband}_lambda.bg): every record line of father.vcf, mother.vcf and proband.vcf (PI, the process-wide
    assembly counter of the reference, aside)."""
    ss = ctx.open_seqset(golden_dir + "/%s_lambda/seqset" % sample)
    rm = ctx.open_readmap(golden_dir + "/%s_lambda/readmap" % sample)
    ref = "".join(x.strip() for x in open(golden_dir + "/father_lambda/reference.fasta") if not x.startswith(">")).upper()
    rmap = tracer.reference_ref_map(ss, [ref])
    o = tracer.TraceOptions(min_overlap=int(0.7 * 150))
    try:
        traced = []
        for rev in (False, True):
            traced += tracer.push_trace(ss, rm, rmap, [(0, ref)], rev_comp=rev, options=o)
    finally:
        rmap.free()
    sections = tracer.process_assemblies(traced, [(0, ref)], options=o)
    kept, lines = tracer.assemblies_coverage(ss, rm, sections, [(0, ref)], vcf_name="lambda")
    want = [x for x in open(golden_dir + "/%s_lambda/%s.vcf" % (sample, sample)) if not x.startswith("#")]
    assert len(lines) == len(want) == n_records

    def without_pi(line):
        f = line.rstrip("\n").split("\t")
        s = f[9].split(":")
        s[3] = "PI"
        return f[:9] + [":".join(s)]
    assert [without_pi(x) for x in lines] == [without_pi(x) for x in want]


def test_synthetic_equals_oracle(ctx):
    """Planted SNPs and indels at 24x: coverage arrays, depths, genotypes and VCF lines of the library == the oracle's."""
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
    kw = dict(min_overlap=70, max_pair_distance=600, read_ahead_distance=3000)
    o_gpu, o_cpu, sc = tracer.TraceOptions(**kw), T.Options(**kw), T.Scaffold(parts)
    from tests.test_oracle_tracer import dropper
    want_in, got_in = [], []
    for rev in (False, True):
        want_in += T.reversable_trace(g, sc, o_cpu, rev_comp=rev, anchor_drop=dropper(o_cpu))
        got_in += tracer.push_trace(ss, rm, rmap, parts, rev_comp=rev, options=o_gpu)
    rmap.free()
    ao = A.AlignOptions(min_overlap=70, read_ahead_distance=3000)
    _, want_sections = A.variants_pipeline(ao, sc, want_in, through_vcf_pad=True)
    got_sections = tracer.process_assemblies(got_in, parts, options=o_gpu)
    for filt in (False, True):
        co = C.CoverageOptions()
        want = C.calc_coverage(g, [int(x) for x in orm.get("read_len")], sc, co, want_sections)
        if filt:
            want = C.simple_genotype_filter(co, want)
        got, lines = tracer.assemblies_coverage(ss, rm, got_sections, parts, genotype_filter=filt, vcf_name="chrS")

        def key(a):
            return (a.left_offset, a.right_offset, a.seq, list(a.coverage), list(a.pair_coverage), a.other_depth, a.other_pair_depth,
                    a.ref_depth) + ((a.strand_count, round(a.genotype_quality, 12)) if filt else ())
        assert [key(a) for a in got] == [key(a) for a in want]
        assert len(got) >= 3
        if filt:
            assert lines == [C.vcf_line(co, "chrS", lambda off, n: ref[off:off + n], a) for a in want]


@pytest.mark.parametrize("paths", [3, 2, 1])
def test_max_coverage_paths(ctx, paths):  # calc_coverage_test.cpp:323-355
    ref, reads, vs = VC.snp_with_other_coverage_case()
    out = run(ctx, reads, [asm(l, r, s) for l, r, s in vs], parts=[(0, ref)], max_coverage_paths=paths)
    assert VC.snp_with_other_coverage_expected(out) == (paths >= 2)


@pytest.mark.parametrize("n_vars", [1, 2, 3, 5, 8, 12])
def test_stress_recipe(ctx, n_vars):  # calc_coverage_test.cpp:370-520
    ref, reads, variants, paired, is_delete, var_count, pair_count, ref_count = VC.stress_case(n_vars, 77 + n_vars)
    ss, rm = world(ctx, reads)
    entry_id, read_len, _, _ = rm.arrays()
    asms = []
    for i, (l, r, s) in enumerate(variants):
        ids = []
        for x in sorted(paired[i]):
            b, e = ss.find_batch([T.revcomp(x)])
            assert int(e[0]) == int(b[0]) + 1
            ids.append([j for j in range(len(entry_id)) if int(entry_id[j]) == int(b[0]) and int(read_len[j]) == len(x)][0])
        asms.append(asm(l, r, s, left_pair_matches=ids))
    o = tracer.CoverageOptions(penalize_directional_coverage=0, max_coverage_paths=1 << min(n_vars, 30))
    out = tracer.assemblies_coverage(ss, rm, asms, [(0, ref)], options=o, genotype_filter=False)
    assert [(a.left_offset, a.right_offset) for a in out] == [(l, r) for l, r, _ in variants]
    for i, a in enumerate(out):
        n = 1 if is_delete[i] else 2
        assert (a.coverage, a.pair_coverage, a.ref_depth) == ([var_count[i]] * n, [pair_count[i]] * n, ref_count[i]), i


def test_discovery_engine_equals_oracle_on_a_120kb_sample(ctx):
    """A 120 kb diploid sample with SNPs, indels (1-50 bp) and structural insertions / deletions (50-1000 bp), 30x paired
    150 bp reads with errors, built by the library's create path (count, correct, seqset, readmap): the reference's whole
    discovery engine through the library -- two pieces x two directions of the push tracer, the serial pipeline with ML
    features, calc_coverage, genotype filter -- gives, VCF line for VCF line, what the oracle's restatement gives on the
    oracle's own seqset and readmap of the same corrected reads."""
    import torch  # noqa: F401
    from biograph_amd import synth
    from tests.test_oracle_tracer import dropper
    G = 120_000
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=78, device="cuda", return_genome=True,
                                            indel_rate=3e-4, sv_rate=5e-5)
    r = ctx.pack_reads_device(seq.data_ptr(), offs.data_ptr(), n)
    ks = ctx.kmer_count(r, 30, 5)
    c, _ = ctx.correct_reads(r, ks)
    ss, _ = ctx.build_seqset(c, 30)
    rm = ctx.build_readmap(ss, c, True)
    cseq, coffs = c.download()
    corrected = O.unpack_strings(cseq, coffs)
    ent = O.Entries.build(cseq, coffs, "cover", 30, threads=8)
    oss = O.Seqset.from_entries(ent, threads=8)
    records = [tuple(s for s in corrected[i:i + 2] if s) for i in range(0, len(corrected), 2)]
    orm = O.Readmap(oss, ent, [rec for rec in records if rec], True)
    ref = "".join("ACGT"[int(x)] for x in genome.cpu().tolist())
    parts = [(0, ref)]
    codes = genome.cpu().numpy().astype(np.uint8)
    g = T.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([G], np.uint64)))
    rmap = tracer.reference_ref_map(ss, [ref])
    mo = 105
    o_gpu, o_cpu, sc = tracer.TraceOptions(min_overlap=mo, output_ml_features=1), T.Options(min_overlap=mo), T.Scaffold(parts)
    want_in, got_in = [], []
    try:
        for start in (0, 60_000):
            for rev in (False, True):
                want_in += T.reversable_trace(g, sc, o_cpu, rev_comp=rev, start_offset=start, limit_offset=start + 60_000,
                                              anchor_drop=dropper(o_cpu))
                got_in += tracer.push_trace(ss, rm, rmap, parts, start=start, limit=start + 60_000, rev_comp=rev, options=o_gpu)
    finally:
        rmap.free()
    _, want_sections = A.variants_pipeline(A.AlignOptions(min_overlap=mo), sc, want_in, through_vcf_pad=True, ml_features=True)
    got_sections = tracer.process_assemblies(got_in, parts, options=o_gpu)
    co = C.CoverageOptions(output_ml_features=True)
    want = C.simple_genotype_filter(co, C.calc_coverage(g, [int(x) for x in orm.get("read_len")], sc, co, want_sections))
    got, lines = tracer.assemblies_coverage(ss, rm, got_sections, parts, options=tracer.CoverageOptions(output_ml_features=1),
                                            vcf_name="chrS")
    assert lines == [C.vcf_line(co, "chrS", lambda off, k: ref[off:off + k], a) for a in want]
    assert len(lines) > 100
    big = [x.split("\t") for x in lines if "SVTYPE" in x]
    assert len(big) >= 3  # structural variants come through as records with END / SVLEN / SVTYPE
    for h in (rm, ss, c, ks, r):
        h.free()


def test_current_default_chain_equals_oracle_on_a_60kb_sample(ctx):
    """`biograph discovery` as the current reference runs it by default (biograph_variants.cpp:188-202): pop tracers over the
    anchor drops of every work item (trace_ref::execute_work), no simple genotype filter, rvg_exclude last -- through the
    library (PushTracer.execute_work -> bg_assemblies_process -> bg_assemblies_coverage with rvg_exclude) and through the
    oracle's restatements, VCF line for VCF line.  Dense indels and structural variants give the pop tracers work."""
    from biograph_amd import synth
    from oracle import pop_tracer as P
    from tests.test_oracle_tracer import dropper
    G = 60_000
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=91, device="cuda", return_genome=True,
                                            indel_rate=1e-3, sv_rate=3e-4)
    r = ctx.pack_reads_device(seq.data_ptr(), offs.data_ptr(), n)
    ks = ctx.kmer_count(r, 30, 5)
    c, _ = ctx.correct_reads(r, ks)
    ss, _ = ctx.build_seqset(c, 30)
    rm = ctx.build_readmap(ss, c, True)
    cseq, coffs = c.download()
    corrected = O.unpack_strings(cseq, coffs)
    ent = O.Entries.build(cseq, coffs, "cover", 30, threads=8)
    oss = O.Seqset.from_entries(ent, threads=8)
    records = [tuple(s for s in corrected[i:i + 2] if s) for i in range(0, len(corrected), 2)]
    orm = O.Readmap(oss, ent, [rec for rec in records if rec], True)
    ref = "".join("ACGT"[int(x)] for x in genome.cpu().tolist())
    parts = [(0, ref)]
    codes = genome.cpu().numpy().astype(np.uint8)
    g = T.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([G], np.uint64)))
    strings = ent.strings()
    rmap = tracer.reference_ref_map(ss, [ref])
    mo = 105
    o_gpu, o_cpu, sc = tracer.TraceOptions(min_overlap=mo, output_ml_features=1), T.Options(min_overlap=mo), T.Scaffold(parts)
    po = P.PopOptions(o_cpu)
    want_in, got_in = [], []
    t = tracer.PushTracer(ss, rm, rmap, o_gpu)
    try:
        for start in (0, 30_000):
            want_in += P.execute_work(g, strings, sc, o_cpu, po, start=start, limit=start + 30_000, anchor_drop=dropper(o_cpu))
            got_in += t.execute_work(parts, start=start, limit=start + 30_000)
    finally:
        t.close()
        rmap.free()
    assert sum(1 for a in got_in if a.from_pop) > 0
    assert [(a.left_offset, a.right_offset, a.seq, a.from_pop) for a in got_in] == [(a.left_offset, a.right_offset, a.seq, "POP" in a.tags)
                                                                                   for a in want_in]
    _, want_sections = A.variants_pipeline(A.AlignOptions(min_overlap=mo), sc, want_in, through_vcf_pad=True, ml_features=True)
    got_sections = tracer.process_assemblies(got_in, parts, options=o_gpu)
    co = C.CoverageOptions(output_ml_features=True, simple_genotype_filter=False, rvg_exclude=True)
    covered = C.simple_genotype_filter(co, C.calc_coverage(g, [int(x) for x in orm.get("read_len")], sc, co, want_sections))
    want = C.rvg_exclude(co, covered)
    got, lines = tracer.assemblies_coverage(ss, rm, got_sections, parts, vcf_name="chrS",
                                            options=tracer.CoverageOptions(output_ml_features=1, simple_genotype_filter=0, rvg_exclude=1))
    assert lines == [C.vcf_line(co, "chrS", lambda off, k: ref[off:off + k], a) for a in want]
    assert 50 < len(lines) < len(covered)  # rvg_exclude dropped something
    for h in (rm, ss, c, ks, r):
        h.free()


def test_walk_chains_batch_with_different_sequences(ctx):
    """bg_walk_chains_batch: chains of different groups, each with its own bases, output offset and stride, in one call
    == each chain walked on its own with bg_walk_chains."""
    import ctypes as CT
    from biograph_amd import capi
    pairs, ref = synthetic_world(ctx, 3000, 23)
    ss, rm, rmap = gpu_world(ctx, pairs, [], [ref])
    rmap.free()

    class Chain(CT.Structure):
        _fields_ = [("begin", CT.c_uint64), ("end", CT.c_uint64), ("code_offset", CT.c_uint64), ("out_offset", CT.c_uint64),
                    ("out_stride", CT.c_uint64), ("size", CT.c_uint32), ("len", CT.c_uint32)]
    lut = {c: i for i, c in enumerate("ACGT")}
    seqs = [ref[100:100 + 37], ref[900:900 + 210], ref[2000:2000 + 1]]
    codes = np.concatenate([np.array([lut[c] for c in s_], np.uint8) for s_ in seqs])
    offs = np.cumsum([0] + [len(s_) for s_ in seqs])
    n = len(seqs)
    chains = (Chain * n)()
    out_words = 0
    layout = []
    for i, s_ in enumerate(seqs):
        stride = 3 + 3 * i  # different strides
        chains[i] = Chain(0, ss.size(), int(offs[i]), out_words, stride, 0, len(s_))
        layout.append((out_words, stride))
        out_words += len(s_) * stride
    out = np.full(out_words, 0xDEADBEEF, np.uint64)
    capi.check(capi.lib().bg_walk_chains_batch(ctx.h, ss.h, n, chains, capi.np_ptr(codes), len(codes), capi.np_ptr(out), out_words))
    for i, s_ in enumerate(seqs):
        start = np.array([[0, ss.size(), 0]], np.uint64)
        one = np.zeros((len(s_), 1, 3), np.uint64)
        capi.check(capi.lib().bg_walk_chains(ctx.h, ss.h, 1, capi.np_ptr(start), capi.np_ptr(codes[offs[i]:offs[i + 1]].copy()), len(s_),
                                             capi.np_ptr(one)))
        at, stride = layout[i]
        got = out[at:at + len(s_) * stride].reshape(len(s_), stride)
        assert (got[:, :3] == one[:, 0, :]).all()
        assert (got[:, 3:] == 0).all()  # the buffer is rewritten as a whole: unnamed words come back 0
    bad = (Chain * 1)(Chain(0, ss.size(), len(codes) - 1, 0, 3, 0, 5))  # reaches past the bases
    assert capi.lib().bg_walk_chains_batch(ctx.h, ss.h, 1, bad, capi.np_ptr(codes), len(codes), capi.np_ptr(out), out_words) != 0
