"""The population configuration (BASELINE.json configs[4]) with all ranks simulated on one GPU (LocalSimComm): samples
of one genome built sharded, merged through the records exchange, readmaps built on the merged replica, discover
shared out by reference chunk.  Parity: the sharded merge gives the single-GPU `bg_seqset_merge` (itself pinned on the
reference's golden merged biograph, tests/test_gpu_merge.py) member for member; a readmap built on the merged seqset
equals the sample's readmap migrated through the merge map (make_readmap.cpp:455-519); chunked discover gives the same
calls whatever the number of ranks."""
import numpy as np
import pytest
import torch

import biograph_amd as bg
from biograph_amd import population, sharded, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SEED = 31


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def _sample_inputs(G, cov, L, sample, P):
    """Each rank's share of one sample's read pairs (the bench's generator, shard=(r, P)) + all of them in rank order."""
    inputs, every = {}, []
    for r in range(P):
        seq, offs, n = synth.make_reads(genome_len=G, coverage=cov, read_len=L, seed=SEED, device="cpu", shard=(r, P),
                                        sample=sample)
        every += O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
        inputs[r] = (seq.cuda(), offs.cuda(), n)
    return inputs, every


def _single_gpu_sample(ctx, reads, prm):
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, prm.kmer_size, prm.min_kmer_count)
    cor, _ = ctx.correct_reads(r, ks, prm.correct.max_corrections, prm.correct.min_good_run, prm.correct.trim_after_portion)
    ss, _ = ctx.build_seqset(cor, prm.kmer_size)
    rm = ctx.build_readmap(ss, cor, True)
    return ss, rm, cor


@pytest.mark.parametrize("P,n_samples,read_lens", [(1, 2, None), (3, 4, None), (8, 3, None), (2, 2, (150, 100))])
def test_sharded_merge_equals_single_gpu_merge(ctx, P, n_samples, read_lens):
    """read_lens: samples of different read lengths (the merged key width is the widest input's,
    seqset_merger.cpp:57-60)."""
    prm = ctx.default_params()
    comm = sharded.LocalSimComm(P)
    ctxs = {r: ctx for r in range(P)}
    rows = {r: [] for r in range(P)}
    singles, max_len = [], 0
    for s in range(n_samples):
        L = read_lens[s] if read_lens else 150
        inputs, every = _sample_inputs(9000, 30, L, s, P)
        out = sharded.sharded_create(comm, ctxs, inputs, prm, gather=False, keep_entry_rows=True)
        st = out[0][1]
        max_len = max(max_len, st["max_len"])
        ss, _ = ctx.create(every)
        assert st["n_entries"] == ss.size() == sum(out[r][0].entry_rows.shape[0] for r in range(P))
        singles.append(ss)
        for r in range(P):
            rows[r].append(out[r][0].entry_rows)
            out[r][0].free()
    want, _ = ctx.merge_seqsets(singles, want_maps=False)
    got = sharded.sharded_merge(comm, ctxs, rows, max_len, gather=True)
    ms, mst = got[0]
    assert mst["n_entries"] == want.size() == ms.size() and mst["n_input_entries"] == sum(s.size() for s in singles)
    assert want.size() > max(s.size() for s in singles)  # the samples differ: the union is larger than any of them
    have, ref = ms.parts(), want.parts()
    for name in bg.PART_NAMES:
        assert have[name] == ref[name], name


def _rows_of(strings, W):
    """Entries as the rows sharded_merge takes: W key words (2 bits a base, first base in the top bits), then the length."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    rows = np.zeros((len(strings), W + 1), np.uint64)
    for i, st in enumerate(strings):
        v = 0
        for ch in st:
            v = (v << 2) | code[ch]
        v <<= 2 * (32 * W - len(st))
        for w in range(W):
            rows[i, w] = (v >> (64 * (W - 1 - w))) & 0xffffffffffffffff
        rows[i, W] = len(st)
    return rows


@pytest.mark.parametrize("P", [2, 3])
def test_sharded_merge_of_the_golden_family(ctx, golden_dir, P):
    """configs[4] on the reference's own data over P ranks: the entries of the golden father / mother / proband seqsets,
    dealt to the ranks in stripes, through sharded_merge == datasets/lambdaToyData/benchmark/family_lambda.bg (entries, sizes,
    shared, fixed; every part == the single-GPU bg_seqset_merge, whose prev bits tests/test_gpu_merge.py compares with the
    file's)."""
    import torch
    names = ["father", "mother", "proband"]
    dm = O.load_seqset_zip(golden_dir + "/family_lambda/seqset")
    ins = [ctx.open_seqset(golden_dir + "/%s_lambda/seqset" % s) for s in names]
    want, _ = ctx.merge_seqsets(ins, want_maps=False)
    comm = sharded.LocalSimComm(P)
    ctxs = {r: ctx for r in range(P)}
    rows = {r: [] for r in range(P)}
    for ss in ins:
        full = _rows_of(ss.entries(), 5)
        for r in range(P):
            rows[r].append(torch.from_numpy(full[r::P].copy().view(np.int64)).cuda())
    got = sharded.sharded_merge(comm, ctxs, rows, 150, gather=True)
    ms, mst = got[0]
    assert mst["n_entries"] == ms.size() == dm["n"] == 103996
    assert ms.entries() == O.decode_entries(dm)
    sz, sh = ms.arrays()
    assert (sz == dm["sizes"]).all() and (sh == dm["shared"]).all()
    assert ms.part("fixed") == np.asarray(dm["fixed"], np.uint64).tobytes()
    have, ref = ms.parts(), want.parts()
    for name in bg.PART_NAMES:
        assert have[name] == ref[name], name


@pytest.mark.parametrize("P", [1, 3])
def test_population_create_readmaps_equal_migrated_readmaps(ctx, tmp_path, P):
    """The whole flow: merged container written in place by the ranks, replicas opened from it, every sample's
    readmap built on the replica by FM search from the all-gathered corrected reads == the single-GPU readmap of the
    sample migrated onto the single-GPU merge."""
    n_samples = 4
    prm = ctx.default_params()
    comm = sharded.LocalSimComm(P)
    ctxs = {r: ctx for r in range(P)}
    sample_inputs, singles = {}, {}
    for s in range(n_samples):
        name = "sample%d" % s
        sample_inputs[name], every = _sample_inputs(9000, 30, 150, s, P)
        singles[name] = _single_gpu_sample(ctx, every, prm)
    names = sorted(sample_inputs)
    path = str(tmp_path / "merged_seqset")
    replicas, readmaps, stats = population.population_create(comm, ctxs, sample_inputs, path, prm)
    want, maps = ctx.merge_seqsets([singles[n][0] for n in names])
    assert stats["merge"]["n_entries"] == want.size()
    for r in range(P):
        assert replicas[r].size() == want.size()
    have, ref = replicas[0].parts(), want.parts()
    for name in bg.PART_NAMES:
        assert have[name] == ref[name], name
    for i, n in enumerate(names):
        mig = ctx.migrate_readmap(singles[n][1], maps[i], want)
        wparts = mig.parts()
        for r in (0, P - 1):
            gparts = readmaps[r][n].parts()
            for part in bg.READMAP_PART_NAMES:
                assert gparts[part] == wparts[part], (n, r, part)


def test_marked_readmap_build_equals_migration_where_plain_lookup_does_not(ctx):
    """A read R of sample 1 whose sample entry is R+T while the merged seqset also holds R+A... from sample 2: the
    migrated row sits on R+T (the image of the sample's entry), the first merged entry with prefix R is R+A.  The marks
    of bg_seqset_mark_inputs select the right one, on resident keys and by FM search on an opened seqset."""
    rng = np.random.default_rng(5)
    core = "".join("ACGT"[x] for x in rng.integers(0, 4, 60))
    s1 = [core + "TTGCA", core, core[5:] + "TTGCAGGA"]           # `core` alone is a read of sample 1
    s2 = [core + "AACCG", core[3:] + "AACCGTT"]
    prm = ctx.default_params(kmer_size=16, min_kmer_count=1)

    def build(reads):
        r = ctx.pack_reads(reads)
        ks = ctx.kmer_count(r, prm.kmer_size, prm.min_kmer_count)
        cor, _ = ctx.correct_reads(r, ks, prm.correct.max_corrections, prm.correct.min_good_run, prm.correct.trim_after_portion)
        ss, _ = ctx.build_seqset(cor, prm.kmer_size)
        return ss, ctx.build_readmap(ss, cor, False), cor

    a, b = build(s1), build(s2)
    merged, maps = ctx.merge_seqsets([a[0], b[0]])
    mig = ctx.migrate_readmap(a[1], maps[0], merged)
    plain = ctx.build_readmap(merged, a[2], False)
    assert plain.parts() != mig.parts()                          # the case is real
    marks = np.zeros(merged.size(), np.uint8)
    marks[maps[0]] = 1
    dm = torch.from_numpy(marks).cuda()
    assert ctx.build_readmap(merged, a[2], False, marks=dm).parts() == mig.parts()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        merged.write(d + "/seqset")
        opened = ctx.open_seqset(d + "/seqset")
        assert ctx.build_readmap(opened, a[2], False, marks=dm).parts() == mig.parts()
        wrong = torch.zeros_like(dm)
        with pytest.raises(bg.capi.BgError):
            ctx.build_readmap(opened, a[2], False, marks=wrong)


def test_sharded_discover_is_independent_of_the_rank_count(ctx):
    """The calls of chunked discover = the union over chunks, whoever ran them; chunking itself only loses calls at
    the piece ends (trace_ref.cpp:44-63 cuts the same way)."""
    G = 60_000
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=SEED, device="cpu", sample=1,
                                            return_genome=True)
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    prm = ctx.default_params()
    ss, rm, cor = _single_gpu_sample(ctx, reads, prm)
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.numpy()].tobytes().decode()
    scaffolds = {"1": ref[:35_000], "X": ref[35_000:]}
    results = []
    for P in (1, 2, 5):
        comm = sharded.LocalSimComm(P)
        got = population.sharded_discover(comm, {r: ss for r in range(P)}, {r: {"s": rm} for r in range(P)}, scaffolds,
                                          min_overlap=105, split_size=10_000)
        results.append(got["s"])
    assert results[0] == results[1] == results[2] and len(results[0]) > 30
    whole = population.sharded_discover(sharded.LocalSimComm(1), {0: ss}, {0: {"s": rm}}, scaffolds, min_overlap=105,
                                        split_size=10 ** 9)["s"]
    lost = set(whole) - set(results[0])
    assert len(lost) <= 0.1 * len(whole)
    # SNPs planted by the generator for sample 1 are found
    sites = {x[1] for x in results[0] if x[0] == "1" and x[2] == x[1] + 1 and len(x[3]) == 1}
    assert len(sites) > 10


def test_sharded_discover_with_the_push_tracer(ctx):
    """engine="push": the work items of trace_ref through bg_tracer_trace, forward and reverse; same calls whoever runs them,
    and the planted SNPs of the sample are inside them."""
    from biograph_amd import tracer
    G = 40_000
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=SEED, device="cpu", sample=2,
                                            return_genome=True)
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    prm = ctx.default_params()
    ss, rm, cor = _single_gpu_sample(ctx, reads, prm)
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.numpy()].tobytes().decode()
    scaffolds = {"1": ref[:25_000] + "N" * 30 + ref[25_030:]}
    rmap = tracer.reference_ref_map(ss, [ref[:25_000], ref[25_030:]])
    try:
        results = []
        for P in (1, 3):
            comm = sharded.LocalSimComm(P)
            got = population.sharded_discover(comm, {r: ss for r in range(P)}, {r: {"s": rm} for r in range(P)}, scaffolds,
                                              min_overlap=105, split_size=8_000, engine="push", ref_maps={r: rmap for r in range(P)})
            results.append(got["s"])
        assert results[0] == results[1] and len(results[0]) > 20
        sites = synth.variant_sites(G, synth.sample_seed(SEED, 2) - 1)
        assert len(sites) >= 1
        # every call is anchored on the left or on the right inside the scaffold
        assert all(0 <= lo < ro <= len(scaffolds["1"]) + 2000 for _, lo, ro, _ in results[0])
    finally:
        rmap.free()


def test_sharded_discovery_vcf_is_the_engine_of_one_process(ctx):
    """configs[4]'s discover step with the reference's engine: (sample, piece) work items traced on P simulated ranks,
    pipeline + coverage + rvg_exclude on the owner of (sample, scaffold): the VCF records do not depend on P and equal what
    one process gets by running all work items itself (the flow of `bgbinary discovery` with its defaults: pop tracers on)."""
    from biograph_amd import tracer
    G = 40_000
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=SEED, device="cpu", sample=2,
                                            return_genome=True)
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    prm = ctx.default_params()
    ss, rm, cor = _single_gpu_sample(ctx, reads, prm)
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.numpy()].tobytes().decode()
    scaffolds = {"1": ref[:25_000] + "N" * 30 + ref[25_030:], "2": ref[30_000:38_000]}
    rmap = tracer.reference_ref_map(ss, [ref[:25_000], ref[25_030:], ref[30_000:38_000]])
    try:
        results = []
        for P in (1, 3):
            comm = sharded.LocalSimComm(P)
            got = population.sharded_discovery_vcf(comm, {r: ss for r in range(P)}, {r: {"a": rm, "b": rm} for r in range(P)},
                                                   scaffolds, split_size=8_000, ref_maps={r: rmap for r in range(P)})
            results.append(got)
        assert results[0] == results[1]
        assert results[0]["a"] == results[0]["b"] and len(results[0]["a"]) > 20
        # the same by hand in one process
        opt = tracer.TraceOptions(min_overlap=105)
        want = []
        session = tracer.PushTracer(ss, rm, rmap, opt)
        for name, sq in scaffolds.items():
            ext = population.extents_of(sq)
            asms = []
            for start in range(0, len(sq), 8_000):  # one work item of trace_ref each: push forward / reverse, pop forward / reverse
                asms += session.execute_work(ext, start=start, limit=min(len(sq), start + 8_000), end_pos=len(sq))
            sections = tracer.process_assemblies(asms, ext, end_pos=len(sq), options=opt)
            want += tracer.assemblies_coverage(ss, rm, sections, ext, end_pos=len(sq), vcf_name=name,
                                               options=tracer.CoverageOptions(simple_genotype_filter=0, rvg_exclude=1))[1]
        session.close()
        assert results[0]["a"] == want
        snps = [x.split("\t") for x in want if len(x.split("\t")[3]) == 1 and len(x.split("\t")[4]) == 1]
        assert len(snps) > 10 and all(f[6] == "PASS" and f[8] == "GT:PG:GQ:PI:OV:DP:AD:PDP:PAD" for f in snps)
    finally:
        rmap.free()
