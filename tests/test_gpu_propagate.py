"""GPU parity of row G4 (readmap queries + seqset_path::propagate_from_end) through the C ABI, against
the oracle's literal restatements, on the reference-written lambda seqset + readmap and on a readmap
built by the GPU path itself."""
import numpy as np
import pytest

import biograph_amd as bg
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(golden_dir):
    ctx = bg.Context(0)
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    g = O.decode_readmap(golden_dir + "/father_lambda/readmap", d)
    ent = O.Entries.from_strings(O.decode_entries(d))
    oss = O.Seqset.from_entries(ent)
    orm = O.Readmap.from_arrays(d["n"], g["entry"], g["read_len"], g["mate_loop_ptr"], g["is_forward"], 150)
    ss = ctx.open_seqset(golden_dir + "/father_lambda/seqset")
    rm = ctx.open_readmap(golden_dir + "/father_lambda/readmap")
    ref = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">"))
    yield ctx, d, g, oss, orm, ss, rm, ref
    ctx.close()


def test_open_readmap_arrays_match_golden(world):
    ctx, d, g, oss, orm, ss, rm, ref = world
    assert rm.size() == g["n_reads"] == 97860 and rm.max_read_len == 150
    e, rl, ml, f = rm.arrays()
    assert (e == g["entry"]).all() and (rl == g["read_len"]).all()
    assert (ml == g["mate_loop_ptr"]).all() and (f == g["is_forward"]).all()
    for name in bg.READMAP_PART_NAMES:   # re-serialising the opened readmap gives the file's own bytes
        assert rm.part(name) == g["raw"][name], name


def test_read_info_and_prefix_reads_match_oracle(world):
    ctx, d, g, oss, orm, ss, rm, ref = world
    rng = np.random.default_rng(11)
    ids = [int(x) for x in rng.choice(g["n_reads"], 3000, replace=False)]
    info = ss.read_info_batch(rm, ids)
    assert info == [orm.read_info(oss, i) for i in ids]
    # ranges: the reads' own entries, truncations of them, and their walk neighbours
    ranges = [t[3] for t in info]
    ranges += [oss.find(O.decode_entries(d)[int(g["entry"][i])][:int(L)]) for i, L in zip(ids[:500], rng.integers(25, 150, 500))]
    ranges.append((0, d["n"], 0))
    got = ss.prefix_reads_batch(rm, ranges, 0, 512)
    for r, ids_got in zip(ranges, got):
        assert ids_got == orm.prefix_reads(oss, r), r
    # read_len_limit raises the floor (readmap.cpp:439)
    r = ranges[0]
    assert ss.prefix_reads_batch(rm, [r], 149, 64)[0] == orm.prefix_reads(oss, r, 149)
    with pytest.raises(bg.BgError):   # readmap of another seqset (:436-438)
        small, _ = ctx.build_seqset(ctx.pack_reads(["ACGTACGTAC"]), 30)
        small.prefix_reads_batch(rm, [(0, 1, 1)])


def test_propagate_from_end_matches_oracle(world):
    ctx, d, g, oss, orm, ss, rm, ref = world
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    n = d["n"]
    rng = np.random.default_rng(5)
    seqs, ends, marks = [], [], []
    for k in range(48):
        a = int(rng.integers(0, len(ref) - 700))
        ln = int(rng.integers(1, 600))
        s = ref[a:a + ln]
        if k % 5 == 0:   # a variant in the middle: the walk has to drop and re-grow context
            s = s[:ln // 2] + "ACGT"[int(rng.integers(4))] + s[ln // 2:]
        seq = np.array([3 - code[c] for c in reversed(s)], np.uint8)
        seqs.append(seq)
        if k % 3 == 0:
            ends.append([(0, n, 0)])
        else:   # start from the rc ranges a neighbouring assembly ends with
            pre = np.array([3 - code[c] for c in reversed(ref[a + ln:a + ln + 150])], np.uint8)
            ends.append(oss.propagate_from_end(pre, [(0, n, 0)])[0][0])
        marks.append(sorted(set(int(x) for x in rng.integers(1, max(len(seq), 2), 3) if 0 < x < len(seq))))
    got = ss.propagate_from_end_batch(seqs, ends, rm, 20, marks, mate_cap=8192)
    plain = ss.propagate_from_end_batch(seqs, ends)
    # without a readmap the assemblies that start from one range run a lane each (k_propagate_single), the others a wave each:
    # the same sets at the marks as with the readmap
    plain_marks = ss.propagate_from_end_batch(seqs, ends, None, 20, marks)
    assert sum(1 for e in ends if len(e) == 1) > 10
    for k in range(len(seqs)):
        assert plain_marks[k][0] == got[k][0] and plain_marks[k][1] == [], k
        want_ent, want_mates = oss.propagate_from_end(seqs[k], ends[k], orm, 20, marks[k])
        want_ent.pop(len(seqs[k]), None) if len(seqs[k]) else None
        ent, mates = got[k]
        assert ent == {o: v for o, v in want_ent.items() if o != len(seqs[k]) or o == 0}, k
        assert mates == sorted(want_mates, key=lambda r: (r[0], -r[1], r[2])), k
        assert plain[k][0] == {0: want_ent[0]} and plain[k][1] == []
    assert sum(len(m) for _, m in got) > 100
    # max_pairs_per_read = 0: every range with a mated prefix read is skipped
    assert all(m == [] for _, m in ss.propagate_from_end_batch(seqs[:8], ends[:8], rm, 0))


def test_readmap_built_on_gpu_answers_like_oracle(world):
    """Same queries on a seqset + readmap built by the GPU create path (not opened from files)."""
    ctx = world[0]
    from biograph_amd import synth
    seq, offs, nreads = synth.make_reads(genome_len=30000, coverage=30, read_len=150, seed=9, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, 30, 5)
    c, _ = ctx.correct_reads(r, ks)
    ss, _ = ctx.build_seqset(c, 30)
    rm = ctx.build_readmap(ss, c, True)
    e, rl, ml, f = rm.arrays()
    ent = O.Entries.from_strings(ss.entries())
    oss = O.Seqset.from_entries(ent)
    orm = O.Readmap.from_arrays(ss.size(), e, rl, ml, f, rm.max_read_len)
    ids = list(range(0, rm.size(), 37))
    info = ss.read_info_batch(rm, ids)
    assert info == [orm.read_info(oss, i) for i in ids]
    got = ss.prefix_reads_batch(rm, [t[3] for t in info], 0, 512)
    assert got == [orm.prefix_reads(oss, t[3]) for t in info]
    cseq, coffs = c.download()
    kept = [s for s in O.unpack_strings(cseq, coffs) if s][:40]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    seqs = [np.array([3 - code[ch] for ch in reversed(s)], np.uint8) for s in kept]
    ends = [[(0, ss.size(), 0)]] * len(seqs)
    res = ss.propagate_from_end_batch(seqs, ends, rm, 20)
    for k in range(len(seqs)):
        we, wm = oss.propagate_from_end(seqs[k], ends[k], orm, 20)
        assert res[k][0] == {0: we[0]} and res[k][1] == sorted(wm, key=lambda r: (r[0], -r[1], r[2]))


def test_open_legacy_readmap_and_seqset(world, golden_dir):
    """Containers written by v3.1.1 (seqset 1.0.0 with raw uint8 sizes, readmap 1.1.0 with raw lengths and
    32-bit loop pointers) open through the same entry points."""
    ctx = world[0]
    d = O.load_seqset_zip(golden_dir + "/e_coli_10000snp/seqset")
    g = O.decode_readmap(golden_dir + "/e_coli_10000snp/readmap", d)
    ss = ctx.open_seqset(golden_dir + "/e_coli_10000snp/seqset")
    rm = ctx.open_readmap(golden_dir + "/e_coli_10000snp/readmap")
    assert ss.size() == 19935 and rm.size() == 16888 and rm.max_read_len == 35
    e, rl, ml, f = rm.arrays()
    assert (e == g["entry"]).all() and (rl == g["read_len"]).all()
    assert (ml == g["mate_loop_ptr"]).all() and (f == g["is_forward"]).all()
    # unpaired: nobody has a mate; rev_comp is the loop partner
    info = ss.read_info_batch(rm, list(range(0, 16888, 211)))
    assert all(not hm and ml[i] == rc for (hm, _, rc, _), i in zip(info, range(0, 16888, 211)))
    # a mate_pair_ptr-only readmap opens (pairing queries work on it), but what needs the mate loop table says what
    # readmap::get_rev_comp says (bio_base/readmap.cpp:248-253)
    legacy = ctx.open_readmap(golden_dir + "/e_coli_merged/readmap_e_coli_test")
    assert legacy.size() > 0
    merged = ctx.open_seqset(golden_dir + "/e_coli_merged/seqset")
    with pytest.raises(bg.BgError) as err:
        merged.read_info_batch(legacy, [0])
    assert "upgrade" in str(err.value)
