"""GPU parity tests: every stage of the HIP path, called through the C ABI (libbgseq.so), compared
bit-for-bit with the CPU oracle on the same seeded inputs and with the reference's golden files."""
import os

import numpy as np
import pytest

import biograph_amd as bg
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def _oracle_stage(reads, k, min_count, maxc=8, mgr=2, trim=0.7):
    seq, offs = O.pack_strings(reads)
    ks = O.KmerSet.count(seq, offs, k, min_count)
    cseq, coffs, keep, stats = O.correct_reads(ks, seq, offs, maxc, mgr, trim)
    return ks, cseq, coffs, keep, stats


def test_pack_roundtrip_and_errors(ctx):
    reads = util.synth_reads(3, n_reads=500, read_len=97, var_len=True, n_rate=0.0, err=0.0)
    reads += ["", "A", "acgtACGT", "T" * 255]
    r = ctx.pack_reads(reads)
    seq, offs = r.download()
    assert O.unpack_strings(seq, offs) == [s.upper() for s in reads]
    assert r.max_len == 255 and r.total_bases == sum(map(len, reads))
    with pytest.raises(bg.BgError) as e:
        ctx.pack_reads(["ACGTXACGT"])
    assert e.value.code == -4
    with pytest.raises(bg.BgError):
        ctx.pack_reads(["A" * 256])
    empty = ctx.pack_reads([])
    assert len(empty) == 0


@pytest.mark.parametrize("seed,k,mc,rl,nrate", [(1, 30, 5, 150, 0.001), (2, 15, 3, 80, 0.01), (3, 31, 2, 100, 0.0),
                                                (4, 5, 1, 40, 0.02), (5, 21, 4, 255, 0.002)])
def test_kmer_count_matches_oracle(ctx, seed, k, mc, rl, nrate):
    reads = util.synth_reads(seed, genome_len=4000, n_reads=1500, read_len=rl, err=0.01, n_rate=nrate, var_len=True)
    seq, offs = O.pack_strings(reads)
    oks = O.KmerSet.count(seq, offs, k, mc)
    ok, of, ofw, orv = oks.arrays()
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, k, mc)
    gk, gf, gc = ks.download()
    assert len(gk) == len(ok)
    assert (gk == ok).all()
    assert (gf == of).all()
    assert (gc == ofw + orv).all()
    allk = O.KmerSet.count(seq, offs, k, 1, keep_all=True)
    assert ks.distinct == len(allk)
    a = allk.arrays()
    assert ks.instances >= int(a[2].sum() + a[3].sum())


@pytest.mark.parametrize("seed,k,err,nrate,maxc,mgr,trim", [(1, 30, 0.004, 0.0005, 8, 2, 0.7),
                                                            (2, 15, 0.03, 0.01, 8, 2, 0.7),
                                                            (3, 21, 0.02, 0.005, 2, 3, 0.9),
                                                            (4, 12, 0.05, 0.02, 8, 0, 0.0),
                                                            (5, 30, 0.01, 0.0, 0, 2, 0.5)])
def test_correction_matches_oracle(ctx, seed, k, err, nrate, maxc, mgr, trim):
    reads = util.synth_reads(seed, genome_len=3000, n_reads=2000, read_len=120, err=err, n_rate=nrate, var_len=True)
    oks, cseq, coffs, keep, stats = _oracle_stage(reads, k, 3, maxc, mgr, trim)
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, k, 3)
    c, st = ctx.correct_reads(r, ks, maxc, mgr, trim)
    gseq, goffs = c.download()
    assert (goffs == coffs).all()
    assert (gseq == cseq).all()
    for key in ("corrected_reads", "corrected_bases", "failed", "modified", "modified_bases", "truncated",
                "dropped_bases"):
        assert st[key] == stats[key], key


def _check_seqset_against_oracle(ss, reads_corrected):
    seq, offs = O.pack_strings(reads_corrected)
    ent = O.Entries.build(seq, offs, "simple")
    oss = O.Seqset.from_entries(ent)
    assert ss.size() == len(ent)
    assert ss.entries() == ent.strings()
    s, sh = ss.arrays()
    os_, osh = oss.arrays()
    assert (s == os_).all() and (sh == osh).all()
    want = oss.parts()
    got = ss.parts()
    for name in bg.PART_NAMES:
        assert got[name] == want[name], name
    return oss


@pytest.mark.parametrize("seed,alphabet,K,rl,var", [(1, "ACGT", 30, 150, False), (2, "AC", 8, 70, True),
                                                     (3, "A", 5, 40, True), (4, "ACGT", 12, 100, True),
                                                     (5, "AT", 31, 255, True), (6, "ACGT", 4, 33, True),
                                                     (7, "AG", 20, 200, False)])
def test_seqset_build_matches_oracle(ctx, seed, alphabet, K, rl, var):
    """Entry set, sizes, shared, prev bits, rank index and packed parts; low-entropy alphabets force
    long equal prefixes (tie refinement rounds) and tandem repeats force self-covering reads."""
    reads = util.synth_reads(seed, genome_len=1500, n_reads=400, read_len=rl, err=0.01, n_rate=0.0,
                             alphabet=alphabet, var_len=var, repeat=(7, 20))
    r = ctx.pack_reads(reads)
    ss, st = ctx.build_seqset(r, K)
    _check_seqset_against_oracle(ss, reads)
    assert st["n_entries"] == ss.size() and st["n_records"] >= st["n_entries"]


ROWS = {"BG_ITEMS_BUILD": "0"}  # the entries through rows (the switches below are that path's; the default is the items path)


@pytest.mark.parametrize("env", [{}, {"BG_ITEMS_KEYS": "1"}, {"BG_VERIFY_CLOSURE": "1"}, {"BG_ITEMS_LEVELS": "3"}, {"BG_ITEMS_EMIT": "plain"}, {"BG_COVER_ITEMS": "0"}, ROWS, dict(ROWS, BG_VERIFY_CLOSURE="1"), dict(ROWS, BG_PREV_FULL_SEARCH="1"),
                                 dict(ROWS, BG_NO_MEDIUM_GROUPS="1"), dict(ROWS, BG_SMALL_TIES_EARLY="1"), dict(ROWS, BG_ENTRIES_UNFUSED="1"),
                                 dict(ROWS, BG_PREV_MERGE="1"), dict(ROWS, BG_RADIX_ONESWEEP="1"), {"BG_PREV_SEARCH_ALL": "1"},
                                 {"BG_PREV_SEARCH_ALL": "1", "BG_VERIFY_CLOSURE": "1"}, dict(ROWS, BG_LOOKBACK_FLAT="1"),
                                 {"BG_COVER_COUNT_ROUNDS": "1"}, dict(ROWS, BG_COVER_COUNT_ROUNDS="1")])
@pytest.mark.parametrize("seed,alphabet,rl", [(11, "ACGT", 150), (12, "AC", 200), (13, "AT", 255)])
def test_seqset_build_search_modes(ctx, monkeypatch, env, seed, alphabet, rl):
    """(BG_PREV_MERGE: the prev bits as one merge of tails against chunks of entries; BG_RADIX_ONESWEEP: radix passes with
    decoupled look-back -- both kept for A/B measurements.  BG_PREV_SEARCH_ALL: every entry's tail searched, as before
    round 5 -- by default the prev bits are set where the sorted records meet and only the listed tails are searched.)
    The prev-bit search on two-word prefixes re-checked against the full-key search (BG_VERIFY_CLOSURE),
    the full-key search alone, the tie handling without the medium-group kernel, the small tie groups ordered by
    k_sort_small_groups instead of inside the entry build's tile, and the unfused entry build: same seqset."""
    for a, b in env.items():
        monkeypatch.setenv(a, b)
    reads = util.synth_reads(seed, genome_len=2500, n_reads=700, read_len=rl, err=0.01, n_rate=0.0,
                             alphabet=alphabet, var_len=True, repeat=(9, 40))
    ss, _ = ctx.build_seqset(ctx.pack_reads(reads), 20)
    _check_seqset_against_oracle(ss, reads)


@pytest.mark.parametrize("env", [{}, ROWS, dict(ROWS, BG_NO_LARGE_GROUPS="1")])
def test_seqset_build_big_tie_groups(ctx, monkeypatch, env):
    """Tie groups of the suffix sorts beyond the medium-group kernel: 1500 and 2800 distinct reads behind one 20-base
    prefix each (the second launch with a CU's whole LDS), 4000 behind a third (more than the LDS takes: the radix
    rounds), 3500 copies of one read and 40 empty reads (groups of equal keys, left as they are) — and the same
    through the radix rounds alone."""
    for a, b in env.items():
        monkeypatch.setenv(a, b)
    rng = np.random.default_rng(77)
    def rnd(n):
        return "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
    reads = []
    for prefix, count in ((rnd(20), 1500), (rnd(20), 2800), (rnd(20), 4000)):
        reads += [prefix + rnd(int(rng.integers(10, 60))) for _ in range(count)]
    reads += [rnd(70)] * 3500 + [""] * 40
    reads += util.synth_reads(5, genome_len=1500, n_reads=300, read_len=80, err=0.0, n_rate=0.0, var_len=True)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    ss, _ = ctx.build_seqset(ctx.pack_reads(reads), 12)
    _check_seqset_against_oracle(ss, reads)


@pytest.mark.parametrize("env", [{}, {"BG_SMALL_TIES_EARLY": "1"}])
def test_seqset_build_tie_groups_around_the_small_limit(ctx, monkeypatch, env):
    """Tie groups of 28 .. 37 records behind one 16-base prefix each (32 is the most the entry build's tile orders
    itself; 33 and up go to the medium-group kernel), 300 of them and the smaller groups their shifted suffixes make, so
    that groups of every size straddle the 512-position tiles of k_gather_dedup_build."""
    for a, b in env.items():
        monkeypatch.setenv(a, b)
    rng = np.random.default_rng(4242)
    def rnd(n):
        return "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
    reads = []
    for g in range(300):
        prefix = rnd(16)
        for _ in range(28 + g % 10):
            reads.append(prefix + rnd(int(rng.integers(4, 40))))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    ss, _ = ctx.build_seqset(ctx.pack_reads(reads), 10)
    _check_seqset_against_oracle(ss, reads)


def test_builder_test_known_answers(ctx):
    # modules/build_seqset/builder_test.cpp:52-123 through the HIP path
    t, trc = O.tseq, O.tseq_rc
    ss, _ = ctx.build_seqset(ctx.pack_reads([t("a")]), 30)
    assert ss.entries() == ["AAAATTAC", "AAATTAC", "AATTAC", "AATTTTAG", "AC", "AG", "ATTAC", "ATTTTAG",
                            "CTAAAATTAC", "GTAATTTTAG", "TAAAATTAC", "TAATTTTAG", "TAC", "TAG", "TTAC", "TTAG",
                            "TTTAG", "TTTTAG"]
    for reads, n in [([t("abcdefg")], 129), ([t("abcd"), t("cdef"), trc("efgh")], 152),
                     ([t("ab"), t("bc"), t("cd"), t("be")], 91), ([t("abc"), t("cde")], 89),
                     ([t("abc"), t("efg")], 99)]:
        for K in (5, 30):
            ss, _ = ctx.build_seqset(ctx.pack_reads(reads), K)
            assert ss.size() == n


def test_empty_and_degenerate_inputs(ctx):
    ss, st = ctx.create([])
    assert ss.size() == 0 and ss.part("fixed") == bytes(40)
    ss, st = ctx.create(["ACGT"])  # shorter than k: nothing survives
    assert ss.size() == 0
    ss, _ = ctx.build_seqset(ctx.pack_reads(["A"]), 30)
    assert ss.entries() == ["A", "T"]
    ss, _ = ctx.build_seqset(ctx.pack_reads(["A" * 255, "A" * 255]), 30)
    _check_seqset_against_oracle(ss, ["A" * 255, "A" * 255])


def test_create_reproduces_golden_ecoli(ctx, golden_dir, tmp_path):
    """golden/e_coli_10000snp.fq -> golden/e_coli_10000snp.bg/seqset, whole create path on the GPU."""
    reads = O.read_fastq(golden_dir + "/e_coli_10000snp/reads.fq")
    ss, st = ctx.create(reads)
    assert st["n_kept_kmers"] == 7108 and st["n_corrected_reads"] == 8444 and st["n_corrected_bases"] == 288464
    assert st["n_entries"] == 19935
    d = O.load_seqset_zip(golden_dir + "/e_coli_10000snp/seqset")
    s, sh = ss.arrays()
    assert (s == d["sizes"]).all() and (sh == d["shared"]).all()
    assert ss.part("fixed") == d["raw"]["fixed"]
    for ch in "ACGT":
        for w in ("bits", "subaccum", "accum"):
            assert ss.part("prev_%s/%s" % (ch, w)) == d["raw"]["prev_%s/%s" % (ch, w)]
    assert ss.entries() == O.decode_entries(d)
    # container round trip + SDK answers of python/biograph/seqset_test.py
    path = str(tmp_path / "seqset")
    ss.write(path, ["biograph", "create", "--in", "reads.fq"])
    members = O.read_spiral_members(path)
    for name, blob in ss.parts().items():
        assert members[name] == blob
    assert members["seqset.json"] == b'{"num_entries":19935}'
    ss2 = ctx.open_seqset(path)
    assert ss2.parts() == ss.parts()
    g = ss2.find("GATTACA")
    assert repr(g) == "<SeqsetEntry 11076-11077: GATTACA>"
    assert ss2.find("GATTACAGATTACA") is None
    assert g.push_front("G") is None
    assert g.push_front("T") == ss2.find("TGATTACA")
    assert g.push_front_drop("G").sequence() == "GGATTAC"
    assert g.push_front_drop("G", 8) is None and g.push_front_drop("G", 7) is None
    assert g.push_front_drop("G", 6).sequence() == "GGATTAC"
    assert ss2.get_entry_by_id(6645).sequence() == "CCCCACCTGACGCGAAAGGATCGCAGT"


def test_open_lambda_golden_current_format(ctx, golden_dir):
    ss = ctx.open_seqset(golden_dir + "/father_lambda/seqset")
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    assert ss.size() == 98006 and ss.max_len == 150
    for name, blob in ss.parts().items():
        assert blob == d["raw"][name], name
    want = O.decode_entries(d)
    assert ss.entries(0, 2000) == want[:2000]
    assert ss.entries(97000, 1006) == want[97000:]


def test_walk_matches_oracle(ctx, golden_dir):
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    oss = O.Seqset.from_arrays(d["sizes"], d["shared"], d["bits"])
    ss = ctx.seqset_from_arrays(d["sizes"], d["shared"], d["bits"])
    n = d["n"]
    rng = np.random.default_rng(9)
    q = 200_000
    begin = rng.integers(0, n, q).astype(np.uint64)
    end = begin + 1
    size = d["sizes"][begin.astype(np.int64)].astype(np.uint32)
    for rounds in range(6):
        base = rng.integers(0, 4, len(begin)).astype(np.uint8)
        minctx = rng.integers(0, 120, len(begin)).astype(np.uint32)
        ob, oe, osz = oss.push_front_drop(begin, end, size, base, minctx)
        gb, ge, gs = ss.push_front_drop_batch(begin, end, size, base, minctx)
        assert (gb == ob).all() and (ge == oe).all() and (gs == osz).all()
        pb, pe, ps = oss.push_front(begin, end, size, base)
        hb, he, hs = ss.push_front_batch(begin, end, size, base)
        assert (hb == pb).all() and (he == pe).all() and (hs == ps).all()
        ok = oe > ob
        begin, end, size = ob[ok], oe[ok], osz[ok]
        assert len(begin) > 0
    ent = O.decode_entries(d)
    qs = [ent[int(i)][int(j):int(j) + 40] for i, j in zip(rng.integers(0, n, 500), rng.integers(0, 60, 500))]
    qs += ["GATTACA" * 3, "", "A"]
    fb, fe = ss.find_batch(qs)
    for s, b, e in zip(qs, fb, fe):
        wb, we, _ = oss.find(s)
        assert (int(b), int(e)) == (wb, we) or (b >= e and wb >= we)


def test_pop_front_matches_find_and_undoes_push_front(ctx, golden_dir):
    """bg_pop_front_batch (seqset_range::pop_front, seqset.cpp:609-628) on the golden lambda seqset: popping the first
    base of a prefix of an entry gives the range find() gives for the rest of the prefix; chains of pops walk an entry
    down to the empty context; pop after push_front gets back the range pushed from (when that push kept its context)."""
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    oss = O.Seqset.from_arrays(d["sizes"], d["shared"], d["bits"])
    ss = ctx.seqset_from_arrays(d["sizes"], d["shared"], d["bits"])
    n = d["n"]
    ent = O.decode_entries(d)
    rng = np.random.default_rng(17)
    ids = rng.integers(0, n, 3000)
    lens = np.array([rng.integers(1, len(ent[int(i)]) + 1) for i in ids])
    seqs = [ent[int(i)][:int(k)] for i, k in zip(ids, lens)]
    start = [oss.find(x) for x in seqs]
    b = np.array([x[0] for x in start], np.uint64)
    e = np.array([x[1] for x in start], np.uint64)
    sz = lens.astype(np.uint32)
    pb, pe, ps = ss.pop_front_batch(b, e, sz)
    for x, gb, ge, gs in zip(seqs, pb, pe, ps):
        wb, we, _ = oss.find(x[1:])
        assert (int(gb), int(ge), int(gs)) == (wb, we, len(x) - 1)
    # whole chains down to the empty context
    cb, ce, cs = b[:200].copy(), e[:200].copy(), sz[:200].copy()
    steps = 0
    while (cs > 0).any():
        live = cs > 0
        nb, ne, ns = ss.pop_front_batch(cb[live], ce[live], cs[live])
        cb[live], ce[live], cs[live] = nb, ne, ns
        steps += 1
    assert steps == int(sz[:200].max()) and (cb == 0).all() and (ce == n).all()
    # pop undoes a push that did not have to drop context
    base = rng.integers(0, 4, len(b)).astype(np.uint8)
    qb, qe, qs = ss.push_front_batch(b, e, sz, base)
    ok = qe > qb
    rb, re_, rs = ss.pop_front_batch(qb[ok], qe[ok], qs[ok])
    assert (rb == b[ok]).all() and (re_ == e[ok]).all() and (rs == sz[ok]).all() and ok.sum() > 500
    # invalid / empty inputs: (0, 0, 0)
    zb, ze, zs = ss.pop_front_batch(np.array([5, 0], np.uint64), np.array([5, n], np.uint64), np.array([3, 0], np.uint32))
    assert (zb == 0).all() and (ze == 0).all() and (zs == 0).all()


def test_synthetic_wgs_shape_full_path(ctx):
    """30x 150 bp paired reads with errors and N's through the whole create path, vs the oracle."""
    import torch
    from biograph_amd import synth
    seq, offs, n = synth.make_reads(genome_len=20000, coverage=30, read_len=150, seed=11, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    oss, oent, info = O.build_seqset_from_fastq_reads(reads)
    ss, st = ctx.create(reads)
    assert st["n_kept_kmers"] == info["kept_kmers"]
    assert st["n_corrected_reads"] == info["corrected_reads"] and st["n_corrected_bases"] == info["corrected_bases"]
    assert ss.size() == len(oent)
    want = oss.parts()
    for name, blob in ss.parts().items():
        assert blob == want[name], name
    assert info["modified"] > 0


def test_bgbinary_create_cli_matches_golden(golden_dir, tmp_path):
    """`bgbinary create` (C++ host over the C ABI) on the reference's golden FASTQ: the written spiral
    container holds byte-identical payload parts to golden/e_coli_10000snp.bg/seqset (old layout
    stores sizes/shared raw: compare decoded arrays) and `bgbinary refmap` runs on the result."""
    import json
    import subprocess
    import sys
    out = tmp_path / "ecoli.bg"
    r = subprocess.run([sys.executable, "-m", "biograph_amd.cli", "create", "--reads",
                        golden_dir + "/e_coli_10000snp/reads.fq", "--out", str(out), "--id", "test_accession_id"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    stats = json.load(open(out / "qc" / "create_stats.json"))
    assert stats["corrected_reads"] == 8444 and stats["corrected_bases"] == 288464 and stats["imported_reads"] == 10000
    mine = O.load_seqset_zip(str(out / "seqset"))
    gold = O.load_seqset_zip(golden_dir + "/e_coli_10000snp/seqset")
    assert mine["n"] == gold["n"] == 19935
    assert (mine["sizes"] == gold["sizes"]).all() and (mine["shared"] == gold["shared"]).all()
    for w in ("bits", "subaccum", "accum"):
        for b in range(4):
            assert (mine[w][b] == gold[w][b]).all()
    assert (mine["fixed"] == gold["fixed"]).all()
    # the sample's readmap: coverage/<sha1>.readmap, listed in bg_info.json, bound to the seqset's uuid,
    # and equal to the oracle's readmap of the same corrected reads (unpaired)
    import hashlib
    info = json.load(open(out / "metadata" / "bg_info.json"))
    sha = info["samples"]["test_accession_id"]
    rm_path = out / "coverage" / (sha + ".readmap")
    assert hashlib.sha1(rm_path.read_bytes()).hexdigest() == sha
    rmm = O.read_spiral_members(str(rm_path))
    ss_uuid = json.loads(O.read_spiral_members(str(out / "seqset"))["file_info.json"])["uuid"]
    assert json.loads(rmm["readmap.json"]) == {"seqset_uuid": ss_uuid}
    reads = [l.strip() for i, l in enumerate(open(golden_dir + "/e_coli_10000snp/reads.fq")) if i % 4 == 1]
    _, cseq, coffs, keep, _ = _oracle_stage(reads, 30, 5)
    kept = [s for s in O.unpack_strings(cseq, coffs) if s]
    ent = O.Entries.from_strings(O.decode_entries(gold))
    orm = O.Readmap(O.Seqset.from_entries(ent), ent, [(s,) for s in kept], False)
    assert json.loads(rmm["read_lengths/packed_varbit_vector.json"])["element_count"] == len(orm) == 2 * 8444
    for name, data in orm.parts().items():
        assert rmm[name] == data, name
    # ... and, read id for read id, to the readmap the reference itself wrote for this FASTQ (format 1.1.0)
    gold_rm = O.decode_readmap(golden_dir + "/e_coli_10000snp/readmap", gold)
    mine_rm = O.decode_readmap(str(rm_path), mine)
    for key in ("entry", "read_len", "is_forward", "mate_loop_ptr"):
        assert (mine_rm[key] == gold_rm[key]).all(), key
    # refusing to overwrite without --force (biograph_create.cpp:497-506)
    r2 = subprocess.run([sys.executable, "-m", "biograph_amd.cli", "create", "--reads",
                         golden_dir + "/e_coli_10000snp/reads.fq", "--out", str(out)], capture_output=True, text=True)
    assert r2.returncode == 1 and "overwrite" in r2.stderr


def test_kmer_count_overflow_fallback_path(ctx, monkeypatch):
    """Buckets whose distinct k-mers do not fit the LDS table are recounted by the sort-based path:
    force it with a 2-slot table and compare with the oracle (and with the default table)."""
    import torch  # noqa: F401
    from biograph_amd import synth
    seq, offs, n = synth.make_reads(genome_len=150000, coverage=20, read_len=100, seed=3, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    s, o = O.pack_strings(reads)
    ok, of, ofw, orv = O.KmerSet.count(s, o, 25, 4).arrays()
    r = ctx.pack_reads(reads)
    for lg in ("1", "3", "13"):
        monkeypatch.setenv("BG_KMER_LDS_LOG2CAP", lg)
        ks = ctx.kmer_count(r, 25, 4)
        gk, gf, gc = ks.download()
        assert len(gk) == len(ok) and (gk == ok).all() and (gf == of).all() and (gc == ofw + orv).all(), lg
        allk = O.KmerSet.count(s, o, 25, 1, keep_all=True)
        assert ks.distinct == len(allk)


def test_pop_front_of_every_entry_is_undone_by_push_front(ctx):
    """seqset_range::pop_front is a select over the prev bits (which entry does push_front(b) map onto `begin`?): here for EVERY
    entry of a built seqset -- every rank of every base goes through the sampled select table (sel256) and the rank blocks.
    pop_front(entry i) must be a range one base shorter, and push_front of i's first base on it must land on a range that holds
    i again (seqset.cpp:319-338 undoes :609-628; seqset.cpp:123-126: every entry is the push of exactly one (range, base))."""
    from biograph_amd import synth
    seq, offs, n = synth.make_reads(genome_len=150_000, coverage=30, read_len=150, seed=4242, device="cuda")
    ss, st = ctx.create_device(seq.data_ptr(), offs.data_ptr(), n, ctx.default_params())
    try:
        N = ss.size()
        assert N > 200_000
        sizes, _ = ss.arrays()
        fixed = np.frombuffer(ss.part("fixed"), np.uint64)
        assert len(fixed) == 5 and int(fixed[4]) == N
        idx = np.arange(N, dtype=np.uint64)
        base = (np.searchsorted(fixed[1:4], idx, side="right")).astype(np.uint8)  # the bucket an entry begins in = its first base
        pb, pe, ps = ss.pop_front_batch(idx, idx + np.uint64(1), sizes.astype(np.uint32))
        assert (ps == sizes - 1).all() and (pe > pb).all()
        qb, qe, qs = ss.push_front_batch(pb, pe, ps, base)
        assert (qb <= idx).all() and (qe > idx).all() and (qs == sizes).all()
        # and for shorter contexts (ranges of many entries): the first 40 bases of every 50th entry
        sel = idx[::50]
        sel = sel[sizes[::50] > 40]
        fb, fe = ss.find_batch([s[:40] for s in (ss.entries(int(i), 1)[0] for i in sel[:400])])
        pb2, pe2, ps2 = ss.pop_front_batch(fb, fe, np.full(len(fb), 40, np.uint32))
        qb2, qe2, qs2 = ss.push_front_batch(pb2, pe2, ps2, base[sel[:400].astype(np.int64)])
        assert (qb2 == fb).all() and (qe2 == fe).all() and (qs2 == 40).all()
    finally:
        ss.free()
