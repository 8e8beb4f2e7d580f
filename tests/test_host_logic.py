"""CPU checks of the kernels' per-thread logic (bg_device_funcs.h compiled with g++) against the
oracle.  These run without a GPU; the same functions run inside the HIP kernels."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import util


def _correct_via_harness(reads, k, min_count, maxc=8, mgr=2, trim=0.7, use_presence=1):
    hh = util.host_harness()
    seq, offs = O.pack_strings(reads)
    ks = O.KmerSet.count(seq, offs, k, min_count)
    kmers, flags, _, _ = ks.arrays()
    out_seq = np.zeros(max(int(offs[-1]), 1), np.uint8)
    out_offs = np.zeros(len(reads) + 1, np.uint64)
    keep = np.zeros(max(len(reads), 1), np.uint8)
    nc = np.zeros(1, np.uint64)
    rc = hh.hh_correct_reads(O._pad(kmers), len(kmers), k, O._pad(seq), offs, len(reads), maxc, mgr, trim, out_seq,
                             out_offs, keep, nc, use_presence)
    assert rc == 0, rc
    oseq, ooffs, okeep, stats = O.correct_reads(ks, seq, offs, maxc, mgr, trim)
    return (out_seq[:int(out_offs[-1])], out_offs, keep[:len(reads)], int(nc[0])), (oseq, ooffs, okeep, stats)


@pytest.mark.parametrize("seed,k,err,nrate", [(1, 15, 0.01, 0.002), (2, 21, 0.03, 0.01), (3, 30, 0.005, 0.0),
                                              (4, 12, 0.05, 0.02), (5, 30, 0.02, 0.005)])
def test_correction_dfs_matches_oracle(seed, k, err, nrate):
    reads = util.synth_reads(seed, genome_len=2000, n_reads=800, read_len=80, err=err, n_rate=nrate, var_len=True)
    for use_presence in (0, 1, 2):  # 2 = the routed search of the sharded build (RoutedTable, probes)
        (hs, ho, hk, hn), (os_, oo, ok, st) = _correct_via_harness(reads, k, 3, use_presence=use_presence)
        assert (hk == ok).all()
        assert (ho == oo).all()
        assert (hs == os_).all()
        assert hn == st["modified_bases"]
        assert st["modified"] > 0  # the case actually exercises substitutions


@pytest.mark.parametrize("maxc,mgr,trim", [(0, 2, 0.7), (1, 1, 0.5), (2, 3, 0.9), (8, 0, 0.0), (4, 2, 1.0)])
def test_correction_params(maxc, mgr, trim):
    reads = util.synth_reads(11, genome_len=1500, n_reads=500, read_len=70, err=0.02, n_rate=0.004)
    for use_presence in (1, 2):
        (hs, ho, hk, hn), (os_, oo, ok, st) = _correct_via_harness(reads, 17, 3, maxc, mgr, trim, use_presence)
        assert (hk == ok).all() and (ho == oo).all() and (hs == os_).all()


def test_correction_golden_ecoli(golden_dir):
    reads = O.read_fastq(golden_dir + "/e_coli_10000snp/reads.fq")
    for use_presence in (1, 2):
        (hs, ho, hk, hn), (os_, oo, ok, st) = _correct_via_harness(reads, 30, 5, use_presence=use_presence)
        assert st["corrected_reads"] == 8444 and st["corrected_bases"] == 288464
        assert (hk == ok).all() and (ho == oo).all() and (hs == os_).all()


def _entries_via_harness(reads, K):
    hh = util.host_harness()
    seq, offs = O.pack_strings(reads)
    h = hh.hh_build_entries(O._pad(seq), offs, len(reads), K)
    try:
        assert hh.hh_entries_err(h) == 0
        n = hh.hh_entries_count(h)
        s = np.zeros(max(hh.hh_entries_total(h), 1), np.uint8)
        o = np.zeros(n + 1, np.uint64)
        sh = np.zeros(max(n, 1), np.uint32)
        bits = [np.zeros(max((n + 63) // 64, 1), np.uint64) for _ in range(4)]
        hh.hh_entries_get(h, s, o, sh, *bits)
        return O.unpack_strings(s[:int(o[-1])], o), sh[:n], [b[:(n + 63) // 64] for b in bits], hh.hh_entries_records(h)
    finally:
        hh.hh_entries_free(h)


@pytest.mark.parametrize("seed,alphabet,K,rl,var", [(1, "ACGT", 12, 60, False), (2, "AC", 8, 70, True),
                                                     (3, "A", 5, 40, True), (4, "ACGT", 30, 150, False),
                                                     (5, "AT", 31, 100, True), (6, "ACGT", 4, 33, True)])
def test_cover_emission_entries_match_all_suffix_definition(seed, alphabet, K, rl, var):
    reads = util.synth_reads(seed, genome_len=1200, n_reads=300, read_len=rl, err=0.01, n_rate=0.0,
                             alphabet=alphabet, var_len=var, repeat=(7, 20))
    seq, offs = O.pack_strings(reads)
    ent = O.Entries.build(seq, offs, "simple")
    want = ent.strings()
    got, shared, bits, nrec = _entries_via_harness(reads, K)
    assert got == want
    ss = O.Seqset.from_entries(ent)
    _, osh = ss.arrays()
    assert (shared == osh).all()
    for b in range(4):
        assert (bits[b] == ss.prev(b, "bits")).all()
    assert nrec <= 2 * sum(len(r) for r in reads)


def test_builder_test_known_answers():
    # modules/build_seqset/builder_test.cpp:52-123
    t, trc = O.tseq, O.tseq_rc
    got, _, _, _ = _entries_via_harness([t("a")], 30)
    assert got == ["AAAATTAC", "AAATTAC", "AATTAC", "AATTTTAG", "AC", "AG", "ATTAC", "ATTTTAG", "CTAAAATTAC",
                   "GTAATTTTAG", "TAAAATTAC", "TAATTTTAG", "TAC", "TAG", "TTAC", "TTAG", "TTTAG", "TTTTAG"]
    for reads, n in [([t("abcdefg")], 129), ([t("abcd"), t("cdef"), trc("efgh")], 152),
                     ([t("ab"), t("bc"), t("cd"), t("be")], 91), ([t("AB"), t("BC"), t("CD"), t("BE")], 91),
                     ([t("abc"), t("cde")], 89), ([t("abc"), t("efg")], 99)]:
        for K in (5, 10, 30):
            got, _, _, _ = _entries_via_harness(reads, K)
            assert len(got) == n


@pytest.mark.parametrize("word_scan", [False, True])
def test_push_front_drop_matches_oracle_on_golden(golden_dir, word_scan):
    import ctypes as C
    hh = util.host_harness(word_scan)
    d = O.load_seqset_zip(golden_dir + "/e_coli_10000snp/seqset")
    ss = O.Seqset.from_arrays(d["sizes"], d["shared"], d["bits"])
    n = d["n"]
    rng = np.random.default_rng(5)
    # query ranges reached by real walks: start from single entries and random finds
    begin = rng.integers(0, n, 4000).astype(np.uint64)
    end = begin + 1
    size = d["sizes"][begin.astype(np.int64)].astype(np.uint32)
    base = rng.integers(0, 4, 4000).astype(np.uint8)
    minctx = rng.integers(0, 36, 4000).astype(np.uint32)
    shared_padded = np.zeros(n + 16, np.uint8)  # (the word-wise searches read the whole 8-byte word an entry lies in)
    shared_padded[:n] = d["shared"]
    for rounds in range(4):
        ob, oe, osz = ss.push_front_drop(begin, end, size, base, minctx)
        sz8 = d["sizes"].astype(np.uint8)
        sh8 = shared_padded[:n]
        arrs = {w: [ss.prev(b, w) for b in range(4)] for w in ("bits", "subaccum", "accum")}
        ptrs = {w: (C.c_void_p * 4)(*[a.ctypes.data for a in arrs[w]]) for w in arrs}
        hb = np.zeros(len(begin), np.uint64)
        he = np.zeros(len(begin), np.uint64)
        hs = np.zeros(len(begin), np.uint32)
        hh.hh_push_front_drop(n, sz8, sh8, ptrs["bits"], ptrs["subaccum"], ptrs["accum"], ss.fixed(), len(begin),
                              begin, end, size, base, minctx, hb, he, hs)
        assert (hb == ob).all() and (he == oe).all() and (hs == osz).all()
        ok = oe > ob
        assert ok.any()
        begin, end, size = ob[ok], oe[ok], osz[ok]
        base = rng.integers(0, 4, len(begin)).astype(np.uint8)
        minctx = rng.integers(0, 20, len(begin)).astype(np.uint32)
