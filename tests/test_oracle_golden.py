"""Pins the CPU oracle to the reference's own committed fixtures and known-answer tests
(SURVEY.md §8c).  No GPU."""
import json

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def ecoli(golden_dir):
    reads = O.read_fastq(golden_dir + "/e_coli_10000snp/reads.fq")
    ss, ent, info = O.build_seqset_from_fastq_reads(reads)
    d = O.load_seqset_zip(golden_dir + "/e_coli_10000snp/seqset")
    return reads, ss, ent, info, d


def test_create_path_reproduces_golden_ecoli(ecoli, golden_dir):
    reads, ss, ent, info, d = ecoli
    stats = json.load(open(golden_dir + "/e_coli_10000snp/create_stats.json"))
    assert info["kept_kmers"] == 7108  # qc/create_log.txt:68 "keeping 7108/9383 kmers"
    assert info["corrected_reads"] == stats["corrected_reads"] == 8444
    assert info["corrected_bases"] == stats["corrected_bases"] == 288464
    assert len(ent) == d["n"] == 19935
    assert ent.strings() == O.decode_entries(d)
    sizes, shared = ss.arrays()
    assert (sizes == d["sizes"]).all() and (shared == d["shared"]).all()
    assert (ss.fixed() == d["fixed"]).all()
    for b in range(4):
        for w in ("bits", "subaccum", "accum"):
            assert (ss.prev(b, w) == d[w][b]).all(), (b, w)


def test_cover_mode_equals_all_suffix_mode(ecoli):
    reads, ss, ent, info, d = ecoli
    ss2, ent2, info2 = O.build_seqset_from_fastq_reads(reads, mode="cover")
    assert ent2.strings() == ent.strings()
    assert info2["n_records"] < info["n_records"]


def test_seqset_py_walk_answers(ecoli):
    # python/biograph/seqset_test.py:17-104
    _, ss, ent, _, _ = ecoli
    assert len(ss) == 19935
    assert ss.find("GATTACA") == (11076, 11077, 7)
    b, e, _ = ss.find("GATTACAGATTACA")
    assert b >= e
    es = ent.strings()
    assert es[0] == "AAAAAAAAGCCCGCACTGTCAGGTGCGGGC"
    assert es[6645] == "CCCCACCTGACGCGAAAGGATCGCAGT"
    assert es[19934] == "TTTTTTTTCGACCAAAGGTAACGAGGTAACAACCA"
    g = ss.find("GATTACA")
    one = lambda x: np.array([x])
    # push_front("G") invalid; push_front("T") == find("TGATTACA")
    ob, oe, osz = ss.push_front(one(g[0]).astype(np.uint64), one(g[1]).astype(np.uint64), one(7).astype(np.uint32),
                                one(2).astype(np.uint8))
    assert ob[0] >= oe[0]
    ob, oe, osz = ss.push_front(one(g[0]).astype(np.uint64), one(g[1]).astype(np.uint64), one(7).astype(np.uint32),
                                one(3).astype(np.uint8))
    assert (int(ob[0]), int(oe[0]), int(osz[0])) == ss.find("TGATTACA")
    # push_front_drop("G") -> GGATTAC; min_ctx 8,7 -> invalid; 6 -> GGATTAC
    want = ss.find("GGATTAC")
    for mc, ok in [(0, True), (8, False), (7, False), (6, True)]:
        ob, oe, osz = ss.push_front_drop(one(g[0]).astype(np.uint64), one(g[1]).astype(np.uint64),
                                         one(7).astype(np.uint32), one(2).astype(np.uint8), one(mc).astype(np.uint32))
        if ok:
            assert (int(ob[0]), int(oe[0]), int(osz[0])) == want
        else:
            assert (int(ob[0]), int(oe[0]), int(osz[0])) == (0, 0, 0)


def test_lambda_golden_row_e_closed_loop_and_container_parts(golden_dir):
    """father_lambda.bg (current container format): re-deriving the arrays from the decoded entry
    strings reproduces every payload part byte for byte."""
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    assert d["n"] == 98006 and d["entry_sizes_meta"]["max_value"] == 150 and d["shared_meta"]["max_value"] == 149
    entries = O.decode_entries(d)
    assert entries == sorted(entries)
    ss = O.Seqset.from_entries(O.Entries.from_strings(entries))
    parts = ss.parts()
    for name, blob in parts.items():
        assert blob == d["raw"][name], name
    # the set is closed: all suffixes of entries are represented (row E), spot-check by find()
    rng = np.random.default_rng(1)
    for i in rng.integers(0, len(entries), 200):
        e = entries[int(i)]
        j = int(rng.integers(0, len(e)))
        b, en, _ = ss.find(e[j:])
        assert en > b


def test_builder_test_known_answers():
    # modules/build_seqset/builder_test.cpp:52-123
    t, trc = O.tseq, O.tseq_rc

    def entries(reads, mode):
        seq, offs = O.pack_strings(reads)
        return O.Entries.build(seq, offs, mode, 10).strings()

    for mode in ("simple", "cover"):
        assert entries([t("a")], mode) == [
            "AAAATTAC", "AAATTAC", "AATTAC", "AATTTTAG", "AC", "AG", "ATTAC", "ATTTTAG", "CTAAAATTAC", "GTAATTTTAG",
            "TAAAATTAC", "TAATTTTAG", "TAC", "TAG", "TTAC", "TTAG", "TTTAG", "TTTTAG"]
        assert len(entries([t("abcdefg")], mode)) == 129
        assert len(entries([t("abcd"), t("cdef"), trc("efgh")], mode)) == 152
        assert len(entries([t("ab"), t("bc"), t("cd"), t("be")], mode)) == 91
        assert len(entries([t("AB"), t("BC"), t("CD"), t("BE")], mode)) == 91
        assert len(entries([t("abc"), t("cde")], mode)) == 89
        assert len(entries([t("abc"), t("efg")], mode)) == 99


def test_kmer_counter_semantics():
    # kmer_counter.h:297-326: N resets the window and clears the first-kmer flag; flags swap when flipped
    reads = ["ACGTACGTAC", "NACGTACGTA", "ACGTNACGTACG"]
    seq, offs = O.pack_strings(reads)
    ks = O.KmerSet.count(seq, offs, k=5, min_count=1)
    kmers, flags, fwd, rev = ks.arrays()
    tot = int(fwd.sum() + rev.sum())
    assert tot == 6 + 5 + 3
    code = {c: i for i, c in enumerate("ACGT")}
    def enc(s):
        v = 0
        for c in s:
            v = v * 4 + code[c]
        return v
    idx = {int(k): i for i, k in enumerate(kmers)}
    # "ACGTA" is its own... rc(ACGTA)=TACGT > ACGTA so canonical is ACGTA, first kmer of read 0 -> fwd flag
    assert flags[idx[enc("ACGTA")]] & 1
    # last kmer of read 0 is "CGTAC": rc = GTACG > CGTAC, not flipped -> rev_starts flag
    assert flags[idx[enc("CGTAC")]] & 2


def test_readmap_oracle_against_lambda_golden(golden_dir):
    """Readmap (make_readmap.cpp:120-366) — groundwork for the next row.  The only committed
    readmap in the current container format (father_lambda, written by v4.0.1-dev) predates the total
    row order of make_readmap.h:187-213 (rows inside one seqset entry follow input-file order there), so
    it pins: the sparse_multi rank vectors byte for byte, the per-entry multiset of (read length,
    is_forward), and the mate-loop cycle structure — not the row order inside an entry group."""
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    g = O.decode_readmap(golden_dir + "/father_lambda/readmap", d)
    assert g["n_reads"] == 97860 and len(g["records"]) == 28939 and sum(len(r) for r in g["records"]) == 48930
    ent = O.Entries.from_strings(O.decode_entries(d))
    ss = O.Seqset.from_entries(ent)
    rm = O.Readmap(ss, ent, g["records"], True)
    assert len(rm) == 97860 and rm.max_read_len == 150
    parts = rm.parts()
    for sel in ("source_to_mid", "dest_to_mid"):
        for w in ("bits", "subaccum", "accum"):
            name = "read_ids/%s/%s" % (sel, w)
            assert parts[name] == g["raw"][name], name
    for name in ("read_lengths/elements", "mate_loop_ptr/elements", "is_forward/packed_data"):
        assert len(parts[name]) == len(g["raw"][name])
    e, rl, isf, mlp = rm.get("entry_id"), rm.get("read_len"), rm.get("is_forward"), rm.get("mate_loop_ptr")
    assert (e == g["entry"]).all()
    mine = sorted(zip(e.tolist(), rl.tolist(), isf.tolist()))
    gold = sorted(zip(g["entry"].tolist(), g["read_len"].tolist(), g["is_forward"].tolist()))
    assert mine == gold

    def cycles(entry, rlen, fwd, ptr):
        seen = np.zeros(len(ptr), bool)
        out = []
        for i in range(len(ptr)):
            if seen[i]:
                continue
            c, j = [], i
            while not seen[j]:
                seen[j] = True
                c.append((int(entry[j]), int(rlen[j]), int(fwd[j])))
                j = int(ptr[j])
            assert j == i  # closed loop
            k = c.index(min(c))
            out.append(tuple(c[k:] + c[:k]))
        return sorted(out)
    cm = cycles(e, rl, isf, mlp)
    cg = cycles(g["entry"], g["read_len"], g["is_forward"], g["mate_loop_ptr"])
    assert len(cm) == len(cg) == 28939
    # same loops up to rotation: a loop is fwd -> rc -> mate fwd -> mate rc (or fwd -> rc)
    assert sorted(tuple(sorted(c)) for c in cm) == sorted(tuple(sorted(c)) for c in cg)


@pytest.fixture(scope="module")
def lambda_world(golden_dir):
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    g = O.decode_readmap(golden_dir + "/father_lambda/readmap", d)
    ent_strs = O.decode_entries(d)
    ent = O.Entries.from_strings(ent_strs)
    ss = O.Seqset.from_entries(ent)
    rm = O.Readmap.from_arrays(d["n"], g["entry"], g["read_len"], g["mate_loop_ptr"], g["is_forward"], 150)
    return d, g, ent_strs, ss, rm


def test_readmap_queries_on_lambda_golden(lambda_world):
    """Oracle restatement of readmap::get_prefix_reads / has_mate / get_mate_rc / get_rev_comp /
    read::get_seqset_entry (bio_base/readmap.cpp:207-290,434-447,538-649) on the reference-written
    lambda readmap, against brute force over the decoded sequences."""
    d, g, ent, ss, rm = lambda_world
    n_ids = g["n_reads"]
    rl, entry = g["read_len"], g["entry"]
    seqs = [ent[int(entry[i])][:int(rl[i])] for i in range(n_ids)]
    comp = str.maketrans("ACGT", "TGCA")
    rng = np.random.default_rng(3)
    ids = rng.choice(n_ids, 300, replace=False)
    by_seq = {}
    for i, s in enumerate(seqs):
        by_seq.setdefault(s, []).append(i)
    for i in map(int, ids):
        has_mate, mate_rc, rev, r = rm.read_info(ss, i)
        assert seqs[rev] == seqs[i].translate(comp)[::-1]
        assert rm.read_info(ss, rev)[2] == i
        assert r == ss.find(seqs[i])                     # ctx_entry(id).truncate(len) == find(read)
        assert bool(g["is_forward"][rev]) != bool(g["is_forward"][i])
        if has_mate:
            # mate_rc of mate_rc is the read itself; the mate's reverse complement is a different read id
            assert rm.read_info(ss, mate_rc)[1] == i
        # get_prefix_reads(range of this read) = every read whose sequence is a prefix of it
        got = rm.prefix_reads(ss, r)
        want = sorted(j for L in range(1, len(seqs[i]) + 1) for j in by_seq.get(seqs[i][:L], []))
        assert sorted(got) == want and i in got
        assert len(set(got)) == len(got)
    # ranges shorter than every read give nothing (readmap.cpp:439-442)
    assert rm.prefix_reads(ss, ss.find("ACGT")) == []
    assert int(rl.min()) >= 30


def test_propagate_from_end_on_lambda_golden(lambda_world, golden_dir):
    """seqset_path::propagate_from_end (variants/assemble.cpp:1345-1418) over slices of the lambda
    reference, against a plain Python walk with the oracle's push_front_drop and brute-force mates."""
    d, g, ent, ss, rm = lambda_world
    ref = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">"))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    n = d["n"]
    seqs_len = g["read_len"]

    def dedup(rs):
        rs = set(rs)
        return sorted(r for r in rs if not any(o != r and o[0] >= r[0] and o[1] <= r[1] and
                                               (o[0] > r[0] or o[1] < r[1] or o[2] > r[2]) for o in rs))
    for a, ln in [(1000, 400), (20000, 333), (48000, 200)]:
        seq = np.array([3 - code[c] for c in reversed(ref[a:a + ln])], np.uint8)   # rev_comp of the slice
        marks = [7, ln // 2]
        got, mates = ss.propagate_from_end(seq, [(0, n, 0)], rm, 20, marks)
        cur = [(0, n, 0)]
        want, want_mates = {ln: cur}, set()
        for off in range(ln - 1, -1, -1):
            b, e, s = (np.array(x) for x in zip(*cur))
            ob, oe, osz = ss.push_front_drop(b, e, s, np.full(len(cur), seq[off], np.uint8), np.zeros(len(cur), np.uint32))
            cur = dedup([(int(x), int(y), int(z)) for x, y, z in zip(ob, oe, osz)])
            if off == 0 or off in marks:
                want[off] = cur
            for r in cur:
                ids = [i for i in rm.prefix_reads(ss, r) if rm.read_info(ss, i)[0]]
                if len(ids) <= 20:
                    want_mates.update(rm.read_info(ss, rm.read_info(ss, i)[1])[3] for i in ids)
        assert got == want
        assert set(mates) == want_mates and len(mates) == len(want_mates)
        assert len(mates) > 0
        # the walk ends on the reference's own bases
        assert all(sz >= 100 for _, _, sz in got[0])


def test_readmap_oracle_reproduces_ecoli_golden_exactly(golden_dir):
    """golden/e_coli_10000snp.bg/coverage/*.readmap (readmap 1.1.0, unpaired reads): the oracle's readmap of
    the corrected reads reproduces the reference's read ids one for one — entry of every read id, lengths,
    orientation and the mate loop pointers (fwd <-> rc 2-cycles claimed in the reference's sequential
    order).  This pins the row order and the claim ranks for unpaired input; pairs are pinned by the
    lambda golden only up to the order inside an entry group (test above)."""
    d = O.load_seqset_zip(golden_dir + "/e_coli_10000snp/seqset")
    g = O.decode_readmap(golden_dir + "/e_coli_10000snp/readmap", d)
    assert g["n_reads"] == 2 * 8444
    reads = [l.strip() for i, l in enumerate(open(golden_dir + "/e_coli_10000snp/reads.fq")) if i % 4 == 1]
    seq, offs = O.pack_strings(reads)
    ks = O.KmerSet.count(seq, offs, 30, 5)
    cseq, coffs, keep, stats = O.correct_reads(ks, seq, offs, 8, 2, 0.7)
    kept = [s for s in O.unpack_strings(cseq, coffs) if s]
    ent = O.Entries.from_strings(O.decode_entries(d))
    rm = O.Readmap(O.Seqset.from_entries(ent), ent, [(s,) for s in kept], False)
    assert len(rm) == g["n_reads"]
    assert (rm.get("entry_id") == g["entry"]).all()
    assert (rm.get("read_len") == g["read_len"]).all()
    assert (rm.get("is_forward") == g["is_forward"]).all()
    assert (rm.get("mate_loop_ptr") == g["mate_loop_ptr"]).all()
    for sel in ("source_to_mid", "dest_to_mid"):
        for w in ("bits", "subaccum", "accum"):
            name = "read_ids/%s/%s" % (sel, w)
            assert rm.parts()[name] == g["raw"][name], name


def test_merge_and_readmap_migration_reproduce_merged_golden(golden_dir):
    """SURVEY §8f.3, pinned on golden/e_coli_merged.bg (`biograph merge` of e_coli_10000snp.bg and a second
    sample whose .bg is not in the tree).  The second sample's corrected reads are recoverable from ITS
    readmap inside the merged biograph (a read = entry prefix), so: row E on those 100-bp reads gives its
    seqset; the oracle's make_mergemap restatement merges it with the e_coli seqset; the result must be
    the golden merged seqset entry for entry; and fast_migrate of the e_coli readmap through the mergemap
    must be the golden's migrated readmap read id for read id."""
    d1 = O.load_seqset_zip(golden_dir + "/e_coli_10000snp/seqset")
    dm = O.load_seqset_zip(golden_dir + "/e_coli_merged/seqset")
    e1 = O.Entries.from_strings(O.decode_entries(d1))
    em = O.decode_entries(dm)
    assert dm["n"] == 232745
    r2 = O.decode_readmap(golden_dir + "/e_coli_merged/readmap_e_coli_test", dm)
    reads2 = sorted(set(r2["read_seqs"]))
    assert r2["n_reads"] == 76094 and len(reads2) == 63808 and {len(s) for s in reads2} == {100}
    seq, offs = O.pack_strings(reads2)
    e2 = O.Entries.build(seq, offs, "simple")
    merged, maps = O.merge_entries([e1, e2])
    assert merged.strings() == em
    assert len(maps[0]) == d1["n"] and len(maps[1]) == len(e2)
    for m in maps:
        assert (np.diff(m.astype(np.int64)) > 0).all()          # injective and order preserving
    # the merged seqset's arrays, rebuilt from the merged entries, are the golden's
    oss = O.Seqset.from_entries(merged)
    s, sh = oss.arrays()
    assert (s == dm["sizes"]).all() and (sh == dm["shared"]).all()
    # prev bits: equal to the golden except where the reference's chunked sweep had a choice — an entry
    # b+t whose tail t is a proper prefix of SEVERAL merged entries gets its bit on the first of them by
    # the builder's rule (builder.cpp:85-108, used here) but on the first entry of a parallel_for chunk
    # when a chunk boundary falls inside that block (seqset_merger.cpp:89-106,111-131): 5 such bits in
    # this file, all verified to be exactly that situation.
    import bisect
    ndiff = 0
    for b in range(4):
        mine = np.unpackbits(oss.prev(b, "bits").view(np.uint8), bitorder="little")[:dm["n"]]
        gold = np.unpackbits(np.asarray(dm["bits"][b]).view(np.uint8), bitorder="little")[:dm["n"]]
        assert mine.sum() == gold.sum()
        only_m, only_g = np.flatnonzero(mine & ~gold), np.flatnonzero(gold & ~mine)
        assert len(only_m) == len(only_g)
        for i, j in zip(only_m, only_g):
            assert i < j
            lcp = min(int(x) for x in sh[i + 1:j + 1])
            cands = ["ACGT"[b] + em[i][:L] for L in range(1, lcp + 1)]
            assert any(em[bisect.bisect_left(em, c)] == c for c in cands)
        ndiff += len(only_m)
    assert ndiff == 5
    # readmap migration of sample 1
    g1 = O.decode_readmap(golden_dir + "/e_coli_10000snp/readmap", d1)
    gm = O.decode_readmap(golden_dir + "/e_coli_merged/readmap_test_accession_id", dm)
    rm1 = O.Readmap.from_arrays(d1["n"], g1["entry"], g1["read_len"], g1["mate_loop_ptr"], g1["is_forward"], 35)
    mig = rm1.migrate(maps[0], dm["n"])
    assert (mig.get("entry_id") == gm["entry"]).all()
    assert (mig.get("read_len") == gm["read_len"]).all()
    assert (mig.get("mate_loop_ptr") == gm["mate_loop_ptr"]).all()
    assert (mig.get("is_forward") == gm["is_forward"]).all()
    for sel in ("source_to_mid", "dest_to_mid"):
        for w in ("bits", "subaccum", "accum"):
            name = "read_ids/%s/%s" % (sel, w)
            assert mig.bitcount(sel, w).tobytes() == gm["raw"][name], name
    # sample 2's own readmap: every read id sits on the merged entry its sequence maps to
    ent2 = e2.strings()
    import bisect
    for i in range(0, r2["n_reads"], 97):
        s_ = r2["read_seqs"][i]
        j = bisect.bisect_left(ent2, s_)
        assert ent2[j].startswith(s_) and int(maps[1][j]) == int(r2["entry"][i])


FAMILY_READMAPS = {"father": "7cb90add2a864f9fa337ee759941743f48433727", "mother": "c6a04bde3351da4a69967082c4a74c2ff0f018ee",
                   "proband": "ec86f985446e2c28e9a58d36573dec46fe2f6402"}  # metadata/bg_info.json of family_lambda.bg


def family_prev_bit_differences(em, sh, mine_bits, gold_bits, n):
    """Prev bits of a merged seqset: the builder's closed form against the file `biograph merge` wrote.  They may differ only
    where the merger's chunked sweep had a choice (seqset_merger.cpp:89-106,111-131: an entry b+t whose tail t is a proper
    prefix of several merged entries): the same number of bits per base, each moved forward inside a block of entries that
    share the tail, and b + a prefix of the block's first entry is an entry.  Returns the number of such bits."""
    import bisect
    ndiff = 0
    for b in range(4):
        mine = np.unpackbits(np.asarray(mine_bits[b]).view(np.uint8), bitorder="little")[:n]
        gold = np.unpackbits(np.asarray(gold_bits[b]).view(np.uint8), bitorder="little")[:n]
        assert mine.sum() == gold.sum()
        only_m, only_g = np.flatnonzero(mine & ~gold), np.flatnonzero(gold & ~mine)
        assert len(only_m) == len(only_g)
        for i, j in zip(only_m, only_g):
            assert i < j
            lcp = min(int(x) for x in sh[i + 1:j + 1])
            cands = ["ACGT"[b] + em[i][:L] for L in range(1, lcp + 1)]
            assert any(em[bisect.bisect_left(em, c)] == c for c in cands)
        ndiff += len(only_m)
    return ndiff


def test_three_sample_merge_reproduces_family_lambda(golden_dir):
    """SURVEY §8c / configs[4]: datasets/lambdaToyData/benchmark/family_lambda.bg = `biograph merge` of father, mother and
    proband (current container format, three readmaps).  The oracle's merge of the three golden seqsets must be the golden
    merged seqset entry for entry (sizes, shared, fixed; prev bits up to the merger's chunk-boundary choice), and fast_migrate
    of each sample's readmap through its mergemap must be that sample's readmap inside the merged biograph."""
    ds = {s: O.load_seqset_zip(golden_dir + "/%s_lambda/seqset" % s) for s in FAMILY_READMAPS}
    dm = O.load_seqset_zip(golden_dir + "/family_lambda/seqset")
    em = O.decode_entries(dm)
    names = ["father", "mother", "proband"]
    ents = [O.Entries.from_strings(O.decode_entries(ds[s])) for s in names]
    merged, maps = O.merge_entries(ents)
    assert merged.strings() == em and len(em) == dm["n"]
    oss = O.Seqset.from_entries(merged)
    s, sh = oss.arrays()
    assert (s == dm["sizes"]).all() and (sh == dm["shared"]).all()
    assert list(oss.fixed()) == [int(x) for x in dm["fixed"]]
    ndiff = family_prev_bit_differences(em, sh, [oss.prev(b, "bits") for b in range(4)], dm["bits"], dm["n"])
    assert ndiff == 31  # of 103 996 x 4 bits: every one checked above to be the merger's chunk-boundary choice
    for k, sname in enumerate(names):
        g1 = O.decode_readmap(golden_dir + "/%s_lambda/readmap" % sname, ds[sname])
        gm = O.decode_readmap(golden_dir + "/family_lambda/coverage/%s.readmap" % FAMILY_READMAPS[sname], dm)
        rm1 = O.Readmap.from_arrays(ds[sname]["n"], g1["entry"], g1["read_len"], g1["mate_loop_ptr"], g1["is_forward"], 150)
        mig = rm1.migrate(maps[k], dm["n"])
        assert (mig.get("entry_id") == gm["entry"]).all(), sname
        assert (mig.get("read_len") == gm["read_len"]).all()
        assert (mig.get("mate_loop_ptr") == gm["mate_loop_ptr"]).all()
        assert (mig.get("is_forward") == gm["is_forward"]).all()
        for sel in ("source_to_mid", "dest_to_mid"):
            for w in ("bits", "subaccum", "accum"):
                name = "read_ids/%s/%s" % (sel, w)
                assert mig.bitcount(sel, w).tobytes() == gm["raw"][name], (sname, name)
