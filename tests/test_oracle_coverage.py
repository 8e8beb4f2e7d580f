"""oracle/coverage.py (path_group, calc_coverage, simple_genotype_filter, the VCF record line) against the reference's
own vectors: modules/variants/calc_coverage_test.cpp:36-330 transcribed (reads unpaired unless noted, coverage is
interbase: len(seq) + 1 values), and every record of datasets/lambdaToyData/benchmark/father.vcf -- GT, GQ, OV, DP, AD,
PDP, PAD and the INFO column -- from the golden father_lambda biograph (PI, the assembly id, is a process-wide counter
there and is not compared).  CPU only."""
import numpy as np
import pytest

from oracle import align as A
from oracle import coverage as C
from oracle import oracle as O
from oracle import tracer as T

tseq, tseq_rc = O.tseq, O.tseq_rc
ALPHA = tseq("abcdefghijklmnopqrstuvwxyz")


def over(letters, n):  # assemble_testutil.h:346-348
    return [n] * (len(tseq(letters)) - 1)


def world(reads, pairs=()):
    """use_reads / use_paired_reads (assemble_testutil.h:159-190): seqset + readmap of the reads; no ref_map needed."""
    all_reads = [r for p in pairs for r in p] + list(reads)
    seq, offs = O.pack_strings(all_reads)
    ent = O.Entries.build(seq, offs, "simple")
    ss = O.Seqset.from_entries(ent)
    rm = O.Readmap(ss, ent, [tuple(p) for p in pairs] + [(r,) for r in reads], True)
    g = T.Graph(ss, rm, np.zeros(len(ss), np.uint8))
    return g, [int(x) for x in rm.get("read_len")], ss, rm


def asm(left, right, seq, **kw):
    a = T.Assembly()
    a.left_offset, a.right_offset, a.seq = left, right, seq
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def read_id_for_seq(g, read_len, ss, seq):  # assemble_testutil.h:161-173
    b, e, s = ss.find(seq)
    assert e == b + 1
    first, last = g.entry_to_index(int(b))
    for rid in range(first, last):
        if read_len[rid] == len(seq):
            return rid
    raise AssertionError("no read id for " + seq)


def run(reads, asms, parts=(), pairs=(), **opts):
    g, rl, ss, rm = world(reads, pairs)
    o = C.CoverageOptions(**dict(dict(penalize_directional_coverage=False), **opts))  # calc_coverage_test.cpp:19
    return C.calc_coverage(g, rl, T.Scaffold(list(parts)), o, asms)


def test_single():
    out = run([tseq("bcdefg")], [asm(0, 1, tseq("abcdefgh"))])
    assert [a.coverage for a in out] == [[0] + over("a", 0) + [0] + over("bcdefg", 1) + [0] + over("h", 0) + [0]]
    assert out[0].pair_coverage == [0] + over("abcdefgh", 0) + [0]


def test_dual_with_a_pair_match():
    reads = [tseq("bcdefg"), tseq("defgh")]
    g, rl, ss, rm = world(reads)
    a = asm(0, 1, tseq("abcdefghi"))
    a.right_pair_matches = [read_id_for_seq(g, rl, ss, tseq_rc("defgh"))]
    out = C.calc_coverage(g, rl, T.Scaffold([]), C.CoverageOptions(penalize_directional_coverage=False), [a])
    assert out[0].coverage == [0] + over("a", 0) + [0] + over("bc", 1) + [1] + over("defg", 2) + [1] + over("h", 1) + [0] + over("i", 0) + [0]
    assert out[0].pair_coverage == [0] + over("abc", 0) + [0] + over("defgh", 1) + [0] + over("i", 0) + [0]


def test_edges_and_exact_full():
    out = run([tseq("abc")], [asm(0, 1, tseq("abcdefg"))])
    assert out[0].coverage == [0] + over("abc", 1) + [0] + over("defg", 0) + [0]
    out = run([tseq("efg")], [asm(0, 1, tseq("abcdefg"))])
    assert out[0].coverage == [0] + over("abcd", 0) + [0] + over("efg", 1) + [0]
    reads = [tseq("abcdefg")]
    g, rl, ss, rm = world(reads)
    a = asm(0, 1, tseq("abcdefg"))
    a.left_pair_matches = [read_id_for_seq(g, rl, ss, tseq_rc("abcdefg"))]
    out = C.calc_coverage(g, rl, T.Scaffold([]), C.CoverageOptions(penalize_directional_coverage=False), [a])
    assert out[0].coverage == [0] + over("abcdefg", 1) + [0] and out[0].pair_coverage == [0] + over("abcdefg", 1) + [0]


def test_split_input():
    out = run([tseq("abcdef"), tseq("cdefg")], [asm(0, 1000, tseq("abcdefg"))])
    assert out[0].coverage == [0] + over("ab", 1) + [1] + over("cdef", 2) + [1] + over("g", 1) + [0]


def test_long_overlap():
    seq = tseq("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ")
    reads_seq = tseq("XXXX") + seq + tseq("XXXX")
    reads = [reads_seq[i:i + 41] for i in range(1, len(reads_seq) - 40, 10)]
    out = run(reads, [asm(40, len(reads_seq) - 40, seq)], parts=[(0, reads_seq)])
    assert out[0].coverage == [4] * (len(seq) + 1)


REF_T = tseq("abcd") + "T" + tseq("efghijklmnopqrstuvwxyz")


@pytest.mark.parametrize("reads,cov,ref_depth", [
    ([tseq("abcd") + "A" + tseq("efg")], [1, 1], None),
    ([tseq("abcd") + "A", tseq("cd") + "A" + tseq("efg")], [2, 1], None),
    ([tseq("abcd") + "A" + tseq("efg"), "A" + tseq("efgh")], [1, 2], 0),
    ([tseq("abcd") + "T" + tseq("efg")], [0, 0], 1),
])
def test_basic_with_ref(reads, cov, ref_depth):
    out = run(reads, [asm(40, 41, "A")], parts=[(0, REF_T)])
    assert len(out) == 1 and out[0].coverage == cov and out[0].other_depth == 0
    if ref_depth is not None:
        assert out[0].ref_depth == ref_depth


def test_insert_between_scaffolds():
    out = run([tseq("abcd"), tseq("cdef"), tseq("efgh"), tseq("ghij"), tseq("ijkl")], [asm(40, 1000, tseq("efgh"))],
              parts=[(0, tseq("abcd")), (1000, tseq("ijkl"))])
    assert out[0].coverage == [1] + over("ef", 2) + [1] + over("gh", 2) + [1]
    assert (out[0].other_depth, out[0].ref_depth) == (0, 0)


@pytest.mark.parametrize("second", [False, True])
@pytest.mark.parametrize("edge", range(38, 44))
def test_part_overlaps_extent_edge(edge, second):  # coverage_extent_edge_test :255-309
    nxt = tseq("ghijklmnopqrstuvwxyz")
    main = tseq("abcdef") + nxt[:edge]
    last_pos = len(main) - 40
    reads = [tseq("abcd"), tseq("cdef"), tseq("efgh"), main[last_pos:last_pos + 40]]
    asms = [asm(60, 61, "T")]
    if second:
        asms.append(asm(1020, 1021, "T"))
    out = run(reads, asms, parts=[(0, main), (1000, tseq("ijkl"))])
    want_ref = 2 if last_pos < 60 else 1
    assert (out[0].coverage, out[0].other_depth, out[0].ref_depth) == ([0, 0], 0, want_ref)
    if second:
        assert len(out) == 2 and (out[1].coverage, out[1].other_depth, out[1].ref_depth) == ([0, 0], 0, 0)
    else:
        assert len(out) == 1


# ---- the reference's own sample ------------------------------------------------------------------------------------
@pytest.mark.parametrize("sample,n_records", [("father", 8), ("mother", 8), ("proband", 17)])
def test_father_vcf_lines_from_the_golden_biograph(golden_dir, sample, n_records):
    """datasets/lambdaToyData/benchmark/{father,mother,proband}_lambda.bg -> every record line of {father,mother,proband}.vcf."""
    d = O.load_seqset_zip(golden_dir + "/%s_lambda/seqset" % sample)
    g = O.decode_readmap(golden_dir + "/%s_lambda/readmap" % sample, d)
    oss = O.Seqset.from_arrays(d["sizes"], d["shared"], d["bits"])
    orm = O.Readmap.from_arrays(d["n"], g["entry"], g["read_len"], g["mate_loop_ptr"], g["is_forward"], 150)
    ref = "".join(x.strip() for x in open(golden_dir + "/father_lambda/reference.fasta") if not x.startswith(">")).upper()
    codes = np.array([b"ACGT".index(bytes([c])) for c in ref.encode()], np.uint8)
    graph = T.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([len(ref)], np.uint64)))
    mo = int(0.7 * 150)
    o, ao, sc = T.Options(min_overlap=mo), A.AlignOptions(min_overlap=mo), T.Scaffold([(0, ref)])
    traced = []
    for rev in (False, True):
        traced += T.reversable_trace(graph, sc, o, rev_comp=rev, anchor_drop=lambda s_, emit: A.AnchorDropper(ao, s_, emit))
    _, sections = A.variants_pipeline(ao, sc, traced, through_vcf_pad=True)
    co = C.CoverageOptions()
    kept = C.simple_genotype_filter(co, C.calc_coverage(graph, [int(x) for x in g["read_len"]], sc, co, sections))
    got = [C.vcf_line(co, "lambda", lambda off, n: ref[off:off + n], a) for a in kept]
    want = [x for x in open(golden_dir + "/%s_lambda/%s.vcf" % (sample, sample)) if not x.startswith("#")]
    assert len(got) == len(want) == n_records

    def without_pi(line):
        f = line.rstrip("\n").split("\t")
        s = f[9].split(":")
        s[3] = "PI"
        return f[:9] + [":".join(s)]
    assert [without_pi(x) for x in got] == [without_pi(x) for x in want]


# ---- calc_coverage_test.cpp:323-355 (max_coverage_paths) and :370-520 (the stress recipe) --------------------------
def snp_with_other_coverage_case():
    ref = tseq("abcd") + "G" + tseq("efghijklmnopqrstuvwxyz")
    reads = [tseq("abcd") + "T" + tseq("efg"), tseq("abcd") + "A" + tseq("efg"), tseq("bcd") + "A" + tseq("efgh")]
    return ref, reads, [(40, 41, "T"), (40, 41, "A")]


def snp_with_other_coverage_expected(out):
    """The expectation of the test for max_coverage_paths >= 2 (other depth of the second "should be 1?" there)."""
    got = sorted((a.seq, a.coverage, a.other_depth, a.ref_depth) for a in out)
    return got == [("A", [2, 2], 0, 0), ("T", [1, 1], 2, 0)]


@pytest.mark.parametrize("paths", [3, 2, 1])
def test_max_coverage_paths(paths):
    ref, reads, vs = snp_with_other_coverage_case()
    out = run(reads, [asm(l, r, s) for l, r, s in vs], parts=[(0, ref)], max_coverage_paths=paths)
    assert snp_with_other_coverage_expected(out) == (paths >= 2)


def stress_case(n_vars, seed):
    """The construction of calc_coverage_stress_test: n_vars sites, each a SNP / insert / delete of one base against
    the reference, reads spanning all sites that carry a random subset of the variants; by construction every variant's
    coverage is its read count, its pair coverage the number of distinct pair-supported reads, the reference depth beside
    it the number of reads without it."""
    rng = np.random.default_rng(seed)
    alpha = "abcdefghijklmnopqrstuvwxyz"
    is_delete = [bool(rng.integers(0, 2)) for _ in range(n_vars)]
    is_insert = [(not d) and bool(rng.integers(0, 2)) for d in is_delete]
    var_count = [int(rng.integers(1, 6)) for _ in range(n_vars)]
    total = sum(var_count)
    ref, variants = tseq("X"), []
    for i in range(n_vars):
        ref += tseq(alpha[i])
        left = len(ref)
        if not is_insert[i]:
            ref += "T"
        variants.append((left, len(ref), "" if is_delete[i] else "A"))
    ref += tseq(alpha[n_vars]) + tseq("X")
    left_counts = list(var_count)
    pair_count, ref_count = [0] * n_vars, [0] * n_vars
    reads, paired = [], [set() for _ in range(n_vars)]
    while total:
        pair_matches = bool(rng.integers(0, 2))
        present = []
        for i in range(n_vars):
            if left_counts[i] and rng.integers(0, 2):
                present.append(True)
                left_counts[i] -= 1
                total -= 1
            else:
                present.append(False)
                ref_count[i] += 1
        seq = tseq("X")
        for i in range(n_vars):
            seq += tseq(alpha[i]) + (variants[i][2] if present[i] else ref[variants[i][0]:variants[i][1]])
        seq += tseq(alpha[n_vars]) + tseq("X")
        off = int(rng.integers(0, 11))
        seq = seq[off:off + len(seq) - 10]
        reads.append(seq)
        if pair_matches:
            for i in range(n_vars):
                if present[i] and seq not in paired[i]:
                    paired[i].add(seq)
                    pair_count[i] += 1
    return ref, reads, variants, paired, is_delete, var_count, pair_count, ref_count


@pytest.mark.parametrize("n_vars", [1, 2, 3, 5, 8])
def test_stress_recipe(n_vars):
    ref, reads, variants, paired, is_delete, var_count, pair_count, ref_count = stress_case(n_vars, 77 + n_vars)
    g, rl, ss, rm = world(reads)
    asms = []
    for i, (l, r, s) in enumerate(variants):
        a = asm(l, r, s)
        a.left_pair_matches = [read_id_for_seq(g, rl, ss, T.revcomp(x)) for x in sorted(paired[i])]
        asms.append(a)
    o = C.CoverageOptions(penalize_directional_coverage=False, max_coverage_paths=1 << min(n_vars, 60))
    out = C.calc_coverage(g, rl, T.Scaffold([(0, ref)]), o, asms)
    assert [(a.left_offset, a.right_offset) for a in out] == [(l, r) for l, r, _ in variants]
    for i, a in enumerate(out):
        n = 1 if is_delete[i] else 2
        assert (a.coverage, a.pair_coverage, a.ref_depth) == ([var_count[i]] * n, [pair_count[i]] * n, ref_count[i]), i
