"""The host drivers of tests/variants_twin.py against the reference's own known answers:
graph_discover/branch_test.cpp:63-98, push_to_pair_test.cpp:63-150 (tseq fixtures) and
python/biograph/variants/graph_discover_test.py:41-56 (8 anchored variants on father_lambda).
The walks are answered by the oracle here (tests/discover_backend.py); tests/test_gpu_discover_drivers.py
repeats them with the GPU kernels behind the same drivers."""
import pytest

from tests import variants_twin as V
from oracle import oracle as O
from tests.discover_backend import OracleWalker, world_for_reads

t = O.tseq
REF = t("abcdefghijklmnopqrstuvwxyz")


def ref_asm(left, seq):
    return V.Assembly(left, left + len(seq), seq, True, ["test"])


def var_asm(left, seq, right):
    return V.Assembly(left, right, seq, False, ["test"])


def run_branch(walker, asms, min_overlap):
    asms = V.update_rc_seqset_entries(sorted(asms, key=V.canon_key), walker, with_mates=True)
    asms = V.discover_branch(asms, walker, min_overlap, "branch_discover")
    asms = V.update_rc_seqset_entries(asms, walker, with_mates=True)
    return [(a.left_offset, a.seq, a.right_offset) for a in asms if "branch_discover" in a.tags]


def run_push(walker, asms, min_overlap, max_pair_distance=1000):
    asms = V.update_rc_seqset_entries(sorted(asms, key=V.canon_key), walker, with_mates=True)
    asms = V.discover_push_to_pair(asms, walker, min_overlap, max_pair_distance, "push_to_pair")
    asms = V.update_rc_seqset_entries(asms, walker, with_mates=True)
    return [(a.left_offset, a.seq, a.right_offset) for a in asms if "push_to_pair" in a.tags]


def make_walker(unpaired, paired=()):
    oss, orm, _ = world_for_reads(unpaired, paired)
    return OracleWalker(oss, orm)


BRANCH_READS = [t("abcdefg"), t("defghij"), t("efghijklm")]


def branch_cases(make):
    w = make(BRANCH_READS)
    assert run_branch(w, [ref_asm(0, t("abcdefghijklm"))], 100) == []                       # branch_test.cpp:63-73
    w = make(BRANCH_READS + [t("ef") + "G" + t("gh")])
    got = run_branch(w, [ref_asm(len(t("a")), t("bcdefghijklm"))], len(t("ef")))            # :75-87
    assert got == [(len(t("a")), t("bcdef") + "G", None)]
    assert run_branch(w, [ref_asm(len(t("a")), t("bcdefghijklm"))], len(t("ef")) + 1) == []  # :89-98


def push_cases(make):
    w = make(BRANCH_READS)
    assert run_push(w, [ref_asm(0, t("abcdefghijklm"))], 100) == []                         # push_to_pair_test.cpp:63-72
    # ref_rejoin :74-95
    w = make(BRANCH_READS + [t("ef") + "G" + t("gh"), t("ghijklmno")])
    got = run_push(w, [var_asm(len(t("abcd")), t("ef") + "G", None), ref_asm(len(t("abcde")), t("fghi"))], len(t("ef")))
    anchor = (len(t("abcd")), t("ef") + "G" + t("ghi"), len(t("abcdefghi")))
    cont = (len(t("abcd")), t("ef") + "G" + t("ghi"), None)
    assert anchor in got and cont in got and all(g in (anchor, cont) for g in got)
    # pair_extend_prev :97-114
    w = make([t("cdef") + "G" + t("gh")], [(t("abc"), V.rev_comp("G" + t("ghi")))])
    got = run_push(w, [ref_asm(0, t("abcd")), var_asm(len(t("abcd")), t("ef") + "G", None)], len(t("ef")))
    assert got == [(len(t("abcd")), t("ef") + "G" + t("ghi"), None)]
    # pair_extend_other_assembly :116-133
    w = make([t("cdef") + "G" + t("gh")], [(t("abcde"), V.rev_comp("G" + t("ghi")))])
    got = run_push(w, [ref_asm(0, t("abcd")), var_asm(len(t("abcd")), t("ef") + "G", None)], len(t("ef")))
    assert got == [(len(t("abcd")), t("ef") + "G" + t("ghi"), None)]
    # pair_extend_within_max_pair_distance :135-153
    w = make([t("pqrst") + "G" + t("uv")], [(t("abc"), V.rev_comp("G" + t("uvw")))])
    got = run_push(w, [ref_asm(0, t("abcd")), ref_asm(len(t("abcd")), t("efghijklmnopqr")),
                       var_asm(len(t("abcdefghijklmnopqr")), t("st") + "G", None)],
                   len(t("ef")), len(t("abcdefghijklmnopqrst")))
    assert got == [(len(t("abcdefghijklmnopqr")), t("st") + "G" + t("uvw"), None)]


def test_branch_discover_known_answers():
    branch_cases(make_walker)


def test_push_to_pair_known_answers():
    push_cases(make_walker)


def lambda_pipeline(walker, ref):
    """graph_discover_test.py:41-56."""
    ext = [(0, ref)]

    def trim_and_update(asms):
        asms = V.graph_trim_ref(asms, ext)
        asms = V.dedup_assemblies(asms)
        return V.update_rc_seqset_entries(asms, walker)     # the Python binding passes no readmap here

    asms = V.make_ref_assemblies(ext, 0, len(ref))
    asms = trim_and_update(asms)
    asms = V.discover_branch(asms, walker, 70, "BRANCH")
    asms = trim_and_update(asms)
    asms = V.discover_push_to_pair(asms, walker, 70, 1000, "PUSH_TO_PAIR", ["BRANCH"])
    asms = trim_and_update(asms)
    return [a for a in asms if a.left_offset is not None and a.right_offset is not None and not a.matches_reference]


def test_graph_discover_on_lambda_golden(golden_dir):
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    g = O.decode_readmap(golden_dir + "/father_lambda/readmap", d)
    oss = O.Seqset.from_arrays(d["sizes"], d["shared"], d["bits"])
    orm = O.Readmap.from_arrays(d["n"], g["entry"], g["read_len"], g["mate_loop_ptr"], g["is_forward"], 150)
    ref = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">"))
    found = lambda_pipeline(OracleWalker(oss, orm), ref.upper())
    assert len(found) == 8, found
    check_against_father_vcf(found, ref.upper(), golden_dir)


def check_against_father_vcf(found, ref, golden_dir):
    """datasets/lambdaToyData/benchmark/father.vcf (8 records, written by the retired `variants` command —
    indicative): every discovered anchored assembly is one of its records (compared as the
    haplotype each produces, which is alignment-normalisation free)."""
    def hap(left, right, alt):
        lo, hi = max(left - 400, 0), min(right + 400, len(ref))
        return lo, ref[lo:left] + alt + ref[right:hi]
    mine = sorted(hap(a.left_offset, a.right_offset, a.seq) for a in found)
    vcf = []
    for line in open(golden_dir + "/father_lambda/father.vcf"):
        if line.startswith("#"):
            continue
        f = line.split("\t")
        pos, r, alt = int(f[1]) - 1, f[3], f[4]
        assert ref[pos:pos + len(r)] == r
        # trim the shared leading bases so that the window matches the assembly's
        k = 0
        while k < min(len(r), len(alt)) and r[k] == alt[k]:
            k += 1
        vcf.append((pos + k, pos + len(r), alt[k:]))
    assert len(vcf) == 8

    def normalised(left, right, alt):   # compare whole-genome edits: same edited sequence <=> same variant
        return ref[:left] + alt + ref[right:]
    assert sorted(normalised(a.left_offset, a.right_offset, a.seq) for a in found) == \
        sorted(normalised(*v) for v in vcf)
    assert mine
