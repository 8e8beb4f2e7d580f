"""oracle/pop_tracer.py (pop_tracer.cpp + the wiring of trace_ref::execute_work) against the reference's own vectors:
modules/variants/trace_ref_test.cpp:230-300 (trace_ref_pop_test: simple, additional_inside_read; forward and reverse pop
tracer).  Reference of the test (:36-55): scaffold "alpha" = tseq(abcdefg) + N x len(tseq(hijklm)) + tseq(nopqrstuvw),
scaffold "num" = tseq(0..9), plus the four single-base scaffolds of reference_testutil.cpp:67-75.  CPU only."""
import numpy as np
import pytest

from oracle import align as A
from oracle import oracle as O
from oracle import pop_tracer as P
from oracle import tracer as T

tseq, tseq_rc = O.tseq, O.tseq_rc
ALPHA = [(0, tseq("abcdefg")), (len(tseq("abcdefghijklm")), tseq("nopqrstuvw"))]
REF = [tseq("abcdefg"), tseq("nopqrstuvw"), tseq("0123456789"), "A", "C", "G", "T"]


def world(pairs):
    reads = [r for p in pairs for r in p]
    seq, offs = O.pack_strings(reads)
    ent = O.Entries.build(seq, offs, "simple")
    ss = O.Seqset.from_entries(ent)
    rm = O.Readmap(ss, ent, [tuple(p) for p in pairs], True)
    codes = np.array([b"ACGT".index(bytes([c])) for s in REF for c in s.encode()], np.uint8)
    starts = np.cumsum([0] + [len(s) for s in REF[:-1]]).astype(np.uint64)
    sizes = np.array([len(s) for s in REF], np.uint64)
    return T.Graph(ss, rm, ss.ref_map_build(codes, starts, sizes)), ent.strings()


def run(pairs, rev):
    g, strings = world(pairs)
    # trace_ref_test.cpp:82-93,236-250
    o = T.Options(min_overlap=30, trace_reference_assemblies=True, forward_pairs_face_inward=False, max_pair_distance=1000)
    po = P.PopOptions(o, min_pair_distance=10, min_pop_overlap=20)
    sc = T.Scaffold(ALPHA)
    ao = A.AlignOptions(min_overlap=o.min_overlap, read_ahead_distance=o.read_ahead_distance, trace_reference_assemblies=True)
    return P.execute_work(g, strings, sc, o, po, anchor_drop=lambda s, emit: A.AnchorDropper(ao, s, emit),
                          skip_pop_fwd=rev, skip_pop_rev=not rev)


def is_asm(a, left, seq, right):  # AssemblyIs (assemble_testutil.h:200-206)
    return (not a.matches_reference) and a.left_offset == left and a.right_offset == right and a.seq == seq


G = "G"


@pytest.mark.parametrize("rev", [False, True])
def test_simple(rev):
    out = run([(tseq_rc("nopq"), tseq("tuvw")),  # full reads before and after the variant: the push tracer's anchors
               (tseq_rc("abcde"), tseq("opq") + G + tseq("R")), (tseq_rc("abcde"), tseq("pq") + G + tseq("RS")),
               (tseq_rc("abcde"), tseq("RS") + G + tseq("tu")), (tseq_rc("abcde"), tseq("S") + G + tseq("tuv"))], rev)
    assert len(out) == 1, out
    a = out[0]
    assert (is_asm(a, len(tseq("abcdefghijklmnopq")), G + tseq("RS") + G + tseq("tuvw"), len(tseq("abcdefghijklmnopqrstuvw")))
            or is_asm(a, len(tseq("abcdefghijklm")), tseq("nopq") + G + tseq("RS") + G, len(tseq("abcdefghijklmnopqrs")))), a
    assert a.tags == {"POP"}


@pytest.mark.parametrize("rev", [False, True])
def test_additional_inside_read(rev):
    out = run([(tseq_rc("nopq"), tseq("tuvw")),
               (tseq_rc("nopq"), tseq("0123")),  # pair information that makes the pop tracer aware of tseq("0123")
               (tseq_rc("abcd"), tseq("opq") + G + tseq("0")), (tseq_rc("abcd"), tseq("pq") + G + tseq("01")),
               (tseq_rc("abcd"), tseq("23") + G + tseq("tu")), (tseq_rc("abcd"), tseq("3") + G + tseq("tuv"))], rev)
    assert len(out) == 1, out
    a = out[0]
    assert (is_asm(a, len(tseq("abcdefghijklmnopq")), G + tseq("0123") + G + tseq("tuvw"), len(tseq("abcdefghijklmnopqrstuvw")))
            or is_asm(a, len(tseq("abcdefghijklm")), tseq("nopq") + G + tseq("0123") + G, len(tseq("abcdefghijklmnopqrs")))), a


def test_std_sort_is_libstdcxx():
    """std_sort against what g++'s std::sort does to keys with many ties (the order of equal keys is the point)."""
    import subprocess
    import tempfile
    import os
    import random
    rng = random.Random(5)
    cases = [[rng.randrange(0, m) for _ in range(n)] for n, m in ((5, 3), (16, 4), (17, 4), (40, 5), (100, 7), (300, 3), (1000, 50))]
    src = r'''
#include <algorithm>
#include <cstdio>
#include <vector>
int main() {
  int n;
  while (scanf("%d", &n) == 1) {
    std::vector<std::pair<int, int>> v(n);
    for (int i = 0; i < n; ++i) { scanf("%d", &v[i].first); v[i].second = i; }
    std::sort(v.begin(), v.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    for (auto& p : v) printf("%d ", p.second);
    printf("\n");
  }
}
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.cpp"), "w").write(src)
        subprocess.run(["g++", "-O1", "-o", os.path.join(d, "s"), os.path.join(d, "s.cpp")], check=True)
        inp = "".join("%d %s\n" % (len(c), " ".join(map(str, c))) for c in cases)
        got = subprocess.run([os.path.join(d, "s")], input=inp, capture_output=True, text=True, check=True).stdout.splitlines()
    for c, line in zip(cases, got):
        v = [(k, i) for i, k in enumerate(c)]
        P.std_sort(v, lambda a, b: a[0] < b[0])
        assert [i for _, i in v] == [int(x) for x in line.split()]
