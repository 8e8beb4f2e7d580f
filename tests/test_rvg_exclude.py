"""rvg_exclude (modules/variants/rvg_exclude.cpp:5-67), the last step of the reference's standard pipeline
(pipeline.cpp:102-104): the library's step (csrc/coverage.cpp, bg_assemblies_rvg_exclude -- host only, runs without a
GPU) against hand-made cases that follow the reference's statements one by one, and against the oracle's restatement
(oracle/coverage.py) on random section lists.  The reference holds no test vector for this step: parity unpinned beyond
the restatement."""
import random

import pytest

from biograph_amd import tracer as BT
from oracle import coverage as C
from oracle import tracer as T


def sect(left, right, seq, aid, pair_cov=None, other_pair_depth=0, ref=False):
    """A covered section as calc_coverage leaves it (interbase coverage: len(seq) + 1 values)."""
    n = len(seq) + 1
    pc = list(pair_cov) if pair_cov is not None else [1] * n
    kw = dict(left_offset=left, right_offset=right, seq=seq, matches_reference=ref, assembly_id=aid)
    t = BT.TraceAssembly(**kw)
    t.coverage, t.pair_coverage = [3] * len(pc), pc
    t.other_depth, t.other_pair_depth, t.ref_depth = 0, other_pair_depth, 2
    return t


def to_oracle(t):
    a = T.Assembly()
    a.left_offset, a.right_offset, a.seq, a.matches_reference = t.left_offset, t.right_offset, t.seq, t.matches_reference
    a.assembly_id = t.assembly_id
    a.coverage, a.pair_coverage = list(t.coverage), list(t.pair_coverage)
    a.other_depth, a.other_pair_depth, a.ref_depth = t.other_depth, t.other_pair_depth, t.ref_depth
    return a


def key(a):
    return (a.left_offset, a.right_offset, a.seq, a.matches_reference, a.assembly_id, tuple(a.pair_coverage))


def run(sections, **kw):
    return BT.rvg_exclude(sections, BT.CoverageOptions(rvg_exclude=1, **kw))


def test_pair_supported_small_variant_passes_and_unsupported_one_is_dropped():
    good = sect(10, 11, "A", 1, [2, 1])
    bad = sect(20, 21, "C", 2, [1, 0])
    out = run([good, bad])
    assert [key(a) for a in out] == [key(good)]


def test_other_pair_depth_rescues():
    a = sect(20, 21, "C", 2, [0, 0], other_pair_depth=1)
    assert [key(x) for x in run([a])] == [key(a)]


def test_reference_sections_pass():
    r = sect(5, 30, "ACGTACGTACGTACGTACGTACGTA", 3, [0] * 26, ref=True)
    assert [key(x) for x in run([r])] == [key(r)]


def test_structural_section_passes_and_releases_its_phase_mates_on_both_sides():
    before = sect(10, 11, "A", 7, [0, 0])      # held back ...
    other = sect(12, 13, "G", 8, [0, 0])       # ... as is this one, of another assembly: dropped at the end
    sv = sect(40, 41, "ACGT" * 6, 7, [0] * 25)  # 24 bases >= vcf_sv_size_threshold: releases `before`
    after = sect(50, 51, "T", 7, [0, 0])       # known in phase by now
    out = run([before, other, sv, after])
    assert [key(a) for a in out] == [key(before), key(sv), key(after)]


def test_deletion_is_measured_as_if_padded():
    # 19 reference bases deleted: reflen 19 -> 20 with the pad base = structural at the default threshold of 20
    d = sect(100, 119, "", 4, [0])
    assert [key(x) for x in run([d])] == [key(d)]
    d18 = sect(100, 118, "", 5, [0])
    assert run([d18]) == []
    assert [key(x) for x in run([d18], vcf_sv_size_threshold=19)] == [key(d18)]


def test_output_stays_sorted_when_a_held_section_is_released_late():
    held = sect(10, 11, "A", 1, [0, 0])
    mid = sect(15, 16, "C", 2, [1, 1])   # may not overtake `held` while it is tracked
    sv = sect(30, 31, "ACGT" * 5, 1, [0] * 21)
    out = run([held, mid, sv])
    assert [a.left_offset for a in out] == [10, 15, 30]


def test_missing_pair_coverage_is_an_error():
    a = sect(20, 21, "C", 2, [])
    with pytest.raises(Exception):
        run([a])


@pytest.mark.parametrize("seed", range(8))
def test_library_equals_oracle_on_random_sections(seed):
    rng = random.Random(9000 + seed)
    for _ in range(60):
        sections = []
        pos = 0
        n_ids = rng.randint(1, 6)
        for _ in range(rng.randint(0, 40)):
            pos += rng.choice([0, 0, 1, 3, 30])
            kind = rng.random()
            if kind < 0.15:
                seq, span = "".join(rng.choice("ACGT") for _ in range(rng.randint(18, 30))), rng.randint(0, 3)
            elif kind < 0.3:
                seq, span = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 2))), rng.randint(17, 25)
            else:
                seq, span = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 3))), rng.randint(0, 3)
            pc = [rng.choice([0, 1, 1, 2]) for _ in range(len(seq) + 1)]
            sections.append(sect(pos, pos + span, seq, rng.randint(1, n_ids), pc, other_pair_depth=rng.choice([0, 0, 0, 1]),
                                 ref=rng.random() < 0.1))
        thr = rng.choice([20, 20, 5, 19])
        got = run(sections, vcf_sv_size_threshold=thr)
        want = C.rvg_exclude(C.CoverageOptions(rvg_exclude=True, vcf_sv_size_threshold=thr), [to_oracle(t) for t in sections])
        assert [key(a) for a in got] == [key(a) for a in want]
