"""trace_ref's work list (modules/variants/trace_ref.cpp:27-105) restated in biograph_amd.population: piece
boundaries, the bidirectional overlap, the execution order of scaffolds, the round-robin share of a rank."""
import pytest

from biograph_amd import population as pop


def test_pieces_cover_the_range_without_overlap():
    work = pop.ref_chunks([("1", 0, 2_500_000)])
    assert work == [("1", 0, 1_000_000), ("1", 1_000_000, 2_000_000), ("1", 2_000_000, 2_500_000)]
    work = pop.ref_chunks([("1", 300, 1000)], split_size=256)
    assert [w[1:] for w in work] == [(300, 556), (556, 812), (812, 1000)]
    assert pop.ref_chunks([("1", 5, 5)]) == []


def test_bidirectional_pieces_overlap_by_the_read_ahead_distance():
    # trace_ref.cpp:54-59: the next piece starts read_ahead_distance before the end of this one
    work = pop.ref_chunks([("1", 0, 2500)], split_size=1000, read_ahead_distance=100, bidir=True)
    assert [w[1:] for w in work] == [(0, 1000), (900, 1900), (1800, 2500)]
    with pytest.raises(ValueError):
        pop.ref_chunks([("1", 0, 10)], split_size=10, read_ahead_distance=10, bidir=True)


def test_scaffold_order_is_trace_refs():
    # numeric names in numeric order; then shorter names first; equal lengths in descending name order (:77-105)
    names = ["hs37d5", "X", "10", "2", "Y", "MT", "1"]
    work = pop.ref_chunks([(n, 0, 10) for n in names])
    assert [w[0] for w in work] == ["1", "2", "10", "Y", "X", "MT", "hs37d5"]
    work = pop.ref_chunks([("2", 0, 25), ("1", 0, 15)], split_size=10)
    assert work == [("1", 0, 10), ("1", 10, 15), ("2", 0, 10), ("2", 10, 20), ("2", 20, 25)]


def test_round_robin_share():
    work = pop.ref_chunks([("1", 0, 100)], split_size=10)
    parts = [pop.chunks_of_rank(work, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == sorted(work) and [len(p) for p in parts] == [4, 3, 3]
    assert parts[0][0] == work[0] and parts[1][0] == work[1]


def test_extents_split_at_n():
    assert pop.extents_of("ACGTNNACNG") == [(0, "ACGT"), (6, "AC"), (9, "G")]
    assert pop.extents_of("NNN") == [] and pop.extents_of("") == []
