"""biograph_amd._capi (pybind11 over the C ABI) against the assertions of the reference's own SDK tests, on the
reference's own fixtures: python/biograph/seqset_test.py:17-104 (golden/e_coli_10000snp.bg),
python/biograph/readmap_test.py:22-105 (golden/e_coli_merged.bg + the unmerged golden) and
python/biograph/variants/graph_discover_test.py:41-56 (father_lambda.bg: 8 anchored variants)."""
import json
import shutil

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch  # noqa: F401  (its HIP runtime must be the one that gets loaded)
    from biograph_amd import _capi
    return _capi


def _bgdir(tmp, name, seqset, readmaps, accession=None):
    d = tmp / name
    (d / "coverage").mkdir(parents=True)
    (d / "metadata").mkdir()
    shutil.copy(seqset, d / "seqset")
    samples = {}
    for i, (sample, path) in enumerate(readmaps):
        sha = "%040x" % (i + 1)
        shutil.copy(path, d / "coverage" / (sha + ".readmap"))
        samples[sample] = sha
    json.dump({"accession_id": accession or name, "samples": samples}, open(d / "metadata" / "bg_info.json", "w"),
              separators=(",", ":"))
    return str(d)


@pytest.fixture(scope="module")
def ecoli(capi, golden_dir, tmp_path_factory):
    g = golden_dir + "/e_coli_10000snp"
    return capi.BioGraph(_bgdir(tmp_path_factory.mktemp("bg"), "e_coli_10000snp.bg", g + "/seqset",
                                [("e_coli_10000snp", g + "/readmap")]))


@pytest.fixture(scope="module")
def merged(capi, golden_dir, tmp_path_factory):
    g = golden_dir + "/e_coli_merged"
    return capi.BioGraph(_bgdir(tmp_path_factory.mktemp("bg"), "e_coli_merged.bg", g + "/seqset",
                                [("test_accession_id", g + "/readmap_test_accession_id"),
                                 ("e_coli_test", g + "/readmap_e_coli_test")]))


def test_seqset_cases(ecoli):
    ss = ecoli.seqset
    gattaca = ss.find("GATTACA")
    assert ss.size() == 19935                                                     # test_size
    empty = ss.empty_entry()                                                      # test_empty_entry
    assert empty is not None and len(empty) == 0 and empty.sequence() == ""
    assert str(empty) == "<SeqsetEntry 0-19935: >" == repr(empty)
    assert empty == ss.find("")
    assert empty.get_begin_entry_id() == 0 and empty.get_end_entry_id() == ss.size()
    assert gattaca is not None and len(gattaca) == 7                              # test_find_existing
    assert gattaca.sequence() == "GATTACA" and gattaca.sequence(3) == "GAT"
    assert str(gattaca) == "<SeqsetEntry 11076-11077: GATTACA>" == repr(gattaca)
    assert gattaca.get_begin_entry_id() == 11076 and gattaca.get_end_entry_id() == 11077
    assert ss.find("GATTACAGATTACA") is None                                      # test_find_not_existing
    for n in (8, 7):                                                              # test_truncate
        t = gattaca.truncate(n)
        assert t == gattaca and not (t != gattaca)
    t = gattaca.truncate(6)
    assert t != gattaca and not (t == gattaca) and t.sequence() == "GATTAC"
    for n, want in enumerate(["", "G", "GA", "GAT", "GATT", "GATTA", "GATTAC", "GATTACA", "GATTACA"]):  # test_sequence
        assert gattaca.sequence(n) == want
    assert gattaca.pop_front().sequence() == "ATTACA"                             # test_pop_front
    r = ss.find("T").pop_front()
    assert r == ss.empty_entry()
    with pytest.raises(RuntimeError):
        r.pop_front()
    with pytest.raises(TypeError):                                                # test_get_entry_by_id
        ss.get_entry_by_id(-1)
    assert ss.get_entry_by_id(0).sequence() == "AAAAAAAAGCCCGCACTGTCAGGTGCGGGC"
    by_id = ss.get_entry_by_id(6645)
    assert by_id.sequence() == "CCCCACCTGACGCGAAAGGATCGCAGT"
    assert by_id.get_begin_entry_id() == 6645 and by_id.get_end_entry_id() == 6646
    assert ss.get_entry_by_id(19934).sequence() == "TTTTTTTTCGACCAAAGGTAACGAGGTAACAACCA"
    with pytest.raises(RuntimeError):
        ss.get_entry_by_id(19935)
    assert gattaca.push_front("G") is None                                        # test_push_front
    assert gattaca.push_front("T") == ss.find("TGATTACA")
    assert gattaca.push_front_drop("G").sequence() == "GGATTAC"                   # test_push_front_drop
    assert gattaca.push_front_drop("G", 8) is None
    assert gattaca.push_front_drop("G", 7) is None
    assert gattaca.push_front_drop("G", 6).sequence() == "GGATTAC"
    assert ss.max_sequence_length() == 35
    assert len({gattaca, ss.find("GATTACA"), empty}) == 2                         # hashable like the reference's


def test_readmap_cases(ecoli, merged):
    ss = merged.seqset
    rm1 = merged.open_readmap("test_accession_id")
    rm2 = merged.open_readmap("e_coli_test")
    unmerged_rm = ecoli.open_readmap("")
    assert rm1.size() == 16888 and rm2.size() == 76094 and unmerged_rm.size() == 16888   # test_size
    assert rm1.min_read_len() == 30 and rm2.min_read_len() == 100                         # test_read_lens
    assert rm1.max_read_len() == 35 and rm2.max_read_len() == 100
    read = rm1.get_read_by_id(123)                                                        # test_read_get_by_id
    assert read.get_read_id() == 123 and len(read) == 35
    assert read.get_seqset_entry().sequence() == "AAAATTACAGAGTACACAACATCCATGAAACGCAT"
    assert len(list(rm1.get_prefix_reads(ss.find("A")))) == 0                             # test_get_empty_prefixes
    rm1.get_read_by_id(16887)                                                             # ..._out_of_range
    with pytest.raises(RuntimeError):
        rm1.get_read_by_id(16888)
    r = ss.find("AAAATTACAGAGTACACAACATCCATGAAACGCAT")                                    # test_get_prefixes_equal
    assert not list(rm2.get_prefix_reads(r))
    reads = list(rm1.get_prefix_reads(r))
    assert [x.get_read_id() for x in reads] == [123, 124]
    assert reads[0].is_original_orientation() and not reads[1].is_original_orientation()
    assert all(x.get_seqset_entry().sequence() == r.sequence() for x in reads)
    r = ss.find("GGATGAAATGAGTTGCCATCTGGTGCTCACCACGG")                                    # test_get_prefixes_prefix
    assert not list(rm2.get_prefix_reads(r))
    assert [(x.get_read_id(), x.get_seqset_entry().sequence()) for x in rm1.get_prefix_reads(r)] == [
        (11160, "GGATGAAATGAGTTGCCATCTGGTGCTCACCACGG"), (11161, "GGATGAAATGAGTTGCCATCTGGTGCTCAC"),
        (11162, "GGATGAAATGAGTTGCCATCTGGTGCTCACC"), (11163, "GGATGAAATGAGTTGCCATCTGGTGCTCACCA"),
        (11164, "GGATGAAATGAGTTGCCATCTGGTGCTCACCACG"), (11165, "GGATGAAATGAGTTGCCATCTGGTGCTCACCACGG")]
    read = rm2.get_read_by_id(63816)                                                      # test_get_mate
    assert read.has_mate() and read.get_mate().get_read_id() == 37698
    read = rm2.get_read_by_id(35906)
    assert not read.has_mate()
    with pytest.raises(RuntimeError):
        read.get_mate()
    with pytest.raises(RuntimeError):                                                     # test_wrong_seqset
        rm1.get_prefix_reads(ecoli.seqset.find("AAAATTACAGAGTACACAACATCCATGAAACGCAT"))
    r = ss.find("TGAAATGAGTTGCCATCTGGTGCTCAC")                                            # test_reads_containing
    assert [(off, x.get_read_id(), x.get_seqset_entry().sequence()) for off, x in rm1.get_reads_containing(r)] == [
        (0, 14544, "TGAAATGAGTTGCCATCTGGTGCTCACCACGG"), (1, 3553, "ATGAAATGAGTTGCCATCTGGTGCTCACCACGG"),
        (3, 11160, "GGATGAAATGAGTTGCCATCTGGTGCTCACCACGG"), (3, 11161, "GGATGAAATGAGTTGCCATCTGGTGCTCAC"),
        (3, 11162, "GGATGAAATGAGTTGCCATCTGGTGCTCACC"), (3, 11163, "GGATGAAATGAGTTGCCATCTGGTGCTCACCA"),
        (3, 11164, "GGATGAAATGAGTTGCCATCTGGTGCTCACCACG"), (3, 11165, "GGATGAAATGAGTTGCCATCTGGTGCTCACCACGG")]
    with pytest.raises(RuntimeError):
        rm1.get_reads_containing(ecoli.seqset.find("TGAAATGAGTTGCCATCTGGTGCTCAC"))
    st1, st2 = rm1.get_pair_stats(), rm2.get_pair_stats()                                 # test_pair_stats
    assert (st1.paired_reads, st1.paired_bases, st1.unpaired_reads, st1.unpaired_bases) == (0, 0, 8444, 288464)
    assert (st2.paired_reads, st2.paired_bases, st2.unpaired_reads, st2.unpaired_bases) == (25762, 2576200, 12285, 1228500)
    cov = rm1.get_approx_seq_coverage("GGATGAAATGAGTTGCCATCTGGTGCTCACCACGG")             # readmap.cpp:350-411
    # the 35-base read (id 11165 and its 34-, 32-, 31-, 30-base prefixes... each read ending inside the sequence counts
    # over its own span): every position is covered by the full read at least
    assert len(cov) == 35 and min(cov) >= 1 and max(cov) >= 2 and cov[0] >= cov[-1]
    with pytest.raises(RuntimeError):   # two samples: a readmap must be named (biograph_dir.cpp:108-111)
        merged.open_readmap("")
    assert merged.metadata.samples.keys() == {"test_accession_id", "e_coli_test"}


def test_graph_discover_generators(capi, golden_dir, tmp_path):
    """python/biograph/variants/graph_discover_test.py:41-56 through the generator functions of the module."""
    g = golden_dir + "/father_lambda"
    bgd = capi.BioGraph(_bgdir(tmp_path, "father_lambda.bg", g + "/seqset", [("father", g + "/readmap")]))
    seqset, rm = bgd.seqset, bgd.open_readmap()
    ref = capi.Reference(golden_dir + "/ref_lambda")

    def dedup(asms):
        prev = None
        for a in asms:
            if prev:
                if prev.left_offset == a.left_offset and prev.right_offset == a.right_offset and prev.seq == a.seq:
                    continue
                yield prev
            prev = a
        if prev:
            yield prev

    def trim_and_update(asms):
        asms = capi.graph_trim_ref(asms, ref, "lambda")
        asms = dedup(asms)
        return list(capi.update_rc_seqset_entries(asms, seqset, enable_self_test=True))

    asms = capi.make_ref_assemblies(ref, "lambda", 0, ref.scaffold_lens["lambda"])
    asms = trim_and_update(asms)
    asms = capi.discover_branch(asms, rm, tag="BRANCH")
    asms = trim_and_update(asms)
    asms = capi.discover_push_to_pair(asms, rm, tag="PUSH_TO_PAIR", discover_tags=["BRANCH"])
    asms = trim_and_update(asms)
    found = [a for a in asms if a.left_offset and a.right_offset and not a.matches_reference]
    assert len(found) == 8, found
    tags = {t for a in found for t in a.tags}
    assert tags <= {"BRANCH", "PUSH_TO_PAIR"} and tags
