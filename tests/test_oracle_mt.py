"""The threaded stages of the oracle (the CPU baseline bench.py times) give exactly what the
single-threaded restatement gives — which tests/test_oracle_golden.py pins on the reference's fixtures."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import util


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_threaded_stages_equal_single_threaded(threads):
    reads = util.synth_reads(31, genome_len=20000, n_reads=5000, read_len=100, err=0.01, n_rate=0.002, var_len=True)
    seq, offs = O.pack_strings(reads)
    ks1 = O.KmerSet.count(seq, offs, 25, 3)
    ksT = O.KmerSet.count(seq, offs, 25, 3, threads=threads)
    for a, b in zip(ks1.arrays(), ksT.arrays()):
        assert (a == b).all()
    c1 = O.correct_reads(ks1, seq, offs, 8, 2, 0.7)
    cT = O.correct_reads(ksT, seq, offs, 8, 2, 0.7, threads=threads)
    assert (c1[0] == cT[0]).all() and (c1[1] == cT[1]).all() and (c1[2] == cT[2]).all() and c1[3] == cT[3]
    e1 = O.Entries.build(c1[0], c1[1], "cover", 25)
    eT = O.Entries.build(c1[0], c1[1], "cover", 25, threads=threads)
    assert len(e1) == len(eT) and e1.n_records == eT.n_records
    assert e1.strings() == eT.strings()
    s1 = O.Seqset.from_entries(e1)
    sT = O.Seqset.from_entries(eT, threads=threads)
    assert s1.parts() == sT.parts()


def test_threaded_golden_ecoli(golden_dir):
    reads = O.read_fastq(golden_dir + "/e_coli_10000snp/reads.fq")
    seq, offs = O.pack_strings(reads)
    ks = O.KmerSet.count(seq, offs, 30, 5, threads=4)
    assert len(ks) == 7108
    cseq, coffs, keep, st = O.correct_reads(ks, seq, offs, threads=4)
    assert st["corrected_reads"] == 8444 and st["corrected_bases"] == 288464
    ent = O.Entries.build(cseq, coffs, "cover", 30, threads=4)
    assert len(ent) == 19935
    ss = O.Seqset.from_entries(ent, threads=4)
    ref = O.Seqset.from_entries(O.Entries.build(cseq, coffs, "cover", 30))
    assert ss.parts() == ref.parts()


def test_readmap_from_packed_equals_readmap_from_strings():
    """The CPU baseline's readmap leg (bench.py cpu_baseline) takes the packed corrected reads as they are; the records it
    derives with numpy are the ones the string-based constructor gets."""
    import numpy as np
    from tests import util
    reads = util.synth_reads(9, genome_len=3000, n_reads=2400, read_len=80, err=0.004, n_rate=0.0005, var_len=True)
    seq, offs = O.pack_strings(reads)
    ks = O.KmerSet.count(seq, offs, 20, 3)
    cseq, coffs, keep, _ = O.correct_reads(ks, seq, offs, 8, 2, 0.7)
    corrected = O.unpack_strings(cseq, coffs)
    assert "" in corrected  # the sample is made so that the correction drops some reads
    ent = O.Entries.build(cseq, coffs, "cover", 20)
    oss = O.Seqset.from_entries(ent)
    for paired in (True, False):
        if paired:
            records = [tuple(s for s in corrected[i:i + 2] if s) for i in range(0, len(corrected), 2)]
        else:
            records = [(s,) for s in corrected if s]
        a = O.Readmap(oss, ent, [r for r in records if r], paired)
        b = O.Readmap.from_packed(oss, ent, cseq, coffs, paired)
        pa, pb = a.parts(), b.parts()
        assert pa.keys() == pb.keys() and all(pa[k] == pb[k] for k in pa)
        # the threaded build (look-ups, row sort as merged stable sorts, first pass on threads): the same readmap
        for threads in (2, 5, 8):
            pc = O.Readmap.from_packed(oss, ent, cseq, coffs, paired, threads=threads).parts()
            assert pa.keys() == pc.keys() and all(pa[k] == pc[k] for k in pa), threads
