"""Reference reader against the reference's own lambda reference directory
(datasets/lambdaToyData/benchmark/ref_lambda: karyotype.json + reference.ref, copied as data) and its FASTA.
Host-only code: runs without a GPU."""
import hashlib
import json
import struct

import pytest

import biograph_amd as bg


def _fasta(golden_dir):
    return "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">")).upper()


def test_reference_directory_matches_fasta_and_karyotype(golden_dir):
    ref = bg.Reference(golden_dir + "/ref_lambda")
    kt = json.load(open(golden_dir + "/ref_lambda/karyotype.json"))
    assert ref.scaffolds == kt["chr_order"] == ["lambda"]
    assert ref.scaffold_lens == {c["name"]: c["len"] for c in kt["chromosomes"]}
    assert ref.supercontigs == [(s["chr"], s["offset"], s["offset"] + s["len"]) for s in kt["supercontigs"]]
    fa = _fasta(golden_dir)
    assert ref.size() == len(fa) == 48502
    assert ref.flat_bases(0, ref.size()) == fa
    assert ref.sequence("lambda", 100, 160) == fa[100:160]
    with pytest.raises(ValueError):
        ref.sequence("lambda", 48000, 49000)
    with pytest.raises(bg.BgError):
        ref.flat_bases(48000, 1000)


def test_spec_file_index_and_md5(golden_dir):
    """The bare .ref: scaffold table of its msgpack index (name = the FASTA record it was built from, md5 of its
    bases as stored in the index) and the 2-bit payload."""
    path = golden_dir + "/ref_lambda/reference.ref"
    ref = bg.Reference(path)
    assert ref.scaffolds == ["lambda:0"] and ref.scaffold_lens == {"lambda:0": 48502}
    assert ref.supercontigs == [("lambda:0", 0, 48502)]
    seq = ref.flat_bases(0, ref.size())
    assert seq == _fasta(golden_dir)
    raw = open(path, "rb").read()
    n = struct.unpack(">I", raw[-4:])[0]
    assert hashlib.md5(seq.encode()).hexdigest().encode() in raw[-4 - n:-4]  # scaffold_t::md5


def test_fasta_with_n_runs(tmp_path):
    p = tmp_path / "x.fa"
    p.write_text(">chr1 description\nACGTNNNNacgtac\nGGNN\n>chr2\nNNTTTT\n")
    ref = bg.Reference(str(p))
    assert ref.scaffolds == ["chr1", "chr2"] and ref.scaffold_lens == {"chr1": 18, "chr2": 6}
    assert ref.supercontigs == [("chr1", 0, 4), ("chr1", 8, 16), ("chr2", 2, 6)]
    assert ref.flat_bases(0, ref.size()) == "ACGT" + "ACGTACGG" + "TTTT"
    assert ref.sequence("chr1", 9, 12) == "CGT"
    q = tmp_path / "crlf.fa"   # CRLF line ends, other IUPAC letters (not part of any supercontig), no newline at the end
    q.write_bytes(b">c1\r\nACGTRYac\r\nGTNN\r\nnnAC")
    ref = bg.Reference(str(q))
    assert ref.scaffolds == ["c1"] and ref.scaffold_lens == {"c1": 16}
    assert ref.supercontigs == [("c1", 0, 4), ("c1", 6, 10), ("c1", 14, 16)]
    assert ref.flat_bases(0, ref.size()) == "ACGT" + "ACGT" + "AC"


def test_errors(tmp_path):
    with pytest.raises(bg.BgError):
        bg.Reference(str(tmp_path / "missing"))
    (tmp_path / "d").mkdir()
    with pytest.raises(bg.BgError):
        bg.Reference(str(tmp_path / "d"))
    (tmp_path / "bad.ref").write_bytes(b"SPR001" + b"\x00" * 20)
    with pytest.raises(bg.BgError):
        bg.Reference(str(tmp_path / "bad.ref"))
