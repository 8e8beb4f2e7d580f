"""BAM input of `bgbinary create` (read_importer.cpp:182-264,387-470 restated without htslib in csrc/cli/bgbinary.cpp),
on the CPU through `bgbinary bam2fastq`: the reference's golden ERR091571.bam against its SAM text (both data files of the
reference's tests; the SAM reduced to qname / flag / seq by tests/golden/bam/make_fixture.py), and hand-made BAMs for
the flag handling -- reverse strand (the IUPAC vector of bio_base/dna_test.cpp:514-519), secondary / supplementary
records, unpaired reads among paired ones, mates in another file, mates that never come."""
import gzip
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGBINARY = os.path.join(ROOT, "biograph_amd", "bgbinary")
COMP = dict(zip("ABCDGHKMNRSTVWY", "TVGHCDMKNYSABWR"))


def iupac_rc(s):
    return "".join(COMP.get(c, c) for c in reversed(s))


def expected_text(records):
    """records = [(qname, flag, seq)] in file order -> the FASTQ text bam2fastq writes."""
    cache, pairs, unpaired, got_paired = {}, [], [], False
    for q, flag, seq in records:
        if flag & (0x100 | 0x800):
            continue
        if seq == "*":
            seq = ""
        if flag & 0x10:
            seq = iupac_rc(seq)
        if flag & 0x1:
            got_paired = True
            if q in cache:
                pairs.append((q, seq, cache.pop(q)))
            else:
                cache[q] = seq
        else:
            unpaired.append((q, seq))

    def rec(q, s):
        return "@%s\n%s\n+\n%s\n" % (q, s, "I" * len(s))

    out = "".join(rec(q, a) + rec(q, b) for q, a, b in pairs)
    for q, s in unpaired + sorted(cache.items()):
        out += rec(q, s) + (rec(q, "") if got_paired else "")
    return out, len(cache)


def write_bam(path, records, refs=(("chr1", 1000),), bgzf_blocks=1):
    """A minimal BAM writer (SAM spec section 4): records = [(qname, flag, seq)]; one or several gzip members."""
    nt16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    body = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", len(refs))
    for name, ln in refs:
        body += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    chunks = [body]
    for q, flag, seq in records:
        name = q.encode() + b"\0"
        packed = bytearray((len(seq) + 1) // 2)
        for i, c in enumerate(seq):
            packed[i >> 1] |= nt16[c] << (4 if i % 2 == 0 else 0)
        rec = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, flag, len(seq), -1, -1, 0) + name + bytes(packed) + \
            bytes([30] * len(seq))
        chunks.append(struct.pack("<i", len(rec)) + rec)
    data = b"".join(chunks)
    with open(path, "wb") as f:  # several gzip members = what BGZF is to a gzip reader
        step = max(1, len(data) // bgzf_blocks)
        for i in range(0, len(data), step):
            f.write(gzip.compress(data[i:i + step]))


def bam2fastq(tmp_path, *bams):
    out = tmp_path / "out.fq"
    cmd = [BGBINARY, "bam2fastq"]
    for b in bams:
        cmd += ["--reads", str(b)]
    r = subprocess.run(cmd + ["--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out.read_text(), r.stderr


@pytest.mark.skipif(not os.path.exists(BGBINARY), reason="bgbinary not built")
def test_golden_bam_equals_its_sam(tmp_path):
    recs = []
    with gzip.open(os.path.join(ROOT, "tests", "golden", "bam", "ERR091571.qname_flag_seq.tsv.gz"), "rt") as f:
        for line in f:
            q, flag, seq = line.rstrip("\n").split("\t")
            recs.append((q, int(flag), seq))
    assert len(recs) == 9775
    want, lacking = expected_text(recs)
    got, log = bam2fastq(tmp_path, os.path.join(ROOT, "tests", "golden", "bam", "ERR091571.bam"))
    assert got == want
    assert "9775 records" in log and "%d without mate" % lacking in log and "paired" in log
    assert sum(1 for _, f, _ in recs if f & 0x10) > 1000  # the reverse-strand path is exercised


@pytest.mark.skipif(not os.path.exists(BGBINARY), reason="bgbinary not built")
def test_flags_iupac_and_pairing_across_files(tmp_path):
    iupac = "YWVTSRNMKHGDCBAABCDHKMNRSTVWY"
    a = [("p1", 0x1 | 0x40, "ACGTACGTAC"), ("sec", 0x1 | 0x100, "AAAA"), ("sup", 0x800, "CCCC"),
         ("rev", 0x10, iupac), ("p2", 0x1 | 0x80 | 0x10, "AACCGGTTA"), ("lonely", 0x1 | 0x40, "GGGGG"),
         ("single", 0, "TTTTTT"), ("p1", 0x1 | 0x80, "TTGCA"), ("empty", 0, "")]
    b = [("p2", 0x1 | 0x40, "CATCATCAT"), ("eq", 0, "=ACGT")]
    write_bam(tmp_path / "a.bam", a, bgzf_blocks=3)
    write_bam(tmp_path / "b.bam", b)
    got, log = bam2fastq(tmp_path, tmp_path / "a.bam", tmp_path / "b.bam")
    want, lacking = expected_text(a + b)
    assert got == want and lacking == 1
    assert "@rev\nRWBASYNKMDHGVTTVGHCDMKNYSABWR\n" in got           # bio_base/dna_test.cpp:514-519
    assert "@p1\nTTGCA\n+\nIIIII\n@p1\nACGTACGTAC\n" in got          # the later mate first (add_paired_read)
    assert "@p2\nCATCATCAT\n+\nIIIIIIIII\n@p2\nTAACCGGTT\n" in got   # mates in different files; reverse strand turned back
    assert "sec" not in got and "sup" not in got
    assert got.endswith("@lonely\nGGGGG\n+\nIIIII\n@lonely\n\n+\n\n")  # no mate: an unpaired read of a paired run
    # an all-unpaired BAM gives plain records
    write_bam(tmp_path / "c.bam", [("x", 0, "ACGT"), ("y", 0x10, "AACC")])
    got, log = bam2fastq(tmp_path, tmp_path / "c.bam")
    assert got == "@x\nACGT\n+\nIIII\n@y\nGGTT\n+\nIIII\n" and "unpaired" in log


@pytest.mark.skipif(not os.path.exists(BGBINARY), reason="bgbinary not built")
def test_rejects_what_is_not_bam(tmp_path):
    p = tmp_path / "x.bam"
    p.write_bytes(gzip.compress(b"@r\nACGT\n+\nIIII\n"))
    r = subprocess.run([BGBINARY, "bam2fastq", "--reads", str(p), "--out", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode != 0 and "not a BAM file" in r.stderr
    write_bam(tmp_path / "t.bam", [("q", 0, "ACGT")])
    data = (tmp_path / "t.bam").read_bytes()
    raw = gzip.decompress(data)
    (tmp_path / "cut.bam").write_bytes(gzip.compress(raw[:-3]))
    r = subprocess.run([BGBINARY, "bam2fastq", "--reads", str(tmp_path / "cut.bam"), "--out", str(tmp_path / "o")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "truncated" in r.stderr
