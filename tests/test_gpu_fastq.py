"""FASTQ import on the GPU (csrc/fastq.hip: row A1 of SURVEY.md section 8, read_importer.cpp:604) and the
`bgbinary create` options around it.  The parser is checked against the oracle's host reader on the reference's
own golden FASTQ and on the layouts the reference accepts (mate files, interleaved, --cut-reads, CRLF)."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import biograph_amd as bg
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGBINARY = os.path.join(ROOT, "biograph_amd", "bgbinary")


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def _fastq(reads, names=None, eol="\n", qual="I"):
    out = []
    for i, r in enumerate(reads):
        out.append("@%s%s%s%s+%s%s%s" % (names[i] if names else "r%d" % i, eol, r, eol, eol, qual * len(r), eol))
    return "".join(out).encode()


def _strings(seq, offs):
    return O.unpack_strings(seq, offs)


def test_golden_fastq_parses_like_the_host_reader(ctx, golden_dir):
    path = golden_dir + "/e_coli_10000snp/reads.fq"
    want = O.read_fastq(path)
    seq, offs = ctx.import_fastq([open(path, "rb").read()])
    assert _strings(seq, offs) == want
    seq2, offs2 = ctx.import_fastq([open(path, "rb").read()], pinned=False)
    assert (seq2 == seq).all() and (offs2 == offs).all()


def test_mate_files_interleaved_and_cut(ctx):
    rng = np.random.default_rng(3)
    reads = ["".join(rng.choice(list("ACGTN"), size=int(rng.integers(1, 200)))) for _ in range(3001)]
    a, b = reads[0:3000:2], reads[1:3000:2]
    # two mate files read in lock step: mates adjacent
    seq, offs = ctx.import_fastq([_fastq(a), _fastq(b)], paired_files=True)
    assert _strings(seq, offs) == reads[:3000]
    # two pairs of mate files
    seq, offs = ctx.import_fastq([_fastq(a[:700]), _fastq(b[:700]), _fastq(a[700:]), _fastq(b[700:])], paired_files=True)
    got = _strings(seq, offs)
    assert got[:1400] == reads[:1400] and got[1400:] == reads[1400:3000]
    # interleaved: natural order; an odd record count is refused
    seq, offs = ctx.import_fastq([_fastq(reads[:3000])], interleaved=True)
    assert _strings(seq, offs) == reads[:3000]
    with pytest.raises(bg.BgError, match="even number"):
        ctx.import_fastq([_fastq(reads)], interleaved=True)
    with pytest.raises(bg.BgError, match="different numbers"):
        ctx.import_fastq([_fastq(a), _fastq(b[:-1])], paired_files=True)
    # --cut-reads 10-100: bases [10, min(100, len)) of every read (read_importer.cpp:157-169)
    long_reads = [r for r in reads if len(r) > 10]
    seq, offs = ctx.import_fastq([_fastq(long_reads)], cut=(10, 100))
    assert _strings(seq, offs) == [r[10:100] for r in long_reads]
    with pytest.raises(bg.BgError, match="cut-reads"):
        ctx.import_fastq([_fastq(["ACGT" * 30, "ACG"])], cut=(10, 100))


def test_line_ends_and_malformed_input(ctx):
    reads = ["ACGTACGTAC", "TTGA", "NNNNACGT"]
    base = _fastq(reads)
    for text in (base, base[:-1], base + b"\n\n\n", _fastq(reads, eol="\r\n"), _fastq(reads, eol="\r\n")[:-2]):
        seq, offs = ctx.import_fastq([text])
        assert _strings(seq, offs) == reads
    # qualities that look like headers or newline-free garbage do not confuse the record structure
    seq, offs = ctx.import_fastq([_fastq(reads, qual="@")])
    assert _strings(seq, offs) == reads
    seq, offs = ctx.import_fastq([b""])
    assert len(offs) == 1
    with pytest.raises(bg.BgError, match="malformed"):
        ctx.import_fastq([base.replace(b"@r1", b"xr1")])
    with pytest.raises(bg.BgError, match="malformed"):
        ctx.import_fastq([base.replace(b"\n+\n", b"\n-\n", 1)])
    with pytest.raises(bg.BgError, match="whole number of records"):
        ctx.import_fastq([base + b"@extra\nACGT\n"])


def _run(args, ok=True):
    env = dict(os.environ)
    r = subprocess.run([BGBINARY] + args, capture_output=True, text=True, env=env, timeout=600)
    assert (r.returncode == 0) == ok, r.stderr[-2000:]
    return r


def _bgdir_parts(ctx, d):
    ss = ctx.open_seqset(os.path.join(d, "seqset"))
    info = json.load(open(os.path.join(d, "metadata", "bg_info.json")))
    (sample, sha), = info["samples"].items()
    rm = ctx.open_readmap(os.path.join(d, "coverage", sha + ".readmap"))
    parts = dict(ss.parts())
    for name in bg.READMAP_PART_NAMES:
        parts["readmap/" + name] = rm.part(name)
    n = rm.size()
    rm.free()
    ss.free()
    return parts, info, n


def test_cli_interleaved_gzip_cut_reads_and_unknown_options(ctx, tmp_path):
    reads = util.synth_reads(21, genome_len=6000, n_reads=2400, read_len=100, err=0.004, n_rate=0.0005)
    a, b = reads[0::2], reads[1::2]
    fa, fb, fi, fz = (str(tmp_path / n) for n in ("a.fq", "b.fq", "inter.fq", "inter.fq.gz"))
    open(fa, "wb").write(_fastq(a))
    open(fb, "wb").write(_fastq(b))
    open(fi, "wb").write(_fastq(reads))
    with gzip.open(fz, "wb") as f:
        f.write(_fastq(reads))
    common = ["--kmer-size", "24", "--min-kmer-count", "3"]
    _run(["create", "--reads", fa, "--pair", fb, "--out", str(tmp_path / "pair.bg"), "--id", 'we"ird\\id'] + common)
    _run(["create", "--reads", fi, "--interleaved", "--out", str(tmp_path / "inter.bg")] + common)
    _run(["create", "--reads", fz, "--interleaved", "--out", str(tmp_path / "gz.bg")] + common)
    _run(["create", "--reads", fi, "--out", str(tmp_path / "unpaired.bg")] + common)
    pp, pinfo, pn = _bgdir_parts(ctx, str(tmp_path / "pair.bg"))
    ip, iinfo, in_ = _bgdir_parts(ctx, str(tmp_path / "inter.bg"))
    zp, zinfo, zn = _bgdir_parts(ctx, str(tmp_path / "gz.bg"))
    up, uinfo, un = _bgdir_parts(ctx, str(tmp_path / "unpaired.bg"))
    # interleaved == mate files == gzip of the same, seqset and readmap (mate loops included) byte for byte
    assert ip == pp and zp == pp
    # the same reads unpaired: same seqset, but a readmap without mates
    assert all(up[k] == pp[k] for k in bg.PART_NAMES)
    assert up["readmap/mate_loop_ptr/elements"] != pp["readmap/mate_loop_ptr/elements"]
    # metadata: escaped id, biograph_id = the seqset's uuid, the command line in the history; the log is there
    assert pinfo["accession_id"] == 'we"ird\\id' and list(pinfo["samples"]) == ['we"ird\\id']
    assert iinfo["accession_id"] == "inter"  # default: the stem of the output directory (biograph_create.cpp:518-520)
    assert len(pinfo["biograph_id"]) == 36 and "create" in pinfo["command_history"][0]
    log = open(str(tmp_path / "pair.bg" / "qc" / "create_log.txt")).read()
    assert "Importing reads" in log and "reads survived read correction" in log
    stats = json.load(open(str(tmp_path / "pair.bg" / "qc" / "create_stats.json")))
    assert stats["imported_reads"] == len(reads) and stats["uuid"] == pinfo["biograph_id"]
    # --cut-reads: the build of the cut reads
    _run(["create", "--reads", fi, "--interleaved", "--cut-reads", "5-85", "--out", str(tmp_path / "cut.bg")] + common)
    cut_ref, _ = ctx.create([r[5:85] for r in reads], kmer_size=24, min_kmer_count=3)
    cp, _, _ = _bgdir_parts(ctx, str(tmp_path / "cut.bg"))
    assert all(cp[k] == cut_ref.part(k) for k in bg.PART_NAMES)
    # what this build does not implement is refused, not ignored
    for bad in (["--interleaved", "--pair", fb], ["--no-such-option", "1"], ["--format", "bam"], ["--cut-reads", "7"]):
        r = _run(["create", "--reads", fi, "--out", str(tmp_path / "bad.bg")] + bad + common, ok=False)
        assert r.stderr.strip()
    assert not os.path.exists(str(tmp_path / "bad.bg"))


def test_cli_create_from_bam(ctx, tmp_path):
    """`bgbinary create --reads X.bam`: the records are decoded on the host (tests/test_bam_import.py pins the decoder on
    the reference's golden BAM / SAM pair), matched by name and imported as interleaved text: same biograph as `create`
    on that text, same seqset as the mate files the BAM was made from; reverse-strand records are turned back."""
    from tests.test_bam_import import iupac_rc, write_bam
    reads = util.synth_reads(22, genome_len=6000, n_reads=2400, read_len=100, err=0.004, n_rate=0.0005)
    a, b = reads[0::2], reads[1::2]
    recs = []
    for i, (x, y) in enumerate(zip(a, b)):
        fx = 0x1 | 0x40 | (0x10 if i % 3 == 0 else 0)
        fy = 0x1 | 0x80 | (0x10 if i % 5 == 0 else 0)
        recs.append(("r%d" % i, fx, iupac_rc(x) if fx & 0x10 else x))
        if i % 7 == 0:
            recs.append(("r%d" % i, 0x1 | 0x100, "ACGTACGT"))  # a secondary alignment: ignored
        recs.append(("r%d" % i, fy, iupac_rc(y) if fy & 0x10 else y))
    bam = str(tmp_path / "reads.bam")
    write_bam(bam, recs, bgzf_blocks=5)
    fa, fb, fq = (str(tmp_path / n) for n in ("a.fq", "b.fq", "from_bam.fq"))
    open(fa, "wb").write(_fastq(a))
    open(fb, "wb").write(_fastq(b))
    common = ["--kmer-size", "24", "--min-kmer-count", "3"]
    _run(["bam2fastq", "--reads", bam, "--out", fq])
    _run(["create", "--reads", bam, "--out", str(tmp_path / "bam.bg")] + common)
    _run(["create", "--reads", fq, "--interleaved", "--out", str(tmp_path / "text.bg")] + common)
    _run(["create", "--reads", fa, "--pair", fb, "--out", str(tmp_path / "pair.bg")] + common)
    bp, binfo, bn = _bgdir_parts(ctx, str(tmp_path / "bam.bg"))
    tp, _, tn = _bgdir_parts(ctx, str(tmp_path / "text.bg"))
    pp, _, pn = _bgdir_parts(ctx, str(tmp_path / "pair.bg"))
    assert bp == tp and bn == tn == pn
    assert all(bp[k] == pp[k] for k in bg.PART_NAMES)
    log = open(str(tmp_path / "bam.bg" / "qc" / "create_log.txt")).read()
    assert "BAM: %d records" % len(recs) in log and "paired" in log
    stats = json.load(open(str(tmp_path / "bam.bg" / "qc" / "create_stats.json")))
    assert stats["imported_reads"] == len(reads)
    r = _run(["create", "--reads", bam, "--pair", fb, "--out", str(tmp_path / "bad.bg")] + common, ok=False)
    assert "BAM" in r.stderr
