"""BASELINE.json configs[1] where the GPU test run sees it: 30x paired 150 bp reads of the 4 641 652 bp synthetic genome
(SURVEY.md section 8d's "ecoli" shape) -> FASTQ -> `bgbinary create` -> `bgbinary discovery`, bit-matched against the CPU
restatement: every container part of seqset and readmap against the threaded oracle (oracle/bg_oracle.cpp, orc_*_mt), and
the VCF record lines of a 16 kb stretch against oracle/tracer.py + align.py + coverage.py run on the oracle's own seqset,
readmap and ref_map of the same reads (whole reference).  The same comparison bench.py's cpu_baseline leg makes, here as a test."""
import os
import subprocess
import sys

import numpy as np
import pytest

import biograph_amd as bg
from oracle import align as OA
from oracle import coverage as OC
from oracle import oracle as O
from oracle import tracer as OT

pytestmark = pytest.mark.gpu

GENOME = 4_641_652
READ_LEN = 150
SEED = 20261002
REGION = 16_000


def _write_fastq(path, rows):
    """rows: uint8 [n, L] ASCII reads -> a FASTQ file with fixed-width names, written in one piece."""
    n, L = rows.shape
    names = np.frombuffer(b"".join(b"@r%08d\n" % i for i in range(n)), np.uint8).reshape(n, 11)
    rec = np.empty((n, 11 + L + 1 + 2 + L + 1), np.uint8)
    rec[:, :11] = names
    rec[:, 11:11 + L] = rows
    rec[:, 11 + L] = 10
    rec[:, 12 + L] = ord("+")
    rec[:, 13 + L] = 10
    rec[:, 14 + L:14 + 2 * L] = ord("I")
    rec[:, 14 + 2 * L] = 10
    rec.tofile(path)


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    from biograph_amd import synth
    d = tmp_path_factory.mktemp("config1")
    dseq, doffs, n, genome = synth.make_reads(genome_len=GENOME, coverage=30, read_len=READ_LEN, seed=SEED, device="cuda",
                                              return_genome=True)
    assert n == 2 * (30 * GENOME // (2 * READ_LEN))  # 464 165 pairs
    seq = np.ascontiguousarray(dseq.cpu().numpy())
    offs = doffs.cpu().numpy().astype(np.uint64)
    rows = seq.reshape(n, READ_LEN)
    _write_fastq(d / "r1.fq", rows[0::2])
    _write_fastq(d / "r2.fq", rows[1::2])
    codes = genome.cpu().numpy().astype(np.uint8)
    ref = np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode()
    (d / "ref.fa").write_text(">chrS\n" + "\n".join(ref[i:i + 60] for i in range(0, GENOME, 60)) + "\n")
    exe = [sys.executable, "-m", "biograph_amd.cli"]
    r = subprocess.run(exe + ["create", "--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq"), "--out", str(d / "s.bg")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the CPU restatement of the create path on the same reads, on the host's threads
    threads = min(O.hardware_threads(), 16)
    ks = O.KmerSet.count(seq, offs, 30, 5, threads=threads)
    cseq, coffs, keep, cstats = O.correct_reads(ks, seq, offs, 8, 2, 0.7, threads=threads)
    ent = O.Entries.build(cseq, coffs, "cover", 30, threads=threads)
    oss = O.Seqset.from_entries(ent, threads=threads)
    orm = O.Readmap.from_packed(oss, ent, cseq, coffs, True)
    return {"dir": d, "exe": exe, "ref": ref, "codes": codes, "oss": oss, "orm": orm, "n_reads": n}


def test_create_cli_bit_matches_the_cpu_restatement(sample):
    """seqset: every payload part; readmap: every payload part (paired, the oracle's total row order)."""
    d = sample["dir"]
    ctx = bg.Context(0)
    try:
        ss = ctx.open_seqset(str(d / "s.bg" / "seqset"))
        want = sample["oss"].parts()
        assert ss.size() == len(sample["oss"])
        for name in bg.PART_NAMES:
            assert ss.part(name) == want[name], name
        rms = [f for f in os.listdir(d / "s.bg" / "coverage") if f.endswith(".readmap")]
        assert len(rms) == 1
        rm = ctx.open_readmap(str(d / "s.bg" / "coverage" / rms[0]))
        want_rm = sample["orm"].parts()
        for name in bg.READMAP_PART_NAMES:
            assert rm.part(name) == want_rm[name], name
        assert ss.size() > 2 * GENOME * 0.9  # about one entry per genome position and strand, and then some
        rm.free()
        ss.free()
    finally:
        ctx.close()


def test_discovery_cli_vcf_lines_match_the_cpu_restatement(sample):
    """`bgbinary discovery --bed` over the first 16 kb (push tracers in both directions, anchor_dropper, pair_counter, serial
    pipeline, calc_coverage, simple genotype filter, VCF records) == the oracle's engine on the oracle's structures."""
    d, ref, codes = sample["dir"], sample["ref"], sample["codes"]
    (d / "r.bed").write_text("chrS\t0\t%d\n" % REGION)
    out = d / "v.vcf"
    r = subprocess.run(sample["exe"] + ["discovery", "--in", str(d / "s.bg"), "--ref", str(d / "ref.fa"), "--out", str(out),
                                        "--bed", str(d / "r.bed"), "--ml-features", "false", "--simple-gt", "true",
                                        "--rvg-exclude", "false", "--enable-pop-tracer", "false"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [x.rstrip("\n") for x in open(out) if not x.startswith("#")]
    oss, orm = sample["oss"], sample["orm"]
    mo = int(0.7 * READ_LEN)
    graph = OT.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([GENOME], np.uint64)))
    o, ao, sc = OT.Options(min_overlap=mo), OA.AlignOptions(min_overlap=mo), OT.Scaffold([(0, ref)])
    traced = []
    for rev in (False, True):
        traced += OT.reversable_trace(graph, sc, o, rev_comp=rev, start_offset=0, limit_offset=REGION,
                                      anchor_drop=lambda s_, emit: OA.AnchorDropper(ao, s_, emit))
    _, sections = OA.variants_pipeline(ao, sc, traced, through_vcf_pad=True)
    co = OC.CoverageOptions()
    kept = OC.simple_genotype_filter(co, OC.calc_coverage(graph, [int(x) for x in orm.get("read_len")], sc, co, sections))
    want = [OC.vcf_line(co, "chrS", lambda off, k: ref[off:off + k], a) for a in kept]

    def no_pi(line):  # PI is the assembly's number in the process: the command numbers over its whole run
        f = line.split("\t")
        s = f[9].split(":")
        s[3] = "PI"
        return "\t".join(f[:9] + [":".join(s)])
    assert len(want) >= 10
    assert [no_pi(x) for x in got] == [no_pi(x.rstrip("\n")) for x in want]
