"""Super-k-mer counting path (csrc/kmer_skm.hip) against the oracle and against the instance path it
replaces: minimiser runs cut at the record capacity (low-complexity reads), every k the path accepts, the
staging-area and record-buffer overflow paths, extra partition levels, and the presence bits it hands to the
read correction (including the buckets whose LDS table overflows)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import biograph_amd as bg
    return bg.Context(0)


def _low_complexity_reads(seed, n_reads=1200, read_len=150):
    rng = np.random.default_rng(seed)
    reads = util.synth_reads(seed, genome_len=2500, n_reads=n_reads // 3, read_len=read_len, err=0.01, n_rate=0.002,
                             alphabet="AC", var_len=True)
    reads += util.synth_reads(seed + 1, genome_len=2500, n_reads=n_reads // 3, read_len=read_len, err=0.005,
                              n_rate=0.001, repeat=(7, 200), var_len=True)
    for _ in range(n_reads // 3):  # homopolymers and dinucleotide runs with a few errors: one minimiser per read
        L = int(rng.integers(40, read_len + 1))
        unit = ["A", "T", "AC", "GT", "ACG"][int(rng.integers(0, 5))]
        r = list((unit * L)[:L])
        for j in range(L):
            if rng.random() < 0.01:
                r[j] = "ACGTN"[int(rng.integers(0, 5))]
        reads.append("".join(r))
    return reads


def _check_count(ctx, reads, k, mc):
    seq, offs = O.pack_strings(reads)
    ok, of, ofw, orv = O.KmerSet.count(seq, offs, k, mc).arrays()
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, k, mc)
    gk, gf, gc = ks.download()
    assert len(gk) == len(ok)
    assert (gk == ok).all() and (gf == of).all() and (gc == ofw + orv).all()
    assert ks.distinct == len(O.KmerSet.count(seq, offs, k, 1, keep_all=True))
    return r, ks


@pytest.mark.parametrize("k", [14, 16, 22, 27, 30, 31])
def test_low_complexity_reads_every_k(ctx, k):
    _check_count(ctx, _low_complexity_reads(k), k, 2)


@pytest.mark.parametrize("k,rl", [(14, 255), (31, 255), (30, 31), (20, 64)])
def test_read_length_edges(ctx, k, rl):
    reads = util.synth_reads(11 + k, genome_len=5000, n_reads=900, read_len=rl, err=0.01, n_rate=0.004, var_len=False)
    reads += ["ACGT" * 3, "N" * rl, "A" * rl, ""]  # shorter than k, all N, one run, empty
    _check_count(ctx, reads, k, 2)


@pytest.mark.parametrize("env", [{"BG_SKM_READS_PER_TILE": "50000"}, {"BG_SKM_CAP": "100"},
                                 # the generator that groups its records by the top byte of their bucket (BG_SKM_BINNED=1;
                                 # measured, left off): regions big enough for any input, with the staging area overfilled as
                                 # well; regions of the default length and of 8 places (they overflow: back to the plain generator)
                                 {"BG_SKM_BINNED": "big"}, {"BG_SKM_BINNED": "big", "BG_SKM_READS_PER_TILE": "50000"},
                                 {"BG_SKM_BINNED": "tiny"}, {"BG_SKM_BINNED": "1"},
                                 {"BG_SKM_BINNED": "big", "BG_KMER_EXTRA_BITS": "3"},
                                 {"BG_KMER_EXTRA_BITS": "3"}, {"BG_KMER_EXTRA_BITS": "5", "BG_KMER_LDS_LOG2CAP": "4"},
                                 {"BG_KMER_SKM": "0"},
                                 # a first table most buckets overflow, a second try that takes some of them and
                                 # leaves the rest to the sort-based count; and the same without a second try
                                 {"BG_KMER_LDS_SLOTS": "12", "BG_KMER_LDS_SLOTS2": "40"},
                                 {"BG_KMER_LDS_SLOTS": "12", "BG_KMER_LDS_SLOTS2": "4000"},
                                 {"BG_KMER_LDS_SLOTS": "12", "BG_SKM_NO_SECOND_TRY": "1"}])
def test_overflow_and_fallback_paths(ctx, monkeypatch, env):
    for a, b in env.items():
        monkeypatch.setenv(a, b)
    reads = util.synth_reads(5, genome_len=20000, n_reads=6000, read_len=120, err=0.01, n_rate=0.002, var_len=True)
    _check_count(ctx, reads + _low_complexity_reads(9, 600), 24, 3)


@pytest.mark.parametrize("env", [{}, {"BG_KMER_LDS_LOG2CAP": "2"}, {"BG_KMER_LDS_LOG2CAP": "7"},
                                 {"BG_KMER_LDS_SLOTS": "10", "BG_KMER_LDS_SLOTS2": "30"},
                                 # the k-mer hash table built region by region in LDS (regions of one and of two
                                 # buckets at this size: chains spill over into the next region)
                                 {"BG_HASH_REGION_MIN_LOG2CAP": "19", "BG_HASH_MIN_LOG2CAP": "19"},
                                 {"BG_HASH_REGION_MIN_LOG2CAP": "19", "BG_HASH_MIN_LOG2CAP": "20"},
                                 {"BG_SCREEN_LOOKUP": "1"}, {"BG_SKM_READS_PER_TILE": "50000"}, {"BG_SKM_BINNED": "big"},
                                 {"BG_SKM_BINNED": "big", "BG_KMER_LDS_LOG2CAP": "2"}])
def test_presence_bits_drive_the_correction(ctx, monkeypatch, env):
    """bg_correct_reads takes the window presence from the count (no look-ups); the buckets whose table
    overflowed get theirs from the finished hash set; BG_SCREEN_LOOKUP=1 is the look-up screen."""
    for a, b in env.items():
        monkeypatch.setenv(a, b)
    k, mc = 22, 3
    reads = util.synth_reads(21, genome_len=4000, n_reads=3000, read_len=130, err=0.02, n_rate=0.004, var_len=True)
    reads += _low_complexity_reads(3, 300)
    seq, offs = O.pack_strings(reads)
    oks = O.KmerSet.count(seq, offs, k, mc)
    cseq, coffs, keep, stats = O.correct_reads(oks, seq, offs, 8, 2, 0.7)
    r, ks = _check_count(ctx, reads, k, mc)
    c, st = ctx.correct_reads(r, ks, 8, 2, 0.7)
    gseq, goffs = c.download()
    assert (goffs == coffs).all() and (gseq == cseq).all()
    for key in ("corrected_reads", "corrected_bases", "failed", "modified", "modified_bases", "truncated",
                "dropped_bases"):
        assert st[key] == stats[key], key
    # a set counted from OTHER reads must not lend its presence bits
    other = ctx.pack_reads(reads[: len(reads) // 2])
    c2, _ = ctx.correct_reads(other, ks, 8, 2, 0.7)
    s2, o2 = O.pack_strings(reads[: len(reads) // 2])
    cseq2, coffs2, _, _ = O.correct_reads(oks, s2, o2, 8, 2, 0.7)
    g2, go2 = c2.download()
    assert (go2 == coffs2).all() and (g2 == cseq2).all()


def test_pool_cache_cap_env():
    """BG_POOL_CAP_GB bounds what the allocator keeps from the driver without using it (a segment that becomes wholly
    free beyond the cap goes back at once): the build still runs and gives the same seqset; once its result is
    freed, what is held stays under the cap."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "import biograph_amd as bg\nfrom biograph_amd import synth\n"
            "ctx = bg.Context(0)\n"
            "seq, offs, n = synth.make_reads(genome_len=300000, coverage=20, read_len=150, seed=4, device='cuda')\n"
            "ss, st = ctx.create_device(seq.data_ptr(), offs.data_ptr(), n, ctx.default_params())\n"
            "h = hashlib.sha256(b''.join(ss.parts()[k] for k in bg.PART_NAMES)).hexdigest()\n"
            "ss.free()\n"
            "print('RESULT', h, ctx.memory()['cached'])\n") % root
    outs = []
    for cap in ("0.002", "64"):
        env = dict(os.environ, BG_POOL_CAP_GB=cap)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][0].split()
        outs.append((line[1], int(line[2])))
    assert outs[0][0] == outs[1][0]
    assert outs[0][1] <= 2_000_000 < outs[1][1]
