"""Sharded (multi-rank) build on ONE GPU: all ranks are simulated in-process (LocalSimComm), every
exchange buffer goes through the same C-ABI phases the RCCL path uses.  The assembled seqset must be
byte-identical to the single-GPU build and to the oracle, for any number of ranks and any split of
the reads."""
import numpy as np
import pytest
import torch

import biograph_amd as bg
from biograph_amd import sharded, synth
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def _shard_inputs(reads, P, rng):
    """Split reads into P uneven shards (one may be empty) as device tensors."""
    n = len(reads)
    cuts = sorted(int(x) for x in rng.integers(0, n + 1, P - 1)) if P > 1 else []
    bounds = [0] + cuts + [n]
    inputs = {}
    keep = []
    for r in range(P):
        part = reads[bounds[r]:bounds[r + 1]]
        seq, offs = O.pack_strings(part)
        dseq = torch.from_numpy(seq.copy()).cuda() if len(seq) else torch.zeros(0, dtype=torch.uint8, device="cuda")
        doffs = torch.from_numpy(offs.astype(np.int64)).cuda()
        keep.append((dseq, doffs))
        inputs[r] = (dseq, doffs, len(part))
    return inputs


@pytest.mark.parametrize("mode", ["routed", "replicated"])
@pytest.mark.parametrize("P", [1, 2, 3, 4, 8])
def test_sharded_build_equals_single_gpu_and_oracle(ctx, P, mode):
    """routed: nothing replicated (kept k-mers stay with their owners, the cover index is range-partitioned);
    replicated: the round-1 design (all-gathered k-mer set and corrected reads)."""
    seq, offs, n = synth.make_reads(genome_len=12000, coverage=30, read_len=150, seed=5, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    rng = np.random.default_rng(P)
    comm = sharded.LocalSimComm(P)
    out = sharded.sharded_create(comm, {r: ctx for r in range(P)}, _shard_inputs(reads, P, rng), mode=mode)
    ss, st = out[0]
    assert st["mode"] == mode
    ref, rst = ctx.create(reads)
    assert st["n_records"] == rst["n_records"]  # the sharded cover rule emits exactly the single-GPU records
    assert st["n_kept_kmers"] == rst["n_kept_kmers"] and st["n_distinct_kmers"] == rst["n_distinct_kmers"]
    assert st["n_kmers"] == rst["n_kmers"]
    assert st["n_corrected_reads"] == rst["n_corrected_reads"] and st["n_corrected_bases"] == rst["n_corrected_bases"]
    assert st["n_entries"] == rst["n_entries"] == ss.size()
    want = ref.parts()
    got = ss.parts()
    for name in bg.PART_NAMES:
        assert got[name] == want[name], name
    oss, oent, info = O.build_seqset_from_fastq_reads(reads)
    assert ss.size() == len(oent)
    oparts = oss.parts()
    for name in bg.PART_NAMES:
        assert got[name] == oparts[name], name
    assert ss.entries(0, 500) == oent.strings()[:500]  # FM navigation over the assembled arrays


@pytest.mark.parametrize("mode", ["routed", "replicated"])
@pytest.mark.parametrize("alphabet,P", [("AC", 3), ("A", 2), ("ACGT", 5)])
def test_sharded_low_entropy_and_short_entries(ctx, alphabet, P, mode):
    """Low-entropy genomes put most records in few leading-base bins (empty shards, cross-shard prefix
    drops, short tails that resolve on a later shard)."""
    reads = util.synth_reads(3, genome_len=900, n_reads=300, read_len=60, err=0.0, n_rate=0.0, alphabet=alphabet,
                             var_len=True, repeat=(5, 30))
    reads += ["ACG", "A", "T", "ACGTA", "GG"]
    kk = 14 if mode == "routed" else 12  # the routed path needs a 12-base minimiser inside the k-mer
    prm = ctx.default_params(kmer_size=kk, min_kmer_count=1)
    rng = np.random.default_rng(1)
    comm = sharded.LocalSimComm(P)
    out = sharded.sharded_create(comm, {r: ctx for r in range(P)}, _shard_inputs(reads, P, rng), prm, mode=mode)
    ss, st = out[0]
    assert st["mode"] == mode
    ref, rst = ctx.create(reads, kmer_size=kk, min_kmer_count=1)
    assert ss.size() == ref.size()
    assert ss.parts() == ref.parts()
    assert len(st["shard_entries"]) == P and sum(st["shard_entries"]) == ss.size()


@pytest.mark.parametrize("env", [{"BG_KMER_SKM": "0"}, {"BG_KMER_LDS_LOG2CAP": "3"}, {"BG_KMER_EXTRA_BITS": "2"},
                                 {"BG_SCREEN_LOOKUP": "1", "BG_SHARD_REPLICATED": "1"},
                                 {"BG_KMER_LDS_LOG2CAP": "3", "BG_SHARD_REPLICATED": "1"},
                                 {"BG_COVER_ROUNDS": "1,2,3,256"}, {"BG_COVER_ROUNDS": "1,256"}])
def test_sharded_kmer_exchange_modes(ctx, monkeypatch, env):
    """The k-mer exchange as instances (BG_KMER_SKM=0) and as super-k-mers, with the owners' counting tables
    overflowing (every record comes back "unknown" and is looked up by its sender), with extra partition
    levels, and with the returned presence bits ignored: same seqset as the single-GPU build."""
    for a, b in env.items():
        monkeypatch.setenv(a, b)
    P = 3
    seq, offs, n = synth.make_reads(genome_len=9000, coverage=25, read_len=120, seed=11, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    prm = ctx.default_params(kmer_size=22, min_kmer_count=3)
    out = sharded.sharded_create(sharded.LocalSimComm(P), {r: ctx for r in range(P)},
                                 _shard_inputs(reads, P, np.random.default_rng(2)), prm)
    ss, st = out[0]
    monkeypatch.undo()
    ref, rst = ctx.create(reads, kmer_size=22, min_kmer_count=3)
    for key in ("n_kept_kmers", "n_distinct_kmers", "n_kmers", "n_corrected_reads", "n_corrected_bases", "n_entries"):
        assert st[key] == rst[key], key
    assert ss.parts() == ref.parts()


@pytest.mark.parametrize("every", [1, 3, 1000000])
def test_sharded_splitter_sample_step(ctx, monkeypatch, every):
    """The records exchange chooses its splitters from a sampled prefix histogram; any sampling step (all records,
    every third, a single record per rank: hence empty shards) gives the same seqset and the same record total."""
    monkeypatch.setattr(sharded, "PREFIX_SAMPLE", every)
    P = 4
    seq, offs, n = synth.make_reads(genome_len=7000, coverage=25, read_len=110, seed=13, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    prm = ctx.default_params(kmer_size=20, min_kmer_count=3)
    out = sharded.sharded_create(sharded.LocalSimComm(P), {r: ctx for r in range(P)},
                                 _shard_inputs(reads, P, np.random.default_rng(3)), prm)
    ss, st = out[0]
    ref, rst = ctx.create(reads, kmer_size=20, min_kmer_count=3)
    assert st["n_records"] == rst["n_records"] and st["n_entries"] == rst["n_entries"]
    assert ss.parts() == ref.parts()


def test_sharded_truncation_narrows_the_word_stride(ctx):
    """Every read ends in NNNN: the corrected reads are 96 bases (3 words) in objects that keep the input's 4-word
    stride — the corrected-read exchange must narrow them (found by tests/fuzz_parity.py)."""
    rng = np.random.default_rng(4)
    g = "".join(rng.choice(list("ACGT"), size=3000))
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    for _ in range(1200):
        s = int(rng.integers(0, len(g) - 96))
        r = g[s:s + 96]
        if rng.random() < 0.5:
            r = r[::-1].translate(comp)
        reads.append(r + "NNNN")
    prm = ctx.default_params(kmer_size=20, min_kmer_count=2)
    P = 3
    out = sharded.sharded_create(sharded.LocalSimComm(P), {r: ctx for r in range(P)},
                                 _shard_inputs(reads, P, np.random.default_rng(3)), prm)
    ss, st = out[0]
    ref, rst = ctx.create(reads, kmer_size=20, min_kmer_count=2)
    assert rst["max_len"] == 96 and st["n_entries"] == rst["n_entries"] == ss.size()
    assert ss.parts() == ref.parts()


@pytest.mark.parametrize("P,alphabet", [(1, "ACGT"), (2, "ACGT"), (3, "ACGT"), (8, "ACGT"), (5, "AC"), (4, "A")])
def test_sharded_slices_written_in_place_equal_single_gpu(ctx, tmp_path, P, alphabet):
    """gather=False: nothing is collected on one rank.  Every rank re-aligns its entries to 512 with its
    neighbours, finalizes its stretch and writes its bytes of every container member in place; the file opens as
    the seqset of the single-GPU build, member for member (uneven and empty shards, shards of a few entries)."""
    if alphabet == "ACGT":
        seq, offs, n = synth.make_reads(genome_len=9000, coverage=30, read_len=150, seed=17, device="cpu")
        reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
        prm = ctx.default_params()
        kw = {}
    else:
        reads = util.synth_reads(7, genome_len=700, n_reads=260, read_len=60, err=0.0, n_rate=0.0, alphabet=alphabet,
                                 var_len=True, repeat=(5, 30))
        reads += ["ACG", "A", "T", "ACGTA", "GG"]
        prm = ctx.default_params(kmer_size=14, min_kmer_count=1)
        kw = {"kmer_size": 14, "min_kmer_count": 1}
    comm = sharded.LocalSimComm(P)
    out = sharded.sharded_create(comm, {r: ctx for r in range(P)}, _shard_inputs(reads, P, np.random.default_rng(P)), prm,
                                 gather=False)
    slices = {r: out[r][0] for r in range(P)}
    st = out[0][1]
    assert sum(s.size() for s in slices.values()) == st["n_entries"]
    assert all(s.first % 512 == 0 for s in slices.values())
    path = str(tmp_path / "seqset")
    sharded.write_slices(comm, slices, path, ["bgbinary", "create"])
    for s in slices.values():
        s.free()
    got = ctx.open_seqset(path)
    ref, rst = ctx.create(reads, **kw)
    assert got.size() == ref.size() == st["n_entries"]
    want = ref.parts()
    have = got.parts()
    for name in bg.PART_NAMES:
        assert have[name] == want[name], name


def _sim_build_memory(P, genome_len, mode):
    """Per-rank peak of the library's allocator (one Context per simulated rank) and the process-wide peak of
    torch's (inputs + exchange buffers of ALL simulated ranks) for one sharded build."""
    ctxs = {r: bg.Context(0) for r in range(P)}
    inputs = {}
    for r in range(P):
        seq, offs, n = synth.make_reads(genome_len=genome_len, coverage=30, read_len=150, seed=23, device="cuda",
                                        shard=(r, P))
        inputs[r] = (seq, offs, n)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    for c in ctxs.values():
        c.memory(reset_peak=True)
    out = sharded.sharded_create(sharded.LocalSimComm(P), ctxs, inputs, mode=mode, gather=False)
    n_entries = out[0][1]["n_entries"]
    for r in range(P):
        out[r][0].free()
    peaks = [ctxs[r].memory()["peak"] for r in range(P)]
    tpeak = torch.cuda.max_memory_allocated() - base
    for c in ctxs.values():
        c.close()
    del inputs, out
    torch.cuda.empty_cache()
    return max(peaks), tpeak, n_entries


def test_routed_per_rank_memory_does_not_grow_with_the_total():
    """Same reads per rank (1 Mb of genome at 30x each), total doubled (4 ranks -> 8 ranks, simulated): in the routed
    build no per-rank allocation follows the TOTAL number of reads or kept k-mers — the library peak of the
    fullest rank and the exchange buffers per rank stay put; the replicated build (all-gathered k-mer set and
    cover index on every rank) grows, which is what the test would catch."""
    per_rank = 1_000_000
    p4, t4, n4 = _sim_build_memory(4, 4 * per_rank, "routed")
    p8, t8, n8 = _sim_build_memory(8, 8 * per_rank, "routed")
    assert n8 > 1.8 * n4
    assert p8 <= 1.15 * p4, (p4, p8)            # the fullest rank's library allocations
    assert t8 / 8 <= 1.15 * t4 / 4, (t4, t8)    # torch side: inputs + exchange buffers, per rank
    q4, u4, _ = _sim_build_memory(4, 4 * per_rank, "replicated")
    q8, u8, _ = _sim_build_memory(8, 8 * per_rank, "replicated")
    assert q8 > 1.3 * q4, (q4, q8)              # the replicated structures follow the total
