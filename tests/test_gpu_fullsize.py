"""The bench workload at its full size (30x chr20-shaped, 1.93 Gbases: BASELINE.json configs[2]) — too big for the
oracle, so the seqset is checked through properties that do not depend on size: the invariants of seqset.cpp /
builder.cpp (sorted, prefix-free, sum of prev popcounts = n, every tail closed), FM navigation (find(entry) lands
on that entry, pop/push round trip), and agreement of two independent code paths (the fused single-GPU build and
the multi-rank build with its exchanges) part for part."""
import hashlib
import os

import numpy as np
import pytest
import torch

import biograph_amd as bg
from biograph_amd import sharded, synth

pytestmark = pytest.mark.gpu
# Two sizes, both in the default `-m gpu` run: the bench workload (BASELINE.json configs[2]) and ONE RANK'S SHARE of
# configs[3] -- a 30x whole human over 8 GPUs = 3.1 Gb / 8 = 387.5 Mb of genome, 11.6 Gbases, 77.5 M reads per rank: past
# 2^32 k-mer windows and base offsets, ~170 GB of HBM at the peak.  BG_FULLSIZE_GENOME=<bases> replaces the first size.
G_CHR20 = int(os.environ.get("BG_FULLSIZE_GENOME", synth.SHAPES["chr20"]))
G_HUMAN_SHARE = synth.SHAPES["human"] // 8


@pytest.fixture(scope="module", params=[G_CHR20, G_HUMAN_SHARE], ids=["chr20", "human_eighth"])
def built(request):
    G = request.param
    os.environ["BG_VERIFY_CLOSURE"] = "1"  # every entry's tail is looked up and must be a prefix of what it finds
    try:
        ctx = bg.Context(0)
        seq, offs, n = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=20261002, device="cuda")
        ss, st = ctx.create_device(seq.data_ptr(), offs.data_ptr(), n, ctx.default_params())
    finally:
        del os.environ["BG_VERIFY_CLOSURE"]
    st = dict(st, genome_len=G)
    yield ctx, ss, st, (seq, offs, n)
    ss.free()
    ctx.close()
    del seq, offs
    torch.cuda.empty_cache()


def chr20_only(st):
    if st["genome_len"] != G_CHR20:
        pytest.skip("run at the bench size only")


def test_fullsize_counters_and_invariants(built):
    ctx, ss, st, (seq, offs, n) = built
    G = st["genome_len"]
    assert st["n_reads"] == n == 2 * (30 * G // 300) and st["n_bases"] == n * 150
    assert st["n_kmers"] == n * (150 - 30 + 1)
    assert 0.95 * n < st["n_corrected_reads"] <= n
    assert G < st["n_kept_kmers"] < 1.05 * G  # the distinct 30-mers of a diploid genome with 1/1000 variants
    N = ss.size()
    assert N == st["n_entries"] and 2 * G < N < 2.4 * G  # both strands
    parts = ss.parts()
    total = 0
    for b in "ACGT":
        bits = np.frombuffer(parts["prev_%s/bits" % b], np.uint8)
        total += int(np.unpackbits(bits).sum(dtype=np.int64))
    assert total == N  # seqset.cpp:123-126: every entry is the pop_front of exactly one (entry, base)
    sizes = np.frombuffer(parts["entry_sizes/elements"], np.uint8)
    shared = np.frombuffer(parts["shared/elements"], np.uint8)
    assert len(sizes) >= N and int(sizes[:N].max()) == 150 and int(sizes[:N].min()) >= 1
    assert shared[0] == 0 and bool((shared[1:N] < sizes[1:N]).all())  # prefix-free: never the whole entry
    assert bool((shared[1:N] <= sizes[:N - 1]).all())


def test_fullsize_sorted_windows_and_navigation(built):
    ctx, ss, st, _ = built
    N = ss.size()
    rng = np.random.default_rng(5)
    for first in [0, N - 4000] + [int(x) for x in rng.integers(0, N - 4000, 6)]:
        ent = ss.entries(first, 4000)
        for a, b in zip(ent, ent[1:]):
            assert a < b and not b.startswith(a)
        b, e = ss.find_batch(ent[::16])
        assert b.tolist() == list(range(first, first + 4000, 16))  # find(entry) begins at the entry itself
        assert bool((e > b).all())
    # pop_front then push_front of the dropped base comes back to a range holding the entry
    first = int(rng.integers(0, N - 2000))
    ent = [s for s in ss.entries(first, 2000) if len(s) > 40]
    tb, te = ss.find_batch([s[1:] for s in ent])
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    nb, ne, nz = ss.push_front_batch(tb, te, np.array([len(s) - 1 for s in ent], np.uint32),
                                     np.array([code[s[0]] for s in ent], np.uint8))
    eb, ee = ss.find_batch(ent)
    assert nb.tolist() == eb.tolist() and ne.tolist() == ee.tolist() and nz.tolist() == [len(s) for s in ent]


def test_fullsize_sharded_build_equals_fused(built):
    """Two ranks (simulated on the one GPU, every buffer through the exchange phases of the C ABI) build the same
    sample: every container part equal to the fused build's."""
    ctx, ss, st, (seq, offs, n) = built
    chr20_only(st)
    want = {k: hashlib.sha256(v).hexdigest() for k, v in ss.parts().items()}
    half = (n // 4) * 2
    cut = half * 150
    inputs = {0: (seq[:cut], offs[:half + 1], half),
              1: (seq[cut:], (offs[half:] - cut).contiguous(), n - half)}
    out = sharded.sharded_create(sharded.LocalSimComm(2), {0: ctx, 1: ctx}, inputs, ctx.default_params())
    ss2, st2 = out[0]
    try:
        for key in ("n_kept_kmers", "n_distinct_kmers", "n_kmers", "n_corrected_reads", "n_corrected_bases",
                    "n_records", "n_entries"):
            assert st2[key] == st[key], key
        got = {k: hashlib.sha256(v).hexdigest() for k, v in ss2.parts().items()}
        assert got == want
    finally:
        ss2.free()


def test_fullsize_eight_rank_routed_build_written_in_slices_equals_fused(built, tmp_path):
    """The shape of BASELINE.json configs[3] -- at chr20 size and at one rank's share of the whole human (11.6 Gbases): EIGHT
    ranks (simulated on the one GPU), uneven shards, the routed build with nothing replicated and nothing gathered --
    every rank writes its slice of one container in place -- and the file opens as the fused build's seqset, member for
    member."""
    ctx, ss, st, (seq, offs, n) = built
    want = {k: hashlib.sha256(v).hexdigest() for k, v in ss.parts().items()}
    P = 8
    rng = np.random.default_rng(8)
    cuts = sorted(int(x) * 2 for x in rng.integers(0, n // 2 + 1, P - 1))
    bounds = [0] + cuts + [n]
    inputs = {}
    for r in range(P):
        a, b = bounds[r], bounds[r + 1]
        inputs[r] = (seq[a * 150:b * 150], (offs[a:b + 1] - a * 150).contiguous(), b - a)
    comm = sharded.LocalSimComm(P)
    out = sharded.sharded_create(comm, {r: ctx for r in range(P)}, inputs, ctx.default_params(), mode="routed", gather=False)
    st8 = out[0][1]
    for key in ("n_kept_kmers", "n_distinct_kmers", "n_kmers", "n_corrected_reads", "n_corrected_bases", "n_records", "n_entries"):
        assert st8[key] == st[key], key
    assert len(st8["shard_entries"]) == P and min(st8["shard_entries"]) > 0.8 * st["n_entries"] / P  # balanced splitters
    path = str(tmp_path / "seqset")
    slices = {r: out[r][0] for r in range(P)}
    sharded.write_slices(comm, slices, path, ["test_gpu_fullsize"])
    for s in slices.values():
        s.free()
    opened = ctx.open_seqset(path)
    try:
        got = {k: hashlib.sha256(v).hexdigest() for k, v in opened.parts().items()}
        assert {k: got[k] for k in bg.PART_NAMES} == {k: want[k] for k in bg.PART_NAMES}
    finally:
        opened.free()


def test_fullsize_native_driver_equals_fused(built, tmp_path):
    """The C++ driver `bgbinary create --gpus N` runs (bg_create_sharded + bg_sharded_write, csrc/sharded.hip over csrc/comm.hip)
    on a rank alone at the bench size and at one rank's share of the whole human: the routed phases, the slice written in place
    -- the container opens as the fused build's seqset, member for member.  (Several real ranks: tests/test_gpu_native_sharded.py,
    2, 3 and 8 processes on a smaller sample.)"""
    ctx, ss, st, (seq, offs, n) = built
    want = {k: hashlib.sha256(v).hexdigest() for k, v in ss.parts().items()}
    comm = sharded.NativeComm(ctx, 0, 1, sharded.NativeComm.unique_id(), "rccl")
    try:
        sh = sharded.native_create(ctx, comm, seq.data_ptr(), offs.data_ptr(), n, ctx.default_params())
        try:
            for key in ("n_kept_kmers", "n_corrected_reads", "n_corrected_bases", "n_records", "n_entries"):
                assert sh.stats[key] == st[key], key
            path = str(tmp_path / "seqset_native")
            sh.write(comm, path, None, ["test_gpu_fullsize"])
        finally:
            sh.free()
    finally:
        comm.free()
    opened = ctx.open_seqset(path)
    try:
        got = {k: hashlib.sha256(v).hexdigest() for k, v in opened.parts().items()}
        assert {k: got[k] for k in bg.PART_NAMES} == {k: want[k] for k in bg.PART_NAMES}
    finally:
        opened.free()


def test_fullsize_discovery_engine_properties():
    """The reference's discovery engine end to end at the bench size (tools/discovery_bench.py: FASTQ -> `bgbinary
    create` -> `bgbinary discovery`, 65 pieces x 2 directions): too big for the oracle, so the VCF is checked against
    what the generator planted -- at least 95 % of the SNP sites are records, no SNP record lies off a planted site, and
    a third of the called sites are homozygous, as planted."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "discovery_bench.py"), "--genome-len", str(synth.SHAPES["chr20"])],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["planted_snps"] > 60000
    assert d["planted_snps_in_vcf"] >= 0.95 * d["planted_snps"]
    assert d["snp_records_not_planted"] == 0
    assert 0.30 < d["planted_snps_called_homozygous"] / d["planted_snps_in_vcf"] < 0.40
    assert d["vcf_records"] > d["snp_records"] > 0  # indels and SVs come out as records too
