"""Sharded readmap of ONE sample (BASELINE.json configs[3] ends with seqset + readmap, like `biograph create`): every rank
holds some of the pairs, the entries are range-partitioned, and the per-read columns the ranks end up with -- put together
-- are the single-GPU readmap (bg_readmap_build, itself pinned on the oracle and the e_coli golden) part for part.  Ranks
are simulated in one process on one GPU (LocalSimComm); every buffer goes through the C-ABI phases the RCCL path uses."""
import ctypes as C

import numpy as np
import pytest
import torch

import biograph_amd as bg
from biograph_amd import sharded, synth
from oracle import oracle as O
from tests import util
from tests.test_gpu_sharded import _shard_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def _single_gpu(ctx, reads, paired, prm):
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, prm.kmer_size, prm.min_kmer_count)
    c, _ = ctx.correct_reads(r, ks)
    ss, _ = ctx.build_seqset(c, prm.kmer_size)
    rm = ctx.build_readmap(ss, c, paired)
    for x in (r, ks, c):
        x.free()
    return ss, rm


def _pair_shards(reads, P, rng):
    """like _shard_inputs, with the cuts on pair boundaries"""
    n = len(reads) // 2
    cuts = sorted(int(x) * 2 for x in rng.integers(0, n + 1, P - 1)) if P > 1 else []
    bounds = [0] + cuts + [len(reads)]
    inputs = {}
    for r in range(P):
        part = reads[bounds[r]:bounds[r + 1]]
        seq, offs = O.pack_strings(part)
        dseq = torch.from_numpy(seq.copy()).cuda() if len(seq) else torch.zeros(0, dtype=torch.uint8, device="cuda")
        inputs[r] = (dseq, torch.from_numpy(offs.astype(np.int64)).cuda(), len(part))
    return inputs


def _check(ctx, out, comm, P, want_ss, want_rm, gathered):
    cols = out[0][1]["readmap_columns"] if gathered else {r: out[r][0].readmap_columns for r in range(P)}
    assert sum(int(c.entry.shape[0]) for c in cols.values()) == cols[0].n_total == want_rm.size()
    assert cols[0].n_entries == want_ss.size()
    assert [c.first for c in cols.values()] == list(np.cumsum([0] + [int(cols[r].entry.shape[0]) for r in range(P)])[:-1])
    rm = sharded.gather_readmap(comm, {r: ctx for r in range(P)}, cols, "uuid-of-the-seqset")[0]
    try:
        assert rm.max_read_len == want_rm.max_read_len
        want = want_rm.parts()
        for name in bg.READMAP_PART_NAMES:
            assert rm.part(name) == want[name], name
    finally:
        rm.free()


@pytest.mark.parametrize("P", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("paired", [True, False])
def test_sharded_readmap_equals_single_gpu(ctx, P, paired):
    seq, offs, n = synth.make_reads(genome_len=15000, coverage=30, read_len=150, seed=11 + P, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    rng = np.random.default_rng(100 + P)
    comm = sharded.LocalSimComm(P)
    prm = ctx.default_params()
    inputs = _pair_shards(reads, P, rng) if paired else _shard_inputs(reads, P, rng)
    out = sharded.sharded_create(comm, {r: ctx for r in range(P)}, inputs, prm, readmap="paired" if paired else "unpaired")
    ss, rm = _single_gpu(ctx, reads, paired, prm)
    try:
        assert out[0][1]["n_read_ids"] == rm.size()
        _check(ctx, out, comm, P, ss, rm, gathered=True)
    finally:
        out[0][0].free()
        ss.free()
        rm.free()


@pytest.mark.parametrize("P,glen,n,rl", [(3, 300, 3000, 60), (4, 90, 2500, 33), (2, 2000, 1500, 80)])
def test_sharded_readmap_with_big_groups_of_identical_reads(ctx, P, glen, n, rl):
    """Small genomes with many reads: big groups of identical rows whose loops are claimed in read-id order
    (make_readmap.cpp:328-392) across ranks; variable lengths, reads the correction drops, slices written in place."""
    reads = util.synth_reads(7 * P, genome_len=glen, n_reads=n, read_len=rl, err=0.004, n_rate=0.0005, var_len=True)
    if len(reads) % 2:
        reads = reads[:-1]
    rng = np.random.default_rng(P)
    comm = sharded.LocalSimComm(P)
    prm = ctx.default_params(kmer_size=20, min_kmer_count=3)
    out = sharded.sharded_create(comm, {r: ctx for r in range(P)}, _pair_shards(reads, P, rng), prm, gather=False, readmap="paired")
    ss, rm = _single_gpu(ctx, reads, True, prm)
    try:
        _check(ctx, out, comm, P, ss, rm, gathered=False)
    finally:
        for r in range(P):
            out[r][0].free()
        ss.free()
        rm.free()


def test_sharded_create_writes_seqset_and_readmap_that_open_together(ctx, tmp_path):
    """The end of `biograph create` for a sharded sample: slices written in place, the readmap bound to the seqset's uuid,
    both open and the readmap's queries run against the opened seqset."""
    seq, offs, n = synth.make_reads(genome_len=20000, coverage=30, read_len=150, seed=3, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    P = 4
    comm = sharded.LocalSimComm(P)
    out = sharded.sharded_create(comm, {r: ctx for r in range(P)}, _pair_shards(reads, P, np.random.default_rng(1)),
                                 ctx.default_params(), gather=False, readmap="paired")
    slices = {r: out[r][0] for r in range(P)}
    uuid = sharded.write_slices(comm, slices, str(tmp_path / "seqset"), ["test"])
    rm = sharded.gather_readmap(comm, {r: ctx for r in range(P)}, {r: slices[r].readmap_columns for r in range(P)}, uuid)[0]
    rm.write(tmp_path / "readmap")
    rm.free()
    for s in slices.values():
        s.free()
    ss = ctx.open_seqset(tmp_path / "seqset")
    rm = ctx.open_readmap(tmp_path / "readmap")
    buf = C.create_string_buffer(64)
    bg.capi.check(bg.capi.lib().bg_seqset_uuid(ss.h, buf, 64))
    assert buf.value.decode() == uuid
    e, rl, ml, f = rm.arrays()
    assert len(e) == rm.size() > 0 and int(e.max()) < ss.size()
    ids = np.arange(0, rm.size(), max(1, rm.size() // 500), dtype=np.uint32)
    info = ss.read_info_batch(rm, ids)  # BG_EINVAL if the readmap belonged to another seqset
    assert len(info) == len(ids) and all(0 <= x[3][0] < x[3][1] <= ss.size() for x in info)
    rm.free()
    ss.free()
