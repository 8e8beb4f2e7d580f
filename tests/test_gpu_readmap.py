"""GPU parity of the readmap build: bg_readmap_build through the C ABI against the oracle's
restatement of make_readmap::create_from_reads, and against the lambda golden's container."""
import hashlib
import json

import numpy as np
import pytest

import biograph_amd as bg
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def _oracle_readmap(corrected, paired):
    """corrected: list of str ('' = dropped).  Returns (oracle Readmap, entries)."""
    kept = [s for s in corrected if s]
    seq, offs = O.pack_strings(kept)
    ent = O.Entries.build(seq, offs, "simple")
    oss = O.Seqset.from_entries(ent)
    if paired:
        records = []
        for i in range(0, len(corrected), 2):
            rec = tuple(s for s in corrected[i:i + 2] if s)
            if rec:
                records.append(rec)
    else:
        records = [(s,) for s in kept]
    return O.Readmap(oss, ent, records, paired), ent


def _compare(rm, orm):
    assert rm.size() == len(orm)
    assert rm.max_read_len == orm.max_read_len
    want = orm.parts()
    got = rm.parts()
    assert sorted(want) == sorted(bg.READMAP_PART_NAMES)
    for name in bg.READMAP_PART_NAMES:
        assert got[name] == want[name], name


@pytest.mark.parametrize("seed,paired,rl,var,glen,n", [(1, True, 100, False, 600, 3000), (2, True, 80, True, 2000, 2000),
                                                        (3, False, 60, True, 1500, 1500), (4, True, 40, False, 120, 4000),
                                                        (5, False, 33, False, 90, 2000)])
def test_readmap_matches_oracle(ctx, seed, paired, rl, var, glen, n):
    """Small genomes with many reads give many identical reads, i.e. big groups of identical rows: the
    deterministic claim ranks (make_readmap.cpp:328-392) are what is under test."""
    reads = util.synth_reads(seed, genome_len=glen, n_reads=n, read_len=rl, err=0.004, n_rate=0.0, var_len=var)
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, 20, 3)
    c, st = ctx.correct_reads(r, ks)
    cseq, coffs = c.download()
    corrected = O.unpack_strings(cseq, coffs)
    assert any(not s for s in corrected) and sum(1 for s in corrected if s) > n // 2
    ss, _ = ctx.build_seqset(c, 20)
    rm = ctx.build_readmap(ss, c, paired)
    orm, ent = _oracle_readmap(corrected, paired)
    assert ss.entries() == ent.strings()
    _compare(rm, orm)


def test_readmap_degenerate(ctx):
    c = ctx.pack_reads([])
    ss, _ = ctx.build_seqset(c, 30)
    rm = ctx.build_readmap(ss, c, True)
    assert rm.size() == 0 and rm.part("read_lengths/elements") == b""
    # a palindromic read is its own reverse complement; identical mates
    reads = ["ACGT" * 10, "ACGT" * 10, "AAAACCCCGGGGTTTT", "", "", "TTTTTTTTTT"]
    c = ctx.pack_reads(reads)
    ss, _ = ctx.build_seqset(c, 30)
    for paired in (True, False):
        rm = ctx.build_readmap(ss, c, paired)
        orm, _ = _oracle_readmap(reads, paired)
        _compare(rm, orm)
    with pytest.raises(bg.BgError):
        ctx.build_readmap(ss, ctx.pack_reads(["ACGT"] * 3), True)  # odd count
    with pytest.raises(bg.BgError) as e:
        ctx.build_readmap(ss, ctx.pack_reads(["GGGGGGGGGGGGGGGGGGGG", "ACGT"]), True)  # not in the seqset
    assert "not found" in str(e.value)


def test_readmap_lambda_golden(ctx, golden_dir, tmp_path):
    """The reads of the lambda golden (decoded from its readmap) through the GPU build: the seqset is the
    golden's, the sparse_multi vectors are byte-equal to the golden file, every part equals the oracle's
    (whose pinning against this golden is in test_oracle_golden.py), and the written container has the
    golden's member list."""
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    g = O.decode_readmap(golden_dir + "/father_lambda/readmap", d)
    flat = []
    for rec in g["records"]:
        flat += [rec[0], rec[1] if len(rec) == 2 else ""]
    c = ctx.pack_reads(flat)
    ss, _ = ctx.build_seqset(c, 30)
    assert ss.size() == d["n"]
    gold_parts = O.read_spiral_members(golden_dir + "/father_lambda/seqset")
    for name in bg.PART_NAMES:
        assert ss.part(name) == gold_parts[name], name
    rm = ctx.build_readmap(ss, c, True)
    assert rm.size() == 97860 and rm.max_read_len == 150
    for sel in ("source_to_mid", "dest_to_mid"):
        for w in ("bits", "subaccum", "accum"):
            name = "read_ids/%s/%s" % (sel, w)
            assert rm.part(name) == g["raw"][name], name
    ent = O.Entries.from_strings(O.decode_entries(d))
    orm = O.Readmap(O.Seqset.from_entries(ent), ent, g["records"], True)
    _compare(rm, orm)
    # container
    ss_path, rm_path = tmp_path / "seqset", tmp_path / "readmap"
    ss.write(ss_path, ["test"])
    sha = rm.write(rm_path, ["bgbinary", "create"])
    assert sha == hashlib.sha1(rm_path.read_bytes()).hexdigest()
    mine = O.read_spiral_members(str(rm_path))
    gold = O.read_spiral_members(golden_dir + "/father_lambda/readmap")
    assert list(mine) == list(gold)
    for name, data in gold.items():
        if name in ("file_info.json", "readmap.json"):
            continue
        if name.endswith(".json"):
            assert json.loads(mine[name]) == json.loads(data), name
        elif name.startswith("read_ids/"):
            assert mine[name] == data, name
        else:
            assert len(mine[name]) == len(data), name
    ss_uuid = json.loads(O.read_spiral_members(str(ss_path))["file_info.json"])["uuid"]
    assert json.loads(mine["readmap.json"]) == {"seqset_uuid": ss_uuid}


def test_readmap_wgs_shape_vs_oracle_and_loop_properties(ctx):
    """30x 150 bp paired reads with errors and N's (the bench workload's shape, 200 kb genome): all parts
    equal the oracle's; and the size-independent properties the bench-size run relies on: mate_loop_ptr is
    a permutation made of 2- and 4-cycles fwd -> rc (-> mate fwd -> mate rc), lengths agree inside each
    (fwd, rc) hop."""
    from biograph_amd import synth
    seq, offs, n = synth.make_reads(genome_len=200000, coverage=30, read_len=150, seed=5, device="cpu")
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, 30, 5)
    c, st = ctx.correct_reads(r, ks)
    ss, _ = ctx.build_seqset(c, 30)
    rm = ctx.build_readmap(ss, c, True)
    cseq, coffs = c.download()
    corrected = O.unpack_strings(cseq, coffs)
    orm, ent = _oracle_readmap(corrected, True)
    _compare(rm, orm)
    nid = rm.size()

    def unpack(blob, bits):
        w = np.frombuffer(blob, np.uint64)
        allbits = np.unpackbits(w.view(np.uint8), bitorder="little")[:nid * bits].reshape(nid, bits)
        return (allbits.astype(np.uint64) << np.arange(bits, dtype=np.uint64)).sum(axis=1)
    ptr = unpack(rm.part("mate_loop_ptr/elements"), int(nid).bit_length()).astype(np.int64)
    rl = unpack(rm.part("read_lengths/elements"), int(rm.max_read_len).bit_length())
    fwd = unpack(rm.part("is_forward/packed_data"), 1)
    assert sorted(ptr.tolist()) == list(range(nid))
    p2 = ptr[ptr]
    p4 = p2[p2]
    assert (p4 == np.arange(nid)).all()
    assert (fwd[ptr] != fwd).all()                       # orientation alternates along a loop
    f = fwd == 1
    assert (rl[ptr[f]] == rl[f]).all()                   # forward row -> its own reverse complement
    assert int((p2 == np.arange(nid)).sum()) + int((p2 != np.arange(nid)).sum()) == nid
    assert (p2 != np.arange(nid)).any() and (p2 == np.arange(nid)).any()   # pairs and orphans both occur


def test_readmap_from_arrays_reproduces_the_built_readmap(ctx):
    """bg_readmap_from_arrays (the pieces of a sharded build put together on one device): the columns of a built
    readmap give back every container part."""
    reads = util.synth_reads(21, genome_len=1500, n_reads=2500, read_len=70, err=0.004, n_rate=0.0, var_len=True)
    r = ctx.pack_reads(reads)
    ks = ctx.kmer_count(r, 20, 3)
    c, _ = ctx.correct_reads(r, ks)
    ss, _ = ctx.build_seqset(c, 20)
    rm = ctx.build_readmap(ss, c, True)
    e, rl, ml, f = rm.arrays()
    assert e.dtype == np.uint64
    again = ctx.readmap_from_arrays(ss.size(), rm.max_read_len, "u", e, rl, ml, f)
    want = rm.parts()
    for name in bg.READMAP_PART_NAMES:
        assert again.part(name) == want[name], name
    # columns that cannot be a readmap are refused
    with pytest.raises(bg.capi.BgError):
        ctx.readmap_from_arrays(ss.size(), rm.max_read_len, "u", e[::-1].copy(), rl, ml, f)
    with pytest.raises(bg.capi.BgError):
        ctx.readmap_from_arrays(int(e.max()), rm.max_read_len, "u", e, rl, ml, f)
    for x in (again, rm, ss, c, ks, r):
        x.free()


def test_readmap_entry_ids_beyond_32_bits_round_trip(ctx, tmp_path):
    """A whole-human seqset has more than 2^32 entries (make_readmap.h:56-90 packs 37-bit entry ids): the readmap of
    entries around and beyond 2^32 is built from its columns, written, opened again and queried -- entry ids, the
    sparse_multi bit vectors and the loops survive."""
    n_ent = (1 << 32) + (1 << 21)
    rng = np.random.default_rng(3)
    ent = np.sort(np.concatenate([rng.integers(0, 1 << 20, 300), rng.integers((1 << 32) - 2000, (1 << 32) + 2000, 600),
                                  rng.integers((1 << 32) + 2000, n_ent, 500), [n_ent - 1, (1 << 32) - 1, 1 << 32]]).astype(np.uint64))
    n = len(ent)
    rl = rng.integers(30, 151, n).astype(np.uint32)
    fwd = (np.arange(n) % 2 == 0).astype(np.uint8)
    loop = np.arange(n, dtype=np.uint32) ^ 1  # unpaired loops: every read points at its reverse complement
    if n & 1:
        loop[-1] = n - 1
    rm = ctx.readmap_from_arrays(n_ent, 150, "beyond-32-bits", ent, rl, loop, fwd)
    path = tmp_path / "big.readmap"
    rm.write(path)
    back = ctx.open_readmap(path)
    e2, rl2, ml2, f2 = back.arrays()
    assert e2.dtype == np.uint64 and (e2 == ent).all() and int(e2.max()) == n_ent - 1 and int((e2 >> 32).sum()) > 0
    assert (rl2 == rl).all() and (ml2 == loop).all() and (f2 == fwd).all()
    want = rm.parts()
    for name in bg.READMAP_PART_NAMES:
        assert back.part(name) == want[name], name
    bits = np.frombuffer(want["read_ids/source_to_mid/bits"], np.uint64)
    assert len(bits) == (n_ent + 63) // 64
    set_pos = np.unique(ent)
    assert int(sum(bin(int(w)).count("1") for w in bits[np.unique(set_pos >> 6).astype(np.int64)])) == len(set_pos)
    acc = np.frombuffer(want["read_ids/source_to_mid/accum"], np.uint64)
    assert int(acc[-1]) == len(set_pos)  # bitcount::finalize: the total in the last accum cell
    back.free()
    rm.free()
