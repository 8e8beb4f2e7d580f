"""GPU parity of the frontier walks the graph_discover drivers are built on
against the oracle's literal restatements, on the reference's lambda golden seqset + reference."""
import numpy as np
import pytest

import biograph_amd as bg
from oracle import oracle as O

pytestmark = pytest.mark.gpu

CODE = {c: i for i, c in enumerate("ACGT")}


@pytest.fixture(scope="module")
def world(golden_dir):
    ctx = bg.Context(0)
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    oss = O.Seqset.from_arrays(d["sizes"], d["shared"], d["bits"])
    ss = ctx.seqset_from_arrays(d["sizes"], d["shared"], d["bits"])
    fa = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">"))
    ref = np.array([CODE[c] for c in fa.upper()], np.uint8)
    yield ctx, d, oss, ss, ref
    ctx.close()


def _frontiers_from_ref(oss, ref, rng, count, width):
    """Frontiers the drivers really see: the ranges reached by walking `width` reference windows."""
    fr = []
    for _ in range(count):
        k = int(rng.integers(1, 6))
        cur = set()
        for _ in range(k):
            p = int(rng.integers(0, len(ref) - width))
            r = (0, len(oss), 0)
            for j in range(width):
                b, e, s = oss.push_front_drop(np.array([r[0]], np.uint64), np.array([r[1]], np.uint64),
                                              np.array([r[2]], np.uint32), np.array([3 - ref[p + j]], np.uint8),
                                              np.array([0], np.uint32))
                r = (int(b[0]), int(e[0]), int(s[0]))
            cur.add(r)
        fr.append(sorted(cur))
    return fr


def _without_packing(call):
    """The same call with a wave an assembly whatever its starting set (BG_WALK_PACK=0: the kernels of rounds 1-5)."""
    import os
    os.environ["BG_WALK_PACK"] = "0"
    try:
        return call()
    finally:
        del os.environ["BG_WALK_PACK"]


def test_packed_walks_equal_a_wave_an_assembly_on_many_traces(world):
    """Thousands of single-range traces next to a few wide ones (sixteen a wave / a lane each against a wave each): branch
    trace, propagate and push-to-pair trace, long enough for seen lists to fill and for repeats to end traces."""
    ctx, d, oss, ss, ref = world
    rng = np.random.default_rng(99)
    n = len(oss)
    starts = rng.integers(0, len(ref) - 1300, 3000)
    seqs = [ref[int(p):int(p) + int(rng.integers(200, 1200))] for p in starts]
    fr = [[(0, n, 0)] for _ in seqs]
    wide = _frontiers_from_ref(oss, ref, rng, 12, 50)
    for k, w in zip(range(5, 3000, 250), wide):
        fr[k] = w
    assert sum(1 for f in fr if len(f) > 1) >= 3
    a = ss.branch_trace_batch(seqs, fr, 40, ev_cap=512)
    b = _without_packing(lambda: ss.branch_trace_batch(seqs, fr, 40, ev_cap=512))
    assert a == b
    assert sum(nev for nev, _, _ in a) > 100
    rc = [(3 - s[::-1]).astype(np.uint8) for s in seqs]
    pa = ss.propagate_from_end_batch(rc, fr)
    pb = _without_packing(lambda: ss.propagate_from_end_batch(rc, fr))
    assert pa == pb
    ends = [list(ent[0]) if ent.get(0) else [] for ent, _ in pa]
    assert sum(1 for e in ends if len(e) == 1) > 2000
    for seen_cap, max_steps in ((256, 400), (24, 400), (300, 120)):
        ta = ss.push_to_pair_trace_batch(ends, 50, max_steps=max_steps, seen_cap=seen_cap)
        tb = _without_packing(lambda: ss.push_to_pair_trace_batch(ends, 50, max_steps=max_steps, seen_cap=seen_cap))
        assert ta == tb
        reasons = {}
        for why, _, _ in ta:
            reasons[why] = reasons.get(why, 0) + 1
        assert len(reasons) >= 2, reasons


def _with_env(name, value, call):
    import os
    os.environ[name] = value
    try:
        return call()
    finally:
        del os.environ[name]


def test_walk_blocks_equal_the_plain_walk(world):
    """The walk blocks (one line per 40 entries: ranks, sizes, shared) against rank4 / sizes / shared (BG_WALK_BLOCKS=0): random
    push_front_drop / push_front / pop_front calls, the reference map, chains from different histories."""
    from biograph_amd import capi
    ctx, d, oss, ss, ref = world
    rng = np.random.default_rng(123)
    n = len(oss)
    m = 200000
    begin = rng.integers(0, n, m).astype(np.uint64)
    width = np.where(rng.random(m) < 0.7, 1, rng.integers(1, 400, m)).astype(np.uint64)
    end = np.minimum(begin + width, np.uint64(n)).astype(np.uint64)
    begin[:50], end[:50] = 0, n  # the whole range, and the last block
    begin[50:100] = n - 1
    end[50:100] = n
    size = np.minimum(d["sizes"][begin.astype(np.int64)].astype(np.uint32), rng.integers(0, 151, m).astype(np.uint32))
    size[:50] = 0
    base = rng.integers(0, 4, m).astype(np.uint8)
    mc = np.where(rng.random(m) < 0.5, 0, rng.integers(0, 120, m)).astype(np.uint32)
    calls = {
        "push_front_drop": lambda: ss.push_front_drop_batch(begin, end, size, base, mc),
        "push_front": lambda: ss.push_front_batch(begin, end, size, base),
        "pop_front": lambda: ss.pop_front_batch(begin, end, np.maximum(size, 1).astype(np.uint32)),
    }
    for name, call in calls.items():
        a = call()
        b = _with_env("BG_WALK_BLOCKS", "0", call)
        for x, y in zip(a, b):
            assert (np.asarray(x) == np.asarray(y)).all(), name
    want = oss.push_front_drop(begin[:20000], end[:20000], size[:20000], base[:20000], mc[:20000])
    got = ss.push_front_drop_batch(begin[:20000], end[:20000], size[:20000], base[:20000], mc[:20000])
    for g, w in zip(got, want):
        assert (g == w).all()
    # the reference map: chains of every length through the four ways
    for chain in (0, 300, 1000):
        a = ss.ref_map_build(ref, [0], [len(ref)], chain)
        assert (_with_env("BG_WALK_BLOCKS", "0", lambda: ss.ref_map_build(ref, [0], [len(ref)], chain)) == a).all(), chain
    # chains that start from different histories (bg_walk_chains): every range after every base
    nch, length = 64, 600
    start = np.zeros((nch, 3), np.uint64)
    start[:, 1] = n
    for i in range(1, nch):
        b, e, s_ = np.array([0], np.uint64), np.array([n], np.uint64), np.array([0], np.uint32)
        p0 = int(rng.integers(0, len(ref) - 200))
        for c in ref[p0:p0 + int(rng.integers(1, 160))]:
            b, e, s_ = ss.push_front_drop_batch(b, e, s_, np.array([3 - c], np.uint8), np.array([0], np.uint32))
        start[i] = (b[0], e[0], s_[0])
    codes = np.ascontiguousarray(ref[5000:5000 + length])

    def chains():
        out = np.zeros((length, nch, 3), np.uint64)
        capi.check(capi.lib().bg_walk_chains(ctx.h, ss.h, nch, capi.np_ptr(start), capi.np_ptr(codes), length, capi.np_ptr(out)))
        return out
    a = chains()
    assert (_with_env("BG_WALK_BLOCKS", "0", chains) == a).all()
    b, e, s_ = start[:, 0].copy(), start[:, 1].copy(), start[:, 2].astype(np.uint32)
    for t in range(0, 200):
        b, e, s_ = ss.push_front_drop_batch(b, e, s_, np.full(nch, 3 - codes[t], np.uint8), np.zeros(nch, np.uint32))
        assert (a[t, :, 0] == b).all() and (a[t, :, 1] == e).all() and (a[t, :, 2] == s_).all(), t


def test_ref_map_build_matches_reference_chunking(world):
    ctx, d, oss, ss, ref = world
    want = oss.ref_map_build(ref, [0], [len(ref)])
    for chain in (0, 300, 1000, 25600):
        got = ss.ref_map_build(ref, [0], [len(ref)], chain)
        assert (got == want).all(), chain
    assert (want & 63).max() >= 1 and ((want & 0x80) != 0).any() and ((want & 0x40) != 0).any()
    # two extents (an N gap in the middle) and a tiny one
    es, ez = [0, 20000, 48000], [15000, 25000, 100]
    assert (ss.ref_map_build(ref, es, ez, 2048) == oss.ref_map_build(ref, es, ez)).all()


def test_trace_one_base_matches_oracle(world):
    ctx, d, oss, ss, ref = world
    rng = np.random.default_rng(4)
    fr = _frontiers_from_ref(oss, ref, rng, 300, 60)
    fr += [[(0, len(oss), 0)], [], [(5, 6, int(d["sizes"][5]))]]
    for rounds in range(5):
        min_ctx = [int(x) for x in rng.integers(0, 80, len(fr))]
        got = ss.trace_one_base_batch(fr, min_ctx)
        nxt = []
        for f, mc, g in zip(fr, min_ctx, got):
            best, cands, n2 = oss.trace_one_base(f, mc)
            assert g[0] == best
            assert g[1] == cands
            assert g[2] == n2
            nxt.append(n2 if n2 else f)
        fr = nxt


def test_branch_trace_matches_oracle(world):
    ctx, d, oss, ss, ref = world
    rng = np.random.default_rng(8)
    seqs, fronts = [], []
    for _ in range(120):
        p = int(rng.integers(200, len(ref) - 500))
        L = int(rng.integers(1, 300))
        # frontier = walk of the 150 bases before the assembly (what rc_seqset_entries.ends() holds)
        r = (0, len(oss), 0)
        for j in range(150):
            b, e, s = oss.push_front_drop(np.array([r[0]], np.uint64), np.array([r[1]], np.uint64),
                                          np.array([r[2]], np.uint32), np.array([3 - ref[p - 150 + j]], np.uint8),
                                          np.array([0], np.uint32))
            r = (int(b[0]), int(e[0]), int(s[0]))
        seqs.append(ref[p:p + L].copy())
        fronts.append([r])
    for mo in (20, 70, 105):
        got = ss.branch_trace_batch(seqs, fronts, mo, ev_cap=512)
        for s_, f, g in zip(seqs, fronts, got):
            n, ev, cur = oss.branch_trace(s_, f, mo)
            assert g[0] == n and g[1] == ev and g[2] == cur
        assert _without_packing(lambda: ss.branch_trace_batch(seqs, fronts, mo, ev_cap=512)) == got
    assert any(g[0] > 0 for g in got)


def test_push_to_pair_trace_matches_oracle(world):
    """Extension traces (the loop of push_to_pair_discover::on_trace) from rc frontiers of the lambda
    reference — a trace from there re-walks the sample's haplotype — incl. the bounded-buffer stop
    reasons and resuming from the last recorded frontier."""
    ctx, d, oss, ss, ref = world
    n = len(oss)
    rng = np.random.default_rng(21)
    fronts = []
    for _ in range(150):
        p = int(rng.integers(200, len(ref) - 200))
        cur = [(0, n, 0)]
        for c in ref[p - 150:p]:   # rc context of ref[p-150:p]: push the complement of the forward bases
            b, e, s = (np.array(x) for x in zip(*cur))
            ob, oe, osz = oss.push_front_drop(b, e, s, np.full(len(cur), 3 - c, np.uint8),
                                              np.zeros(len(cur), np.uint32))
            cur = sorted({(int(x), int(y), int(z)) for x, y, z in zip(ob, oe, osz) if y > x})[:16]
        fronts.append(cur)
    fronts += _frontiers_from_ref(oss, ref, rng, 50, 60)
    fronts.append([])
    fronts.append([(0, n, 0)])
    reasons = set()
    for mo, steps in [(70, 64), (105, 300), (30, 40)]:
        got = ss.push_to_pair_trace_batch(fronts, mo, max_steps=steps)
        for f, g in zip(fronts, got):
            want = oss.push_to_pair_trace(f, mo, steps)
            assert g == want, (f, mo)
            reasons.add(g[0])
        assert _without_packing(lambda: ss.push_to_pair_trace_batch(fronts, mo, max_steps=steps)) == got
    assert {0, 2} <= reasons
    f = fronts[0]
    full = oss.push_to_pair_trace(f, 70, 64)
    assert len(full[1]) > 20
    part = ss.push_to_pair_trace_batch([f], 70, max_steps=64, range_cap=10)[0]
    assert part[0] == 4 and 0 < len(part[1]) < len(full[1])
    rest = ss.push_to_pair_trace_batch([part[2][-1]], 70, max_steps=64 - len(part[1]))[0]
    assert part[1] + rest[1] == full[1] and part[2] + rest[2] == full[2]
