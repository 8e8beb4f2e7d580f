"""The entry build from 16-byte items (csrc/items_build.hip) on what its ordinary path does not take -- a bucket of more
than 4096 records, runs of hundreds and thousands of equal first words (a tandem repeat under deep coverage) -- against the
oracle, and against the build through rows (BG_ITEMS_BUILD=0), part by part."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GEN = r"""
import numpy as np
def make_reads(genome_len, copies, element, coverage=30, read_len=150, seed=5):
    # a random genome with `copies` exact copies of one `element`-base sequence at scattered places: the records that start
    # inside a copy share their first words and part ways in the flanks
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, genome_len).astype(np.uint8)
    el = rng.integers(0, 4, element).astype(np.uint8)
    step = genome_len // copies
    for c in range(copies):
        at = c * step + int(rng.integers(0, max(1, step - element - 1)))
        g[at:at + element] = el
    n = genome_len * coverage // read_len
    starts = rng.integers(0, genome_len - read_len, n)
    rev = rng.random(n) < 0.5
    lut = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for s_, r_ in zip(starts, rev):
        x = g[s_:s_ + read_len]
        if r_:
            x = 3 - x[::-1]
        out.append(lut[x].tobytes().decode())
    return out
"""

CHILD = GEN + r"""
import json, os, sys, hashlib
sys.path.insert(0, %(root)r)
import biograph_amd as bg
reads = make_reads(%(genome)d, %(copies)d, %(element)d)
ctx = bg.Context(0)
ss, st = ctx.create(reads, kmer_size=30, min_kmer_count=2, max_corrections=8, min_good_run=2, trim_after_portion=0.7)
print(json.dumps({"n": ss.size(), "parts": {k: hashlib.sha1(bytes(v)).hexdigest() for k, v in ss.parts().items()}}))
"""


def _build(mode, genome, copies, element):
    env = dict(os.environ, BG_ITEMS_BUILD=mode, BG_DEBUG_PREV="1", BG_ITEMS_CHECK="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "genome": genome, "copies": copies, "element": element}],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


@pytest.mark.parametrize("genome,copies,element", [(1_200_000, 5000, 150), (400_000, 700, 90)])
def test_repeats_through_items_equal_rows_and_oracle(genome, copies, element):
    import hashlib
    from oracle import oracle as O
    items, err = _build("1", genome, copies, element)
    assert "entries from items: yes" in err
    # the input does reach the parts of the path this test is about
    chk = [l for l in err.splitlines() if l.startswith("items check")][-1]
    assert "out of order 0, LCP notes missing or wrong 0" in chk, chk
    if copies >= 5000:  # (runs of thousands of equal first words: left to the rows machinery; the smaller family stays in the
        assert " 0 stretches" not in chk, chk  # bucket sort's bitonic network and the workgroup run kernel)
    rows, err0 = _build("0", genome, copies, element)
    assert "entries from items: no" in err0
    ns = {}
    exec(GEN, ns)
    reads = ns["make_reads"](genome, copies, element)
    oss, oent, info = O.build_seqset_from_fastq_reads(reads, 30, 2, 8, 2, 0.7)
    want = {k: hashlib.sha1(bytes(v)).hexdigest() for k, v in oss.parts().items() if k in rows["parts"]}  # (the oracle adds seqset.json)
    assert len(want) >= 15
    assert rows["n"] == len(oent) and rows["parts"] == want, ("rows", [k for k in want if rows["parts"].get(k) != want[k]])
    assert items["n"] == len(oent) and items["parts"] == want, ("items", [k for k in want if items["parts"].get(k) != want[k]])


def test_shard_entry_points_on_a_fresh_create():
    """A seqset straight from bg_create holds no key table (the items path makes it on demand): the exported shard entry
    points that read keys -- boundary rows, entry rows, tail rows -- must make it themselves instead of reading a null
    pointer, from two threads at once as well, and agree with the entries' strings."""
    import threading
    import ctypes as C
    import numpy as np
    import biograph_amd as bg
    from biograph_amd import capi
    from tests import util
    reads = util.synth_reads(3, genome_len=4000, n_reads=1500, read_len=100, err=0.003, n_rate=0.0)
    ctx = bg.Context(0)
    ss, _ = ctx.create(reads, kmer_size=30, min_kmer_count=3)
    L = capi.lib()
    n = ss.size()
    assert n > 0
    W = L.bg_seqset_words(ss.h)
    assert W == 4
    first = [np.zeros(W + 1, np.uint64) for _ in range(2)]
    last = [np.zeros(W + 1, np.uint64) for _ in range(2)]
    rc = [None, None]

    def call(i):  # the first keyed access of a fresh seqset from two host threads
        rc[i] = L.bg_seqset_boundary_rows(ctx.h, ss.h, capi.np_ptr(first[i]), capi.np_ptr(last[i]))

    th = [threading.Thread(target=call, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert rc == [0, 0], capi.last_error() if hasattr(capi, "last_error") else rc
    assert (first[0] == first[1]).all() and (last[0] == last[1]).all()
    rows = bg.DeviceArray(ctx, n * (W + 1) * 8)
    capi.check(L.bg_seqset_entry_rows(ctx.h, ss.h, C.c_void_p(rows.ptr)))
    h = rows.to_numpy(np.uint64, n * (W + 1)).reshape(n, W + 1)
    ents = ss.entries()

    def decode(row):
        ln = int(row[W]) & 0xff
        return "".join("ACGT"[(int(row[j // 32]) >> (62 - 2 * (j % 32))) & 3] for j in range(ln))

    for i in (0, 1, n // 2, n - 1):
        assert decode(h[i]) == ents[i]
    assert decode(first[0]) == ents[0] and decode(last[0]) == ents[-1]
    capi.check(L.bg_seqset_tail_rows(ctx.h, ss.h, C.c_void_p(rows.ptr)))
    t = rows.to_numpy(np.uint64, n * (W + 1)).reshape(n, W + 1)
    for i in (0, n // 3, n - 1):
        assert decode(t[i]) == ents[i][1:]
    rows.free()
    ctx.close()
