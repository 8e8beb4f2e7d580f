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
