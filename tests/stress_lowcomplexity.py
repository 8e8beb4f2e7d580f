"""Build time and oracle parity on low-complexity input: a 2 Mb random genome with a 5 kb poly-A tract, a 5 kb
(AC)n tract and a 3 kb (AAT)n tract, 30x reads, plus one read duplicated 20 000 times (PCR pile-up) and 2 000
all-N reads.  Usage: python tests/stress_lowcomplexity.py [check]   (check = compare a 200 kb version with the oracle)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import biograph_amd as bg


def make(G, seed=1, dup=20000, nn=2000):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, G, dtype=np.uint8)
    t = G // 10
    a = min(5000, G // 40)
    g[t:t + a] = 0
    g[2 * t:2 * t + a] = np.tile([0, 1], a // 2 + 1)[:a]
    g[3 * t:3 * t + 3 * (a // 5)] = np.tile([0, 0, 3], a // 5 + 1)[:3 * (a // 5)]
    n = 30 * G // 150
    starts = rng.integers(0, G - 150, n)
    idx = starts[:, None] + np.arange(150)[None, :]
    r = g[idx]
    flip = rng.random(n) < 0.5
    r[flip] = 3 - r[flip][:, ::-1]
    err = rng.random(r.shape) < 0.002
    r = np.where(err, (r + rng.integers(1, 4, r.shape, dtype=np.uint8)) & 3, r)
    asc = np.frombuffer(b"ACGT", np.uint8)[r]
    asc = np.concatenate([asc, np.repeat(asc[:1], dup, axis=0), np.full((nn, 150), ord("N"), np.uint8)])
    rng.shuffle(asc)
    return asc


def main():
    ctx = bg.Context(0)
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        from oracle import oracle as O
        asc = make(200_000, dup=300, nn=50)
        reads = [bytes(x).decode() for x in asc]
        ss, st = ctx.create(reads)
        oss, oent, info = O.build_seqset_from_fastq_reads(reads)
        ok = ss.size() == len(oent) and all(ss.parts()[k] == oss.parts()[k] for k in bg.PART_NAMES)
        print("oracle parity on the 200 kb version:", ok, st["n_entries"])
        return 0 if ok else 1
    asc = make(2_000_000)
    n = asc.shape[0]
    seq = bg.DeviceArray.from_numpy(ctx, np.ascontiguousarray(asc.reshape(-1)))
    offs = bg.DeviceArray.from_numpy(ctx, np.arange(0, (n + 1) * 150, 150, dtype=np.uint64))
    for rep in range(2):
        ctx.set_profiling(rep == 1)
        t = time.perf_counter()
        ss, st = ctx.create_device(seq.ptr, offs.ptr, n)
        dt = time.perf_counter() - t
        ss.free()
    tm = ctx.timings()
    print("%d reads: %.1f ms (%.2f Gbases/s)" % (n, dt * 1e3, n * 150 / dt / 1e9))
    print({k: st[k] for k in ("n_records", "n_tied_records", "n_entries", "n_kept_kmers", "n_corrected_reads")})
    print(", ".join("%s %.1f" % (x["name"], x["ms"]) for x in sorted(tm, key=lambda x: -x["ms"])[:10]))
    return 0


sys.exit(main())
