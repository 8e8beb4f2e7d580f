#!/usr/bin/env python3
"""The CPU baseline's stages (threaded oracle) on the bench's CPU sample for several thread counts: which count the host
actually runs fastest at (more threads than physical cores / memory channels only add contention)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from biograph_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4_641_652
dseq, doffs, n = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=20261002, device="cpu")
seq = np.ascontiguousarray(dseq.numpy())
offs = doffs.numpy().astype(np.uint64)
for T in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["16", "32", "64", "128", "256"])]:
    if T > O.hardware_threads():
        continue
    st = {}
    t0 = time.perf_counter()
    ks = O.KmerSet.count(seq, offs, 30, 5, threads=T)
    st["kmerization"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    cseq, coffs, keep, _ = O.correct_reads(ks, seq, offs, 8, 2, 0.7, threads=T)
    st["read_correction"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ent = O.Entries.build(cseq, coffs, "cover", 30, threads=T)
    oss = O.Seqset.from_entries(ent, threads=T)
    st["make_seqset"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.Readmap.from_packed(oss, ent, cseq, coffs, True, threads=T)
    st["make_readmap"] = time.perf_counter() - t0
    tot = sum(st.values())
    print("threads %3d: %.2f s = %.4f Gbases/s  %s" % (T, tot, int(offs[-1]) / tot / 1e9, {k: round(v, 2) for k, v in st.items()}), flush=True)
