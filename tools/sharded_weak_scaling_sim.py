"""Weak-scaling estimate of the sharded build WITHOUT the links: P simulated ranks in one process on one GPU
(LocalSimComm), each holding 1/P of the pairs of a P x G genome.  Per-rank time = total / P; the growth over
P = 1 is work that follows the TOTAL instead of the rank's share (in the replicated mode: the hash set over all kept
k-mers and the cover index over all reads; the routed mode has none).  Usage: python
tools/sharded_weak_scaling_sim.py [G_per_rank] [routed|replicated]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BG_SHARD_TIMING"] = "1"
import torch

import biograph_amd as bg
from biograph_amd import sharded, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
MODE = sys.argv[2] if len(sys.argv) > 2 else "routed"
os.environ.pop("BG_SHARD_TIMING", None)
bg.use_shared_arena()  # as bench.py does for sharded runs: torch's buffers come out of the library's arena
ctx = bg.Context(0, torch.cuda.current_stream().cuda_stream)
base = None
for P in (1, 2, 4, 8):
    inputs = {}
    for r in range(P):
        seq, offs, n = synth.make_reads(genome_len=G * P, coverage=30, read_len=150, seed=7, device="cuda", shard=(r, P))
        inputs[r] = (seq, offs, n)
    ctxs = {r: ctx for r in range(P)}
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        comm = sharded.LocalSimComm(P)
        out = sharded.sharded_create(comm, ctxs, inputs, mode=MODE, gather=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        for r in range(P):
            out[r][0].free()
    per_rank = dt / P
    base = base or per_rank
    off = max(comm.off_rank_bytes) / 1e9
    # link time if nothing overlapped: 7 xGMI links x ~50 GB/s usable each way per GPU
    print("%s P=%d  total %.1f ms  per rank %.1f ms  efficiency (no links) %.2f  | off-rank bytes of the fullest rank %.2f GB"
          " = %.1f ms at 350 GB/s" % (MODE, P, dt * 1e3, per_rank * 1e3, base / per_rank, off, off / 350 * 1e3), flush=True)
    del inputs
    ctx.trim()
