"""One timed sharded build with P ranks simulated in this process (LocalSimComm): phase times (BG_SHARD_TIMING)
and kernel-family times.  Usage: python tools/sharded_sim_once.py <genome bases per rank> <P[,P...]>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BG_SHARD_TIMING"] = "1"
import torch
import biograph_amd as bg
from biograph_amd import sharded, synth
G = int(sys.argv[1]); Ps = [int(x) for x in sys.argv[2].split(",")]
ctx = bg.Context(0)
for P in Ps:
    inputs = {}
    for r in range(P):
        seq, offs, n = synth.make_reads(genome_len=G * P, coverage=30, read_len=150, seed=7, device="cuda", shard=(r, P))
        inputs[r] = (seq, offs, n)
    ctxs = {r: ctx for r in range(P)}
    for rep in range(2):
        ctx.set_profiling(rep == 1)
        torch.cuda.synchronize(); t = time.perf_counter()
        out = sharded.sharded_create(sharded.LocalSimComm(P), ctxs, inputs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        out[0][0].free()
    tm = ctx.timings()
    ctx.set_profiling(False)
    print("kernel families (ms, all ranks): " + ", ".join("%s %.1f" % (x["name"], x["ms"]) for x in sorted(tm, key=lambda x: -x["ms"])[:24]))
    print("P=%d total %.1f ms per rank %.1f ms" % (P, dt * 1e3, dt / P * 1e3), flush=True)
    del inputs; ctx.trim()
