"""Build time on a repeat-heavy genome: a 300-base element copied exactly `copies` times into a random genome (an
Alu-like family without divergence: every 16-base prefix of the element heads a tie group of copies x coverage
records, every k-mer of it has copies x coverage occurrences).  Usage: python tools/stress_repeats.py [genome] [copies] [divergence per copy]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import biograph_amd as bg
from biograph_amd import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
div = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
seq, offs, n = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=3, device="cuda", repeats=(300, copies, div))
ctx = bg.Context(0)
prm = ctx.default_params()
for rep in range(2):
    ctx.set_profiling(rep == 1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    ss, st = ctx.create_device(seq.data_ptr(), offs.data_ptr(), n, prm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ss.free()
tm = ctx.timings()
print("genome %d, %d copies of a 300-base element, divergence %.2f (%.1f %% of the genome): %.1f ms, %.2f Gbases/s" % (
    G, copies, div, 100.0 * copies * 300 / G, dt * 1e3, st["n_bases"] / dt / 1e9))
print({k: st[k] for k in ("n_records", "n_tied_records", "n_entries", "n_kept_kmers")})
print(", ".join("%s %.1f" % (x["name"], x["ms"]) for x in sorted(tm, key=lambda x: -x["ms"])[:12]))
