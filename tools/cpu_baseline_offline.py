#!/usr/bin/env python3
"""The CPU restatement (oracle/, every hardware thread) on the BENCH's own sample, once, offline: `python tools/cpu_baseline_offline.py
[genome_len] > profiles/rNN_cpu_baseline_chr20.json` -- bench.py's cpu_baseline leg on the chr20-shaped reads of the timed step
(~2.5 min of CPU work; the leg inside the default bench run uses an E. coli-sized sample to stay within a few minutes)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import biograph_amd as bg
from biograph_amd import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else synth.SHAPES["chr20"]
ctx = bg.Context(0)
out = bench.cpu_baseline(ctx, G, 30, 150, 20261002, torch.device("cuda", 0))
out["note"] = "the bench's own workload (configs[2]); bench.py's in-run leg uses the smaller default sample"
print(json.dumps(out, indent=1))
ctx.close()
