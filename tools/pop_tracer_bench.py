#!/usr/bin/env python3
"""Times one work item of trace_ref with the pop tracer (PushTracer.execute_work: push forward, push reverse, pop forward, pop
reverse) on 1 Mb pieces of a synthetic 30x sample; BG_TRACER_TIMING=1 prints the library's own split per call."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=4_000_000)
    ap.add_argument("--chunk-size", type=int, default=1_000_000)
    ap.add_argument("--chunks", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    import biograph_amd as bg
    from biograph_amd import synth, tracer
    dev = torch.device("cuda", 0)
    ctx = bg.Context(0)
    G = args.genome_len
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=20261002, device=dev, return_genome=True)
    prm = ctx.default_params()
    r = ctx.pack_reads_device(seq.data_ptr(), offs.data_ptr(), n)
    ks = ctx.kmer_count(r, prm.kmer_size, prm.min_kmer_count)
    cor, _ = ctx.correct_reads(r, ks, prm.correct.max_corrections, prm.correct.min_good_run, prm.correct.trim_after_portion)
    ss, _ = ctx.build_seqset(cor, prm.kmer_size)
    rm = ctx.build_readmap(ss, cor, True)
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.cpu().numpy()].tobytes().decode()
    rmap = tracer.reference_ref_map(ss, [ref])
    session = tracer.PushTracer(ss, rm, rmap, tracer.TraceOptions(min_overlap=105))
    rows = []
    for c in range(args.chunks):
        start, limit = c * args.chunk_size, min(G, (c + 1) * args.chunk_size)
        t0 = time.perf_counter()
        push = session.execute_work([(0, ref)], start=start, limit=limit, pop_fwd=False, pop_rev=False)
        t1 = time.perf_counter()
        both = session.execute_work([(0, ref)], start=start, limit=limit)
        t2 = time.perf_counter()
        rows.append({"start": start, "push_only_s": round(t1 - t0, 3), "push_and_pop_s": round(t2 - t1, 3), "push_assemblies": len(push),
                     "pop_assemblies": sum(1 for a in both if a.from_pop)})
    print(json.dumps({"genome_len": G, "pieces": rows}))
    session.close()


if __name__ == "__main__":
    main()
