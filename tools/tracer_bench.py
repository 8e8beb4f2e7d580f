#!/usr/bin/env python3
"""Times the push tracer (bg_push_trace, row f.2) on reference chunks of a synthetic sample: seqset +
readmap built on the GPU, ref_map over the reference, then --chunks chunks of --chunk-size bases, forward and reverse.
Prints one JSON object (traces, device rounds, assemblies, planted SNPs recovered among the non-reference assemblies)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=8_000_000)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--chunk-size", type=int, default=1_000_000)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--no-anchor-drop", action="store_true", help="leave the anchor_dropper step out (raw dead ends)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import biograph_amd as bg
    from biograph_amd import synth, tracer
    dev = torch.device("cuda", 0)
    ctx = bg.Context(0)
    G = args.genome_len
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=args.coverage, read_len=args.read_len, seed=args.seed,
                                            device=dev, return_genome=True)
    prm = ctx.default_params()
    r = ctx.pack_reads_device(seq.data_ptr(), offs.data_ptr(), n)
    ks = ctx.kmer_count(r, prm.kmer_size, prm.min_kmer_count)
    cor, _ = ctx.correct_reads(r, ks, prm.correct.max_corrections, prm.correct.min_good_run, prm.correct.trim_after_portion)
    ss, _ = ctx.build_seqset(cor, prm.kmer_size)
    rm = ctx.build_readmap(ss, cor, True)
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.cpu().numpy()].tobytes().decode()
    t0 = time.perf_counter()
    rmap = tracer.reference_ref_map(ss, [ref])
    t_rmap = time.perf_counter() - t0
    opt = tracer.TraceOptions(min_overlap=int(0.7 * args.read_len), anchor_drop=0 if args.no_anchor_drop else 1)
    t0 = time.perf_counter()
    session = tracer.PushTracer(ss, rm, rmap, opt)
    t_open = time.perf_counter() - t0
    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + 1)
    snp = torch.rand(G, generator=g, device=dev) < 0.001
    snp_pos = set(torch.nonzero(snp).flatten().cpu().tolist())
    out = {"genome_len": G, "entries": ss.size(), "reads": len(rm) // 2 if hasattr(rm, "__len__") else None,
           "ref_map_seconds": round(t_rmap, 3), "session_open_seconds": round(t_open, 3), "chunk_size": args.chunk_size, "chunks": []}
    for c in range(args.chunks):
        start = c * args.chunk_size
        limit = min(G, start + args.chunk_size)
        if start >= G:
            break
        row = {"start": start, "limit": limit}
        for rev in (False, True):
            t0 = time.perf_counter()
            asms, st = session.trace([(0, ref)], start=start, limit=limit, rev_comp=rev, with_stats=True)
            dt = time.perf_counter() - t0
            var = [a for a in asms if not a.matches_reference]
            hit = set()
            for a in var:
                if a.right_anchor_len and len(a.seq) == a.right_offset - a.left_offset:
                    refseq = ref[a.left_offset:a.right_offset]
                    for i, (x, y) in enumerate(zip(a.seq, refseq)):
                        if x != y and (a.left_offset + i) in snp_pos:
                            hit.add(a.left_offset + i)
            planted = len([p for p in snp_pos if start + 300 < p < limit - 300])
            row["rev" if rev else "fwd"] = {
                "seconds": round(dt, 2), "traces": st["traces"], "device_rounds": st["device_rounds"],
                "device_queries": st["device_queries"], "search_steps": st["step_count"], "assemblies": len(asms),
                "non_reference": len(var), "anchor_drops": len([a for a in var if not a.right_anchor_len]),
                "planted_snps_in_chunk": planted, "planted_snps_in_assemblies": len(hit),
                "traces_per_s": round(st["traces"] / dt, 1), "ref_bases_per_s": round((limit - start) / dt, 1)}
        out["chunks"].append(row)
    session.close()
    rmap.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
