#!/usr/bin/env python3
"""BASELINE.json configs[4]: N samples of one genome (synth.py sample seeds +10...), each built sharded over the ranks,
merged through the records exchange into one container, readmaps built on every rank's replica, discover shared out by
reference chunk (biograph_amd/population.py).  Prints one JSON line with the stage times.

  python tools/population_bench.py --genome-len 8000000                     # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29700 \
      tools/population_bench.py --shape human                               # one rank per GPU over RCCL
  python tools/population_bench.py --sim-ranks 8 --genome-len 2000000       # 8 ranks simulated on one GPU

The discover leg covers --discover-bases of the reference per sample (default: all of it)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ecoli", choices=["tiny", "ecoli", "chr20", "human"])
    ap.add_argument("--genome-len", type=int, default=0)
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--sim-ranks", type=int, default=0, help="simulate this many ranks in one process (LocalSimComm)")
    ap.add_argument("--discover-bases", type=int, default=-1, help="reference bases walked per sample (0 = skip)")
    ap.add_argument("--split-size", type=int, default=1_000_000)
    ap.add_argument("--engine", default="graph", choices=["graph", "push", "vcf"],
                    help="graph = the graph-discover pipeline per chunk; push = the production push tracer (both directions); "
                         "vcf = push tracer + pipeline + coverage + genotype: VCF records per sample")
    ap.add_argument("--out", default="/dev/shm/bg_population_seqset")
    ap.add_argument("--check", action="store_true", help="compare with the single-GPU merge (small inputs only)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import biograph_amd as bg
    from biograph_amd import population, sharded, synth
    G = args.genome_len or synth.SHAPES[args.shape]
    contigs = synth.contig_lengths(G, args.shape) if args.shape == "human" else None
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
        bg.use_shared_arena()
        ctx = bg.Context(local, torch.cuda.current_stream(dev).cuda_stream)
        comm = sharded.TorchDistComm(dev)
        ctxs = {comm.rank: ctx}
    else:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        bg.use_shared_arena()
        P = max(args.sim_ranks, 1)
        comm = sharded.LocalSimComm(P)
        ctx = bg.Context(0, torch.cuda.current_stream(dev).cuda_stream)
        ctxs = {r: ctx for r in range(P)}
    ranks = comm.local_ranks
    world = comm.world
    names = ["sample%d" % i for i in range(args.samples)]

    def clock():
        torch.cuda.synchronize()
        comm.barrier()
        return time.perf_counter()

    t0 = clock()
    inputs, genome, n_bases = {}, None, 0
    for i, name in enumerate(names):
        inputs[name] = {}
        for r in ranks:
            seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=args.coverage, read_len=args.read_len,
                                                    seed=args.seed, device=dev, shard=(r, world), sample=i,
                                                    return_genome=True, contigs=contigs)
            inputs[name][r] = (seq, offs, n)
            n_bases += int(seq.numel())
    t_gen = clock() - t0
    n_bases = sharded._all_sum(comm, {r: n_bases // len(ranks) for r in ranks})

    t0 = clock()
    replicas, readmaps, stats = population.population_create(comm, ctxs, inputs, args.out, timings=True)
    t_create = clock() - t0
    tm = stats["seconds"]
    out = {"config": {"workload": "population: %d samples x %dx of a %d-base genome, %d-base paired reads" %
                      (args.samples, args.coverage, G, args.read_len), "shape": args.shape, "ranks": world,
                      "simulated_ranks": bool(args.sim_ranks), "backend": "rccl" if world_env > 1 else "in-process"},
           "n_bases": n_bases, "generate_seconds": round(t_gen, 2),
           "create_seconds": round(t_create, 3), "stage_seconds": {k: round(v, 3) for k, v in tm.items()},
           "build_merge_gbases_per_s": round(n_bases / max(tm["build"] + tm["merge"] + tm["write"], 1e-9) / 1e9, 3),
           "sample_entries": {n: stats["samples"][n]["n_entries"] for n in names},
           "merged_entries": stats["merge"]["n_entries"], "merged_shards": stats["merge"]["shard_entries"],
           "device_memory": {str(r): ctxs[r].memory() for r in ranks[:1]}}
    del inputs

    if args.check:
        if 0 in ranks:
            singles = []
            for i in range(args.samples):
                # the same pairs the ranks drew: shard (r, world) of sample i, in rank order
                parts = [synth.make_reads(genome_len=G, coverage=args.coverage, read_len=args.read_len, seed=args.seed,
                                          device=dev, shard=(r, world), sample=i, contigs=contigs) for r in range(world)]
                seq = torch.cat([p[0] for p in parts])
                n = sum(p[2] for p in parts)
                offs = torch.arange(0, (n + 1) * args.read_len, args.read_len, dtype=torch.int64, device=dev)
                ss, _ = ctxs[0].create_device(seq.data_ptr(), offs.data_ptr(), n)
                singles.append(ss)
            want, _ = ctxs[0].merge_seqsets(singles, want_maps=False)
            out["equals_single_gpu_merge"] = all(replicas[0].part(p) == want.part(p) for p in bg.PART_NAMES)
            for s in singles:
                s.free()
            want.free()

    if args.discover_bases != 0:
        region = G if args.discover_bases < 0 else min(G, args.discover_bases)
        codes = genome[:region].cpu().numpy()
        ref = np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode()
        scaffolds = {"1": ref}
        if contigs is not None:
            scaffolds, at = {}, 0
            for i, ln in enumerate(contigs):
                if at >= region:
                    break
                scaffolds[str(i + 1)] = ref[at:min(at + ln, region)]
                at += ln
        ref_maps = None
        if args.engine in ("push", "vcf"):
            from biograph_amd import tracer
            whole = [x for name in scaffolds for _, x in population.extents_of(scaffolds[name])]
            ref_maps = {r: tracer.reference_ref_map(replicas[r], whole) for r in ranks}
        t0 = clock()
        if args.engine == "vcf":  # the reference's engine end to end: VCF records per sample
            vcf = population.sharded_discovery_vcf(comm, replicas, readmaps, scaffolds, split_size=args.split_size, ref_maps=ref_maps)
            t_rank = time.perf_counter() - t0
            t_disc = clock() - t0
            n_calls = sum(len(v) for v in vcf.values())
        else:
            mine = population.sharded_discover(comm, replicas, readmaps, scaffolds, min_overlap=int(0.7 * args.read_len),
                                               split_size=args.split_size, gather=False, engine=args.engine, ref_maps=ref_maps)
            t_rank = time.perf_counter() - t0
            t_disc = clock() - t0
            n_calls = sharded._all_sum(comm, {r: sum(len(v) for v in mine[r].values()) for r in ranks})
        out["discover"] = {"engine": args.engine, "reference_bases_per_sample": region, "split_size": args.split_size,
                           "work_items": args.samples * len(population.ref_chunks([(k, 0, len(v)) for k, v in scaffolds.items()],
                                                                                  args.split_size)),
                           "seconds": round(t_disc, 2), "this_rank_busy_seconds": round(t_rank, 2),
                           "anchored_variants": n_calls, "anchored_variants_per_s": round(n_calls / max(t_disc, 1e-9), 1)}
    if 0 in ranks:
        print(json.dumps(out))
        try:
            os.unlink(args.out)
        except OSError:
            pass
    if world_env > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
