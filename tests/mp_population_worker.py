"""Worker of tests/test_gpu_sharded_mp.py::test_multiprocess_population: one rank of the population configuration
(2 samples built sharded, sharded merge, readmaps on the merged replica, chunk-sharded discover) as separate
processes.  All ranks share cuda:0; collectives over gloo on host copies, or RCCL with "rccl" as second argument."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import biograph_amd as bg  # noqa: E402
from biograph_amd import population, sharded, synth  # noqa: E402

G, SEED, SAMPLES = 30000, 11, 2


def main():
    out_path = sys.argv[1]
    rccl = len(sys.argv) > 2 and sys.argv[2] == "rccl"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev) if rccl else dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = bg.Context(0, torch.cuda.current_stream(dev).cuda_stream)
    comm = sharded.TorchDistComm(dev, stage_through_host=not rccl)
    inputs, mine = {}, {}
    genome = None
    for s in range(SAMPLES):
        seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=SEED, device=dev,
                                                shard=(rank, world), sample=s, return_genome=True)
        inputs["s%d" % s] = {rank: (seq, offs, n)}
        mine["s%d" % s] = (seq.cpu().numpy(), n)
    path = out_path + ".merged_seqset"
    replicas, readmaps, stats = population.population_create(comm, {rank: ctx}, inputs, path)
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.cpu().numpy()].tobytes().decode()
    calls = population.sharded_discover(comm, replicas, readmaps, {"1": ref}, min_overlap=105, split_size=4000)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        prm = ctx.default_params()
        singles = []
        for s in range(SAMPLES):
            name = "s%d" % s
            allseq = np.concatenate([g[name][0] for g in gathered])
            total = sum(g[name][1] for g in gathered)
            dseq = torch.from_numpy(allseq).to(dev)
            doffs = torch.arange(0, (total + 1) * 150, 150, dtype=torch.int64, device=dev)
            r = ctx.pack_reads_device(dseq.data_ptr(), doffs.data_ptr(), total)
            ks = ctx.kmer_count(r, prm.kmer_size, prm.min_kmer_count)
            cor, _ = ctx.correct_reads(r, ks, prm.correct.max_corrections, prm.correct.min_good_run,
                                       prm.correct.trim_after_portion)
            ss, _ = ctx.build_seqset(cor, prm.kmer_size)
            singles.append((ss, ctx.build_readmap(ss, cor, True)))
        want, maps = ctx.merge_seqsets([x[0] for x in singles])
        same_seqset = all(replicas[rank].part(p) == want.part(p) for p in bg.PART_NAMES)
        same_readmaps = True
        one_rank = {}
        for s in range(SAMPLES):
            mig = ctx.migrate_readmap(singles[s][1], maps[s], want)
            got = readmaps[rank]["s%d" % s].parts()
            for part, blob in mig.parts().items():
                if got[part] != blob:
                    same_readmaps = False
                    print("sample", s, "readmap member differs:", part, len(got[part]), len(blob), file=sys.stderr)
            one_rank["s%d" % s] = mig
        alone = population.sharded_discover(sharded.LocalSimComm(1), {0: want}, {0: one_rank}, {"1": ref}, min_overlap=105,
                                            split_size=4000)
        json.dump({"same_seqset": bool(same_seqset), "same_readmaps": bool(same_readmaps),
                   "same_calls": alone == calls, "n_calls": {k: len(v) for k, v in calls.items()},
                   "n_entries": stats["merge"]["n_entries"], "want_entries": want.size(),
                   "shards": stats["merge"]["shard_entries"]}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
