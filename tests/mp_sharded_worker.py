"""Worker of tests/test_gpu_sharded_mp.py: one rank of a real multi-process sharded build.  All ranks
share cuda:0; collectives run over gloo on host copies (TorchDistComm(stage_through_host=True)), or, with
"rccl" as the second argument, over RCCL itself (one rank per GPU)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import biograph_amd as bg  # noqa: E402
from biograph_amd import sharded, synth  # noqa: E402


def main():
    out_path = sys.argv[1]
    rccl = len(sys.argv) > 2 and sys.argv[2] == "rccl"  # the real backend (one rank per GPU: world 1 on a 1-GPU box)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if rccl:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    seq, offs, n = synth.make_reads(genome_len=40000, coverage=30, read_len=150, seed=9, device=dev,
                                    shard=(rank, world))
    ctx = bg.Context(0, torch.cuda.current_stream(dev).cuda_stream)
    comm = sharded.TorchDistComm(dev, stage_through_host=not rccl)
    out = sharded.sharded_create(comm, {rank: ctx}, {rank: (seq, offs, n)})
    ss, st = out[rank]
    # the same build again without the gather: every process writes its slice of one container in place
    # ... and the sample's readmap is built sharded too (routed look-ups, exchanged claims) and put together on rank 0
    out2 = sharded.sharded_create(comm, {rank: ctx}, {rank: (seq, offs, n)}, gather=False, readmap="paired")
    sliced_path = out_path + ".seqset"
    uuid = sharded.write_slices(comm, {rank: out2[rank][0]}, sliced_path, ["mp_sharded_worker"])
    rm = sharded.gather_readmap(comm, {rank: ctx}, {rank: out2[rank][0].readmap_columns}, uuid)[rank]
    out2[rank][0].free()
    # every rank also ships its reads to rank 0 so the single-GPU build of the same sample can be compared
    gathered = [None] * world
    dist.all_gather_object(gathered, (seq.cpu().numpy(), n))
    if rank == 0:
        allseq = np.concatenate([g[0] for g in gathered])
        total = sum(g[1] for g in gathered)
        dseq = torch.from_numpy(allseq).to(dev)
        doffs = torch.arange(0, (total + 1) * 150, 150, dtype=torch.int64, device=dev)
        ref, rst = ctx.create_device(dseq.data_ptr(), doffs.data_ptr(), total)
        same = all(ss.part(p) == ref.part(p) for p in bg.PART_NAMES)
        opened = ctx.open_seqset(sliced_path)
        same_sliced = all(opened.part(p) == ref.part(p) for p in bg.PART_NAMES)
        opened.free()
        # the single-GPU readmap of the same pairs in the same order (make_reads shards by whole pairs, rank after rank)
        r_ = ctx.pack_reads_device(dseq.data_ptr(), doffs.data_ptr(), total)
        prm = ctx.default_params()
        ks = ctx.kmer_count(r_, prm.kmer_size, prm.min_kmer_count)
        c_, _ = ctx.correct_reads(r_, ks)
        ss1, _ = ctx.build_seqset(c_, prm.kmer_size)
        rm1 = ctx.build_readmap(ss1, c_, True)
        same_readmap = rm.size() == rm1.size() and all(rm.part(p) == rm1.part(p) for p in bg.READMAP_PART_NAMES)
        for x in (rm1, ss1, c_, ks, r_, rm):
            x.free()
        json.dump({"same": bool(same), "same_sliced": bool(same_sliced), "same_readmap": bool(same_readmap), "n_entries": ss.size(), "ref_entries": ref.size(), "stats": st,
                   "shards": st["shard_entries"]}, open(out_path, "w"))
        ref.free()
        ss.free()
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
