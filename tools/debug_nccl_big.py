import os, sys, torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for lg in (26, 27, 28, 29, 30):
    n = 1 << lg
    for width in (1, 6):
        rows = n // width
        x = torch.arange(rows * width, dtype=torch.int64, device=dev).reshape(rows, width)
        out = torch.empty_like(x)
        dist.all_to_all_single(out, x, output_split_sizes=[rows], input_split_sizes=[rows])
        torch.cuda.synchronize()
        ok = bool(torch.equal(out, x))
        bad = int((out != x).sum().item()) if not ok else 0
        print("a2a unequal-api lg", lg, "width", width, "bytes", x.numel() * 8, "ok", ok, "bad", bad, flush=True)
        del out
        outs = [torch.empty_like(x)]
        dist.all_gather(outs, x)
        torch.cuda.synchronize()
        print("   all_gather ok", bool(torch.equal(outs[0], x)), flush=True)
        del x, outs
dist.destroy_process_group()
