"""Time and peak memory of the synthetic sample generator at the bench's 1-GPU and 8-GPU genome sizes (rank 0's shard)."""
import time, torch, sys
sys.path.insert(0, '.')
from biograph_amd import synth
torch.cuda.init()
for world in (1, 8):
    torch.cuda.synchronize(); t = time.time()
    seq, offs, n, ref = synth.make_reads(genome_len=64444167 * world, coverage=30, read_len=150, seed=20261002, device="cuda", shard=(0, world), return_genome=True)
    torch.cuda.synchronize()
    print(world, n, seq.numel(), ref.numel(), round(time.time() - t, 2), "s", round(torch.cuda.max_memory_allocated() / 1e9, 1), "GB peak")
    del seq, offs, ref
