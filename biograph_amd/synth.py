"""Seeded synthetic WGS-shaped reads, generated with torch on any device.

Genome: i.i.d. uniform ACGT with 0.5 % of its length copied as exact 2-5 kb duplications.  Sample:
two haplotypes, the second carrying substitutions at 1/1000 bp (SNPs; indels/SVs of the full plan
are not planted yet — DESIGN.md "synthetic input").  Reads: paired 2x`read_len`, insert
N(400,50) clipped to [200,800], FR orientation, `coverage`x, substitution error 0.2 %/base, N 0.01 %.
Output = concatenated ASCII (uint8) + offsets (int64 holding the uint64 values), mates adjacent.
"""
import torch

SHAPES = {
    "tiny": 200_000,
    "ecoli": 4_641_652,
    "chr20": 64_444_167,
}


def make_genome(n, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    genome = torch.randint(0, 4, (n,), generator=g, device=device, dtype=torch.uint8)
    # exact duplications: 0.5 % of the length in 2-5 kb pieces
    cpu = torch.Generator()
    cpu.manual_seed(seed + 7)
    todo = n // 200
    while todo > 2000 and n > 20000:
        ln = int(torch.randint(2000, 5001, (1,), generator=cpu))
        src = int(torch.randint(0, n - ln, (1,), generator=cpu))
        dst = int(torch.randint(0, n - ln, (1,), generator=cpu))
        genome[dst:dst + ln] = genome[src:src + ln].clone()
        todo -= ln
    return genome


def make_reads(genome_len=SHAPES["tiny"], coverage=30, read_len=150, seed=20261002, device="cpu", err=0.002,
               n_rate=0.0001, snp_rate=0.001, chunk_pairs=1 << 20, shard=(0, 1), return_genome=False, repeats=None):
    """Returns (seq uint8 [n_reads*read_len], offs int64 [n_reads+1], n_reads).

    shard=(rank, world): the same sample (genome + haplotypes from `seed`) on every rank, each rank
    drawing its own 1/world of the read pairs (read RNG seeded with seed + 2 + 1000*rank)."""
    dev = torch.device(device)
    hap_a = make_genome(genome_len, seed, dev)
    if repeats:  # (unit_len, copies[, divergence]): a high-copy element (stress input, tools/stress_repeats.py)
        unit_len, copies = repeats[0], repeats[1]
        div = repeats[2] if len(repeats) > 2 else 0.0
        cg = torch.Generator()
        cg.manual_seed(seed + 9)
        unit = torch.randint(0, 4, (unit_len,), generator=cg, dtype=torch.uint8)
        starts = torch.randint(0, genome_len - unit_len, (copies,), generator=cg)
        copies_t = unit[None, :].repeat(copies, 1)
        if div > 0:  # every copy carries its own substitutions
            mut = torch.rand(copies_t.shape, generator=cg) < div
            copies_t = torch.where(mut, (copies_t + torch.randint(1, 4, copies_t.shape, generator=cg, dtype=torch.uint8)) & 3,
                                   copies_t)
        idx = (starts[:, None] + torch.arange(unit_len)[None, :]).reshape(-1).to(dev)
        hap_a[idx] = copies_t.reshape(-1).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1)
    snp = torch.rand(genome_len, generator=g, device=dev) < snp_rate
    shift = torch.randint(1, 4, (genome_len,), generator=g, device=dev, dtype=torch.uint8)
    hap_b = torch.where(snp, (hap_a + shift) & 3, hap_a)
    haps = torch.stack([hap_a, hap_b])  # [2, G]
    del snp, shift

    rank, world = shard
    n_pairs = int(coverage * genome_len // (2 * read_len))
    n_pairs = n_pairs // world + (1 if rank < n_pairs % world else 0)
    n_reads = 2 * n_pairs
    out = torch.empty(n_reads * read_len, dtype=torch.uint8, device=dev)
    letters = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)  # ACGT
    g.manual_seed(seed + 2 + 1000 * rank)
    ar = torch.arange(read_len, device=dev)
    done = 0
    while done < n_pairs:
        m = min(chunk_pairs, n_pairs - done)
        insert = (torch.randn(m, generator=g, device=dev) * 50 + 400).clamp(200, 800).long()
        insert = torch.maximum(insert, torch.full_like(insert, read_len))
        # integer draw: float32 uniforms have 24 bits, which would quantise the starts of a >16 Mb genome
        start = torch.randint(0, max(genome_len - 800, 1), (m,), generator=g, device=dev)
        hap = torch.randint(0, 2, (m,), generator=g, device=dev)
        flip = torch.rand(m, generator=g, device=dev) < 0.5  # which strand the fragment came from
        # mate 1: forward from fragment start; mate 2: reverse complement from fragment end (FR)
        idx1 = start[:, None] + ar[None, :]
        idx2 = (start + insert - 1)[:, None] - ar[None, :]
        b1 = haps[hap[:, None], idx1]
        b2 = 3 - haps[hap[:, None], idx2]
        first = torch.where(flip[:, None], b2, b1)
        second = torch.where(flip[:, None], b1, b2)
        pair = torch.stack([first, second], dim=1).reshape(2 * m, read_len)  # mates adjacent
        e = torch.rand(pair.shape, generator=g, device=dev)
        sub = torch.randint(1, 4, pair.shape, generator=g, device=dev, dtype=torch.uint8)
        pair = torch.where(e < err, (pair + sub) & 3, pair)
        asc = letters[pair.long()]
        asc = torch.where((e >= err) & (e < err + n_rate), torch.full_like(asc, 78), asc)  # 'N'
        out[2 * done * read_len: 2 * (done + m) * read_len] = asc.reshape(-1)
        done += m
    offs = torch.arange(0, (n_reads + 1) * read_len, read_len, dtype=torch.int64, device=dev)
    if return_genome:
        return out, offs, n_reads, hap_a
    return out, offs, n_reads
