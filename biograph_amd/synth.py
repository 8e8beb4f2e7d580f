"""Seeded synthetic WGS-shaped reads, generated with torch on any device.

Genome: i.i.d. uniform ACGT with 0.5 % of its length copied as exact 2-5 kb duplications.  Sample: a diploid
copy of it with planted variants (seed + 1): SNPs 1/1000 bp, indels of 1-50 bp 1/10 000 bp, structural
insertions / deletions of 50-1000 bp 1/1 000 000 bp, two thirds of the sites heterozygous (on one random
haplotype), the rest on both.  Reads: paired 2x`read_len`, insert
N(400,50) clipped to [200,800], FR orientation, `coverage`x, substitution error 0.2 %/base, N 0.01 %.
Output = concatenated ASCII (uint8) + offsets (int64 holding the uint64 values), mates adjacent.
"""
import numpy as np
import torch

SHAPES = {
    "tiny": 200_000,
    "ecoli": 4_641_652,
    "chr20": 64_444_167,
    "human": 3_100_000_000,
}

# GRCh38 primary assembly, chr1..22, X, Y (bases): the 24 contigs of the "human" shape are proportional to these
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
          46709983, 50818468, 156040895, 57227415]


def contig_lengths(genome_len, shape):
    """Contigs of a shape: one, except "human" (24, proportional to GRCh38)."""
    if shape != "human":
        return [genome_len]
    tot = float(sum(GRCH38))
    lens = [int(genome_len * x / tot) for x in GRCH38]
    lens[0] += genome_len - sum(lens)
    return lens


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


def variant_sites(G, seed, indel_rate=1e-4, sv_rate=1e-6, het_frac=2.0 / 3.0, rng=None):
    """The planted indels and structural variants of a G-base genome, in position order:
    (position, length, is_insertion, heterozygous, haplotype of a heterozygous site).  Host-side and seeded, so
    a scorer can ask for the truth set without the haplotypes (bench.py)."""
    if rng is None:
        rng = np.random.default_rng(seed + 1)
    if G < 8000:
        return []
    n_indel = int(rng.binomial(G, indel_rate)) if indel_rate > 0 else 0
    n_sv = int(rng.binomial(G, sv_rate)) if sv_rate > 0 else 0
    pos = rng.integers(2000, G - 3000, n_indel + n_sv)
    length = np.concatenate([rng.integers(1, 51, n_indel), rng.integers(50, 1001, n_sv)])
    is_ins = rng.random(n_indel + n_sv) < 0.5
    het = rng.random(n_indel + n_sv) < het_frac
    hap_of = rng.integers(0, 2, n_indel + n_sv)
    order = np.argsort(pos, kind="stable")
    sites, last_end = [], 0
    for i in order:
        p, ln = int(pos[i]), int(length[i])
        if p < last_end + 10:
            continue  # would touch the previous site
        sites.append((p, ln, bool(is_ins[i]), bool(het[i]), int(hap_of[i])))
        last_end = p + (0 if is_ins[i] else ln)
    return sites


def plant_variants(reference, seed, g, snp_rate=0.001, indel_rate=1e-4, sv_rate=1e-6, het_frac=2.0 / 3.0):
    """The two haplotypes of the sample.  Sites are drawn on the host (numpy, seed + 1: the same on every rank),
    the bases on `reference`'s device with generator `g`.  Indels and structural variants do not overlap and keep
    2 kb clear of the ends; genomes under 8 kb get SNPs only."""
    dev = reference.device
    G = int(reference.numel())
    rng = np.random.default_rng(seed + 1)
    haps = []
    # SNPs: site mask, substituted base, zygosity
    snp = torch.rand(G, generator=g, device=dev) < snp_rate
    shift = torch.randint(1, 4, (G,), generator=g, device=dev, dtype=torch.uint8)
    zyg = torch.rand(G, generator=g, device=dev)  # < het_frac: heterozygous, on the haplotype `which` names
    which = torch.rand(G, generator=g, device=dev) < 0.5
    alt = (reference + shift) & 3
    for h in (0, 1):
        on = snp & ((zyg >= het_frac) | (which == bool(h)))
        haps.append(torch.where(on, alt, reference))
    del snp, shift, zyg, which, alt
    if G < 8000 or (indel_rate <= 0 and sv_rate <= 0):
        return haps[0], haps[1]
    sites = variant_sites(G, seed, indel_rate, sv_rate, het_frac, rng)
    ins_total = sum(ln for _, ln, ins, _, _ in sites if ins)
    ins_bases = torch.randint(0, 4, (max(ins_total, 1),), generator=g, device=dev, dtype=torch.uint8)
    out = []
    for h in (0, 1):
        pieces, cur, used = [], 0, 0
        for p, ln, ins, is_het, hp in sites:
            mine = (not is_het) or hp == h
            if ins:
                if mine:
                    pieces.append(haps[h][cur:p])
                    pieces.append(ins_bases[used:used + ln])
                    cur = p
                used += ln  # both haplotypes of a homozygous insertion carry the same bases
            elif mine:
                pieces.append(haps[h][cur:p])
                cur = p + ln
        pieces.append(haps[h][cur:])
        out.append(torch.cat(pieces))
    return out[0], out[1]


def sample_seed(seed, sample):
    """The seed of sample `sample` of the population config: variant_sites(G, sample_seed(seed, i) - 1) is its truth set
    (variant_sites and plant_variants add the 1 themselves)."""
    return seed + 10 + int(sample)


def make_reads(genome_len=SHAPES["tiny"], coverage=30, read_len=150, seed=20261002, device="cpu", err=0.002,
               n_rate=0.0001, snp_rate=0.001, chunk_pairs=1 << 20, shard=(0, 1), return_genome=False, repeats=None,
               indel_rate=1e-4, sv_rate=1e-6, het_frac=2.0 / 3.0, contigs=None, sample=None):
    """Returns (seq uint8 [n_reads*read_len], offs int64 [n_reads+1], n_reads).

    contigs: lengths of the contigs the genome is cut into (fragments do not span a contig end).

    shard=(rank, world): the same sample (genome + haplotypes from `seed`) on every rank, each rank
    drawing its own 1/world of the read pairs (read RNG seeded with seed + 2 + 1000*rank).

    sample=i (population config, SURVEY.md section 8d: "4 samples from the same genome, seeds +10...+13"): the genome
    still comes from `seed`; the sample's variants are drawn with seed + 10 + i and its reads with
    seed + 110 + i + 1000*rank."""
    dev = torch.device(device)
    sseed = seed if sample is None else sample_seed(seed, sample) - 1
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
    g.manual_seed(sseed + 1)
    reference = hap_a
    hap_a, hap_b = plant_variants(reference, sseed, g, snp_rate, indel_rate, sv_rate, het_frac)
    len_a, len_b = int(hap_a.numel()), int(hap_b.numel())
    haps = torch.zeros((2, max(len_a, len_b)), dtype=torch.uint8, device=dev)  # [2, longest haplotype]
    haps[0, :len_a] = hap_a
    haps[1, :len_b] = hap_b
    del hap_a, hap_b
    hap_len = min(len_a, len_b)  # fragments start where both haplotypes have 800 bases ahead

    rank, world = shard
    n_pairs = int(coverage * genome_len // (2 * read_len))
    n_pairs = n_pairs // world + (1 if rank < n_pairs % world else 0)
    n_reads = 2 * n_pairs
    out = torch.empty(n_reads * read_len, dtype=torch.uint8, device=dev)
    letters = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)  # ACGT
    g.manual_seed(sseed + 2 + 1000 * rank + (0 if sample is None else 99))
    ar = torch.arange(read_len, device=dev)
    done = 0
    while done < n_pairs:
        m = min(chunk_pairs, n_pairs - done)
        insert = (torch.randn(m, generator=g, device=dev) * 50 + 400).clamp(200, 800).long()
        insert = torch.maximum(insert, torch.full_like(insert, read_len))
        # integer draw: float32 uniforms have 24 bits, which would quantise the starts of a >16 Mb genome
        start = torch.randint(0, max(hap_len - 800, 1), (m,), generator=g, device=dev)
        if contigs is not None and len(contigs) > 1:  # a fragment that would cross a contig end is moved inside
            ends = torch.cumsum(torch.tensor(contigs, dtype=torch.int64, device=dev), 0)
            cid = torch.searchsorted(ends, start, right=True).clamp_(max=len(contigs) - 1)
            start = torch.minimum(start, (ends[cid] - 800).clamp_(min=0))
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
        return out, offs, n_reads, reference
    return out, offs, n_reads
