"""The synthetic sample generator (biograph_amd/synth.py, SURVEY.md 8d recipe): seeded, the same sample on every
rank, planted variants where variant_sites says they are.  CPU only."""
import numpy as np
import torch

from biograph_amd import synth


def _haps(G, seed):
    ref = synth.make_genome(G, seed, torch.device("cpu"))
    g = torch.Generator()
    g.manual_seed(seed + 1)
    a, b = synth.plant_variants(ref, seed, g)
    return ref, a, b


def test_sample_is_seeded_and_identical_across_ranks():
    r0 = synth.make_reads(genome_len=60000, coverage=6, read_len=100, seed=3, device="cpu", shard=(0, 2), return_genome=True)
    r0b = synth.make_reads(genome_len=60000, coverage=6, read_len=100, seed=3, device="cpu", shard=(0, 2), return_genome=True)
    r1 = synth.make_reads(genome_len=60000, coverage=6, read_len=100, seed=3, device="cpu", shard=(1, 2), return_genome=True)
    assert torch.equal(r0[0], r0b[0]) and torch.equal(r0[3], r1[3])  # same reads again; same reference on rank 1
    assert not torch.equal(r0[0], r1[0])  # but its own read pairs
    whole = synth.make_reads(genome_len=60000, coverage=6, read_len=100, seed=3, device="cpu")
    assert r0[2] + r1[2] == whole[2] and r0[2] % 2 == 0 and r1[2] % 2 == 0  # pairs are never split
    assert set(np.unique(whole[0].numpy())) <= set(b"ACGTN")
    assert whole[1].tolist() == list(range(0, (whole[2] + 1) * 100, 100))


def test_planted_variants_match_the_truth_list():
    G, seed = 400000, 11
    ref, a, b = _haps(G, seed)
    sites = synth.variant_sites(G, seed)
    assert len(sites) > 20 and all(2000 <= p < G - 3000 for p, *_ in sites)
    assert all(sites[i][0] < sites[i + 1][0] for i in range(len(sites) - 1))
    for h, hap in ((0, a), (1, b)):
        net = sum((ln if ins else -ln) for p, ln, ins, het, hp in sites if not het or hp == h)
        assert hap.numel() == G + net
    het = sum(1 for s in sites if s[3])
    assert 0.45 < het / len(sites) < 0.85  # two thirds heterozygous
    # before the first site the haplotypes differ from the reference by SNPs only, at about 1/1000
    p0 = sites[0][0]
    da = int((a[:p0] != ref[:p0]).sum())
    db = int((b[:p0] != ref[:p0]).sum())
    both = int(((a[:p0] != ref[:p0]) & (b[:p0] != ref[:p0])).sum())
    assert da + db - both < p0 * 0.003 and (p0 < 3000 or da + db > 0)
    # a deletion site: the haplotype that carries it continues with the reference after the deleted bases
    shift = {0: 0, 1: 0}
    for p, ln, ins, is_het, hp in sites:
        for h, hap in ((0, a), (1, b)):
            mine = (not is_het) or hp == h
            if mine and not ins:
                q = p + shift[h]
                seg_h = hap[q:q + 40]
                seg_r = ref[p + ln:p + ln + 40]
                assert int((seg_h != seg_r).sum()) <= 3  # SNPs may fall in the window
            if mine:
                shift[h] += ln if ins else -ln


def test_small_genomes_get_snps_only():
    ref, a, b = _haps(5000, 2)
    assert a.numel() == b.numel() == 5000 and synth.variant_sites(5000, 2) == []


def test_population_samples_share_the_genome_and_differ_in_variants():
    """4 samples from the same genome, seeds +10..+13."""
    seed, G = 77, 60_000
    outs = [synth.make_reads(G, coverage=2, read_len=100, seed=seed, sample=i, return_genome=True) for i in range(4)]
    base = synth.make_reads(G, coverage=2, read_len=100, seed=seed, return_genome=True)
    for o in outs:
        assert torch.equal(o[3], base[3])  # one reference
    seqs = [bytes(o[0].numpy().tobytes()) for o in outs]
    assert len(set(seqs)) == 4 and bytes(base[0].numpy().tobytes()) not in seqs
    sites = [synth.variant_sites(G, synth.sample_seed(seed, i) - 1) for i in range(4)]
    assert len({tuple(s) for s in sites}) == 4
    again = synth.make_reads(G, coverage=2, read_len=100, seed=seed, sample=2)
    assert torch.equal(again[0], outs[2][0])
