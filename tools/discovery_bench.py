#!/usr/bin/env python3
"""`bgbinary create` + `bgbinary discovery --engine push` end to end on a synthetic sample, as a user would run them:
FASTQ files and a reference FASTA in /dev/shm -> S.bg -> VCF.  Prints one JSON object: wall seconds of both commands,
the discovery command's own stage breakdown (refmap / tracing / pipeline / coverage), VCF records, and how many of the
planted SNPs (bench.py's recipe: a site per 1000 bases, a third homozygous) are records with the planted genotype class.
  python tools/discovery_bench.py [--genome-len 8000000] [--coverage 30] [--read-len 150]"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=8_000_000)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    import numpy as np
    import torch
    from biograph_amd import synth
    dev = torch.device("cuda", 0)
    G, L = args.genome_len, args.read_len
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=args.coverage, read_len=L, seed=args.seed, device=dev,
                                            return_genome=True)
    tmp = tempfile.mkdtemp(prefix="bg_disc_", dir="/dev/shm")
    reads = seq.view(n, L)
    paths = []
    for mate in (0, 1):
        m = n // 2
        rec = 1 + 12 + 1 + L + 3 + L + 1
        t = torch.empty((m, rec), dtype=torch.uint8, device=dev)
        t[:, 0] = 64
        ids = torch.arange(m, device=dev, dtype=torch.int64)
        for d in range(12):
            t[:, 1 + d] = ((ids // (10 ** (11 - d))) % 10 + 48).to(torch.uint8)
        t[:, 13] = 10
        t[:, 14:14 + L] = reads[mate::2]
        t[:, 14 + L] = 10
        t[:, 15 + L] = 43
        t[:, 16 + L] = 10
        t[:, 17 + L:17 + 2 * L] = 73
        t[:, 17 + 2 * L] = 10
        p = os.path.join(tmp, "reads_%d.fq" % (mate + 1))
        t.cpu().numpy().tofile(p)
        paths.append(p)
        del t
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.cpu().numpy()].tobytes().decode()
    fa = os.path.join(tmp, "ref.fa")
    with open(fa, "w") as f:
        f.write(">chrS\n")
        for i in range(0, G, 60):
            f.write(ref[i:i + 60] + "\n")
    # the same bases as eight scaffolds (ref8.fa): a reference of several chromosomes, for what overlaps between scaffolds
    with open(os.path.join(tmp, "ref8.fa"), "w") as f:
        step = (G + 7) // 8
        for k in range(8):
            f.write(">chr%d\n" % (k + 1))
            for i in range(k * step, min(G, (k + 1) * step), 60):
                f.write(ref[i:min(i + 60, (k + 1) * step)] + "\n")
    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + 1)
    snp = torch.rand(G, generator=g, device=dev) < 0.001
    snp_pos = set(torch.nonzero(snp).flatten().cpu().tolist())
    del seq, reads, genome, snp
    torch.cuda.empty_cache()
    exe = os.path.join(ROOT, "biograph_amd", "bgbinary")
    bg = os.path.join(tmp, "s.bg")
    t0 = time.perf_counter()
    r = subprocess.run([exe, "create", "--reads", paths[0], "--pair", paths[1], "--out", bg], capture_output=True, text=True)
    t_create = time.perf_counter() - t0
    if r.returncode:
        print("create FAILED", r.stderr[-500:])
        return 1
    vcf = os.path.join(tmp, "v.vcf")
    t0 = time.perf_counter()
    r = subprocess.run([exe, "discovery", "--in", bg, "--ref", fa, "--out", vcf, "--threads", str(args.threads),
                        # the planted-genotype checks need GT, i.e. the simple genotype filter (off by default, as in the reference)
                        "--simple-gt", "true"], capture_output=True, text=True)
    t_disc = time.perf_counter() - t0
    if r.returncode:
        print("discovery FAILED", r.stderr[-500:])
        return 1
    summary = [x for x in r.stderr.splitlines() if "variants over" in x]
    stages = {}
    if summary:
        m = re.search(r"refmap ([\d.]+), sessions [\d.]+, tracing ([\d.]+) \[(\d+) assemblies\], pipeline ([\d.]+) \[(\d+) sections\], coverage \+ genotype ([\d.]+)", summary[-1])
        if m:
            stages = {"refmap_s": float(m.group(1)), "tracing_s": float(m.group(2)), "traced_assemblies": int(m.group(3)),
                      "pipeline_s": float(m.group(4)), "sections": int(m.group(5)), "coverage_genotype_s": float(m.group(6))}
    stages["stderr"] = [x for x in r.stderr.splitlines() if "discovery:" in x or "variants over" in x]
    recs = [x.rstrip("\n").split("\t") for x in open(vcf) if not x.startswith("#")]
    snps = {int(f[1]) - 1: f[9].split(":")[0] for f in recs if len(f[3]) == 1 and len(f[4]) == 1}
    out = {"genome_len": G, "coverage": args.coverage, "read_len": L, "reads": int(n),
           "create_wall_s": round(t_create, 2), "discovery_wall_s": round(t_disc, 2), "discovery_stages": stages,
           "vcf_records": len(recs), "snp_records": len(snps), "planted_snps": len(snp_pos),
           "planted_snps_in_vcf": len(set(snps) & snp_pos), "snp_records_not_planted": len(set(snps) - snp_pos),
           "planted_snps_called_homozygous": sum(1 for p, gt in snps.items() if p in snp_pos and gt == "1/1"),
           "ref_bases_per_s_discovery": round(G / t_disc, 1), "vcf_records_per_s": round(len(recs) / t_disc, 1)}
    print(json.dumps(out))
    if not args.keep:
        subprocess.run(["rm", "-rf", tmp])
    return 0


if __name__ == "__main__":
    sys.exit(main())
