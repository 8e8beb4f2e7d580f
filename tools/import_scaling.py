#!/usr/bin/env python3
"""`bgbinary create` from the chr20-shaped FASTQ files with 1, 2, 4 and 8 ranks ON ONE GPU (--comm shm): what the host side
of an N-rank import costs -- every rank counts the lines of its n / N bytes of each file, seeks to its first record and
uploads its stretch -- when all ranks read the same files through one page cache and share one PCIe link.  (The build and
the exchanges of N ranks on one device say nothing about N devices; the import column is the measurement.)
  python tools/import_scaling.py [--genome-len 64444167] [--ranks 1,2,4,8]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=64_444_167)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--read-len", type=int, default=150)
    args = ap.parse_args()
    import torch
    from biograph_amd import synth
    dev = torch.device("cuda", 0)
    G, L = args.genome_len, args.read_len
    seq, offs, n, _ = synth.make_reads(genome_len=G, coverage=30, read_len=L, seed=20261002, device=dev, return_genome=True)
    tmp = tempfile.mkdtemp(prefix="bg_imp_", dir="/dev/shm")
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
    fastq_bytes = sum(os.path.getsize(p) for p in paths)
    del seq, reads
    torch.cuda.empty_cache()
    exe = os.path.join(ROOT, "biograph_amd", "bgbinary")
    rows = []
    for ranks in [int(x) for x in args.ranks.split(",")]:
        out = os.path.join(tmp, "s%d.bg" % ranks)
        cmd = [exe, "create", "--reads", paths[0], "--pair", paths[1], "--out", out]
        if ranks > 1:
            cmd += ["--gpus", str(ranks), "--comm", "shm"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode:
            rows.append({"ranks": ranks, "error": r.stderr[-400:]})
            print(json.dumps(rows[-1]), flush=True)
            continue
        st = json.load(open(os.path.join(out, "qc", "create_stats.json")))
        tm = {k: v for d in st["timings"] for k, v in d.items()}
        rows.append({"ranks": ranks, "process_wall_s": round(wall, 3), "import_s": tm["import"], "gpu_build_s": tm["gpu_build"],
                     "write_s": tm["write"], "fastq_GB_per_s_import": round(fastq_bytes / tm["import"] / 1e9, 2),
                     "gbases_per_s_import": round(st["imported_bases"] / tm["import"] / 1e9, 2)})
        print(json.dumps(rows[-1]), flush=True)
        subprocess.run(["rm", "-rf", out])
    print(json.dumps({"fastq_bytes": fastq_bytes, "host_threads": os.cpu_count(), "rows": rows}))
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
