#!/usr/bin/env python3
"""`bgbinary create` on the bench sample's FASTQ files (as bench.py's end_to_end leg), run several times with the environment
given on the command line (KEY=VALUE ...): the CLI's stage timers per run."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from biograph_amd import synth


class A:
    seed = 20261002


def main():
    dev = torch.device("cuda", 0)
    G = int(os.environ.get("E2E_GENOME", synth.SHAPES["chr20"]))
    seq, offs, n = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=A.seed, device=dev)
    envs = [dict(kv.split("=", 1) for kv in arg.split(",") if kv) for arg in (sys.argv[1:] or [""])]
    # write the files once by borrowing the leg's writer: run it with a stub that keeps the directory
    tmp = tempfile.mkdtemp(prefix="bg_e2e_", dir="/dev/shm")
    reads = seq.view(n, 150)
    paths = []
    for mate in (0, 1):
        m = n // 2
        rec = 1 + 12 + 1 + 150 + 3 + 150 + 1
        t = torch.empty((m, rec), dtype=torch.uint8, device=dev)
        t[:, 0] = 64
        ids = torch.arange(m, device=dev, dtype=torch.int64)
        for d in range(12):
            t[:, 1 + d] = ((ids // (10 ** (11 - d))) % 10 + 48).to(torch.uint8)
        t[:, 13] = 10
        t[:, 14:164] = reads[mate::2]
        t[:, 164] = 10
        t[:, 165] = 43
        t[:, 166] = 10
        t[:, 167:317] = 73
        t[:, 317] = 10
        p = os.path.join(tmp, "reads_%d.fq" % (mate + 1))
        t.cpu().numpy().tofile(p)
        paths.append(p)
        del t
    del seq, reads
    torch.cuda.empty_cache()
    exe = os.path.join(ROOT, "biograph_amd", "bgbinary")
    for env in envs:
        for rep in range(2):
            out = os.path.join(tmp, "s.bg")
            subprocess.run(["rm", "-rf", out])
            t0 = time.perf_counter()
            r = subprocess.run([exe, "create", "--reads", paths[0], "--pair", paths[1], "--out", out], capture_output=True, text=True,
                               env=dict(os.environ, **env))
            wall = time.perf_counter() - t0
            if r.returncode:
                print("FAILED", r.stderr[-300:])
                continue
            st = json.load(open(os.path.join(out, "qc", "create_stats.json")))
            tm = {k: v for d in st["timings"] for k, v in d.items()}
            log = [l for l in open(os.path.join(out, "qc", "create_log.txt")) if "moving the files" in l or "written" in l.lower()]
            print(env, "wall %.3f" % wall, tm, "|", " ".join(x.strip()[-90:] for x in log))
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
