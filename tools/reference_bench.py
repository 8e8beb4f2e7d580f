#!/usr/bin/env python3
"""Times `bgbinary reference` (FASTA import + flat reference on the host, BWT on the GPU) on a synthetic genome of the
bench's shape.  Usage: tools/reference_bench.py [genome_len]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from biograph_amd import synth
    G = int(sys.argv[1]) if len(sys.argv) > 1 else synth.SHAPES["chr20"]
    genome = synth.make_genome(G, 20261002, "cuda")
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.cpu().numpy()]
    path = "/tmp/bench_ref.fa"
    with open(path, "wb") as f:
        f.write(b">20\n")
        lines = ref[:G - G % 80].reshape(-1, 80)
        f.write(b"\n".join(bytes(r) for r in lines))
        f.write(b"\n" + bytes(ref[G - G % 80:]) + b"\n")
    subprocess.run(["rm", "-rf", "/tmp/bench_refdir"])
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "biograph_amd", "bgbinary"), "reference", "--in", path, "--refdir", "/tmp/bench_refdir"],
                       capture_output=True, text=True)
    dt = time.time() - t0
    print(r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "", "| total %.2f s, rc %d, genome %d" % (dt, r.returncode, G))
    print({f: os.path.getsize("/tmp/bench_refdir/" + f) for f in sorted(os.listdir("/tmp/bench_refdir"))})


if __name__ == "__main__":
    main()
