"""The build through 16-byte items against the build through rows on the same reads: every part of the seqset (sizes, shared,
the four prev bit vectors with their rank tables) must be the same bytes.  Two child processes (the switch is read once a
process): BG_ITEMS_BUILD=1 and =0.  Usage: python tools/items_vs_rows.py [chr20 | repeats | <genome length>]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(what, path):
    """the reads, once (the generator with planted repeats is not the same from process to process), for both children"""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from biograph_amd import synth
    if what == "repeats":
        seq, offs, n = synth.make_reads(genome_len=20_000_000, coverage=30, read_len=150, seed=3, device="cuda", repeats=(300, 5000, 0.0))
    else:
        g = synth.SHAPES["chr20"] if what == "chr20" else int(what)
        seq, offs, n = synth.make_reads(genome_len=g, coverage=30, read_len=150, seed=1, device="cuda")
    np.save(path + ".seq.npy", seq.cpu().numpy())
    np.save(path + ".offs.npy", offs.cpu().numpy())


def child(what, path):
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    import torch
    import biograph_amd as bg
    seq = torch.from_numpy(np.load(path + ".seq.npy")).cuda()
    offs = torch.from_numpy(np.load(path + ".offs.npy")).cuda()
    n = int(offs.numel()) - 1
    ctx = bg.Context(0)
    prm = ctx.default_params()
    ms = None
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ss, st = ctx.create_device(seq.data_ptr(), offs.data_ptr(), n, prm)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3
        parts = {k: hashlib.sha1(bytes(v)).hexdigest() for k, v in sorted(ss.parts().items())}
        ss.free()
    print(json.dumps({"ms": round(ms, 1), "n_entries": st["n_entries"], "n_records": st["n_records"], "parts": parts}))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "chr20"
    path = "/dev/shm/bg_items_vs_rows_%d" % os.getpid()
    if len(sys.argv) > 3 and sys.argv[2] == "--child":
        return child(what, sys.argv[3])
    if len(sys.argv) > 3 and sys.argv[2] == "--make":
        return make(what, sys.argv[3])
    out = {}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), what, "--make", path], capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            sys.exit(1)
        for mode in ("1", "0"):
            env = dict(os.environ, BG_ITEMS_BUILD=mode, BG_DEBUG_PREV="1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), what, "--child", path], capture_output=True, text=True, env=env, timeout=900)
            if r.returncode != 0:
                print(r.stderr[-3000:])
                sys.exit(1)
            out[mode] = json.loads(r.stdout.strip().splitlines()[-1])
            took = [l for l in r.stderr.splitlines() if l.startswith("entries from items")]
            out[mode]["path"] = took[-1] if took else "?"
    finally:
        for suffix in (".seq.npy", ".offs.npy"):
            if os.path.exists(path + suffix):
                os.unlink(path + suffix)
    same = out["1"]["parts"] == out["0"]["parts"]
    print("%s: items %.1f ms (%s), rows %.1f ms; %d entries from %d records; every part the same bytes: %s" % (
        what, out["1"]["ms"], out["1"]["path"], out["0"]["ms"], out["1"]["n_entries"], out["1"]["n_records"], "yes" if same else "NO"))
    if not same:
        for k in out["1"]["parts"]:
            if out["1"]["parts"][k] != out["0"]["parts"].get(k):
                print("  differs:", k)
        sys.exit(1)


main()
