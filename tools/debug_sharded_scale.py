import sys, time
sys.path.insert(0, ".")
import torch
import biograph_amd as bg
from biograph_amd import sharded, synth
ctx = bg.Context(0)
for G in [int(x) for x in sys.argv[1:]]:
    seq, offs, n = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=1, device="cuda")
    torch.cuda.synchronize()
    t = time.time()
    try:
        out = sharded.sharded_create(sharded.LocalSimComm(1), {0: ctx}, {0: (seq, offs, n)})
        ss, st = out[0]
        print(G, "ok", st["n_entries"], st["n_records"], round(time.time() - t, 2), flush=True)
        ref, rst = ctx.create_device(seq.data_ptr(), offs.data_ptr(), n)
        print("   single", rst["n_entries"], rst["n_records"], ss.part("fixed") == ref.part("fixed"), flush=True)
        ss.free(); ref.free()
    except Exception as e:
        print(G, "FAIL", e, flush=True)
    del seq, offs
    ctx.trim()
