"""Per-step wall time of the graph-discover pipeline (C++ drivers over the GPU walks) on the first `region` bases of a
chr20-shaped sample.  Usage: python tools/discover_profile.py [genome_len] [region]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import biograph_amd as bg
from biograph_amd import synth, variants as V
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
region = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
dev = torch.device("cuda", 0)
seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=20261002, device=dev, return_genome=True)
ctx = bg.Context(0, torch.cuda.current_stream(dev).cuda_stream)
ss, st, rm = ctx.create_sample_device(seq.data_ptr(), offs.data_ptr(), n, True) if hasattr(ctx, "create_sample_device") else (None, None, None)
if ss is None:
    r = ctx.pack_reads_device(seq.data_ptr(), offs.data_ptr(), n)
    ks = ctx.kmer_count(r, 30, 5)
    c, _ = ctx.correct_reads(r, ks)
    ss, st = ctx.build_seqset(c, 30)
    rm = ctx.build_readmap(ss, c, True)
region = min(region, G)
ref = "".join("ACGT"[int(x)] for x in genome[:region].cpu().tolist())
ext = [(0, ref)]
t0 = time.perf_counter()
def step(name, f):
    global t0
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    print("%-28s %8.3f s   assemblies %d" % (name, t - t0, len(lst)), flush=True)
    t0 = t
lst = V.AssemblyList()
for with_rm in (None, rm):
    lst = V.AssemblyList()
    print("--- readmap:", "yes" if with_rm is not None else "no")
    t0 = time.perf_counter()
    step("make_ref_assemblies", lambda: lst.make_ref_assemblies(ext, 0, region))
    step("trim+dedup", lambda: lst.graph_trim_ref(ext).dedup())
    step("update_rc", lambda: lst.update_rc_seqset_entries(ss))
    step("discover_branch", lambda: lst.discover_branch(ss, with_rm, 105, "BRANCH"))
    step("trim+dedup", lambda: lst.graph_trim_ref(ext).dedup())
    step("update_rc", lambda: lst.update_rc_seqset_entries(ss))
    step("discover_push_to_pair", lambda: lst.discover_push_to_pair(ss, with_rm, 105, 1000, "PUSH_TO_PAIR", ["BRANCH"]))
    step("trim+dedup", lambda: lst.graph_trim_ref(ext).dedup())
    step("update_rc", lambda: lst.update_rc_seqset_entries(ss))
    found = [a for a in lst.to_python(False) if a.left_offset is not None and a.right_offset is not None and not a.matches_reference]
    print("anchored variants:", len(found))
