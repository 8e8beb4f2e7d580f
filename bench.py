#!/usr/bin/env python3
"""bench.py — seqset build throughput on synthetic 30x WGS-shaped 150 bp reads.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole create data path (pack -> k-mer count -> correct -> suffix sort ->
seqset arrays; bg_create in include/bgseq.h) over one batch of reads that is already resident in HBM
as ASCII.  Rank 0 prints ONE JSON line.  See DESIGN.md §Measurement for the byte accounting.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# algorithmic bytes of the stage a kernel family belongs to, from the step's counters
SURVEY_8D_BYTES = {
    "entries_dedup_build": lambda st: 28 * st["n_records"],
    "prev_bits": lambda st: 2.625 * st["n_entries"],
    # S2 = "write + read one 8-byte canonical k-mer record once": the write is the k-merizer's (here: sk_generate), the read the counter's
    "skm_count_lds": lambda st: 8 * st["n_kmers"],
    "skm_generate": lambda st: 8 * st["n_kmers"],
    "suffix_cover_count": lambda st: 28 * st["n_records"],
    "correct_dfs": lambda st: 0.5 * st["n_bases"] + 8 * st["n_kmers"],
    "items_partition": lambda st: 28 * st["n_records"], "items_tie_runs": lambda st: 28 * st["n_records"],
    "items_entries": lambda st: 28 * st["n_records"], "items_bucket_sort": lambda st: 28 * st["n_records"],
}
SURVEY_8D_FORMULA = {
    "entries_dedup_build": "S4 = 28 x N_rec", "prev_bits": "S5 = 2.625 x N_ent", "skm_count_lds": "S2 = 16 x N_kmers, the counter's half: 8 x N_kmers read",
    "skm_generate": "S2 = 16 x N_kmers, the k-merizer's half: 8 x N_kmers written", "suffix_cover_count": "S4 = 28 x N_rec", "correct_dfs": "S3 = 0.5 x N_bases + 8 x N_kmers",
    "items_partition": "S4 = 28 x N_rec", "items_tie_runs": "S4 = 28 x N_rec", "items_entries": "S4 = 28 x N_rec",
    "items_bucket_sort": "S4 = 28 x N_rec",
}
# kernel families of the step by stage of SURVEY.md section 8(d)
STAGE_OF_FAMILY = [("S1 pack", ("pack_",)), ("S2 k-mer count", ("skm_", "kmer_")), ("S3 correct", ("correct_",)),
                   ("S4+S5 suffix sort, dedup, prev bits, arrays", ("suffix_", "radix_", "entries_", "items_", "prev_bits", "seqset_"))]


def stage_table(kernels, stats, steps):
    """Per stage of SURVEY.md section 8(d): its algorithmic bytes over the device time of its kernel families, as a fraction of
    the HBM peak (the radix passes belong to S4 except the two of the k-mer hash build, which are small)."""
    alg = {"S1 pack": 1.25 * stats["n_bases"], "S2 k-mer count": 16 * stats["n_kmers"],
           "S3 correct": 0.5 * stats["n_bases"] + 8 * stats["n_kmers"],
           "S4+S5 suffix sort, dedup, prev bits, arrays": 28 * stats["n_records"] + 2.625 * stats["n_entries"]}
    out = {}
    for stage, prefixes in STAGE_OF_FAMILY:
        ms = sum(v["ms_per_step"] for k, v in kernels.items() if k.startswith(prefixes))
        if ms > 0:
            out[stage] = {"ms_per_step": round(ms, 2), "algorithmic_GB": round(alg[stage] / 1e9, 2),
                          "frac_of_hbm_peak": round(alg[stage] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def cpu_baseline(ctx, sample_genome, coverage, read_len, seed, dev):
    """The CPU restatement of the path (oracle/, kind="port": the reference itself needs Bazel + Boost + abseil,
    DESIGN.md section 2) on the node's host cores, SURVEY.md section 8(d): every stage on a std::thread pool of
    hardware_concurrency() threads, wall time per stage under the reference's stage names
    (modules/io/runtime_stats.h:13 as used by biograph_create.cpp:540-783), on a bounded sample of the same workload
    shape (E. coli-sized by default: 139 Mbases), and its seqset + readmap compared part by part with the GPU
    build of the same reads in the same run.  Also times the graph-discover drivers over the oracle's walks on a
    stretch of the sample's reference (the CPU baseline of the discover leg)."""
    import numpy as np
    import biograph_amd as bg
    from biograph_amd import synth
    from oracle import oracle as O
    # The port's stages have serial sections (the merges behind the per-partition counts, the readmap's row order) and do not
    # get faster beyond ~16 threads on the GPU box's 2 x 64-core host (profiles/r03_cpu_baseline_thread_sweep.txt: 13.3 s on
    # 16 threads, 14.2 s on 64, 16.9 s on 256 for this sample): the baseline runs on the count it is fastest at.
    # (round 4: the readmap's look-ups, row sort and first pass and the entry set's dedup / concatenation run on the threads too;
    # the stages are flat from 16 to 256 threads on the 2 x 64-core host -- 10.4 / 10.3 / 10.1 / 10.8 s on 16 / 64 / 128 / 256,
    # profiles/r04_cpu_baseline_thread_sweep.txt -- so the baseline runs on every hardware thread, as SURVEY.md section 8d asks)
    threads = min(O.hardware_threads(), int(os.environ.get("BG_CPU_BASELINE_THREADS", str(O.hardware_threads()))))
    dseq, doffs, n, genome = synth.make_reads(genome_len=sample_genome, coverage=coverage, read_len=read_len, seed=seed,
                                              device=dev, return_genome=True)
    t0 = time.perf_counter()
    seq = np.ascontiguousarray(dseq.cpu().numpy())
    offs = doffs.cpu().numpy().astype(np.uint64)
    t_import = time.perf_counter() - t0
    stages = {}
    t0 = time.perf_counter()
    ks = O.KmerSet.count(seq, offs, 30, 5, threads=threads)
    stages["kmerization"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    cseq, coffs, keep, cstats = O.correct_reads(ks, seq, offs, 8, 2, 0.7, threads=threads)
    stages["read_correction"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ent = O.Entries.build(cseq, coffs, "cover", 30, threads=threads)
    oss = O.Seqset.from_entries(ent, threads=threads)
    stages["make_seqset"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    orm = O.Readmap.from_packed(oss, ent, cseq, coffs, True, threads=threads)
    stages["make_readmap"] = time.perf_counter() - t0
    dt = sum(stages.values())
    bases = int(offs[-1])
    # the same reads through the GPU path: every container part of seqset and readmap must be equal
    r = ctx.pack_reads_device(dseq.data_ptr(), doffs.data_ptr(), n)
    gks = ctx.kmer_count(r, 30, 5)
    gc, _ = ctx.correct_reads(r, gks)
    gss, _ = ctx.build_seqset(gc, 30)
    grm = ctx.build_readmap(gss, gc, True)
    want = oss.parts()
    same = all(gss.part(p) == want[p] for p in bg.PART_NAMES)
    want_rm = orm.parts()
    same_rm = all(grm.part(p) == want_rm[p] for p in bg.READMAP_PART_NAMES)
    for h in (gc, gks, r):
        h.free()
    out = {"value": bases / dt / 1e9, "unit": "Gbases/s", "cores": threads, "node_cores": os.cpu_count(), "kind": "port",
           "sample": "%d reads x %d bp (%.1f Mbases, %dx of a %d bp synthetic genome), %.1f s on %d threads, %d entries" % (
               n, read_len, bases / 1e6, coverage, sample_genome, dt, threads, len(oss)),
           "stage_seconds": {k: round(v, 3) for k, v in stages.items()},
           "import_seconds_device_to_host": round(t_import, 3),
           "equals_gpu_build": {"seqset_parts": bool(same), "readmap_parts": bool(same_rm)}}
    # discover leg on the CPU: the drivers (tests/variants_twin.py, the Python twin of csrc/discover.cpp) over the
    # oracle's literal walks, on the first stretch of the sample's reference
    try:
        from tests import variants_twin as V
        from tests.discover_backend import OracleWalker
        region = min(30000, int(genome.numel()))
        ref = "".join("ACGT"[int(c)] for c in genome[:region].cpu().tolist())
        ext = [(0, ref)]
        w = OracleWalker(oss, orm)
        t0 = time.perf_counter()
        asms = V.make_ref_assemblies(ext, 0, region)
        asms = V.update_rc_seqset_entries(V.dedup_assemblies(V.graph_trim_ref(asms, ext)), w)
        asms = V.discover_branch(asms, w, 105, "BRANCH")
        asms = V.update_rc_seqset_entries(V.dedup_assemblies(V.graph_trim_ref(asms, ext)), w)
        asms = V.discover_push_to_pair(asms, w, 105, 1000, "PUSH_TO_PAIR", ["BRANCH"])
        asms = V.update_rc_seqset_entries(V.dedup_assemblies(V.graph_trim_ref(asms, ext)), w)
        ddt = time.perf_counter() - t0
        found = [a for a in asms if a.left_offset is not None and a.right_offset is not None and not a.matches_reference]
        out["discover"] = {"region_bases": region, "seconds": round(ddt, 2), "anchored_variants": len(found),
                           "anchored_variants_per_s": round(len(found) / ddt, 2), "cores": 1,
                           "what": "graph-discover drivers (Python twin) over the oracle's walks, min_overlap 105"}
    except Exception as e:  # the baseline of the discover leg must not take the build's line down
        out["discover"] = {"error": repr(e)}
    # the reference's discovery engine on the CPU: the oracle's push tracer (both directions, anchor_dropper,
    # pair_counter), serial pipeline, calc_coverage, genotype filter and VCF lines on the first stretch of the
    # reference, and the same through the library on the GPU build of the same reads: the lines must be equal
    try:
        from oracle import align as OA
        from oracle import coverage as OC
        from oracle import tracer as OT
        from biograph_amd import tracer as GT
        region = min(12000, int(genome.numel()))
        codes = genome[:region].cpu().numpy().astype(np.uint8)
        ref = np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode()
        mo = int(0.7 * read_len)
        t0 = time.perf_counter()
        graph = OT.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([region], np.uint64)))
        o, ao, sc = OT.Options(min_overlap=mo), OA.AlignOptions(min_overlap=mo), OT.Scaffold([(0, ref)])
        traced = []
        for rev in (False, True):
            traced += OT.reversable_trace(graph, sc, o, rev_comp=rev, anchor_drop=lambda s_, emit: OA.AnchorDropper(ao, s_, emit))
        _, sections = OA.variants_pipeline(ao, sc, traced, through_vcf_pad=True)
        co = OC.CoverageOptions()
        kept = OC.simple_genotype_filter(co, OC.calc_coverage(graph, [int(x) for x in orm.get("read_len")], sc, co, sections))
        want = [OC.vcf_line(co, "chrS", lambda off, k: ref[off:off + k], a) for a in kept]
        edt = time.perf_counter() - t0
        rmap = GT.reference_ref_map(gss, [ref])
        t0 = time.perf_counter()
        gopt = GT.TraceOptions(min_overlap=mo)
        gtraced = []
        for rev in (False, True):
            gtraced += GT.push_trace(gss, grm, rmap, [(0, ref)], rev_comp=rev, options=gopt)
        gsections = GT.process_assemblies(gtraced, [(0, ref)], options=gopt)
        _, got = GT.assemblies_coverage(gss, grm, gsections, [(0, ref)], vcf_name="chrS")
        gdt = time.perf_counter() - t0
        rmap.free()
        out["discovery_engine"] = {"region_bases": region, "seconds": round(edt, 2), "vcf_records": len(want), "cores": 1,
                                   "vcf_records_per_s": round(len(want) / max(edt, 1e-9), 2),
                                   "gpu_seconds_same_region": round(gdt, 3), "equals_gpu_vcf_lines": bool(got == want),
                                   "what": "oracle/tracer.py + align.py + coverage.py (pure Python restatement), min_overlap %d" % mo}
    except Exception as e:
        out["discovery_engine"] = {"error": repr(e)}
    for h in (grm, gss):
        h.free()
    return out


def driver_leg(ctx, ss, genome, dev, args, region=None, rm=None):
    """The whole graph-discover pipeline of the reference's graph_discover_test.py (make_ref_assemblies ->
    trim/update -> discover_branch -> ... -> discover_push_to_pair; host drivers biograph_amd/variants.py over
    the kernels) on the first `region` bases of the reference, scored against the SNPs synth.py planted (on one
    haplotype or both; its indels and structural variants land in `other_calls`).  Host bookkeeping in C++
    (csrc/discover.cpp); the chain of rc paths is broken by verified speculation, so a step is a few large
    launches plus single ones around variants."""
    import torch
    from biograph_amd import variants as V  # AssemblyList: the C++ drivers (csrc/discover.cpp)
    from biograph_amd.synth import variant_sites as synth_sites
    import numpy as np
    G = int(genome.numel())
    region = G if region is None else min(region, G)   # the whole reference of the configuration by default
    ref = np.frombuffer(b"ACGT", np.uint8)[genome[:region].cpu().numpy()].tobytes().decode()
    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + 1)
    snp = torch.rand(G, generator=g, device=dev) < 0.001
    shift = torch.randint(1, 4, (G,), generator=g, device=dev, dtype=torch.uint8)
    pos_t = torch.nonzero(snp[:region]).flatten()
    alt_t = ((genome[pos_t].to(torch.int64) + shift[pos_t].to(torch.int64)) & 3).cpu().tolist()
    truth = {p: "ACGT"[a] for p, a in zip(pos_t.cpu().tolist(), alt_t) if 300 < p < region - 300}
    ext = [(0, ref)]
    t0 = time.perf_counter()
    lst = V.AssemblyList().make_ref_assemblies(ext, 0, region)          # C++ drivers (csrc/discover.cpp)
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    lst.discover_branch(ss, rm, 105, "BRANCH")
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    lst.discover_push_to_pair(ss, rm, 105, 1000, "PUSH_TO_PAIR", ["BRANCH"])  # with the readmap: mates anchor the traces
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    dt = time.perf_counter() - t0
    found = [a for a in lst.to_python(False)
             if a.left_offset is not None and a.right_offset is not None and not a.matches_reference]
    snps = {a.left_offset: a.seq for a in found if a.right_offset == a.left_offset + 1 and len(a.seq) == 1}
    hit = sum(1 for p, b in truth.items() if snps.get(p) == b)
    # planted indels / structural variants: a call with the same net length within 60 bases of the site (a
    # call may sit anywhere in the repeat the event can slide along, and may carry flanking matches)
    import bisect
    planted = [(p, (ln if ins else -ln)) for p, ln, ins, _, _ in synth_sites(G, args.seed) if p + ln < region - 100]
    nets = sorted((a.left_offset, len(a.seq) - (a.right_offset - a.left_offset)) for a in found
                  if len(a.seq) != a.right_offset - a.left_offset)
    lefts = [x[0] for x in nets]
    indel_hit = 0
    for p, net in planted:
        i = bisect.bisect_left(lefts, p - 60)
        while i < len(nets) and nets[i][0] <= p + 60:
            if nets[i][1] == net:
                indel_hit += 1
                break
            i += 1
    return {"region_bases": region, "whole_reference": region == G, "seconds": round(dt, 2), "anchored_variants": len(found),
            "anchored_variants_per_s": round(len(found) / dt, 1),
            "planted_snps_in_region": len(truth), "planted_snps_found": hit,
            "planted_indels_svs_in_region": len(planted), "planted_indels_svs_found": indel_hit,
            "other_calls": len(found) - hit - indel_hit, "min_overlap": 105}


def push_tracer_leg(ss, rm, ref_map_dev, genome, args, chunks=2, chunk_size=1_000_000):
    """The production push tracer (csrc/tracer.cpp + csrc/align.cpp) on the first `chunks` work
    items of trace_ref (scaffold_split_size pieces), forward and reverse, with anchor_dropper and pair_counter: seconds per
    piece and direction, how many of the planted SNPs of the pieces lie inside the non-reference assemblies it emits, and
    how many come out of the serial pipeline (sort, ref trim, dedup, align, split) as 1-base variant sections."""
    import numpy as np
    import torch
    from biograph_amd import tracer

    class _Dev:  # reference_ref_map's return type: the ref_map bytes are resident already
        ptr = ref_map_dev.data_ptr()
    G = int(genome.numel())
    region = min(G, chunks * chunk_size)
    ref = np.frombuffer(b"ACGT", np.uint8)[genome[:min(G, region + 200_000)].cpu().numpy()].tobytes().decode()
    g = torch.Generator(device=genome.device)
    g.manual_seed(args.seed + 1)
    snp = torch.rand(G, generator=g, device=genome.device) < 0.001
    snp_pos = set(torch.nonzero(snp[:region]).flatten().cpu().tolist())
    t0 = time.perf_counter()
    session = tracer.PushTracer(ss, rm, _Dev, tracer.TraceOptions(min_overlap=int(0.7 * args.read_len)))
    t_open = time.perf_counter() - t0
    secs, traces, rounds, asms_n, hit = [], 0, 0, 0, set()
    everything = []
    for c in range(chunks):
        start, limit = c * chunk_size, min(region, (c + 1) * chunk_size)
        if start >= limit:
            break
        for rev in (False, True):
            t0 = time.perf_counter()
            asms, st = session.trace([(0, ref)], start=start, limit=limit, rev_comp=rev, with_stats=True)
            secs.append(round(time.perf_counter() - t0, 3))
            traces += st["traces"]
            rounds += st["device_rounds"]
            everything += asms
            for a in asms:
                if a.matches_reference:
                    continue
                asms_n += 1
                if a.right_anchor_len and len(a.seq) == a.right_offset - a.left_offset:
                    for i, (x, y) in enumerate(zip(a.seq, ref[a.left_offset:a.right_offset])):
                        if x != y and (a.left_offset + i) in snp_pos:
                            hit.add(a.left_offset + i)
    session.close()
    # the serial pipeline behind the tracers (csrc/align.cpp): sort, ref trim, dedup, align, split
    t0 = time.perf_counter()
    sections = tracer.process_assemblies(everything, [(0, ref)], steps=tracer.STEP_PIPELINE)
    t_pipe = time.perf_counter() - t0
    variants = [a for a in sections if not a.matches_reference]
    called = {a.left_offset for a in variants if a.right_offset == a.left_offset + 1 and len(a.seq) == 1}
    # calc_coverage + simple_genotype_filter (csrc/coverage.cpp: path groups walked on the device)
    t0 = time.perf_counter()
    kept = tracer.assemblies_coverage(ss, rm, sections, [(0, ref)])
    t_cov = time.perf_counter() - t0
    genotyped = {a.left_offset: a.strand_count for a in kept if a.right_offset == a.left_offset + 1 and len(a.seq) == 1}
    total = sum(secs)
    return {"reference_bases": region, "pieces": len(secs) // 2, "directions": 2, "session_open_seconds": round(t_open, 3),
            "seconds_per_piece_and_direction": secs, "traces": traces, "device_rounds": rounds,
            "non_reference_assemblies": asms_n, "planted_snps_in_region": len(snp_pos),
            "planted_snps_inside_assemblies": len(hit), "ref_bases_per_s_per_direction": round(2 * region / max(total, 1e-9), 1),
            "pipeline_seconds": round(t_pipe, 3), "pipeline_assemblies_in": len(everything),
            "pipeline_variant_sections": len(variants), "planted_snps_called_as_1_base_variants": len(called & snp_pos),
            "coverage_genotype_seconds": round(t_cov, 3), "vcf_records": len(kept),
            "planted_snps_in_vcf_records": len(set(genotyped) & snp_pos),
            "planted_snps_genotyped_homozygous": sum(1 for p, sc in genotyped.items() if p in snp_pos and sc == 2),
            "min_overlap": int(0.7 * args.read_len),
            "what": "tracer + anchor_dropper + pair_counter per piece and direction, then sorter + ref_trimmer + deduper + "
                    "aligner + align_splitter + normalizer + exact_deduper + vcf_padder over all of them, then calc_coverage + "
                    "simple_genotype_filter (the records ploidless_vcf_export writes)"}


def discover_leg(ctx, seq, offs, n_reads, params, genome, dev, args=None):
    """Overlap-walk throughput on the seqset just built (N=1 only, outside the build's timed region):
    ref_map::build over both strands of the reference haplotype (2 x G dependent push_front_drop calls in
    chains of a length the library chooses: 1024 bases here) and branch_discover::on_trace over the whole reference in 1000-base assemblies
    (every 1-base branch = a candidate variant site)."""
    import ctypes as C
    import torch
    from biograph_amd import capi
    L = capi.lib()
    # staged build so that the corrected reads are still resident for the readmap (make_readmap, §8f.1)
    r = ctx.pack_reads_device(seq.data_ptr(), offs.data_ptr(), n_reads)
    ks = ctx.kmer_count(r, params.kmer_size, params.min_kmer_count)
    corrected, _ = ctx.correct_reads(r, ks, params.correct.max_corrections, params.correct.min_good_run,
                                     params.correct.trim_after_portion)
    r.free()
    ks.free()
    ss, st = ctx.build_seqset(corrected, params.kmer_size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rm = ctx.build_readmap(ss, corrected, True)
    torch.cuda.synchronize()
    t_rm = time.perf_counter() - t0
    n_ids = rm.size()
    corrected.free()
    G = int(genome.numel())
    n = ss.size()
    out = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    es = (C.c_uint64 * 1)(0)
    ez = (C.c_uint64 * 1)(G)
    vp = C.c_void_p
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.check(L.bg_ref_map_build(ctx.h, ss.h, vp(genome.data_ptr()), 1, es, ez, 0, vp(out.data_ptr())))
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    marked = int((out[:n] & 63 != 0).sum().item())
    unique = int(((out[:n] & 63) == 1).sum().item())
    # N_pfd and N_drop of the walk's algorithmic bytes (52 N_pfd + 60 N_drop), counted by the same
    # kernel once more with the measurement counters on (untimed: two contended atomics)
    cnt2 = (C.c_uint64 * 2)()
    count_walk = os.environ.get("BG_BENCH_NO_WALK_COUNTERS") != "1"  # (off under the profiler: the counting run would be averaged in)
    ref_counts = None
    if count_walk:
        out2 = torch.empty_like(out)
        capi.check(L.bg_walk_counters(ctx.h, 1, None))
        capi.check(L.bg_ref_map_build(ctx.h, ss.h, vp(genome.data_ptr()), 1, es, ez, 0, vp(out2.data_ptr())))
        capi.check(L.bg_walk_counters(ctx.h, 0, cnt2))
        ref_counts = (int(cnt2[0]), int(cnt2[1]))
        del out2
    # branch trace: assemblies of 1000 reference bases, each preceded by a 256-base warm-up walk from the
    # empty frontier (events inside the warm-up are not counted)
    A, WARM = 1000, 256
    starts = torch.arange(0, G - A - WARM, A, device=dev, dtype=torch.int64)
    na = int(starts.numel())
    soff = (torch.arange(0, na + 1, device=dev, dtype=torch.int64) * (A + WARM)).contiguous()
    idx = (starts[:, None] + torch.arange(0, A + WARM, device=dev)[None, :]).reshape(-1)
    codes = genome[idx].contiguous()
    foff = torch.arange(0, na + 1, device=dev, dtype=torch.int32)
    fb = torch.zeros(na, dtype=torch.int64, device=dev)
    fe = torch.full((na,), n, dtype=torch.int64, device=dev)
    fs = torch.zeros(na, dtype=torch.int32, device=dev)
    cap = 64
    ev_off = torch.empty(na * cap, dtype=torch.int32, device=dev)
    ev_base = torch.empty(na * cap, dtype=torch.uint8, device=dev)
    n_ev = torch.empty(na, dtype=torch.int32, device=dev)
    n_out = torch.empty(na, dtype=torch.uint8, device=dev)
    ob = torch.empty(na * 16, dtype=torch.int64, device=dev)
    oe = torch.empty(na * 16, dtype=torch.int64, device=dev)
    osz = torch.empty(na * 16, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.check(L.bg_branch_trace_batch(ctx.h, ss.h, na, vp(soff.data_ptr()), vp(codes.data_ptr()), vp(foff.data_ptr()),
                                       vp(fb.data_ptr()), vp(fe.data_ptr()), vp(fs.data_ptr()), 105, cap,
                                       vp(ev_off.data_ptr()), vp(ev_base.data_ptr()), vp(n_ev.data_ptr()),
                                       vp(n_out.data_ptr()), vp(ob.data_ptr()), vp(oe.data_ptr()), vp(osz.data_ptr())))
    torch.cuda.synchronize()
    t_br = time.perf_counter() - t0
    br_counts = None
    if count_walk:
        capi.check(L.bg_walk_counters(ctx.h, 1, None))
        capi.check(L.bg_branch_trace_batch(ctx.h, ss.h, na, vp(soff.data_ptr()), vp(codes.data_ptr()), vp(foff.data_ptr()),
                                           vp(fb.data_ptr()), vp(fe.data_ptr()), vp(fs.data_ptr()), 105, cap,
                                           vp(ev_off.data_ptr()), vp(ev_base.data_ptr()), vp(n_ev.data_ptr()),
                                           vp(n_out.data_ptr()), vp(ob.data_ptr()), vp(oe.data_ptr()), vp(osz.data_ptr())))
        capi.check(L.bg_walk_counters(ctx.h, 0, cnt2))
        br_counts = (int(cnt2[0]), int(cnt2[1]))
    k = torch.minimum(n_ev, torch.full_like(n_ev, cap))
    mask = torch.arange(cap, device=dev)[None, :] < k[:, None]
    sites = int(((ev_off.view(na, cap) >= WARM) & mask).sum().item())
    # every branch site -> its 1-base branch assembly (discover_extend_right, discover.cpp:157-175): rc
    # ranges by propagate_from_end over the 255 reference bases before the site + the alt base (row G4),
    # then the push-to-pair extension trace from there (row G2 loop), up to 256 bases
    ev_ok = ((ev_off.view(na, cap) >= WARM) & mask)
    aidx, eidx = torch.nonzero(ev_ok, as_tuple=True)
    ns = int(aidx.numel())
    tr_counts, t_tr = None, 0.0
    ext = None
    if ns:
        pos = starts[aidx] + ev_off.view(na, cap)[aidx, eidx].to(torch.int64)       # absolute site position
        alt = ev_base.view(na, cap)[aidx, eidx]
        LW = 256
        win = (pos[:, None] - (LW - 1) + torch.arange(0, LW - 1, device=dev)[None, :]).clamp_(min=0)
        fwd = torch.cat([genome[win], alt[:, None]], dim=1)                          # assembly sequence
        pcodes = (3 - fwd.flip(1)).to(torch.uint8).contiguous().view(-1)             # its reverse complement
        psoff = (torch.arange(0, ns + 1, device=dev, dtype=torch.int64) * LW).contiguous()
        pfoff = torch.arange(0, ns + 1, device=dev, dtype=torch.int32)
        pfb = torch.zeros(ns, dtype=torch.int64, device=dev)
        pfe = torch.full((ns,), n, dtype=torch.int64, device=dev)
        pfs = torch.zeros(ns, dtype=torch.int32, device=dev)
        pn = torch.empty(ns, dtype=torch.uint8, device=dev)
        pb = torch.empty(ns * 16, dtype=torch.int64, device=dev)
        pe = torch.empty(ns * 16, dtype=torch.int64, device=dev)
        ps = torch.empty(ns * 16, dtype=torch.int32, device=dev)
        nmates = torch.empty(ns, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        capi.check(L.bg_propagate_from_end_batch(ctx.h, ss.h, None, ns, vp(psoff.data_ptr()), vp(pcodes.data_ptr()),
                                                 vp(pfoff.data_ptr()), vp(pfb.data_ptr()), vp(pfe.data_ptr()),
                                                 vp(pfs.data_ptr()), 20, None, None, None, None, None, None,
                                                 vp(pn.data_ptr()), vp(pb.data_ptr()), vp(pe.data_ptr()),
                                                 vp(ps.data_ptr()), 1, vp(nmates.data_ptr()), None, None, None))
        torch.cuda.synchronize()
        t_prop = time.perf_counter() - t0
        keep = torch.arange(16, device=dev)[None, :] < pn[:, None].to(torch.int64)
        tfoff = torch.zeros(ns + 1, dtype=torch.int32, device=dev)
        tfoff[1:] = torch.cumsum(pn.to(torch.int32), 0)
        tb, te, tz = pb.view(ns, 16)[keep].contiguous(), pe.view(ns, 16)[keep].contiguous(), ps.view(ns, 16)[keep].contiguous()
        MAXS, RC, SC = 256, 1024, 256
        o_ns = torch.empty(ns, dtype=torch.int32, device=dev)
        o_why = torch.empty(ns, dtype=torch.uint8, device=dev)
        o_bases = torch.empty(ns * MAXS, dtype=torch.uint8, device=dev)
        o_cnt = torch.empty(ns * MAXS, dtype=torch.uint8, device=dev)
        o_nr = torch.empty(ns, dtype=torch.int32, device=dev)
        o_rb = torch.empty(ns * RC, dtype=torch.int64, device=dev)
        o_re = torch.empty(ns * RC, dtype=torch.int64, device=dev)
        o_rs = torch.empty(ns * RC, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        capi.check(L.bg_push_to_pair_trace_batch(ctx.h, ss.h, ns, vp(tfoff.data_ptr()), vp(tb.data_ptr()),
                                                 vp(te.data_ptr()), vp(tz.data_ptr()), 105, MAXS, RC, SC,
                                                 vp(o_ns.data_ptr()), vp(o_why.data_ptr()), vp(o_bases.data_ptr()),
                                                 vp(o_cnt.data_ptr()), vp(o_nr.data_ptr()), vp(o_rb.data_ptr()),
                                                 vp(o_re.data_ptr()), vp(o_rs.data_ptr())))
        torch.cuda.synchronize()
        t_tr = time.perf_counter() - t0
        if count_walk:  # N_pfd / N_drop of the same launch, counted in an untimed repeat
            capi.check(L.bg_walk_counters(ctx.h, 1, None))
            capi.check(L.bg_push_to_pair_trace_batch(ctx.h, ss.h, ns, vp(tfoff.data_ptr()), vp(tb.data_ptr()),
                                                     vp(te.data_ptr()), vp(tz.data_ptr()), 105, MAXS, RC, SC,
                                                     vp(o_ns.data_ptr()), vp(o_why.data_ptr()), vp(o_bases.data_ptr()),
                                                     vp(o_cnt.data_ptr()), vp(o_nr.data_ptr()), vp(o_rb.data_ptr()),
                                                     vp(o_re.data_ptr()), vp(o_rs.data_ptr())))
            capi.check(L.bg_walk_counters(ctx.h, 0, cnt2))
            tr_counts = (int(cnt2[0]), int(cnt2[1]))
        steps_total = int(o_ns.sum().item())
        ext = {"sites": ns, "propagate_seconds": round(t_prop, 4), "propagate_pfd_per_s": round(ns * LW / t_prop / 1e9, 3),
               "trace_seconds": round(t_tr, 4), "trace_steps": steps_total,
               "trace_pfd_per_s": round(steps_total * 4 / t_tr / 1e9, 3),
               "traces_reaching_max_steps": int((o_why == 2).sum().item()),
               "traces_ended_no_extension": int((o_why == 0).sum().item()),
               "traces_ended_seen_range": int((o_why == 1).sum().item()),
               # candidate SITES (one-base branches) through branch -> propagate -> extension; the anchored variants of the
               # whole pipeline, host bookkeeping included, are graph_discover_driver.anchored_variants_per_s
               "candidate_sites_per_s_kernels_only": round(ns / (t_br + t_prop + t_tr), 1)}
    # the older graph-discover pipeline over the whole reference takes ~45 s and calls fewer of the planted variants than the
    # engine `bgbinary discovery` runs (end_to_end_from_fastq.discovery): only on request
    drv = driver_leg(ctx, ss, genome, dev, args, rm=rm) if (args is not None and getattr(args, "graph_discover_driver", False)) else None
    push = push_tracer_leg(ss, rm, out, genome, args) if args is not None else None
    rm.free()
    ss.free()
    pfd_ref = 2 * G + 2 * 256 * (G // 4096 + 1)
    pfd_br = na * (A + WARM) * 4

    def walk_roofline(kernel, counts, seconds):
        # one push_front_drop without a drop = 52 bytes, every iteration of its drop loop + 60
        if counts is None:
            return None
        n_pfd, n_drop = counts
        alg = 52 * n_pfd + 60 * n_drop
        r = {"bound": "hbm", "kernel": kernel, "n_pfd": n_pfd, "n_drop": n_drop, "algorithmic_bytes": alg,
             "achieved": round(alg / seconds / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(alg / seconds / 1e9 / HBM_PEAK_GBS, 4), "seconds": round(seconds, 4),
             "limiter": "dependent random 64-byte sectors: a chain is one push_front_drop after the other (2 rank look-ups of 3 "
                        "words each + sizes per call), %d chains in flight" % (2 * (G // 1024 + 1) if kernel == "k_ref_map_chains" else (ns if kernel == "k_push_to_pair_single" else na))}
        r.update(pmc_traffic_walk(kernel, seconds * 1e3))
        return r
    roof_ref = walk_roofline("k_ref_map_chains", ref_counts, t_ref)
    roof_br = walk_roofline("k_branch_trace", br_counts, t_br)
    roof_tr = walk_roofline("k_push_to_pair_single", tr_counts, t_tr) if t_tr > 0 else None
    return {"roofline": roof_ref, "roofline_branch_trace": roof_br, "roofline_push_to_pair_trace": roof_tr,
            "readmap": {"seconds": round(t_rm, 4), "read_ids": n_ids, "read_ids_per_s": round(n_ids / t_rm / 1e6, 1),
                        "unit": "M read ids/s (paired, both orientations)"},
            "ref_map": {"seconds": round(t_ref, 4), "ref_bases_per_s": round(2 * G / t_ref / 1e9, 3),
                        "unit": "Gbases/s (both strands)", "push_front_drop_per_s": round(pfd_ref / t_ref / 1e9, 3),
                        "entries_marked": marked, "entries_unique": unique,
                        "algorithmic_GBps": round(pfd_ref * 52 / t_ref / 1e9, 1)},
            "branch_discover": {"seconds": round(t_br, 4), "assemblies": na, "min_overlap": 105,
                                "variants_per_s": round(sites / t_br, 1), "branch_sites": sites,
                                "push_front_drop_per_s": round(pfd_br / t_br / 1e9, 3),
                                "algorithmic_GBps": round(pfd_br * 52 / t_br / 1e9, 1)},
            "push_to_pair_extension": ext, "graph_discover_driver": drv, "push_tracer": push}


def end_to_end_leg(seq, n_reads, read_len, dev, args, genome=None):
    """`bgbinary create` from FASTQ FILES ("report both"): the reads of the bench sample written
    as two mate files, then the CLI as a user runs it — files mapped, their bytes moved to the GPU through pinned
    staging buffers, records parsed and paired on the GPU (csrc/fastq.hip), build, seqset + readmap written.
    Reported: the CLI's own stage timers (qc/create_stats.json) and the wall time of the whole process."""
    import shutil
    import subprocess
    import tempfile
    import torch
    if n_reads % 2 or int(seq.numel()) != n_reads * read_len:
        return None
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 6 * n_reads * read_len else None
    tmp = tempfile.mkdtemp(prefix="bg_e2e_", dir=base)
    try:
        n = n_reads // 2
        rec = 1 + 12 + 1 + read_len + 3 + read_len + 1
        reads = seq.view(n_reads, read_len)
        paths = []
        for mate in (0, 1):
            t = torch.empty((n, rec), dtype=torch.uint8, device=dev)
            t[:, 0] = 64  # '@'
            ids = torch.arange(n, device=dev, dtype=torch.int64)
            for d in range(12):
                t[:, 1 + d] = ((ids // (10 ** (11 - d))) % 10 + 48).to(torch.uint8)
            t[:, 13] = 10
            t[:, 14:14 + read_len] = reads[mate::2]
            t[:, 14 + read_len] = 10
            t[:, 15 + read_len] = 43  # '+'
            t[:, 16 + read_len] = 10
            t[:, 17 + read_len:17 + 2 * read_len] = 73  # 'I'
            t[:, 17 + 2 * read_len] = 10
            p = os.path.join(tmp, "reads_%d.fq" % (mate + 1))
            t.cpu().numpy().tofile(p)
            del t
            paths.append(p)
        fastq_bytes = sum(os.path.getsize(p) for p in paths)
        exe = os.path.join(ROOT, "biograph_amd", "bgbinary")
        out = os.path.join(tmp, "sample.bg")
        t0 = time.perf_counter()
        r = subprocess.run([exe, "create", "--reads", paths[0], "--pair", paths[1], "--out", out], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": r.stderr[-500:]}
        st = json.load(open(os.path.join(out, "qc", "create_stats.json")))
        tm = {k: v for d in st["timings"] for k, v in d.items()}
        bases = st["imported_bases"]
        discovery = discovery_cli_leg(exe, out, tmp, genome, args) if genome is not None else None
        return {"command": "bgbinary create --reads R1.fq --pair R2.fq --out S.bg", "fastq_bytes": fastq_bytes, "discovery": discovery,
                "files_on": "tmpfs" if base else "disk", "imported_reads": st["imported_reads"], "seqset_entries": st["seqset_entries"],
                "seconds": {"import_files_to_gpu_and_parse": tm["import"], "gpu_build_incl_readmap": tm["gpu_build"],
                            "write_seqset_and_readmap": tm["write"], "cli_total": tm["total"], "process_wall": round(wall, 3)},
                "gbases_per_s_import_plus_build": round(bases / (tm["import"] + tm["gpu_build"]) / 1e9, 3),
                "gbases_per_s_process_wall": round(bases / wall / 1e9, 3),
                "note": "cold process: includes HIP start-up and first-touch hipMalloc (30 us/MiB), which a warm "
                        "bench step does not pay"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def discovery_cli_leg(exe, bg_dir, tmp, genome, args):
    """`bgbinary discovery` (the reference's engine: push tracer in both directions over every 1 Mb piece, serial
    pipeline, calc_coverage, genotype filter, VCF records) on the .bg the create leg just wrote, against the sample's
    reference as a FASTA: wall time, the command's own stage times, and the records against the planted SNPs."""
    import re
    import subprocess
    import numpy as np
    import torch
    G = int(genome.numel())
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.cpu().numpy()].tobytes().decode()
    fa = os.path.join(tmp, "ref.fa")
    with open(fa, "w") as f:
        f.write(">chrS\n")
        f.write("\n".join(ref[i:i + 60] for i in range(0, G, 60)))
        f.write("\n")
    vcf = os.path.join(tmp, "v.vcf")
    t0 = time.perf_counter()
    # --simple-gt true: GT is needed for the planted-genotype figures (the reference's default leaves every record 0/1)
    r = subprocess.run([exe, "discovery", "--in", bg_dir, "--ref", fa, "--out", vcf, "--simple-gt", "true"], capture_output=True, text=True)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": r.stderr[-500:]}
    out = {"command": "bgbinary discovery --in S.bg --ref ref.fa --out V.vcf --simple-gt true", "process_wall": round(wall, 2)}
    m = re.search(r"on (\d+) tracer session.*refmap ([\d.]+), sessions ([\d.]+), tracing ([\d.]+) \[(\d+) assemblies\], pipeline ([\d.]+) "
                  r"\[(\d+) sections\], coverage \+ genotype ([\d.]+)", r.stderr)
    if m:
        out.update({"tracer_sessions": int(m.group(1)), "refmap_s": float(m.group(2)), "sessions_open_s": float(m.group(3)),
                    "tracing_s": float(m.group(4)), "traced_assemblies": int(m.group(5)), "pipeline_s": float(m.group(6)),
                    "sections": int(m.group(7)), "coverage_genotype_s": float(m.group(8))})
    recs = [x.split("\t") for x in open(vcf) if not x.startswith("#")]
    g = torch.Generator(device=genome.device)
    g.manual_seed(args.seed + 1)
    planted = set(torch.nonzero(torch.rand(G, generator=g, device=genome.device) < 0.001).flatten().cpu().tolist())
    snps = {int(f[1]) - 1: f[9].split(":")[0] for f in recs if len(f[3]) == 1 and len(f[4]) == 1}
    out.update({"vcf_records": len(recs), "vcf_records_per_s": round(len(recs) / wall, 1), "planted_snps": len(planted),
                "planted_snps_in_vcf": len(set(snps) & planted), "snp_records_not_planted": len(set(snps) - planted),
                "planted_snps_called_homozygous": sum(1 for p, gt in snps.items() if p in planted and gt == "1/1"),
                "ref_bases_per_s": round(G / wall, 1)})
    return out


KERNEL_OF_FAMILY = {"radix_scatter_k": "rs_part_scatter_kernel",
                    "radix_scatter_kv": "rs_scatter_kernel<unsigned int, true>", "radix_hist": "rs_hist_kernel",
                    "correct_screen": "k_correct_screen", "entries_dedup_build": "k_gather_dedup_build",
                    "skm_count_lds": "sk_count_lds_kernel", "skm_generate": "sk_generate_kernel",
                    "suffix_cover_count": "k_cover_count", "prev_bits": "k_prev_bits_pre2",
                    "suffix_small_ties": "k_sort_small_groups", "entries_dedup": "k_gather_lcp_keep",
                    "items_partition": "it_scatter_kernel", "items_tie_runs": "k_items_tie_small", "items_entries": "k_items_entries",
                    "items_bucket_sort": "k_items_bucket_sort"}


LIMITER_OF_FAMILY = {"skm_count_lds": "VALU issue (rolling canonical k-mer + LDS probe per window)",
                     "skm_generate": "VALU issue (minimiser of every window)",
                     "entries_dedup_build": "random 128-byte lines (a 48-byte row drags 1.25 lines; a tile of 512 rows brings 20 more as halo for the "
                                            "tie groups it orders) + look-back latency",
                     "suffix_cover_count": "L1 -> L2 request rate (21 dependent probes per read)",
                     "radix_scatter_k": "hbm stream", "skm_part_scatter": "hbm stream", "suffix_emit": "hbm stream",
                     "items_partition": "hbm stream (16 bytes a record read twice, written once, a level)",
                     "items_bucket_sort": "LDS (a counting sort and a ranking inside bins per bucket)",
                     "items_tie_runs": "random 64-byte sectors (the rest of the string of every record that shares its first word)",
                     "items_entries": "hbm stream"}


def pmc_traffic(family, avg_launch_ms):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r*_chr20_rocprofv3_summary.json: separate FETCH_SIZE / WRITE_SIZE runs; KiB units; FETCH_SIZE doubled as
    MI355X_MICROARCH.md section HBM prescribes for gfx950).  bench.py cannot collect PMC counters itself, so the figure is
    only attached when the profile is of THIS kernel as it runs now: same kernel name and an average duration within
    15 % of the one just measured; otherwise traffic stays null and traffic_source says why."""
    import glob
    pat = KERNEL_OF_FAMILY.get(family)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_chr20_rocprofv3_summary.json")))
    if not pat or not files:
        return {"traffic": None, "traffic_source": "no committed PMC profile for this kernel"}
    for e in json.load(open(files[-1])):
        if pat in e["kernel"] and "fetch_bytes_per_launch_x2" in e:
            prof_ms = e["avg_us"] / 1e3
            if abs(prof_ms - avg_launch_ms) > 0.15 * avg_launch_ms:
                return {"traffic": None, "traffic_source": "%s is stale: %s took %.2f ms there, %.2f ms now" % (
                    os.path.basename(files[-1]), pat, prof_ms, avg_launch_ms)}
            return {"traffic": int(e["fetch_bytes_per_launch_x2"] + e.get("write_bytes_per_launch", 0)),
                    "traffic_fetch_raw": int(e["fetch_bytes_per_launch_raw"]),
                    "traffic_write": int(e.get("write_bytes_per_launch", 0)),
                    "traffic_source": os.path.basename(files[-1]) + " (kernel name and duration match this run)"}
    return {"traffic": None, "traffic_source": "%s has no PMC entry for %s" % (os.path.basename(files[-1]), pat)}


def pmc_traffic_walk(kernel, live_ms):
    """HBM traffic of a walk kernel from the committed counter profile of the discover leg (tools/profile_walk.sh ->
    profiles/r*_chr20_walk_rocprofv3_summary.json): attached when the profile's duration agrees with this run's within 25 %."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_chr20_walk_rocprofv3_summary.json")))
    if not files:
        return {"traffic": None, "traffic_source": "no committed walk profile"}
    for e in json.load(open(files[-1])):
        if kernel in e["kernel"] and "fetch_bytes_per_launch_raw" in e:
            ms = e["avg_us"] / 1e3
            if abs(ms - live_ms) > 0.25 * live_ms:
                return {"traffic": None, "traffic_source": "%s: %s took %.2f ms there, %.2f ms here" % (os.path.basename(files[-1]), kernel, ms, live_ms)}
            return {"traffic": int(e["fetch_bytes_per_launch_x2"] + e.get("write_bytes_per_launch", 0)),
                    "traffic_fetch_raw": int(e["fetch_bytes_per_launch_raw"]), "traffic_write": int(e.get("write_bytes_per_launch", 0)),
                    "traffic_source": os.path.basename(files[-1]) + " (kernel name and duration match this run)"}
    return {"traffic": None, "traffic_source": "%s has no PMC entry for %s" % (os.path.basename(files[-1]), kernel)}


def self_launch(n, json_fd):
    """Runs `python -m torch.distributed.run --nproc-per-node n bench.py <same flags>` as a child process and
    relays the child's single JSON line (fd 1 of the child) and its exit code.  The parent never touches the GPU."""
    import socket
    import subprocess
    dc = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        have = int(dc.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        have = 0
    if have < n:
        print("bench.py: --gpus %d needs %d devices, this node has %d" % (n, n, have), file=sys.stderr)
        return 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env)
    out, _ = p.communicate()
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        os.write(json_fd, (lines[-1] + "\n").encode())
    return p.returncode if p.returncode or lines else 5


def main():
    # stdout must carry exactly ONE JSON line: native libraries (RCCL prints a version banner) write to
    # fd 1 too, so fd 1 is pointed at stderr for the whole run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", default="chr20", choices=["tiny", "ecoli", "chr20", "human"],
                    help="human = ONE 3.1 Gb sample (24 contigs) over all ranks: BASELINE.json configs[3], needs 8 GPUs")
    ap.add_argument("--genome-len", type=int, default=0, help="override the shape's genome length")
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--cpu-sample-genome", type=int, default=4_641_652,
                    help="genome of the CPU baseline sample (default: E. coli-sized, 139 Mbases at 30x)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-discover", action="store_true", help="skip the overlap-walk (discover) leg")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the `bgbinary create` from FASTQ files leg")
    ap.add_argument("--no-shared-arena", action="store_true",
                    help="sharded runs: keep torch's own caching allocator beside the library's arena")
    ap.add_argument("--graph-discover-driver", action="store_true",
                    help="also time the graph-discover drivers (branch + push_to_pair) over the whole reference (~45 s)")
    ap.add_argument("--no-sharded-readmap", action="store_true", help="sharded runs: skip the untimed build-with-readmap leg")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the sharded (RCCL) build path even with one rank (smoke test of the collectives)")
    ap.add_argument("--comm", default="native", choices=["native", "torch"],
                    help="sharded runs: native = bg_create_sharded over bg_comm (rccl.h inside libbgseq: what `bgbinary create "
                         "--gpus N` runs); torch = biograph_amd/sharded.py over torch.distributed")
    ap.add_argument("--shortcut", action="store_true",
                    help="--force-sharded with one rank: let the rank's collectives be copies (default: BG_COMM_NO_SHORTCUT=1, "
                         "every collective goes through RCCL to itself)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as the driver invokes it: this process stays GPU-free (no torch import, no
        # HIP call) and starts one rank per GPU as a CHILD process, relaying its one JSON line and exit code.
        sys.exit(self_launch(args.gpus, json_fd))
    if world != args.gpus:
        print("bench.py: WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)

    import torch
    import biograph_amd as bg
    from biograph_amd import synth

    shared_arena = False
    native = args.comm == "native"
    if (world > 1 or args.force_sharded) and native and not args.shortcut:
        os.environ.setdefault("BG_COMM_NO_SHORTCUT", "1")
    if (world > 1 or args.force_sharded) and not args.no_shared_arena and not native:
        # the exchange buffers (torch tensors) and the library's buffers come out of one arena per device
        shared_arena = bg.use_shared_arena()
    if not torch.cuda.is_available():
        print("bench.py: no GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    if torch.cuda.device_count() < world and not os.environ.get("BG_BENCH_SHARE_DEVICE"):
        print("bench.py: --gpus %d needs %d devices, this node has %d" % (world, world, torch.cuda.device_count()),
              file=sys.stderr)
        sys.exit(4)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    # Weak scaling: ONE sample whose genome is `world` x the shape's length; every rank holds 1/world of
    # its read pairs (per-GPU reads fixed) and the ranks build ONE seqset together (biograph_amd/sharded.py:
    # k-mers, records and tails cross GPUs by RCCL all_to_all, kept k-mers / corrected reads by all_gather).
    # "human" is the whole-human configuration itself (3.1 Gb over however many ranks there are), not a per-rank size.
    genome_len = (args.genome_len or synth.SHAPES[args.shape]) * (1 if args.shape == "human" else world)
    if args.shape == "human" and not args.genome_len and world < 8:
        print("bench.py: --shape human is the 8-GPU configuration (93 Gbases; %d rank(s) cannot hold it)" % world,
              file=sys.stderr)
        sys.exit(2)
    seq, offs, n_reads, genome = synth.make_reads(genome_len=genome_len, coverage=args.coverage,
                                                  read_len=args.read_len, seed=args.seed, device=dev,
                                                  shard=(rank, world), return_genome=True,
                                                  contigs=synth.contig_lengths(genome_len, args.shape))
    torch.cuda.synchronize()
    n_bases = int(seq.numel())
    stream = torch.cuda.current_stream(dev).cuda_stream
    ctx = bg.Context(local_rank, stream)
    params = ctx.default_params()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ncomm = None
    comm = None              # (torch mode) sharded.py's communicator
    sharded_mode = None      # "native" | "torch" | "replicas" once a multi-rank run has settled on one
    sharded_fallback = None  # why the ranks do not build one sample together, when they do not
    if world > 1 or args.force_sharded:
        from biograph_amd import sharded

    def fused_step():
        ss, st = ctx.create_device(seq.data_ptr(), offs.data_ptr(), n_reads, params)
        ss.free()
        return st

    def make_step(mode):
        nonlocal ncomm, comm
        if mode in os.environ.get("BG_BENCH_FAIL_MODES", "").split(","):  # (tests: the fall-back chain)
            raise RuntimeError("injected by BG_BENCH_FAIL_MODES")
        if mode == "native":
            # ONE C call per rank and step: bg_create_sharded over bg_comm (csrc/sharded.hip, csrc/comm.hip: rccl.h inside the
            # library); torch.distributed only carries the communicator's id to the ranks and the barriers around the timed region
            ident = [sharded.NativeComm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)
            ncomm = sharded.NativeComm(ctx, rank, world, ident[0], "rccl")

            def step():
                sh = sharded.native_create(ctx, ncomm, seq.data_ptr(), offs.data_ptr(), n_reads, params)
                st = sh.stats
                sh.free()
                return st
            return step
        if mode == "torch":
            comm = sharded.TorchDistComm(dev)

            def step():
                # nothing is gathered: every rank ends the step with its 512-aligned slice of the seqset in HBM
                out = sharded.sharded_create(comm, {rank: ctx}, {rank: (seq, offs, n_reads)}, params, gather=False)
                sl, st = out[rank]
                sl.free()
                return st
            return step
        return fused_step

    if world > 1 or args.force_sharded:
        # The sharded build has run between a rank and itself and between processes sharing one GPU, never between two devices
        # before the driver's scaling run: a mode that cannot start on every rank (an exception, not a hang) gives way to the
        # next -- the library's driver, the Python orchestration over torch.distributed, then N independent fused builds
        # ("replicas", labelled as such in config.parallelism) -- instead of ending the run without a line.
        step = None
        for mode in (["native", "torch"] if native else ["torch"]) + ["replicas"]:
            err = None
            try:
                if mode == "replicas" and world > 1:
                    # every rank its own whole sample of the shape (a rank's share of the big sample is 1 / world of its coverage)
                    seq = offs = None
                    torch.cuda.empty_cache()
                    genome_len = args.genome_len or synth.SHAPES[args.shape]
                    seq, offs, n_reads, genome = synth.make_reads(genome_len=genome_len, coverage=args.coverage, read_len=args.read_len,
                                                                  seed=args.seed + rank, device=dev, shard=(0, 1), return_genome=True,
                                                                  contigs=synth.contig_lengths(genome_len, args.shape))
                    torch.cuda.synchronize()
                    n_bases = int(seq.numel())
                step = make_step(mode)
                if mode != "replicas":
                    step()  # (an extra, untimed step: the probe)
            except Exception as e:  # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, str(e)[:300])
            flag = torch.tensor([1.0 if err else 0.0], device=dev, dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() == 0.0:
                sharded_mode = mode
                break
            if rank == 0 or err:
                print("bench.py: rank %d: the %s sharded build did not start (%s); trying the next mode" % (rank, mode, err or "another rank failed"),
                      file=sys.stderr)
            sharded_fallback = "%s: %s" % (mode, err or "failed on another rank")
            ncomm = comm = None
    else:
        step = fused_step

    stats = None
    for _ in range(args.warmup):
        stats = step()
    ctx.set_profiling(True)  # HIP events around every kernel family, on the stream the kernels run on
    barrier()
    comm0 = ncomm.stats() if ncomm is not None else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step()
    barrier()
    elapsed = time.perf_counter() - t0
    comm_line = None
    if ncomm is not None:
        c1 = ncomm.stats()
        per = lambda k: (c1[k] - comm0[k]) / max(args.steps, 1)  # noqa: E731
        tb_ = torch.tensor([per("bytes_through_backend")], device=dev, dtype=torch.float64)
        dist.all_reduce(tb_, op=dist.ReduceOp.SUM)
        comm_line = {"backend": {0: "rccl", 1: "shm", 2: "none (a rank alone: copies)"}[c1["backend"]], "rccl_version": c1["rccl_version"],
                     "collectives_per_step_rank0": per("collectives"), "bytes_through_rccl_per_step_all_ranks": int(tb_.item()),
                     "bytes_kept_local_per_step_rank0": int(per("bytes_kept_local")),
                     "host_seconds_in_comm_calls_per_step_rank0": round(per("seconds_in_calls"), 4),
                     "no_shortcut": os.environ.get("BG_COMM_NO_SHORTCUT") == "1",
                     "path": "bg_create_sharded over bg_comm (csrc/sharded.hip, csrc/comm.hip: grouped ncclSend / ncclRecv, pieces <= 1 GiB - 4 KiB)"}
    timings = ctx.timings()
    ctx.set_profiling(False)
    mem = ctx.memory()
    try:
        mem["torch_peak"] = 0 if shared_arena else int(torch.cuda.max_memory_allocated(dev))
    except Exception:
        mem["torch_peak"] = 0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([n_bases], device=dev, dtype=torch.float64)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = float(tb.item())
    else:
        total_bases = float(n_bases)

    sharded_rm = None
    if sharded_mode in ("native", "torch") and not args.no_sharded_readmap:
        # What `biograph create` ends with, outside the timed region: the same build once more with the sample's READMAP
        # sharded like the seqset (routed entry look-ups, rows owned by their entry's rank, the claim hops as exchanged
        # claims), then the ranks' columns put together on rank 0 (bg_readmap_from_arrays).
        barrier()
        t1 = time.perf_counter()
        if ncomm is not None:
            sh = sharded.native_create(ctx, ncomm, seq.data_ptr(), offs.data_ptr(), n_reads, params, readmap="paired")
            barrier()
            t2 = time.perf_counter()
            sharded_rm = {"read_ids": sh.stats["n_read_ids"], "entries": sh.stats["n_entries"],
                          "entry_ids_need_bits": int(sh.stats["n_entries"]).bit_length(),
                          "build_with_readmap_ms": round((t2 - t1) * 1e3, 1), "seqset_only_ms": round(elapsed / max(args.steps, 1) * 1e3, 1),
                          "note": "untimed leg: one bg_create_sharded with the sample's readmap (its columns stay sharded; "
                                  "bg_sharded_write gathers them on rank 0)"}
            sh.free()
        else:
            out = sharded.sharded_create(comm, {rank: ctx}, {rank: (seq, offs, n_reads)}, params, gather=False, readmap="paired")
            sl, st_rm = out[rank]
            barrier()
            t2 = time.perf_counter()
            cols = sl.readmap_columns
            rm = sharded.gather_readmap(comm, {rank: ctx}, {rank: cols}, "bench")[rank]
            barrier()
            t3 = time.perf_counter()
            sharded_rm = {"read_ids": int(cols.n_total), "entries": int(cols.n_entries), "entry_ids_need_bits": int(cols.n_entries).bit_length(),
                          "build_with_readmap_ms": round((t2 - t1) * 1e3, 1), "seqset_only_ms": round(elapsed / max(args.steps, 1) * 1e3, 1),
                          "gather_on_rank0_ms": round((t3 - t2) * 1e3, 1),
                          "note": "untimed leg: one build with readmap='paired' (sharded_readmap), then gather_readmap on rank 0"}
            if rm is not None:
                rm.free()
            sl.free()
            del out, cols
    e2e = None
    if world == 1 and not args.force_sharded and not args.no_end_to_end:
        e2e = end_to_end_leg(seq, n_reads, args.read_len, dev, args, genome=None if args.no_discover else genome)
    discover = None
    if world == 1 and not args.no_discover:
        discover = discover_leg(ctx, seq, offs, n_reads, params, genome, dev, args)
    if rank == 0:
        # dominant kernel family by device time
        fam = max(timings, key=lambda t: t["ms"]) if timings else None
        roofline = None
        if fam and fam["launches"]:
            avg_ms = fam["ms"] / fam["launches"]
            bytes_per_launch = fam["bytes"] / fam["launches"]
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": fam["name"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_launch_ms": round(avg_ms, 4), "launches_per_step": fam["launches"] / max(args.steps, 1),
                        "algorithmic_bytes_per_launch": int(bytes_per_launch),
                        # what actually limits the kernel (SQ counters via tools/pmc_kernel.sh):
                        # the super-k-mer kernels issue one wave64 VALU instruction per CU clock, the rest wait on
                        # dependent random 128-byte lines
                        "limiter": LIMITER_OF_FAMILY.get(fam["name"], "hbm random access (latency / line granularity)")}
        if roofline and args.shape == "chr20" and not args.genome_len:
            roofline.update(pmc_traffic(roofline["kernel"], roofline["avg_launch_ms"]))
        # `achieved`, `frac` and `algorithmic_bytes_per_launch` follow SURVEY.md section 8(d): the bytes the kernel's STAGE
        # stands for in the survey's formula (S4 = 28 bytes per suffix record, the reference's 14-byte record written and read
        # once) over the launch time.  What the kernel must move as the data is laid out here (its rows, its outputs) is kept
        # beside it as `kernel_layout_bytes_per_launch` / `frac_kernel_layout`.
        if roofline:
            s8 = SURVEY_8D_BYTES.get(roofline["kernel"])
            roofline["kernel_layout_bytes_per_launch"] = roofline["algorithmic_bytes_per_launch"]
            roofline["frac_kernel_layout"] = roofline["frac"]
            if s8:
                b8 = s8(stats) / max(roofline["launches_per_step"], 1.0)
                ach = b8 / (roofline["avg_launch_ms"] * 1e-3) / 1e9
                roofline["algorithmic_bytes_per_launch"] = int(b8)
                roofline["achieved"] = round(ach, 1)
                roofline["frac"] = round(ach / HBM_PEAK_GBS, 4)
                roofline["bytes_formula"] = "SURVEY.md 8(d): " + SURVEY_8D_FORMULA.get(roofline["kernel"], "")
                # schema 2 (since round 5): `frac` / `achieved` price the kernel by the survey's bytes of its STAGE (which a stage's
                # other kernels are charged with as well); `frac_bytes_moved` (= frac_kernel_layout) is what the kernel itself moves;
                # the stage's bytes over the time of ALL its kernels is `stage_frac_survey_8d` -- the figure no kernel can flatter.
                roofline["schema"] = 2
                roofline["frac_bytes_moved"] = roofline["frac_kernel_layout"]
                if roofline["kernel"].startswith("skm_"):
                    roofline["note"] = ("SURVEY 8(d) prices the k-mer stage as one 8-byte record per k-mer written and read once; this kernel is the "
                                        "writing half, but what it writes is super-k-mers (1.6 bytes a k-mer): the counters show fewer bytes than the "
                                        "formula (frac_kernel_layout is what it moves), and it is bound by vector-instruction issue, not by HBM "
                                        "(DESIGN.md section 5); the stage as a whole: build_roofline.stages_survey_8d")
        total_ms = sum(t["ms"] for t in timings) or 1.0
        kernels = {t["name"]: {"ms_per_step": round(t["ms"] / max(args.steps, 1), 3),
                               "launches_per_step": t["launches"] / max(args.steps, 1),
                               "GBps": round(t["bytes"] / (t["ms"] * 1e-3) / 1e9, 1) if t["ms"] > 0 else None}
                   for t in sorted(timings, key=lambda t: -t["ms"])}
        if roofline:
            for stage, prefixes in STAGE_OF_FAMILY:
                if roofline["kernel"].startswith(prefixes):
                    row = stage_table(kernels, stats, args.steps).get(stage)
                    if row:
                        roofline["stage"] = stage
                        roofline["stage_frac_survey_8d"] = row["frac_of_hbm_peak"]
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            cpu = cpu_baseline(ctx, args.cpu_sample_genome, args.coverage, args.read_len, args.seed, dev)
        # whole-build algorithmic bytes: 1.25 N_bases + 16 N_kmers + (0.5 N_bases + 8 N_kmers)
        # + 28 N_rec + 2.625 N_ent
        alg = (1.25 * stats["n_bases"] + 16 * stats["n_kmers"] + 0.5 * stats["n_bases"] + 8 * stats["n_kmers"]
               + 28 * stats["n_records"] + 2.625 * stats["n_entries"])
        step_s = elapsed / max(args.steps, 1)
        line = {
            "metric": "seqset_build_gbases_per_s", "value": round(total_bases * args.steps / elapsed / 1e9, 4),
            "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "synthetic 30x %s 150 bp paired reads: seqset build (create data path)"
                                   % (("%s-shaped" % args.shape) if not args.genome_len else
                                      "%d-base genome (--genome-len; contig shape of '%s')" % (genome_len, args.shape)),
                       "genome_len_total": genome_len, "coverage": args.coverage, "read_len": args.read_len,
                       "reads_per_gpu": n_reads, "bases_per_gpu": n_bases, "kmer_size": 30, "min_kmer_count": 5,
                       "sharded_mode": sharded_mode,
                       "parallelism": "1 GPU (fused build)" if world == 1 and not args.force_sharded else
                       ("%d independent fused builds, one per GPU (replicas: the sharded build did not start -- %s)" % (world, sharded_fallback))
                       if sharded_mode == "replicas" else
                       "%d rank(s), ONE sample sharded (routed; only the kept k-mer set is copied to every rank, when it fits a "
                       "sixteenth of the device): all_to_all over RCCL of super-k-mer records (+ presence bits back), all_gather "
                       "of the kept k-mers (8 bytes each; BG_SHARD_KMERSET=routed: correction probes + answers instead), oriented "
                       "reads by leading bases, cover candidates per round (+ covered / keep bytes), entry tails (+ need-full "
                       "bytes); all_gather otherwise only of shard boundaries, cover filters and a few counters" % world},
            "roofline": roofline, "cpu_baseline": cpu,
            "device_memory": {"peak_live_GB": round(mem["peak"] / 1e9, 2), "cached_GB": round(mem["cached"] / 1e9, 2),
                              "torch_peak_GB": round(mem["torch_peak"] / 1e9, 2),
                              "shared_arena": shared_arena,
                              "note": "rank 0; shared_arena: torch allocates from the library's arena and peak_live is the "
                                      "rank's whole footprint; else peak_live = the library's allocator, torch_peak = input "
                                      "reads + exchange buffers, and their sum bounds the footprint"},
            "rccl_ranks": world if dist is not None else 0, "comm": comm_line,
            "build_roofline": {"algorithmic_bytes_per_step": int(alg), "bytes_per_base": round(alg / max(stats["n_bases"], 1), 2),
                               "achieved_GBps": round(alg / step_s / 1e9, 1),
                               "frac_of_hbm_peak": round(alg / step_s / 1e9 / (HBM_PEAK_GBS * world), 5),
                               "kernel_time_frac_of_wall": round(total_ms / 1e3 / elapsed, 3),
                               "stages_survey_8d": stage_table(kernels, stats, args.steps)},
            "end_to_end_from_fastq": e2e, "discover": discover, "sharded_readmap": sharded_rm, "counters": stats, "kernels": kernels,
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if ncomm is not None:
        ncomm.free()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
