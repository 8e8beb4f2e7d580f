"""Randomised differential tests of the two rows added in round 2 (python tests/fuzz_population_tracer.py [seconds] [seed]):

  default            bg_push_trace (csrc/tracer.cpp over the device walks) against the oracle's restatement of the
                     reference's push tracer (oracle/tracer.py): random genomes with repeats, a diploid sample with SNPs /
                     indels, read errors, gapped scaffolds, random assemble_options (overlap, pair distance, read-ahead
                     distance, rejoin / step limits, ambiguous-reference switches), both directions, whole scaffold and
                     random chunks -- every assembly, in order, with anchors, pair matches, scores and step counts.
  FUZZ_POPULATION=1  the sharded merge (1-5 simulated ranks, 2-4 samples, mixed read lengths) against the single-GPU
                     merge, and a sample's readmap built on the merged seqset among the marked entries against the
                     migration of its own readmap.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import biograph_amd as bg  # noqa: E402
from biograph_amd import sharded, synth, tracer  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import align as A  # noqa: E402
from oracle import coverage as CV  # noqa: E402
from oracle import tracer as T  # noqa: E402


def full_key(a):
    return a.key() + (sorted(a.left_pair_matches), sorted(a.right_pair_matches), a.score, a.trace_steps,
                      a.left_anchor_ambiguous_bases)


def tracer_case(ctx, seed):
    rng = np.random.default_rng(seed)
    G = int(rng.integers(1500, 7000))
    rl = int(rng.choice([60, 80, 100, 150]))
    cov = int(rng.integers(8, 30))
    repeats = (int(rng.integers(80, 400)), int(rng.integers(2, 5)), float(rng.choice([0.0, 0.01, 0.03]))) if rng.random() < 0.5 else None
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=cov, read_len=rl, seed=seed, device="cpu", return_genome=True,
                                            err=float(rng.choice([0.0, 0.001, 0.004])), n_rate=0.0, repeats=repeats,
                                            snp_rate=float(rng.choice([0.001, 0.004])), indel_rate=float(rng.choice([1e-4, 1e-3])))
    reads = O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
    pairs = list(zip(reads[0::2], reads[1::2]))
    singles = []
    if rng.random() < 0.3:  # some reads lose their mate
        cut = int(rng.integers(1, max(2, len(pairs) // 4)))
        singles = [p[0] for p in pairs[:cut]]
        pairs = pairs[cut:]
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.numpy()].tobytes().decode()
    # scaffold: 1-3 extents with gaps, possibly not starting at 0
    cuts = sorted(int(x) for x in rng.integers(100, G - 100, int(rng.integers(0, 3))))
    parts, at = [], int(rng.integers(0, 50)) if rng.random() < 0.3 else 0
    for c in cuts + [G]:
        gap = int(rng.integers(1, 60))
        if c - at > gap + 30:
            parts.append((at, ref[at:c - gap]))
        at = c
    if not parts:
        parts = [(0, ref)]
    flat = [r for p in pairs for r in p] + singles
    packed = ctx.pack_reads(flat)
    ss, _ = ctx.build_seqset(packed, 30)
    rm_reads = []
    for a, b in pairs:
        rm_reads += [a, b]
    for r in singles:
        rm_reads += [r, ""]
    rm = ctx.build_readmap(ss, ctx.pack_reads(rm_reads), True)
    rmap = tracer.reference_ref_map(ss, [ref])
    sq, of = O.pack_strings(flat)
    ent = O.Entries.build(sq, of, "simple")
    oss = O.Seqset.from_entries(ent)
    orm = O.Readmap(oss, ent, [tuple(p) for p in pairs] + [(r,) for r in singles], True)
    codes = np.array([b"ACGT".index(bytes([c])) for c in ref.encode()], np.uint8)
    g = T.Graph(oss, orm, oss.ref_map_build(codes, np.array([0], np.uint64), np.array([len(ref)], np.uint64)))
    kw = dict(min_overlap=int(rl * float(rng.choice([0.4, 0.6, 0.7, 0.85]))),
              max_pair_distance=int(rng.choice([150, 400, 1000])),
              read_ahead_distance=int(rng.choice([200, 1000, 100000])),
              max_rejoins=int(rng.choice([1, 2, 5])), max_search_steps=int(rng.choice([30, 200, 1000])),
              max_ambiguous_search_steps=int(rng.choice([5, 100])), initial_search_steps=int(rng.choice([5, 100])),
              max_next_paths=int(rng.choice([4, 1024])), max_branches_between_pairs=int(rng.choice([0, 2, 5])),
              max_bases_between_pairs=int(rng.choice([40, 300])), max_ambiguous_bases=int(rng.choice([20, 300])),
              trace_ambiguous_ref=int(rng.random() < 0.3), ignore_ambiguous_ref_pairs=int(rng.random() < 0.7),
              trace_dead_ends=int(rng.random() < 0.8), forward_pairs_face_inward=int(rng.random() < 0.8),
              trace_reference_assemblies=int(rng.random() < 0.7), max_pairs_per_read=int(rng.choice([1, 20])))
    akw = dict(anchor_drop=int(rng.random() < 0.8), min_anchor_drop_overlap=int(rng.choice([5, 15, 30])),
               max_anchor_kmer_size=int(rng.choice([2, 12, 31])), ref_align_factor=int(rng.choice([2, 4])),
               max_ref_align_bases=int(rng.choice([10, 30])), output_ml_features=int(rng.random() < 0.5))
    o_gpu = tracer.TraceOptions(**kw, **akw)
    o_cpu = T.Options(**{k: (bool(v) if k in ("trace_ambiguous_ref", "ignore_ambiguous_ref_pairs", "trace_dead_ends",
                                              "forward_pairs_face_inward", "trace_reference_assemblies") else v)
                         for k, v in kw.items()})
    sc = T.Scaffold(parts)
    end = sc.end_pos
    ok = True
    ad = None
    whole_want, whole_got = [], []
    if akw["anchor_drop"]:
        ao = A.AlignOptions(min_overlap=kw["min_overlap"], read_ahead_distance=kw["read_ahead_distance"],
                            trace_reference_assemblies=bool(kw["trace_reference_assemblies"]),
                            min_anchor_drop_overlap=akw["min_anchor_drop_overlap"], max_kmer_size=akw["max_anchor_kmer_size"],
                            ref_align_factor=akw["ref_align_factor"], max_ref_align_bases=akw["max_ref_align_bases"])
        ad = lambda s_, emit: A.AnchorDropper(ao, s_, emit)  # noqa: E731
    try:
        for rev in (False, True):
            ranges = [(0, None)]
            a = int(rng.integers(0, end))
            ranges.append((a, min(end, a + int(rng.integers(50, 2500)))))
            for start, limit in ranges:
                want = T.reversable_trace(g, sc, o_cpu, rev_comp=rev, start_offset=start, limit_offset=limit, anchor_drop=ad)
                got = tracer.push_trace(ss, rm, rmap, parts, start=start, limit=limit, rev_comp=rev, options=o_gpu)
                if (start, limit) == (0, None):
                    whole_want += want
                    whole_got += got
                if [full_key(x) for x in got] != [full_key(x) for x in want]:
                    print("TRACER MISMATCH seed", seed, "rev", rev, "range", start, limit, kw, "G", G, "rl", rl, akw, "parts",
                          [(o, len(s)) for o, s in parts], flush=True)
                    for i, (x, y) in enumerate(zip(got, want)):
                        if full_key(x) != full_key(y):
                            print("  first difference at", i, "\n   gpu   ", full_key(x), "\n   oracle", full_key(y), flush=True)
                            break
                    else:
                        print("  lengths", len(got), len(want), flush=True)
                    ok = False
                    break
            if not ok:
                break
        if ok and akw["anchor_drop"]:
            ok = pipeline_and_coverage(rng, seed, g, orm, sc, parts, ss, rm, kw, akw, o_gpu, whole_want, whole_got)
    finally:
        rmap.free()
    return ok, sum(1 for _ in [0])


COUNTS = {"sections": 0, "covered": 0, "vcf_lines": 0, "refused": 0}


def pipeline_and_coverage(rng, seed, g, orm, sc, parts, ss, rm, kw, akw, o_gpu, want_in, got_in):
    """Both directions of the whole scaffold through the serial pipeline, calc_coverage, simple_genotype_filter and the
    VCF record line: library == oracle (oracle/align.py, oracle/coverage.py)."""
    ao = A.AlignOptions(min_overlap=kw["min_overlap"], read_ahead_distance=kw["read_ahead_distance"],
                        trace_reference_assemblies=bool(kw["trace_reference_assemblies"]),
                        min_anchor_drop_overlap=akw["min_anchor_drop_overlap"], max_kmer_size=akw["max_anchor_kmer_size"],
                        ref_align_factor=akw["ref_align_factor"], max_ref_align_bases=akw["max_ref_align_bases"])
    try:
        _, want_sections = A.variants_pipeline(ao, sc, want_in, through_vcf_pad=True, ml_features=bool(akw["output_ml_features"]))
    except AssertionError:
        want_sections = None  # a CHECK of the reference: the library must refuse as well
    try:
        got_sections = tracer.process_assemblies(got_in, parts, end_pos=sc.end_pos, options=o_gpu)
    except Exception:
        got_sections = None
    if (want_sections is None) != (got_sections is None):
        print("PIPELINE MISMATCH seed", seed, "one side refused", want_sections is None, got_sections is None, flush=True)
        return False
    if want_sections is None:
        COUNTS["refused"] += 1
        return True
    COUNTS["sections"] += len(want_sections)

    def skey(a):
        ml = getattr(a, "ml", None)
        return (a.left_offset, a.right_offset, a.seq, bool(a.matches_reference), sorted(a.left_pair_matches),
                sorted(a.right_pair_matches), a.score, a.min_overlap, a.trace_steps, getattr(a, "merged", 0),
                None if ml is None else tuple(ml[:4]) + (round(ml[4], 6), round(ml[5], 6), ml[6]))
    if [skey(a) for a in got_sections] != [skey(a) for a in want_sections]:
        print("PIPELINE MISMATCH seed", seed, kw, akw, len(got_sections), len(want_sections), flush=True)
        return False
    ckw = dict(max_coverage_paths=int(rng.choice([1, 2, 4, 64])), penalize_directional_coverage=int(rng.random() < 0.7),
               min_depth_portion=float(rng.choice([0.05, 0.23, 0.4])), min_avg_pair_depth=float(rng.choice([0, 1, 3])),
               min_read_depth=int(rng.choice([0, 1, 3])), vcf_sv_size_threshold=int(rng.choice([5, 20])),
               output_ml_features=akw["output_ml_features"])
    co = CV.CoverageOptions(**{k: (bool(v) if k in ("penalize_directional_coverage", "output_ml_features") else v) for k, v in ckw.items()})
    for filt in (False, True):
        try:
            want = CV.calc_coverage(g, [int(x) for x in orm.get("read_len")], sc, co, want_sections)
            if filt:
                want = CV.simple_genotype_filter(co, want)
        except AssertionError:
            want = None
        try:
            got, lines = tracer.assemblies_coverage(ss, rm, got_sections, parts, end_pos=sc.end_pos,
                                                    options=tracer.CoverageOptions(**ckw), genotype_filter=filt, vcf_name="s")
        except Exception:
            got = None
        if (want is None) != (got is None):
            print("COVERAGE MISMATCH seed", seed, "one side refused", want is None, got is None, ckw, flush=True)
            return False
        if want is None:
            continue

        def ckey(a):
            return (a.left_offset, a.right_offset, a.seq, list(a.coverage), list(a.pair_coverage), a.other_depth,
                    a.other_pair_depth, a.ref_depth) + ((a.strand_count, round(a.genotype_quality, 12)) if filt else ())
        if [ckey(a) for a in got] != [ckey(a) for a in want]:
            print("COVERAGE MISMATCH seed", seed, ckw, kw, akw, len(got), len(want), flush=True)
            for x, y in zip(got, want):
                if ckey(x) != ckey(y):
                    print("   gpu   ", ckey(x), "\n   oracle", ckey(y), flush=True)
                    break
            return False
        COUNTS["covered"] += len(want)
        if filt:
            COUNTS["vcf_lines"] += len(want)

            def sub(off, n):
                out = ["N"] * n
                for o_, s_ in parts:
                    for p_ in range(max(off, o_), min(off + n, o_ + len(s_))):
                        out[p_ - off] = s_[p_ - o_]
                return "".join(out)
            if lines != [CV.vcf_line(co, "s", sub, a) for a in want]:
                print("VCF LINE MISMATCH seed", seed, ckw, flush=True)
                return False
    return True


def population_case(ctx, seed):
    rng = np.random.default_rng(seed)
    P = int(rng.integers(1, 6))
    n_samples = int(rng.integers(2, 5))
    G = int(rng.integers(800, 5000))
    comm = sharded.LocalSimComm(P)
    ctxs = {r: ctx for r in range(P)}
    rows = {r: [] for r in range(P)}
    singles, max_len = [], 0
    k = int(rng.choice([14, 20, 30]))
    prm = ctx.default_params(kmer_size=k, min_kmer_count=int(rng.choice([1, 2, 5])))
    for s in range(n_samples):
        rl = int(rng.choice([60, 100, 150]))
        every, inputs = [], {}
        for r in range(P):
            seq, offs, n = synth.make_reads(genome_len=G, coverage=int(rng.integers(10, 30)), read_len=rl, seed=seed % 100000,
                                            device="cpu", shard=(r, P), sample=s, err=float(rng.choice([0.0, 0.003])))
            every += O.unpack_strings(seq.numpy(), offs.numpy().astype(np.uint64))
            inputs[r] = (seq.cuda(), offs.cuda(), n)
        out = sharded.sharded_create(comm, ctxs, inputs, prm, gather=False, keep_entry_rows=True)
        max_len = max(max_len, out[0][1]["max_len"])
        r_ = ctx.pack_reads(every)
        ks = ctx.kmer_count(r_, prm.kmer_size, prm.min_kmer_count)
        cor, _ = ctx.correct_reads(r_, ks, prm.correct.max_corrections, prm.correct.min_good_run, prm.correct.trim_after_portion)
        ss, _ = ctx.build_seqset(cor, prm.kmer_size)
        singles.append((ss, ctx.build_readmap(ss, cor, True), cor))
        for r in range(P):
            rows[r].append(out[r][0].entry_rows)
            out[r][0].free()
    want, maps = ctx.merge_seqsets([x[0] for x in singles])
    got = sharded.sharded_merge(comm, ctxs, rows, max_len, gather=True)[0][0]
    if got.size() != want.size() or any(got.part(p) != want.part(p) for p in bg.PART_NAMES):
        print("MERGE MISMATCH seed", seed, "P", P, "samples", n_samples, "G", G, "k", k, flush=True)
        return False, 0
    i = int(rng.integers(0, n_samples))
    marks = np.zeros(want.size(), np.uint8)
    marks[maps[i]] = 1
    dm = torch.from_numpy(marks).cuda()
    mig = ctx.migrate_readmap(singles[i][1], maps[i], want)
    if ctx.build_readmap(want, singles[i][2], True, marks=dm).parts() != mig.parts():
        print("MARKED READMAP MISMATCH seed", seed, "sample", i, flush=True)
        return False, 0
    return True, 1


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = bg.Context(0)
    case = population_case if os.environ.get("FUZZ_POPULATION") else tracer_case
    t0 = time.time()
    n = int(os.environ.get("FUZZ_FIRST", "0"))
    done = 0
    while time.time() - t0 < budget:
        ok, _ = case(ctx, seed0 * 100000 + n)
        if not ok:
            return 1
        n += 1
        done += 1
    print("%d random %s cases equal (%.0f s)" % (done, "population" if case is population_case else "tracer", time.time() - t0),
          flush=True)
    if case is not population_case:
        print("  behind the tracers: %(sections)d pipeline sections, %(covered)d coverage results, %(vcf_lines)d VCF lines equal; "
              "%(refused)d inputs refused by both sides" % COUNTS, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
