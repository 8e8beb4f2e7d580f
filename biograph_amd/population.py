"""The population configuration (BASELINE.json configs[4]): several samples of one genome,
each built sharded over the ranks, merged into one seqset without leaving the GPUs, and discovered per sample over
the merged seqset with the reference cut into chunks that the ranks share out.

  build    sharded.sharded_create per sample (keep_entry_rows, keep_corrected)         -> slices per rank
  merge    sharded.sharded_merge: the samples' entries through the records exchange     -> merged slices, written
           in place into one container (sharded.write_slices); = `biograph merge` (biograph_merge.cpp:189-290)
  readmaps every rank opens the merged seqset (a replica: discover reads it at random) and builds each sample's
           readmap on it from the all-gathered corrected reads.  The reference migrates the per-sample readmaps
           through the merge maps (make_readmap.cpp:455-519): a read goes to the image of ITS SAMPLE's entry, which
           need not be the first merged entry that has the read as a prefix.  The merge therefore leaves one byte per
           (sample, merged entry) -- "an entry of the sample maps here" (bg_seqset_mark_inputs) -- and the reads
           are looked up among the marked entries (bg_readmap_build_marked): same rows as the migration
           (tests/test_gpu_population.py), no per-entry map exchanged.
  discover trace_ref cuts every scaffold into scaffold_split_size pieces (trace_ref.cpp:27-63), orders them
           (trace_ref.cpp:77-105) and hands them to a thread pool; here the pieces go round-robin to the ranks, every
           rank runs the graph-discover pipeline on its pieces (no data-path collective), the calls are gathered at
           the end.

The communicators are those of sharded.py (LocalSimComm in tests, TorchDistComm = RCCL)."""
import re

import torch

from . import sharded


SCAFFOLD_SPLIT_SIZE = 1_000_000  # assemble_options::scaffold_split_size (assemble.h)


def _scaffold_order_key(name):
    """trace_ref::assemble's order of scaffolds (trace_ref.cpp:77-105): numeric names in numeric order, then the
    rest, shorter names first ("X", "Y" before "hs37d5"), equal lengths in DESCENDING name order."""
    # atoi_or_maxint = std::stoi (trace_ref.cpp:71-77): leading whitespace, an optional sign and the numeric PREFIX
    # count ("1_KI270706v1_random" -> 1, "22abc" -> 22); no digits or a value outside int -> maxint
    m = re.match(r"\s*[+-]?\d+", name)
    num = int(m.group(0)) if m else 2 ** 31 - 1
    if not -2 ** 31 <= num <= 2 ** 31 - 1:
        num = 2 ** 31 - 1
    return (num, len(name), tuple(-ord(c) for c in name))


def ref_chunks(scaffolds, split_size=SCAFFOLD_SPLIT_SIZE, read_ahead_distance=0, bidir=False):
    """The work list of trace_ref: scaffolds = [(name, start, limit)] (add_scaffold_range; limit is clipped by the
    caller to the scaffold's end) -> [(name, start, limit)] in execution order.  With the bidirectional tracer
    consecutive pieces overlap by read_ahead_distance (trace_ref.cpp:54-59)."""
    if split_size <= 0:
        raise ValueError("ref_chunks: split_size must be positive")
    if bidir and read_ahead_distance >= split_size:
        raise ValueError("ref_chunks: read_ahead_distance must be smaller than split_size")
    work = []
    for name, start, limit in scaffolds:
        if limit < start:
            raise ValueError("ref_chunks: scaffold %s: limit %d < start %d" % (name, limit, start))
        a = start
        while a < limit:
            lim = min(limit, a + split_size)
            work.append((name, a, lim))
            a = min(lim, a + split_size - read_ahead_distance) if bidir else lim
    work.sort(key=lambda w: (_scaffold_order_key(w[0]), w[1]))
    return work


def chunks_of_rank(work, rank, world):
    """Round-robin: neighbouring pieces (similar cost) land on different ranks."""
    return work[rank::world]


def extents_of(seq):
    """The N-free stretches of a scaffold (the reference's supercontigs): [(offset, bases)]."""
    ext, run = [], 0
    for i, c in enumerate(seq + "N"):
        if c in "ACGT":
            run += 1
            continue
        if run:
            ext.append((i - run, seq[i - run:i]))
        run = 0
    return ext


def discover_chunk(ss, rm, extents, start, limit, min_overlap, max_pair_distance=1000):
    """The graph-discover pipeline (python/biograph tests' graph_discover flow; bgbinary discovery runs the same
    calls per scaffold) on reference positions [start, limit) -> [(left_offset, right_offset, seq)] of the anchored
    non-reference assemblies, scaffold coordinates."""
    from . import variants as V
    lst = V.AssemblyList().make_ref_assemblies(extents, start, limit)
    lst.graph_trim_ref(extents).dedup().update_rc_seqset_entries(ss)
    lst.discover_branch(ss, rm, min_overlap, "BRANCH")
    lst.graph_trim_ref(extents).dedup().update_rc_seqset_entries(ss)
    lst.discover_push_to_pair(ss, rm, min_overlap, max_pair_distance, "PUSH_TO_PAIR", ["BRANCH"])
    lst.graph_trim_ref(extents).dedup().update_rc_seqset_entries(ss)
    return [(a.left_offset, a.right_offset, a.seq) for a in lst.to_python(False)
            if a.left_offset is not None and a.right_offset is not None and not a.matches_reference]


def push_trace_chunk(session, extents, end_pos, start, limit):
    """One work item of trace_ref with the push tracer (trace_ref.cpp:142-206 without the pop passes): forward and
    reverse -> [(left_offset, right_offset, seq)] of the non-reference assemblies, scaffold coordinates."""
    out = []
    for rev in (False, True):
        for a in session.trace(extents, start=start, limit=limit, end_pos=end_pos, rev_comp=rev):
            if not a.matches_reference:
                out.append((a.left_offset, a.right_offset, a.seq))
    return out


def sharded_discover(comm, seqsets, readmaps, scaffolds, min_overlap, split_size=SCAFFOLD_SPLIT_SIZE,
                     max_pair_distance=1000, gather=True, engine="graph", ref_maps=None):
    """seqsets[r] = rank r's replica of the (merged) seqset, readmaps[r] = {sample: Readmap on that replica},
    scaffolds = {name: bases}.  Work items = (sample, chunk), shared out round-robin.  Returns, on every rank,
    {sample: [(scaffold, left, right, seq)]} sorted (gather=True), or each rank's own calls {r: {...}}.
    engine: "graph" = the graph-discover pipeline per chunk; "push" = the production push tracer (biograph_amd.tracer),
    which needs ref_maps[r] = the ref_map of rank r's replica over the whole reference (tracer.reference_ref_map)."""
    ranks = comm.local_ranks
    samples = sorted(readmaps[ranks[0]])
    chunks = ref_chunks([(name, 0, len(seq)) for name, seq in scaffolds.items()], split_size)
    work = [(s, c) for s in samples for c in chunks]
    ext = {name: extents_of(seq) for name, seq in scaffolds.items()}
    if engine not in ("graph", "push"):
        raise ValueError("sharded_discover: engine must be 'graph' or 'push'")
    if engine == "push" and ref_maps is None:
        raise ValueError("sharded_discover: the push tracer needs ref_maps")
    mine = {}
    for r in ranks:
        calls = {s: [] for s in samples}
        sessions = {}
        for s, (name, a, lim) in chunks_of_rank(work, r, comm.world):
            if engine == "push":
                from . import tracer
                if s not in sessions:
                    sessions[s] = tracer.PushTracer(seqsets[r], readmaps[r][s], ref_maps[r],
                                                    tracer.TraceOptions(min_overlap=min_overlap, max_pair_distance=max_pair_distance))
                found = push_trace_chunk(sessions[s], ext[name], len(scaffolds[name]), a, lim)
            else:
                found = discover_chunk(seqsets[r], readmaps[r][s], ext[name], a, lim, min_overlap, max_pair_distance)
            for lo, ro, sq in found:
                calls[s].append((name, lo, ro, sq))
        for t in sessions.values():
            t.close()
        mine[r] = calls
    if not gather:
        return mine
    every = comm.all_gather_obj(mine)[ranks[0]]
    return {s: sorted(x for part in every for x in part[s]) for s in samples}


_ASM_FIELDS = ("left_offset", "right_offset", "seq", "left_anchor_len", "right_anchor_len", "min_overlap", "trace_steps",
               "left_anchor_ambiguous_bases", "matches_reference", "score", "left_pair_matches", "right_pair_matches",
               "from_pop", "assembly_id")


def sharded_discovery_vcf(comm, seqsets, readmaps, scaffolds, min_overlap=None, split_size=SCAFFOLD_SPLIT_SIZE,
                          max_pair_distance=1000, ref_maps=None, trace_options=None, coverage_options=None, pop_tracer=True):
    """The reference's discovery engine over a population (configs[4]): every rank traces its share of the work items
    (sample, reference piece) with the push tracer in both directions; the assemblies of a (sample, scaffold) go to
    the rank that owns that pair, which runs the serial pipeline, calc_coverage and the genotype filter on its replica
    and formats the VCF records (what one `bgbinary discovery --engine push` does per sample, with its defaults: the pop
    tracers over every work item's anchor drops (pop_tracer=False: push only), no simple genotype filter, rvg_exclude --
    `biograph_variants.cpp:188-202`; pass coverage_options to change the last two).  Returns, on every rank,
    {sample: [VCF record lines in scaffold order]}.  Assemblies cross ranks as Python objects (a few kB per piece):
    all_gather_obj; the GPU work (tracing, coverage walks) never leaves its rank.  The result does not depend on the
    number of ranks: the owner orders what it receives by (piece start, direction, position in the piece's output)."""
    from . import tracer
    ranks = comm.local_ranks
    samples = sorted(readmaps[ranks[0]])
    names = list(scaffolds)
    chunks = ref_chunks([(name, 0, len(seq)) for name, seq in scaffolds.items()], split_size)
    work = [(s, c) for s in samples for c in chunks]
    ext = {name: extents_of(seq) for name, seq in scaffolds.items()}
    if ref_maps is None:
        raise ValueError("sharded_discovery_vcf: the push tracer needs ref_maps")

    def options_for(rm):
        if trace_options is not None:
            return trace_options
        mo = min_overlap if min_overlap is not None else int(0.7 * rm.max_read_len)  # biograph_variants.cpp:674
        return tracer.TraceOptions(min_overlap=mo, max_pair_distance=max_pair_distance)

    traced = {}
    for r in ranks:
        mine, sessions = [], {}
        for s, (name, a, lim) in chunks_of_rank(work, r, comm.world):
            if s not in sessions:
                sessions[s] = tracer.PushTracer(seqsets[r], readmaps[r][s], ref_maps[r], options_for(readmaps[r][s]))
            # one work item of trace_ref: push forward, push reverse, then the two pop tracers over their anchor drops
            found = sessions[s].execute_work(ext[name], start=a, limit=lim, end_pos=len(scaffolds[name]), pop_fwd=pop_tracer,
                                             pop_rev=pop_tracer)
            for i, x in enumerate(found):
                # assembly ids come from a per-process counter: made unique over the ranks (rvg_exclude groups by them)
                x.assembly_id = x.assembly_id * comm.world + r
                mine.append(((s, name), (a, i), tuple(getattr(x, f) for f in _ASM_FIELDS)))
        for t in sessions.values():
            t.close()
        traced[r] = mine
    every = comm.all_gather_obj(traced)[ranks[0]]
    pairs = [(s, name) for s in samples for name in names]
    lines = {}
    for r in ranks:
        out = {}
        for k, (s, name) in enumerate(pairs):
            if k % comm.world != r:
                continue
            items = sorted((order, fields) for part in every for key, order, fields in part if key == (s, name))
            asms = [tracer.TraceAssembly(**dict(zip(_ASM_FIELDS, fields))) for _, fields in items]
            opt = options_for(readmaps[r][s])
            sections = tracer.process_assemblies(asms, ext[name], end_pos=len(scaffolds[name]), options=opt)
            cov = coverage_options if coverage_options is not None else tracer.CoverageOptions(simple_genotype_filter=0, rvg_exclude=1)
            _, recs = tracer.assemblies_coverage(seqsets[r], readmaps[r][s], sections, ext[name], end_pos=len(scaffolds[name]),
                                                 options=cov, vcf_name=name)
            out[(s, name)] = recs
        lines[r] = out
    merged = {}
    for part in comm.all_gather_obj(lines)[ranks[0]]:
        merged.update(part)
    return {s: [x for name in names for x in merged[(s, name)]] for s in samples}


def gathered_readmap(comm, ctxs, seqsets, corrected, marks=None, is_paired=True):
    """One sample's readmap on every rank's replica of the merged seqset.  marks[r] = uint8 [shard entries]: the
    sample's row of SeqsetSlice.input_marks (None: the seqset is the sample's own).  corrected[r] = (words, lens, max_len) of
    rank r's corrected reads (SeqsetSlice.corrected of sharded_create(keep_corrected=True)); read ids run through the
    ranks in rank order, as if the shards' FASTQ files had been given to one `biograph create` one after the other."""
    ranks = comm.local_ranks
    W = max(row[0] for row in comm.gather_ints({r: [int(corrected[r][0].shape[1])] for r in ranks}))
    max_len = max(row[0] for row in comm.gather_ints({r: [int(corrected[r][2])] for r in ranks}))

    def widen(t):
        if t.shape[1] == W:
            return t
        return torch.cat([t, torch.zeros((t.shape[0], W - t.shape[1]), dtype=t.dtype, device=t.device)], dim=1)

    gw = comm.all_gather_v({r: widen(corrected[r][0]) for r in ranks})
    gl = comm.all_gather_v({r: corrected[r][1] for r in ranks})
    gm = comm.all_gather_v({r: marks[r].contiguous() for r in ranks}) if marks is not None else None
    out = {}
    for r in ranks:
        dev = corrected[r][0].device
        words = torch.cat([g.to(dev) for g in gw[r]])
        lens = torch.cat([g.to(dev) for g in gl[r]])
        reads = sharded.import_reads(ctxs[r], words, lens, max_len)
        del words, lens
        m = torch.cat([g.to(dev) for g in gm[r]]).contiguous() if gm is not None else None
        if m is not None and m.shape[0] != seqsets[r].size():
            raise ValueError("gathered_readmap: %d marks for %d entries" % (m.shape[0], seqsets[r].size()))
        out[r] = ctxs[r].build_readmap(seqsets[r], reads, is_paired, marks=m)
        reads.free()
        del m
    return out


def population_create(comm, ctxs, sample_inputs, path, params=None, mode=None, argv=("bgbinary", "merge"), timings=False):
    """sample_inputs = {sample: inputs of sharded_create}.  Builds every sample, merges, writes the merged seqset
    to `path` (a file all ranks of the node can open; /dev/shm for a bench) and returns
    (per-rank replicas {r: Seqset}, {r: {sample: Readmap}}, stats).  timings: stats["seconds"] = wall time of the
    stages (device synchronised and ranks at a barrier between stages)."""
    import time
    ranks = comm.local_ranks
    rows = {r: [] for r in ranks}
    kept = {}
    stats = {"samples": {}, "seconds": {}}
    max_len = 0
    last = [None]

    def stage(name):
        if not timings:
            return
        torch.cuda.synchronize()
        comm.barrier()
        now = time.perf_counter()
        if last[0] is not None:
            stats["seconds"][last[0][0]] = stats["seconds"].get(last[0][0], 0.0) + now - last[0][1]
        last[0] = (name, now) if name else None

    stage("build")
    for name in sorted(sample_inputs):
        out = sharded.sharded_create(comm, ctxs, sample_inputs[name], params, mode=mode, gather=False,
                                     keep_entry_rows=True, keep_corrected=True)
        st = out[ranks[0]][1]
        stats["samples"][name] = st
        max_len = max(max_len, st["max_len"])
        kept[name] = {}
        for r in ranks:
            sl = out[r][0]
            rows[r].append(sl.entry_rows)
            kept[name][r] = sl.corrected
            sl.entry_rows = sl.corrected = None
            sl.free()
    stage("merge")
    merged = sharded.sharded_merge(comm, ctxs, rows, max_len, gather=False)
    del rows
    slices = {r: merged[r][0] for r in ranks}
    marks = {r: slices[r].input_marks[0] for r in ranks}
    stats["merge"] = merged[ranks[0]][1]
    stage("write")
    sharded.write_slices(comm, slices, path, list(argv))
    for sl in slices.values():
        sl.free()
    comm.barrier()
    stage("open_replicas")
    replicas = {r: ctxs[r].open_seqset(path) for r in ranks}
    stage("readmaps")
    readmaps = {r: {} for r in ranks}
    for i, name in enumerate(sorted(sample_inputs)):  # input i of the merge = the i-th name
        rms = gathered_readmap(comm, ctxs, replicas, kept[name], {r: marks[r][i] for r in ranks})
        kept[name] = None
        for r in ranks:
            readmaps[r][name] = rms[r]
    stage(None)
    return replicas, readmaps, stats
