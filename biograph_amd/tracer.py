"""Row f.2: the reference's production push tracer behind the C ABI (csrc/tracer.cpp, kernels in csrc/walk.hip) and the
head of the assembly pipeline behind it (csrc/align.cpp).  `push_trace` = reversable_tracer::assemble(start, limit)
(variants/reversable_tracer.cpp:76-145 -> tracer::assemble, tracer.cpp:195-318): tracer -> anchor_dropper ->
pair_counter.  `process_assemblies` = the serial pipeline of variants/pipeline.cpp:40-57,92-95 up to the
align_splitter.  The work list of a whole reference is population.ref_chunks (trace_ref.cpp:27-105)."""
import ctypes as C

import numpy as np

from . import capi

_I64_FIELDS = ["ambiguous_branch_cost", "rejoin_local_cost", "base_cost", "increase_max_between_pair_cost", "max_cost",
               "pairs_used_cost", "traverse_ref_cost", "decrease_overlap_cost", "size_change_cost", "dead_end_cost",
               "anchor_drop_score", "pair_match_score", "min_overlap_score"]
_U64_FIELDS = ["max_pairs_per_read", "max_next_paths", "max_ambiguous_bases", "min_overlap", "read_ahead_distance",
               "scaffold_split_size", "max_branches_between_pairs", "max_rejoins", "max_bases_between_pairs",
               "max_search_steps", "max_ambiguous_search_steps", "initial_search_steps", "max_search_steps_per_read",
               "max_pair_distance"]
_INT_FIELDS = ["forward_pairs_face_inward", "ignore_ambiguous_ref_pairs", "trace_ambiguous_ref", "trace_dead_ends",
               "only_trace_forward", "trace_reference_assemblies"]


class TraceOptions(C.Structure):
    """bg_trace_options = the assemble_options the push tracer reads (variants/assemble.h:475-720)."""
    _fields_ = ([(n, C.c_int64) for n in _I64_FIELDS] + [(n, C.c_uint64) for n in _U64_FIELDS] +
                [(n, C.c_int) for n in _INT_FIELDS] + [("anchor_drop_size_multiplier", C.c_double),
                                                        ("min_anchor_drop_overlap", C.c_uint64),
                                                        ("ref_align_factor", C.c_int32), ("max_ref_align_bases", C.c_int32),
                                                        ("anchor_drop", C.c_int32), ("max_anchor_kmer_size", C.c_int32),
                                                        ("output_ml_features", C.c_int32), ("min_pop_overlap", C.c_uint64),
                                                        ("pop_tracer_offset_slop", C.c_uint64), ("min_pair_distance", C.c_int64)])

    def __init__(self, **kw):
        super().__init__()
        capi.lib().bg_trace_options_default(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise AttributeError("no assemble option %r" % k)
            setattr(self, k, v)


_STAT_FIELDS = ["traces", "ref_reads", "ambiguous_ref_reads", "step_count", "output_count", "found_pairs", "matched_pairs",
                "max_branch_cost", "suboptimal_path_prune", "next_paths_too_big", "extend_ambiguous_rejoin",
                "search_not_fast_enough", "too_many_ambiguous_bases", "too_many_steps", "too_many_ambiguous_steps",
                "too_far_without_pair", "exceeded_branch_limit", "prune_ambiguous_ref", "device_rounds", "device_queries"]


class _Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in _STAT_FIELDS]


class _Asm(C.Structure):
    _fields_ = [("left_offset", C.c_int64), ("right_offset", C.c_int64), ("score", C.c_int64),
                ("left_anchor_len", C.c_int32), ("right_anchor_len", C.c_int32), ("seq", C.c_void_p), ("seq_len", C.c_uint64),
                ("min_overlap", C.c_uint32), ("trace_steps", C.c_uint32), ("left_anchor_ambiguous_bases", C.c_uint32),
                ("matches_reference", C.c_int32), ("rc_read_ids", C.POINTER(C.c_uint32)),
                ("left_pair_matches", C.POINTER(C.c_uint32)), ("right_pair_matches", C.POINTER(C.c_uint32)),
                ("n_rc_read_ids", C.c_uint64), ("n_left_pair_matches", C.c_uint64), ("n_right_pair_matches", C.c_uint64),
                ("assembly_id", C.c_uint64), ("from_pop", C.c_int32)]


class TraceAssembly:
    """variants::assembly as the tracer, pair_counter and the steps of csrc/align.cpp fill it.  aligned_variants =
    [(left_offset, right_offset, seq)] (variants::aligned_var) after the aligner."""

    def __init__(self, a=None, **kw):
        self.aligned_variants = []
        self.ml, self.merged = None, 0  # assembly_ml_features as a tuple, assemblies merged into this one
        self.assembly_id = 0  # assemble.h:295; 0 on a hand-made assembly = the library allocates one
        self.from_pop = False  # assembly::tags: {"POP"} (pop tracer) or {"PUSH"}
        if a is None:  # a hand-made assembly (the reference's unit tests build them this way)
            self.left_offset = self.right_offset = self.left_anchor_len = self.right_anchor_len = 0
            self.seq, self.min_overlap, self.trace_steps, self.left_anchor_ambiguous_bases = "", 0, 0, 0
            self.matches_reference, self.score = False, 0
            self.rc_read_ids, self.left_pair_matches, self.right_pair_matches = [], [], []
            for k, v in kw.items():
                if not hasattr(self, k):
                    raise AttributeError("no assembly field %r" % k)
                setattr(self, k, v)
            return
        self.left_offset, self.right_offset = int(a.left_offset), int(a.right_offset)
        self.left_anchor_len, self.right_anchor_len = int(a.left_anchor_len), int(a.right_anchor_len)
        self.seq = C.string_at(a.seq, a.seq_len).decode() if a.seq_len else ""
        self.min_overlap, self.trace_steps = int(a.min_overlap), int(a.trace_steps)
        self.left_anchor_ambiguous_bases = int(a.left_anchor_ambiguous_bases)
        self.matches_reference = bool(a.matches_reference)
        self.score = int(a.score)
        self.rc_read_ids = [int(a.rc_read_ids[i]) for i in range(a.n_rc_read_ids)]
        self.left_pair_matches = [int(a.left_pair_matches[i]) for i in range(a.n_left_pair_matches)]
        self.right_pair_matches = [int(a.right_pair_matches[i]) for i in range(a.n_right_pair_matches)]
        self.assembly_id = int(a.assembly_id)
        self.from_pop = bool(a.from_pop)

    def key(self):
        return (self.left_offset, self.right_offset, self.left_anchor_len, self.right_anchor_len, self.seq,
                self.matches_reference, self.min_overlap)

    def __repr__(self):
        return "TraceAssembly[%d:%d) anchors %d/%d %s%s" % (self.left_offset, self.right_offset, self.left_anchor_len,
                                                            self.right_anchor_len, self.seq,
                                                            " REF" if self.matches_reference else "")


def reference_ref_map(seqset, sequences):
    """ref_map::build over a whole reference = list of ACGT strings (N-free stretches) -> device array of n bytes
    (DeviceArray; the caller frees it)."""
    import biograph_amd as bg
    ctx = seqset.ctx
    codes = np.concatenate([np.frombuffer(s.encode(), np.uint8) for s in sequences]) if sequences else np.zeros(0, np.uint8)
    lut = np.zeros(256, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    dref = bg.DeviceArray.from_numpy(ctx, np.ascontiguousarray(lut[codes]))
    out = bg.DeviceArray(ctx, max(seqset.size(), 1))
    es = np.cumsum([0] + [len(s) for s in sequences[:-1]]).astype(np.uint64) if sequences else np.zeros(0, np.uint64)
    ez = np.array([len(s) for s in sequences], np.uint64)
    try:
        capi.check(capi.lib().bg_ref_map_build(ctx.h, seqset.h, C.c_void_p(dref.ptr), len(ez), capi.np_ptr(es), capi.np_ptr(ez),
                                               0, C.c_void_p(out.ptr)))
    finally:
        dref.free()
    return out


class CoverageOptions(C.Structure):
    """bg_coverage_options = the assemble_options calc_coverage, simple_genotype_filter and the VCF export read."""
    _fields_ = [("max_coverage_paths", C.c_uint32), ("penalize_directional_coverage", C.c_int32),
                ("simple_genotype_filter", C.c_int32), ("min_pair_depth", C.c_int32), ("min_read_depth", C.c_int32),
                ("vcf_sv_size_threshold", C.c_uint32), ("min_depth_portion", C.c_double), ("min_avg_pair_depth", C.c_double),
                ("output_ml_features", C.c_int32), ("rvg_exclude", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        capi.lib().bg_coverage_options_default(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise AttributeError("no coverage option %r" % k)
            setattr(self, k, v)


class _Cov(C.Structure):
    _fields_ = [("coverage", C.POINTER(C.c_int32)), ("pair_coverage", C.POINTER(C.c_int32)), ("n_coverage", C.c_uint64),
                ("n_pair_coverage", C.c_uint64), ("has_coverage", C.c_int32), ("other_depth", C.c_int32),
                ("other_pair_depth", C.c_int32), ("ref_depth", C.c_int32), ("strand_count", C.c_int32),
                ("genotype_quality", C.c_double)]


class _Ml(C.Structure):
    _fields_ = [("has_ml", C.c_int32), ("score", C.c_int32), ("refspan", C.c_int32), ("lanch", C.c_int32), ("ranch", C.c_int32),
                ("refgc", C.c_float), ("altgc", C.c_float), ("alt_len", C.c_uint64), ("merged", C.c_uint32)]


def _read_result(h):
    L = capi.lib()
    out = []
    a = _Asm()
    cv = _Cov()
    ml = _Ml()
    lo, ro, sp, sl = C.c_int64(), C.c_int64(), C.c_void_p(), C.c_uint64()
    for i in range(L.bg_trace_result_size(h)):
        capi.check(L.bg_trace_result_get(h, i, C.byref(a)))
        t = TraceAssembly(a)
        capi.check(L.bg_trace_result_ml(h, i, C.byref(ml)))
        t.merged = int(ml.merged)
        if ml.has_ml:  # (score, refspan, lanch, ranch, refgc, altgc, alt_seq length): assembly_ml_features
            t.ml = (int(ml.score), int(ml.refspan), int(ml.lanch), int(ml.ranch), float(ml.refgc), float(ml.altgc), int(ml.alt_len))
        capi.check(L.bg_trace_result_coverage(h, i, C.byref(cv)))
        if cv.has_coverage:
            t.coverage = [int(cv.coverage[j]) for j in range(cv.n_coverage)]
            t.pair_coverage = [int(cv.pair_coverage[j]) for j in range(cv.n_pair_coverage)]
            t.other_depth, t.other_pair_depth, t.ref_depth = int(cv.other_depth), int(cv.other_pair_depth), int(cv.ref_depth)
            t.strand_count, t.genotype_quality = int(cv.strand_count), float(cv.genotype_quality)
        for j in range(L.bg_trace_result_var_count(h, i)):
            capi.check(L.bg_trace_result_var(h, i, j, C.byref(lo), C.byref(ro), C.byref(sp), C.byref(sl)))
            t.aligned_variants.append((int(lo.value), int(ro.value), C.string_at(sp.value, sl.value).decode() if sl.value else ""))
        out.append(t)
    return out


def _u32_array(xs):
    return (C.c_uint32 * max(len(xs), 1))(*[int(x) for x in xs])


STEP_ANCHOR_DROP, STEP_SORT, STEP_TRIM_REF, STEP_DEDUP, STEP_ALIGN, STEP_SPLIT = 1, 2, 4, 8, 16, 32
STEP_NORMALIZE, STEP_EXACT_DEDUP, STEP_VCF_PAD = 64, 128, 256
STEP_THROUGH_SPLIT = STEP_SORT | STEP_TRIM_REF | STEP_DEDUP | STEP_ALIGN | STEP_SPLIT
STEP_PIPELINE = STEP_THROUGH_SPLIT | STEP_NORMALIZE | STEP_EXACT_DEDUP | STEP_VCF_PAD  # pipeline.cpp:40-57,92-98


def _make_result(assemblies):
    L = capi.lib()
    src = C.c_void_p()
    capi.check(L.bg_trace_result_new(C.byref(src)))
    try:
        for t in assemblies:
            seq = t.seq.encode()
            ids, lm, rm = _u32_array(t.rc_read_ids), _u32_array(t.left_pair_matches), _u32_array(t.right_pair_matches)
            a = _Asm(t.left_offset, t.right_offset, t.score, t.left_anchor_len, t.right_anchor_len,
                     C.cast(C.c_char_p(seq), C.c_void_p), len(seq), t.min_overlap, t.trace_steps, t.left_anchor_ambiguous_bases,
                     int(t.matches_reference), ids, lm, rm, len(t.rc_read_ids), len(t.left_pair_matches), len(t.right_pair_matches),
                     int(getattr(t, "assembly_id", 0)), int(bool(getattr(t, "from_pop", False))))
            capi.check(L.bg_trace_result_add(src, C.byref(a)))
            if getattr(t, "coverage", None) is not None:  # a section that has been through calc_coverage
                cov = (C.c_int32 * max(len(t.coverage), 1))(*t.coverage)
                pc = (C.c_int32 * max(len(t.pair_coverage), 1))(*t.pair_coverage)
                cv = _Cov(cov, pc, len(t.coverage), len(t.pair_coverage), 1, t.other_depth, t.other_pair_depth, t.ref_depth,
                          getattr(t, "strand_count", 0), getattr(t, "genotype_quality", 0.0))
                capi.check(L.bg_trace_result_set_coverage(src, L.bg_trace_result_size(src) - 1, C.byref(cv)))
            f = getattr(t, "ml", None)
            if f is not None or getattr(t, "merged", 0):
                m = _Ml(int(f is not None), *((f[0], f[1], f[2], f[3], f[4], f[5], f[6]) if f is not None else (0, 0, 0, 0, 0.0, 0.0, 0)),
                        int(getattr(t, "merged", 0)))
                capi.check(L.bg_trace_result_set_ml(src, L.bg_trace_result_size(src) - 1, C.byref(m)))
    except Exception:
        L.bg_trace_result_free(src)
        raise
    return src


def _extent_arrays(extents, end_pos):
    n = len(extents)
    offs = (C.c_int64 * max(n, 1))(*[int(o) for o, _ in extents])
    bufs = [s.encode() for _, s in extents]
    ptrs = (C.c_char_p * max(n, 1))(*bufs)
    lens = (C.c_uint64 * max(n, 1))(*[len(b) for b in bufs])
    calc_end = max([o + len(s) for o, s in extents] + [0])
    return n, offs, ptrs, lens, (calc_end if end_pos is None else end_pos), bufs


def assemblies_coverage(seqset, readmap, sections, extents, end_pos=None, options=None, genotype_filter=True, vcf_name=None):
    """bg_assemblies_coverage: calc_coverage (+ simple_genotype_filter) over the sections that leave process_assemblies
    for one scaffold.  -> [TraceAssembly] with coverage / pair_coverage / other_depth / other_pair_depth / ref_depth
    (/ strand_count / genotype_quality); with vcf_name also the VCF record lines (bg_vcf_record, PI = 0) as a second list."""
    L = capi.lib()
    src = _make_result(sections)
    out = C.c_void_p()
    try:
        n, offs, ptrs, lens, end_pos, _keep = _extent_arrays(extents, end_pos)
        opt = options if options is not None else CoverageOptions()
        capi.check(L.bg_assemblies_coverage(seqset.ctx.h, seqset.h, readmap.h, src, n, offs, ptrs, lens, end_pos, C.byref(opt),
                                            int(bool(genotype_filter)), C.byref(out)))
        res = _read_result(out)
        if vcf_name is None:
            return res
        lines = []
        ln = C.c_uint64()
        for i in range(len(res)):
            capi.check(L.bg_vcf_record(out, i, vcf_name.encode(), n, offs, ptrs, lens, end_pos, C.byref(opt), 0, None, 0, C.byref(ln)))
            buf = C.create_string_buffer(ln.value + 1)
            capi.check(L.bg_vcf_record(out, i, vcf_name.encode(), n, offs, ptrs, lens, end_pos, C.byref(opt), 0, buf, ln.value + 1,
                                       C.byref(ln)))
            lines.append(buf.value.decode())
        return res, lines
    finally:
        L.bg_trace_result_free(src)
        if out:
            L.bg_trace_result_free(out)


def rvg_exclude(sections, options=None):
    """bg_assemblies_rvg_exclude: the rvg_exclude step (rvg_exclude.cpp) over covered sections of one scaffold (host only)."""
    L = capi.lib()
    src = _make_result(sections)
    out = C.c_void_p()
    try:
        opt = options if options is not None else CoverageOptions()
        capi.check(L.bg_assemblies_rvg_exclude(src, C.byref(opt), C.byref(out)))
        return _read_result(out)
    finally:
        L.bg_trace_result_free(src)
        if out:
            L.bg_trace_result_free(out)


def process_assemblies(assemblies, extents, end_pos=None, options=None, steps=STEP_PIPELINE):
    """bg_assemblies_process: the assemblies of one scaffold's work items (TraceAssembly objects, forward coordinates)
    through the chosen steps of the pipeline behind the tracers, in the pipeline's order: anchor_dropper, sorter,
    ref_trimmer, deduper, aligner, align_splitter, normalizer, exact_deduper, vcf_padder.  -> [TraceAssembly] as they
    leave the last step."""
    L = capi.lib()
    src = C.c_void_p()
    capi.check(L.bg_trace_result_new(C.byref(src)))
    out = C.c_void_p()
    try:
        for t in assemblies:
            seq = t.seq.encode()
            ids, lm, rm = _u32_array(t.rc_read_ids), _u32_array(t.left_pair_matches), _u32_array(t.right_pair_matches)
            a = _Asm(t.left_offset, t.right_offset, t.score, t.left_anchor_len, t.right_anchor_len,
                     C.cast(C.c_char_p(seq), C.c_void_p), len(seq), t.min_overlap, t.trace_steps, t.left_anchor_ambiguous_bases,
                     int(t.matches_reference), ids, lm, rm, len(t.rc_read_ids), len(t.left_pair_matches), len(t.right_pair_matches),
                     int(getattr(t, "assembly_id", 0)), int(bool(getattr(t, "from_pop", False))))
            capi.check(L.bg_trace_result_add(src, C.byref(a)))
        n = len(extents)
        offs = (C.c_int64 * max(n, 1))(*[int(o) for o, _ in extents])
        bufs = [s.encode() for _, s in extents]
        ptrs = (C.c_char_p * max(n, 1))(*bufs)
        lens = (C.c_uint64 * max(n, 1))(*[len(b) for b in bufs])
        calc_end = max([o + len(s) for o, s in extents] + [0])
        end_pos = calc_end if end_pos is None else end_pos
        ins = (C.c_void_p * 1)(src)
        opt = options if options is not None else TraceOptions()
        capi.check(L.bg_assemblies_process(ins, 1, n, offs, ptrs, lens, end_pos, C.byref(opt), int(steps), C.byref(out)))
        return _read_result(out)
    finally:
        L.bg_trace_result_free(src)
        if out:
            L.bg_trace_result_free(out)


def process_assemblies_streamed(batches, extents, end_pos=None, options=None, steps=STEP_PIPELINE, release_early=None, take=False):
    """bg_pipeline_open / feed / finish: `batches` = [(assemblies, mark), ...] fed in order (the mark: no later assembly starts
    before it).  -> ([TraceAssembly] as process_assemblies gives them for all batches together, None), or (None, k) when batch k
    held an assembly behind a mark already given, or (release_early = a margin in bases: bg_pipeline_release_early) moved a
    variant to before what the normalizer had let go -- k = len(batches) when the finish found that; the caller then runs
    process_assemblies over everything.  take: what has left the steps is taken after every batch (bg_pipeline_take) and put
    together here; the third value returned is then how many assemblies had left before the finish."""
    L = capi.lib()
    n, offs, ptrs, lens, end_pos, _keep = _extent_arrays(extents, end_pos)
    opt = options if options is not None else TraceOptions()
    pl = C.c_void_p()
    capi.check(L.bg_pipeline_open(n, offs, ptrs, lens, end_pos, C.byref(opt), int(steps), C.byref(pl)))
    out = C.c_void_p()
    taken = []
    try:
        if release_early is not None:
            capi.check(L.bg_pipeline_release_early(pl, int(release_early)))
        for k, (asms, mark) in enumerate(batches):
            src = _make_result(asms)
            try:
                behind = C.c_int(0)
                ins = (C.c_void_p * 1)(src)
                capi.check(L.bg_pipeline_feed(pl, ins, 1, int(mark), C.byref(behind)))
                if behind.value:
                    return (None, k, len(taken)) if take else (None, k)
            finally:
                L.bg_trace_result_free(src)
            if take:
                part = C.c_void_p()
                capi.check(L.bg_pipeline_take(pl, C.byref(part)))
                try:
                    taken += _read_result(part)
                finally:
                    L.bg_trace_result_free(part)
        capi.check(L.bg_pipeline_finish(pl, C.byref(out)))
        if not out:
            return (None, len(batches), len(taken)) if take else (None, len(batches))
        res = taken + _read_result(out)
        return (res, None, len(taken)) if take else (res, None)
    finally:
        L.bg_pipeline_close(pl)
        if out:
            L.bg_trace_result_free(out)


class PushTracer:
    """A tracer session (bg_tracer_open): the seqset / readmap / ref_map arrays the host part consults are copied once,
    trace() is one work item of trace_ref (a scaffold piece in one direction)."""

    def __init__(self, seqset, readmap, ref_map, options=None):
        self.ctx = seqset.ctx
        self._keep = (seqset, readmap, ref_map)
        self.h = C.c_void_p()
        opt = options if options is not None else TraceOptions()
        capi.check(capi.lib().bg_tracer_open(self.ctx.h, seqset.h, readmap.h, C.c_void_p(ref_map.ptr), C.byref(opt), C.byref(self.h)))

    def trace(self, extents, start=0, limit=None, end_pos=None, rev_comp=False, pair_count=True, with_stats=False,
              half_aligned=None, asynchronous=False):
        """extents = [(offset, ACGT string)] of one scaffold, ascending.  -> [TraceAssembly] in output order (+ stats).
        half_aligned: a list that receives (offset, right_anchor, sequence, assembly_id) of what the anchor_dropper could not
        anchor on the right (bg_trace_result_half_aligned; assemble_options::report_half_aligned_func).
        asynchronous: bg_tracer_trace_async + bg_trace_result_wait (the host steps behind the tracer on the session's helper thread)."""
        L = capi.lib()
        n = len(extents)
        offs = (C.c_int64 * max(n, 1))(*[int(o) for o, _ in extents])
        bufs = [s.encode() for _, s in extents]
        ptrs = (C.c_char_p * max(n, 1))(*bufs)
        lens = (C.c_uint64 * max(n, 1))(*[len(b) for b in bufs])
        calc_end = max([o + len(s) for o, s in extents] + [0])
        end_pos = calc_end if end_pos is None else end_pos
        limit = end_pos if limit is None else limit
        h = C.c_void_p()
        call = L.bg_tracer_trace_async if asynchronous else L.bg_tracer_trace
        capi.check(call(self.h, n, offs, ptrs, lens, end_pos, start, limit, int(bool(rev_comp)), int(bool(pair_count)), C.byref(h)))
        try:
            if asynchronous:
                capi.check(L.bg_trace_result_wait(h))
            out = _read_result(h)
            if half_aligned is not None:
                for i in range(L.bg_trace_result_half_aligned_count(h)):
                    off, right, ln, aid = C.c_int64(), C.c_int(), C.c_uint64(), C.c_uint64()
                    sp = C.c_char_p()
                    capi.check(L.bg_trace_result_half_aligned(h, i, C.byref(off), C.byref(right), C.byref(sp), C.byref(ln), C.byref(aid)))
                    half_aligned.append((off.value, bool(right.value), C.string_at(sp, ln.value).decode(), aid.value))
            if not with_stats:
                return out
            st = _Stats()
            capi.check(L.bg_trace_result_stats(h, C.byref(st)))
            return out, {k: int(getattr(st, k)) for k in _STAT_FIELDS}
        finally:
            L.bg_trace_result_free(h)

    def execute_work(self, extents, start=0, limit=None, end_pos=None, push_fwd=True, push_rev=True, pop_fwd=True, pop_rev=True):
        """One work item of trace_ref with the pop tracer over the anchor drops (trace_ref::execute_work, trace_ref.cpp:141-221;
        `biograph discovery --enable-pop-tracer`): push forward, push reverse, then bg_pop_trace forward and reverse over the
        drops the push tracers reported.  -> [TraceAssembly] in that order (pop assemblies have from_pop set)."""
        L = capi.lib()
        n = len(extents)
        offs = (C.c_int64 * max(n, 1))(*[int(o) for o, _ in extents])
        bufs = [s.encode() for _, s in extents]
        ptrs = (C.c_char_p * max(n, 1))(*bufs)
        lens = (C.c_uint64 * max(n, 1))(*[len(b) for b in bufs])
        calc_end = max([o + len(s) for o, s in extents] + [0])
        end_pos = calc_end if end_pos is None else end_pos
        limit = end_pos if limit is None else limit
        held, out = [], []
        try:
            for rev, on in ((0, push_fwd), (1, push_rev)):
                if not on:
                    continue
                h = C.c_void_p()
                capi.check(L.bg_tracer_trace(self.h, n, offs, ptrs, lens, end_pos, start, limit, rev, 1, C.byref(h)))
                held.append(h)
                out.extend(_read_result(h))
            push = (C.c_void_p * max(len(held), 1))(*[h.value for h in held])
            for rev, on in ((0, pop_fwd), (1, pop_rev)):
                if not on:
                    continue
                h = C.c_void_p()
                capi.check(L.bg_pop_trace(self.h, n, offs, ptrs, lens, end_pos, start, limit, rev, push, len(held), C.byref(h)))
                try:
                    out.extend(_read_result(h))
                finally:
                    L.bg_trace_result_free(h)
            return out
        finally:
            for h in held:
                L.bg_trace_result_free(h)

    def close(self):
        if self.h:
            capi.lib().bg_tracer_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def push_trace(seqset, readmap, ref_map, extents, start=0, limit=None, end_pos=None, rev_comp=False, pair_count=True,
               options=None, with_stats=False, half_aligned=None):
    """One-shot form (bg_push_trace semantics): ref_map = reference_ref_map(...)."""
    t = PushTracer(seqset, readmap, ref_map, options)
    try:
        return t.trace(extents, start, limit, end_pos, rev_comp, pair_count, with_stats, half_aligned)
    finally:
        t.close()
