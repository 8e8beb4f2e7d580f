"""ORACLE (test infrastructure only; never imported by the product path): a CPU restatement of the coverage and
genotype end of the reference's assembly pipeline --

  path_group              modules/variants/path_group.cpp:21-277   paths through the seqset carrying "distant objects"
  calc_coverage           modules/variants/calc_coverage.cpp:9-533 interbase read / pair coverage of every variant
                                                                  section, depth of the reference and of the other
                                                                  alleles beside it (pipeline.cpp:99-101)
  simple_genotype_filter  modules/variants/simple_genotype_filter.cpp:27-151
  ploidless_vcf_export    modules/variants/ploidless_vcf_export.cpp:12-166 (one record line)

The reference drives results out of destructors (a coverage tracker reports when the last path lets go of it, an
assembly is emitted when its last tracker has reported).  Nothing a tracker accumulates can change after the last
path dropped it, and everything the destructors compute is a minimum or a sum, so this restatement keeps every tracker
until the end and reports then; the output is sorted in canon order, which is the order the reference's sorted output
queue (with its tracked left offsets) produces.

Pinned on modules/variants/calc_coverage_test.cpp (tests/test_oracle_coverage.py) and on the DP / AD / PDP / PAD / GT
columns of datasets/lambdaToyData/benchmark/father.vcf from the golden father_lambda biograph."""
from .align import SortedOutput, canon_key, check, copy_assembly
from .tracer import _ScaffoldIt, COMP, INT_MAX


class CoverageOptions:
    """assemble_options fields (assemble.h:504,607,700-719 and the depth limits of simple_genotype_filter)."""

    def __init__(self, **kw):
        self.max_coverage_paths = 4
        self.penalize_directional_coverage = True
        self.simple_genotype_filter = True
        self.min_depth_portion = .23
        self.min_avg_pair_depth = 1
        self.min_pair_depth = 0
        self.min_read_depth = 1
        self.vcf_sv_size_threshold = 20
        self.output_ml_features = False  # assemble.h:730; biograph_variants.cpp:686 switches it on
        self.rvg_exclude = False  # assemble.h:699; biograph_variants.cpp:194,689: --rvg-exclude, true by default there
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("no coverage option %r" % k)
            setattr(self, k, v)


# ---- path_group ---------------------------------------------------------------------------------------------------
class _Path:
    __slots__ = ("objects", "unprop")

    def __init__(self):
        self.objects = {}  # adjusted distance -> [dobj] (insertion order, no duplicates)
        self.unprop = 0

    def copy(self):
        p = _Path()
        p.objects = {k: list(v) for k, v in self.objects.items()}
        p.unprop = self.unprop
        return p


def _dobj_add(lst, dobj):  # dobj_set::add(dobj) :231-241
    for i, x in enumerate(lst):
        if x is dobj:
            del lst[i]
            break
    lst.append(dobj)


def _dobj_add_all(lst, more):  # dobj_set::add(dobj_set&&) :243-268
    if not more:
        return lst
    if not lst:
        return list(more)
    ids = {id(x) for x in more}
    return [x for x in lst if id(x) not in ids] + list(more)


class PathGroup:
    def __init__(self, graph, listener, max_size, start=None):
        self.g, self.listener, self.max_size = graph, listener, max_size
        self.cur = {}  # (size, begin) -> (range, path); iterated in cur_comparer order: size descending, begin ascending
        if start is not None:
            self._add_path(start, _Path())

    @staticmethod
    def _key(r):
        return (-r[2], r[0])

    def _ordered(self):
        return [self.cur[k] for k in sorted(self.cur)]

    def _add_path(self, r, p):  # :41-52
        k = self._key(r)
        if k not in self.cur:
            self.cur[k] = (r, p)
        else:
            self.cur[k] = (r, self._merge(r, p, self.cur[k][1]))

    def _merge(self, r, p1, p2):  # :60-80
        if p1.unprop != p2.unprop:
            if p1.unprop:
                self._flush_path(r, p1)
            if p2.unprop:
                self._flush_path(r, p2)
        check(p1.unprop == p2.unprop, "path_group: merged paths keep different distances")
        for k, v in p2.objects.items():
            p1.objects[k] = _dobj_add_all(p1.objects.get(k, []), v)
        return p1

    @staticmethod
    def _flush_path(r, p):  # :82-95
        if not p.unprop:
            return
        p.objects = {k + p.unprop: v for k, v in p.objects.items() if k + p.unprop <= r[2]}
        p.unprop = 0

    def flush(self):  # :54-58
        for r, p in self.cur.values():
            self._flush_path(r, p)

    def add_distant_object(self, dobj, size):  # :97-104
        for r, p in self.cur.values():
            k = -(p.unprop + size)
            p.objects.setdefault(k, [])
            _dobj_add(p.objects[k], dobj)

    def add_sequence(self, seq):
        for b in seq:
            self.add_base(b)

    def add_base(self, b):  # :113-151
        old = self._ordered()
        self.cur = {}
        for r, p in old:
            p.unprop += 1
            nr = self.g.push_front_drop(r, COMP[b], 0)
            check(nr is not None, "path_group: push_front_drop(.., 0) gave an invalid range")
            self._add_path(nr, p)
        if self.max_size and len(self.cur) > self.max_size:  # trim_cur :153-169: the shortest contexts go
            for k in sorted(self.cur)[self.max_size:]:
                del self.cur[k]
        for r, p in self._ordered():
            if p.objects and min(p.objects) + p.unprop > r[2]:
                self._flush_path(r, p)
            self.listener.on_seqset_entry(r, self)

    def visit_distant_objects(self, r, visit):  # :171-201
        p = self.cur[self._key(r)][1]
        done = set()
        for k in sorted(p.objects):
            dist = k + p.unprop
            if dist > r[2]:
                break
            for dobj in p.objects[k]:
                if id(dobj) not in done:
                    done.add(id(dobj))
                    visit(dobj, dist)

    def join(self, rhs):  # :203-220
        for r, p in rhs._ordered():
            self._add_path(r, p)
        rhs.cur = {}

    def split(self):  # :222-234
        out = PathGroup(self.g, self.listener, self.max_size)
        out.cur = {k: (r, p.copy()) for k, (r, p) in self.cur.items()}
        return out


# ---- calc_coverage ------------------------------------------------------------------------------------------------
class _Accum:  # coverage_accum (calc_coverage.cpp:23-88)
    def __init__(self, size):
        self.starts, self.ends = [0] * size, [0] * size

    def size(self):
        return len(self.starts)

    def add(self, start, end):
        n = len(self.starts)
        start = max(start, 0)
        if end >= n:
            end = n - 1
        if start > n - 1 or end < 0:
            return False
        self.starts[start] += 1
        self.ends[end] += 1
        return True

    def coverage(self):
        out, cur, n = [], 0, len(self.starts)
        for i in range(n + 1):
            if i < n:
                cur += self.starts[i]
            if i > 0:
                cur -= self.ends[i - 1]
            if i == n:
                check(cur == 0, "coverage_accum: starts and ends do not balance")
            else:
                out.append(cur)
        return out


class _Shared:  # shared_assembly (calc_coverage.h:44-58)
    def __init__(self, a):
        self.a = a
        self.pair_read_ids = set(a.left_pair_matches) | set(a.right_pair_matches)
        self.other_min = []
        self.other_ref_coverage = []


class _Tracker:  # cov_tracker (calc_coverage.h:60-69)
    def __init__(self, cov_size, var=None):
        self.var = var
        self.others = []
        self.fwd, self.rev = _Accum(cov_size), _Accum(cov_size)
        self.pair = _Accum(cov_size) if var is not None else None


class CalcCoverage:
    def __init__(self, graph, read_len, scaffold, options):
        """graph: oracle.tracer.Graph; read_len: per read id."""
        self.g, self.sc, self.o = graph, scaffold, options
        self.read_len = read_len
        self.min_read_len = min(read_len) if len(read_len) else 0
        self.max_read_len = graph.max_read_len
        self.it = _ScaffoldIt(scaffold)
        self.cur_offset = self.it.offset
        self.need_coverage_to = 0
        self.need_ref_dobj = False
        self.active = []  # (right_offset, shared, path group), by right offset, equal ones in arrival order
        self.trackers, self.shared = [], []
        self._init_ref_pg()

    # -- listener (calc_coverage.cpp:455-527)
    def on_seqset_entry(self, r, pg):
        if r[2] < self.min_read_len:
            return
        first, last = self.g.entry_to_index(r[0])
        if first == last:
            return

        def visit(cov, distance):
            for rid in range(first, last):
                rl = self.read_len[rid]
                if rl > r[2]:
                    continue
                acc = cov.fwd if self.g.is_forward[rid] else cov.rev
                start = distance - rl + 1 + acc.size() - 1
                end = distance - 1 + acc.size() - 1
                if acc.add(start, end) and cov.var is not None and rid in cov.var.pair_read_ids:
                    check(cov.pair.add(start, end), "pair coverage outside the read coverage")
        pg.visit_distant_objects(r, visit)

    def _init_ref_pg(self):  # :362-370
        self.ref_pg = PathGroup(self.g, self, self.o.max_coverage_paths, self.g.ctx_begin())

    def add(self, a):  # on_assembly :206-279
        if a.matches_reference:
            return
        self._skip_to(a.left_offset - self.max_read_len)
        while self.cur_offset < a.left_offset:  # advance_to :281-287
            self._advance_towards(a.left_offset)
        sa = _Shared(a)
        self.shared.append(sa)
        t = _Tracker(len(a.seq) + 1, sa)
        t.others = [x[1] for x in self.active]
        self.trackers.append(t)
        pg = self.ref_pg.split()
        pg.add_distant_object(t, len(a.seq))
        pg.add_sequence(a.seq)
        pg.flush()
        self.need_coverage_to = max(self.need_coverage_to, a.right_offset + self.max_read_len)
        i = len(self.active)
        while i > 0 and self.active[i - 1][0] > a.right_offset:
            i -= 1
        self.active.insert(i, (a.right_offset, sa, pg))
        self.need_ref_dobj = True

    def _add_ref_dobj(self, ref_len):  # :289-319
        if self.active:
            t = _Tracker(ref_len + 1)
            t.others = [x[1] for x in self.active]
            self.trackers.append(t)
            self.ref_pg.add_distant_object(t, ref_len)
        self.need_ref_dobj = False

    def _flush_active_to_here(self):  # :321-343
        while self.active and self.active[0][0] <= self.cur_offset:
            if self.need_ref_dobj:
                self._add_ref_dobj(0)
            self.ref_pg.join(self.active.pop(0)[2])

    def _skip_to(self, offset):  # :345-360
        while self.cur_offset < offset:
            if not self.active and self.cur_offset >= self.need_coverage_to:
                if not self.it.at_end() and self.it.offset < offset:
                    self.it.skip_to(offset)
                    self.cur_offset = self.it.offset
                else:
                    self.cur_offset = offset
                self._init_ref_pg()
            else:
                self._advance_towards(offset)

    def _advance_towards(self, target):  # :372-453
        self._flush_active_to_here()
        check(target > self.cur_offset, "calc_coverage: advancing backwards")
        if self.active and target > self.active[0][0]:
            target = self.active[0][0]
        check(target > self.cur_offset, "calc_coverage: an active assembly ends behind the cursor")
        it = self.it
        if it.at_end():
            self.cur_offset = target
            return
        if it.first_in_extent() and self.cur_offset < it.offset:
            if target >= it.offset:
                self.cur_offset = it.offset
                self._init_ref_pg()
            else:
                self.cur_offset = target
            return
        self._flush_active_to_here()
        extent_end = it.sc.extents[it.ext][0] + len(it.sc.extents[it.ext][1])
        target = min(target, extent_end)
        if self.cur_offset < self.need_coverage_to < target:
            target = self.need_coverage_to
        ref_len = target - self.cur_offset
        check(ref_len > 0, "calc_coverage: empty reference stretch")
        self._add_ref_dobj(ref_len)
        while self.cur_offset < target:
            self.ref_pg.add_base(it.base())
            it.advance()
            self.cur_offset += 1
        self.ref_pg.flush()

    def finish(self):
        """~calc_coverage (:529-534), then what the trackers' and the assemblies' destructors report (:105-204)."""
        self._skip_to(INT_MAX)
        check(not self.active, "calc_coverage: assemblies still active at the end")
        for t in self.trackers:
            f, r = t.fwd.coverage(), t.rev.coverage()
            both = [self._sum_dir_depth(x, y) for x, y in zip(f, r)]
            if t.var is not None:
                pair = t.pair.coverage()
                for other in t.others:
                    other.other_min.append((min(both), min(pair)))
                t.var.a.coverage, t.var.a.pair_coverage = both, pair
            else:
                check(t.others, "calc_coverage: reference coverage nobody asked for")
                for other in t.others:
                    other.other_ref_coverage.append(both)
        out = []
        for sa in self.shared:
            a = sa.a
            a.other_depth = min([m[0] for m in sa.other_min], default=0)
            a.other_pair_depth = min([m[1] for m in sa.other_min], default=0)
            check(sa.other_ref_coverage, "calc_coverage: an assembly without reference coverage")
            a.ref_depth = min(min(c) for c in sa.other_ref_coverage)
            out.append(a)
        out.sort(key=canon_key)
        return out

    def _sum_dir_depth(self, fwd, rev):  # :9-21
        if self.o.penalize_directional_coverage:
            total = fwd + rev
            if abs(fwd - rev) * 4 > total * 3:
                return min(fwd, rev) * 2
            return total
        return fwd + rev


def calc_coverage(graph, read_len, scaffold, options, assemblies):
    c = CalcCoverage(graph, read_len, scaffold, options)
    for a in assemblies:
        c.add(copy_assembly(a))
    return c.finish()


# ---- simple_genotype_filter (simple_genotype_filter.cpp:27-151) -----------------------------------------------------
def simple_genotype_filter(o, assemblies):
    out = []
    for a in assemblies:
        if a.matches_reference:
            continue
        if not o.simple_genotype_filter:
            out.append(a)
            continue
        check(a.coverage, "simple_genotype_filter: assembly without coverage")
        min_depth = min(a.coverage)
        min_pair = min(a.pair_coverage) if a.pair_coverage else 0
        avg_pair = sum(a.pair_coverage) * 1. / len(a.pair_coverage) if a.pair_coverage else 0
        discard = lambda: out.append(a) if "POP" in a.tags else None  # noqa: E731  (:9-25: pop assemblies pass untouched)
        if avg_pair < o.min_avg_pair_depth or min_pair < o.min_pair_depth or min_depth < o.min_read_depth:
            discard()
            continue
        total = min_depth + a.other_depth + a.ref_depth
        if not total:
            discard()
            continue
        min_p, alt_p, ref_p = min_depth * 1. / total, a.other_depth * 1. / total, a.ref_depth * 1. / total
        f = o.min_depth_portion
        if min_p < f:
            discard()
            continue
        if ref_p > f:
            a.strand_count, a.genotype_quality = 1, min(min_p, ref_p) / (min_p + ref_p)
        elif alt_p > f:
            a.strand_count, a.genotype_quality = 1, min(min_p, alt_p) / (min_p + alt_p)
        else:
            a.strand_count, a.genotype_quality = 2, min_p
        out.append(a)
    return out


# ---- rvg_exclude (rvg_exclude.cpp:5-67; the last step of add_standard_variants_pipeline, pipeline.cpp:102-104) ----------
class RvgExclude(SortedOutput):
    def __init__(self, o, emit):
        super().__init__(emit)
        self.o = o
        self.backlog = {}  # assembly id -> [assemblies held back]
        self.known_inphase = set()

    def add(self, a):  # on_assembly :14-67
        self.flush_sorted_to(a.left_offset)
        if a.matches_reference:
            self.sort_and_output(a)
            return
        reflen = (a.right_offset - a.left_offset) & 0xffffffff  # unsigned
        seqlen = len(a.seq)
        if not reflen or not seqlen:  # compare as if VCF padded
            reflen += 1
            seqlen += 1
        if seqlen >= self.o.vcf_sv_size_threshold or reflen >= self.o.vcf_sv_size_threshold:
            self.known_inphase.add(a.assembly_id)  # structural: out it goes, with what waited for it
            for old in self.backlog.pop(a.assembly_id, []):
                self.untrack_left_offset(old.left_offset)
                self.sort_and_output(old)
            self.sort_and_output(a)
            return
        if a.other_pair_depth or a.assembly_id in self.known_inphase:
            output_this = True
        else:
            check(a.pair_coverage, "rvg_exclude: assembly without pair coverage")
            output_this = all(a.pair_coverage)
        if output_this:
            self.sort_and_output(a)
            return
        self.track_left_offset(a.left_offset)
        self.backlog.setdefault(a.assembly_id, []).append(a)

    def finish(self):  # flush() :5-12
        for held in self.backlog.values():
            for a in held:
                self.untrack_left_offset(a.left_offset)
        self.backlog = {}
        self.flush()


def rvg_exclude(o, assemblies):
    out = []
    step = RvgExclude(o, out.append)
    for a in assemblies:
        step.add(a)
    step.finish()
    return out


# ---- ploidless_vcf_export::on_assembly (ploidless_vcf_export.cpp:12-166) ------------------------------------------------
def vcf_line(o, scaffold_name, subscaffold_str, a, assembly_id=0):
    """subscaffold_str(offset, len) -> reference bases.  -> one VCF record line (without ML features / assembly ids)."""
    if a.matches_reference:
        return None
    min_depth = min(a.coverage) if a.coverage else 0
    min_pair = min(a.pair_coverage) if a.pair_coverage else 0
    var_seq, ref_seq, left = a.seq, subscaffold_str(a.left_offset, a.right_offset - a.left_offset), a.left_offset
    if not var_seq or not ref_seq:
        if left == 0:
            b = subscaffold_str(a.right_offset, 1)
            var_seq, ref_seq = var_seq + b, ref_seq + b
        else:
            b = subscaffold_str(a.left_offset - 1, 1)
            var_seq, ref_seq, left = b + var_seq, b + ref_seq, left - 1
    gt, pg = {0: ("0/1", "0|1"), 1: ("0/1", "0|1"), 2: ("1/1", "1|1")}[a.strand_count]
    info = "NS=1"
    if len(var_seq) >= o.vcf_sv_size_threshold or len(ref_seq) >= o.vcf_sv_size_threshold:
        svlen = len(var_seq) - len(ref_seq)
        info += ";END=%d;SVLEN=%d;SVTYPE=%s" % (left + len(ref_seq), svlen, "DEL" if svlen < 0 else "INS" if svlen > 0 else "CPX")
    sample = "%s:%s:%d:%d:%d:%d:%d,%d:%d:%d,%d" % (
        gt, pg, int(a.genotype_quality * 100), assembly_id, a.min_overlap, a.other_depth + min_depth + a.ref_depth,
        a.ref_depth, min_depth, a.other_pair_depth + min_pair, 0, min_pair)
    fmt = "GT:PG:GQ:PI:OV:DP:AD:PDP:PAD"
    if o.output_ml_features:  # ploidless_vcf_export.cpp:128-130,155-165 (LARANCH prints the LEFT anchor, as there)
        check(getattr(a, "ml", None) is not None, "missing ML features on an assembly")
        fmt += ":LASCORE:LAREFSPAN:LARANCH:LALANCH:LAREFGC:LAALTGC:LAALTSEQLEN:NUMASM"
        f = a.ml
        sample += ":%d:%d:%d:%d:%f:%f:%d:%d" % (f[0], f[1], f[2], f[3], f[4], f[5], f[6], getattr(a, "merged", 0) + 1)
    return "%s\t%d\t.\t%s\t%s\t100\tPASS\t%s\t%s\t%s\n" % (scaffold_name, left + 1, ref_seq, var_seq, info, fmt, sample)
