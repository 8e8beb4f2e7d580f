"""The discover drivers (tests/variants_twin.py) with the GPU kernels behind them, against the same known
answers of the reference as tests/test_discover_drivers.py (branch_test.cpp, push_to_pair_test.cpp,
graph_discover_test.py + father.vcf), plus GPU == oracle on every intermediate assembly list."""
import pytest

import biograph_amd as bg
from biograph_amd.variants import AssemblyList
from tests import variants_twin as V
from oracle import oracle as O
from tests import test_discover_drivers as T
from tests.discover_backend import OracleWalker

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = bg.Context(0)
    yield c
    c.close()


def gpu_walker_for_reads(ctx):
    def make(unpaired, paired=()):
        flat = []
        for r in unpaired:
            flat += [r, ""]                      # a single = a pair whose mate was dropped
        for a, b in paired:
            flat += [a, b]
        reads = ctx.pack_reads(flat)
        ss, _ = ctx.build_seqset(reads, 30)
        rm = ctx.build_readmap(ss, reads, True)
        return V.Walker(ss, rm)
    return make


def test_branch_discover_known_answers_gpu(ctx):
    T.branch_cases(gpu_walker_for_reads(ctx))


def test_push_to_pair_known_answers_gpu(ctx):
    T.push_cases(gpu_walker_for_reads(ctx))


def test_graph_discover_on_lambda_golden_gpu(ctx, golden_dir):
    ss = ctx.open_seqset(golden_dir + "/father_lambda/seqset")
    rm = ctx.open_readmap(golden_dir + "/father_lambda/readmap")
    ref = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">")).upper()
    found = T.lambda_pipeline(V.Walker(ss, rm), ref)
    assert len(found) == 8
    T.check_against_father_vcf(found, ref, golden_dir)
    # and the same assemblies, in the same order, as the oracle-backed run
    d = O.load_seqset_zip(golden_dir + "/father_lambda/seqset")
    g = O.decode_readmap(golden_dir + "/father_lambda/readmap", d)
    oss = O.Seqset.from_arrays(d["sizes"], d["shared"], d["bits"])
    orm = O.Readmap.from_arrays(d["n"], g["entry"], g["read_len"], g["mate_loop_ptr"], g["is_forward"], 150)
    want = T.lambda_pipeline(OracleWalker(oss, orm), ref)
    assert [(a.left_offset, a.seq, a.right_offset, a.rc_entries) for a in found] == \
        [(a.left_offset, a.seq, a.right_offset, a.rc_entries) for a in want]


# ---- the C++ drivers (csrc/discover.cpp) ---------------------------------------------------------------
def _sig(asms):
    return [(a.left_offset, a.seq, a.right_offset, a.matches_reference, tuple(sorted(a.tags)), a.rc_entries) for a in asms]


def test_native_drivers_known_answers(ctx):
    """branch_test.cpp / push_to_pair_test.cpp through the C++ drivers; every step's full assembly list (incl.
    rc paths) equals the Python twin's."""
    make = gpu_walker_for_reads(ctx)
    t = O.tseq

    def both(w, asms, step):
        py = V.update_rc_seqset_entries(sorted(asms, key=V.canon_key), w, with_mates=True)
        nat = AssemblyList(asms).update_rc_seqset_entries(w.ss, w.rm)
        assert _sig(nat.to_python()) == _sig(py)
        py = step(py, w)
        if step is branch_py:
            nat.discover_branch(w.ss, w.rm, w._mo, "branch_discover")
        else:
            nat.discover_push_to_pair(w.ss, w.rm, w._mo, w._mpd, "push_to_pair")
        assert _sig(nat.to_python()) == _sig(py)
        py = V.update_rc_seqset_entries(py, w, with_mates=True)
        nat.update_rc_seqset_entries(w.ss, w.rm)
        assert _sig(nat.to_python()) == _sig(py)
        return [(a.left_offset, a.seq, a.right_offset) for a in nat.to_python(False) if a.tags & {"branch_discover", "push_to_pair"}]

    def branch_py(asms, w):
        return V.discover_branch(asms, w, w._mo, "branch_discover")

    def push_py(asms, w):
        return V.discover_push_to_pair(asms, w, w._mo, w._mpd, "push_to_pair")

    w = make(T.BRANCH_READS + [t("ef") + "G" + t("gh")])
    w._mo, w._mpd = len(t("ef")), 1000
    assert both(w, [T.ref_asm(len(t("a")), t("bcdefghijklm"))], branch_py) == [(len(t("a")), t("bcdef") + "G", None)]
    w = make(T.BRANCH_READS + [t("ef") + "G" + t("gh"), t("ghijklmno")])
    w._mo, w._mpd = len(t("ef")), 1000
    got = both(w, [T.var_asm(len(t("abcd")), t("ef") + "G", None), T.ref_asm(len(t("abcde")), t("fghi"))], push_py)
    anchor = (len(t("abcd")), t("ef") + "G" + t("ghi"), len(t("abcdefghi")))
    cont = (len(t("abcd")), t("ef") + "G" + t("ghi"), None)
    assert anchor in got and cont in got and all(g in (anchor, cont) for g in got)
    w = make([t("pqrst") + "G" + t("uv")], [(t("abc"), V.rev_comp("G" + t("uvw")))])
    w._mo, w._mpd = len(t("ef")), len(t("abcdefghijklmnopqrst"))
    got = both(w, [T.ref_asm(0, t("abcd")), T.ref_asm(len(t("abcd")), t("efghijklmnopqr")),
                   T.var_asm(len(t("abcdefghijklmnopqr")), t("st") + "G", None)], push_py)
    assert got == [(len(t("abcdefghijklmnopqr")), t("st") + "G" + t("uvw"), None)]


def test_native_graph_discover_on_lambda_golden(ctx, golden_dir):
    ss = ctx.open_seqset(golden_dir + "/father_lambda/seqset")
    rm = ctx.open_readmap(golden_dir + "/father_lambda/readmap")
    ref = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">")).upper()
    ext = [(0, ref)]
    lst = AssemblyList().make_ref_assemblies(ext, 0, len(ref))
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    lst.discover_branch(ss, rm, 70, "BRANCH")
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    lst.discover_push_to_pair(ss, rm, 70, 1000, "PUSH_TO_PAIR", ["BRANCH"])
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    allasm = lst.to_python()
    found = [a for a in allasm if a.left_offset is not None and a.right_offset is not None and not a.matches_reference]
    assert len(found) == 8
    T.check_against_father_vcf(found, ref, golden_dir)
    # identical, assembly for assembly and rc path for rc path, to the Python twin over the same kernels
    w = V.Walker(ss, rm)

    def trim_and_update(asms):
        return V.update_rc_seqset_entries(V.dedup_assemblies(V.graph_trim_ref(asms, ext)), w)
    py = trim_and_update(V.make_ref_assemblies(ext, 0, len(ref)))
    py = trim_and_update(V.discover_branch(py, w, 70, "BRANCH"))
    py = trim_and_update(V.discover_push_to_pair(py, w, 70, 1000, "PUSH_TO_PAIR", ["BRANCH"]))
    assert _sig(allasm) == _sig(py)


@pytest.mark.parametrize("ref_arg,chrom", [("/father_lambda/reference.fasta", "lambda:0"), ("/ref_lambda", "lambda"),
                                           ("/ref_lambda/reference.ref", "lambda:0")])
@pytest.mark.parametrize("engine", ["push", "graph"])
def test_bgbinary_discovery_cli_on_lambda(golden_dir, tmp_path, ref_arg, chrom, engine):
    """--ref takes a FASTA, a reference directory (karyotype.json + reference.ref: scaffold names from the
    karyotype) or a bare .ref"""
    """`bgbinary discovery` on the lambda golden laid out as a .bg directory: 8 VCF records, the variants of
    lambdaToyData/benchmark/father.vcf."""
    import json
    import shutil
    import subprocess
    import sys
    bgd = tmp_path / "father.bg"
    (bgd / "coverage").mkdir(parents=True)
    (bgd / "metadata").mkdir()
    shutil.copy(golden_dir + "/father_lambda/seqset", bgd / "seqset")
    shutil.copy(golden_dir + "/father_lambda/readmap", bgd / "coverage" / "abc.readmap")
    json.dump({"accession_id": "father", "samples": {"father": "abc"}}, open(bgd / "metadata" / "bg_info.json", "w"),
              separators=(",", ":"))
    out = tmp_path / "v.vcf"
    r = subprocess.run([sys.executable, "-m", "biograph_amd.cli", "discovery", "--in", str(bgd), "--ref",
                        golden_dir + ref_arg, "--out", str(out), "--engine", engine] +
                       # push: the command's defaults (0.7 x read length); father.vcf predates the ML feature columns
                       # and was written with the simple genotype filter and without rvg_exclude (today's defaults are the other way round)
                       (["--min-overlap", "70"] if engine == "graph" else ["--ml-features", "false", "--simple-gt", "true", "--rvg-exclude", "false", "--enable-pop-tracer", "false"]),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">")).upper()

    def edits(path):
        res = []
        for line in open(path):
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            pos, rr, alt = int(f[1]) - 1, f[3], f[4]
            assert ref[pos:pos + len(rr)] == rr
            res.append(ref[:pos] + alt + ref[pos + len(rr):])
        return sorted(res)
    mine, gold = edits(out), edits(golden_dir + "/father_lambda/father.vcf")
    assert len(mine) == 8 and mine == gold
    assert {l.split("\t")[0] for l in open(out) if not l.startswith("#")} == {chrom}
    if engine == "push":
        # the reference's engine: every record line of father.vcf but for the scaffold name given by --ref and the PI
        # column (the reference's process-wide assembly counter; here the record's ordinal)
        def fields(path):
            res = []
            for line in open(path):
                if line.startswith("#"):
                    continue
                f = line.rstrip("\n").split("\t")
                smp = f[9].split(":")
                smp[3] = "PI"
                res.append(f[1:9] + [":".join(smp)])
            return res
        assert fields(out) == fields(golden_dir + "/father_lambda/father.vcf")
        head = [l for l in open(out) if l.startswith("#")]
        assert head[0] == "##fileformat=VCFv4.1\n" and head[-1] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tfather\n"
        assert "##contig=<ID=%s,length=48502>\n" % chrom in head


def test_native_drivers_equal_oracle_twin_on_a_240kb_sample(ctx):
    """A 240 kb diploid sample with SNPs, indels (1-50 bp) and structural insertions / deletions (50-1000 bp), 30x paired
    150 bp reads with errors: the whole pipeline through the C++ drivers over the GPU walks (batched predictions,
    deferred walks, multi-pass sweep of update_rc_seqset_entries) gives, assembly for assembly and rc path for rc path,
    what the Python twin gives over the ORACLE's literal walks on the oracle's own seqset and readmap of the same
    reads — at every step of the pipeline, with mates."""
    import numpy as np
    import torch
    from biograph_amd import synth
    G = 240_000
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=150, seed=77, device="cuda", return_genome=True,
                                            indel_rate=3e-4, sv_rate=5e-5)
    r = ctx.pack_reads_device(seq.data_ptr(), offs.data_ptr(), n)
    ks = ctx.kmer_count(r, 30, 5)
    c, _ = ctx.correct_reads(r, ks)
    ss, _ = ctx.build_seqset(c, 30)
    rm = ctx.build_readmap(ss, c, True)
    # the oracle's world from the same corrected reads (the builds themselves are compared elsewhere)
    cseq, coffs = c.download()
    corrected = O.unpack_strings(cseq, coffs)
    ent = O.Entries.build(cseq, coffs, "cover", 30, threads=8)
    oss = O.Seqset.from_entries(ent, threads=8)
    records = [tuple(s for s in corrected[i:i + 2] if s) for i in range(0, len(corrected), 2)]
    orm = O.Readmap(oss, ent, [rec for rec in records if rec], True)
    op = oss.parts()
    assert all(op[k] == ss.part(k) for k in bg.PART_NAMES)
    ref = "".join("ACGT"[int(x)] for x in genome.cpu().tolist())
    ext = [(0, ref)]
    w = OracleWalker(oss, orm)

    def trim_and_update(asms):
        return V.update_rc_seqset_entries(V.dedup_assemblies(V.graph_trim_ref(asms, ext)), w)

    lst = AssemblyList().make_ref_assemblies(ext, 0, len(ref))
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    py = trim_and_update(V.make_ref_assemblies(ext, 0, len(ref)))
    assert _sig(lst.to_python()) == _sig(py)
    lst.discover_branch(ss, rm, 105, "BRANCH")
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    py = trim_and_update(V.discover_branch(py, w, 105, "BRANCH"))
    assert _sig(lst.to_python()) == _sig(py)
    lst.discover_push_to_pair(ss, rm, 105, 1000, "PUSH_TO_PAIR", ["BRANCH"])
    lst.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss)
    py = trim_and_update(V.discover_push_to_pair(py, w, 105, 1000, "PUSH_TO_PAIR", ["BRANCH"]))
    got = lst.to_python()
    assert _sig(got) == _sig(py)
    found = [a for a in got if a.left_offset is not None and a.right_offset is not None and not a.matches_reference]
    sizes = sorted(abs(len(a.seq) - (a.right_offset - a.left_offset)) for a in found)
    assert len(found) > 150 and sizes[-1] >= 50  # SNPs, indels and at least one structural variant were anchored
    # the pipeline with the readmap at the update steps too (mates collected on the rc paths)
    lst2 = AssemblyList().make_ref_assemblies(ext, 0, 40000)
    lst2.graph_trim_ref(ext).dedup().update_rc_seqset_entries(ss, rm)
    py2 = V.update_rc_seqset_entries(V.dedup_assemblies(V.graph_trim_ref(V.make_ref_assemblies(ext, 0, 40000), ext)), w, with_mates=True)
    assert _sig(lst2.to_python()) == _sig(py2)


def test_bgbinary_discovery_push_engine_with_bed(golden_dir, tmp_path):
    """--bed restricts the push engine to the given ranges (trace_ref::add_scaffold_range): the records of father.vcf
    whose positions lie inside, unchanged."""
    import json
    import shutil
    import subprocess
    import sys
    bgd = tmp_path / "father.bg"
    (bgd / "coverage").mkdir(parents=True)
    (bgd / "metadata").mkdir()
    shutil.copy(golden_dir + "/father_lambda/seqset", bgd / "seqset")
    shutil.copy(golden_dir + "/father_lambda/readmap", bgd / "coverage" / "abc.readmap")
    json.dump({"accession_id": "father", "samples": {"father": "abc"}}, open(bgd / "metadata" / "bg_info.json", "w"),
              separators=(",", ":"))
    bed = tmp_path / "r.bed"
    # (ranges wide enough for the mates of the reads over a variant to be traced too: max_pair_distance = 1000)
    bed.write_text("lambda\t500\t11000\nlambda\t28500\t32500\n")
    out = tmp_path / "v.vcf"
    r = subprocess.run([sys.executable, "-m", "biograph_amd.cli", "discovery", "--in", str(bgd), "--ref", golden_dir + "/ref_lambda",
                        "--out", str(out), "--bed", str(bed), "--max-ploids", "2", "--ml-features", "false", "--simple-gt", "true", "--rvg-exclude", "false", "--enable-pop-tracer", "false"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

    def fields(path, keep):
        res = []
        for line in open(path):
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            if not keep(int(f[1])):
                continue
            smp = f[9].split(":")
            smp[3] = "PI"
            res.append(f[1:9] + [":".join(smp)])
        return res
    inside = lambda pos: 500 < pos <= 11000 or 28500 < pos <= 32500  # noqa: E731
    want = fields(golden_dir + "/father_lambda/father.vcf", inside)
    assert len(want) == 6 and fields(out, lambda pos: True) == want
    # the command's defaults are the current reference's (biograph_variants.cpp:188-202): pop tracer on, rvg_exclude on, no
    # simple genotype filter -- every record of father.vcf is still there (unfiltered, so possibly among others)
    r = subprocess.run([sys.executable, "-m", "biograph_amd.cli", "discovery", "--in", str(bgd), "--ref", golden_dir + "/ref_lambda",
                        "--out", str(out), "--ml-features", "false", "-f"], capture_output=True, text=True)  # (-f: the output exists)
    assert r.returncode == 0, r.stderr
    assert "pop tracer on" in r.stderr
    mine = {tuple(l.split("\t")[1:5]) for l in open(out) if not l.startswith("#")}
    gold = {tuple(l.split("\t")[1:5]) for l in open(golden_dir + "/father_lambda/father.vcf") if not l.startswith("#")}
    assert gold <= mine, sorted(gold - mine)


def test_bgbinary_discovery_ml_feature_columns(golden_dir, tmp_path):
    """The command's default output carries the ML feature columns the current reference always writes
    (biograph_variants.cpp:686, ploidless_vcf_export.cpp:128-165): the father.vcf records, each with LASCORE ... NUMASM."""
    import json
    import shutil
    import subprocess
    import sys
    bgd = tmp_path / "father.bg"
    (bgd / "coverage").mkdir(parents=True)
    (bgd / "metadata").mkdir()
    shutil.copy(golden_dir + "/father_lambda/seqset", bgd / "seqset")
    shutil.copy(golden_dir + "/father_lambda/readmap", bgd / "coverage" / "abc.readmap")
    json.dump({"accession_id": "father", "samples": {"father": "abc"}}, open(bgd / "metadata" / "bg_info.json", "w"),
              separators=(",", ":"))
    out = tmp_path / "v.vcf"
    r = subprocess.run([sys.executable, "-m", "biograph_amd.cli", "discovery", "--in", str(bgd), "--ref", golden_dir + "/ref_lambda",
                        "--out", str(out), "--simple-gt", "true", "--rvg-exclude", "false", "--enable-pop-tracer", "false"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    recs = [l.rstrip("\n").split("\t") for l in open(out) if not l.startswith("#")]
    gold = [l.rstrip("\n").split("\t") for l in open(golden_dir + "/father_lambda/father.vcf") if not l.startswith("#")]
    assert len(recs) == len(gold) == 8
    ref = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">")).upper()
    for f, g in zip(recs, gold):
        assert f[1:8] == g[1:8]
        assert f[8] == g[8] + ":LASCORE:LAREFSPAN:LARANCH:LALANCH:LAREFGC:LAALTGC:LAALTSEQLEN:NUMASM"
        smp, ml = f[9].split(":")[:9], f[9].split(":")[9:]
        want = g[9].split(":")
        assert smp[:3] + smp[4:] == want[:3] + want[4:] and len(ml) == 8
        refspan, altlen, numasm = int(ml[1]), int(ml[6]), int(ml[7])
        assert refspan > 100 and altlen > 100 and numasm >= 2   # found by several traces in both directions
        assert 0.0 < float(ml[4]) < 1.0 and 0.0 < float(ml[5]) < 1.0
    assert any(l.startswith("##FORMAT=<ID=NUMASM") for l in open(out))


def test_bgbinary_create_then_discovery_is_the_same_on_any_number_of_sessions(tmp_path):
    """FASTQ -> `bgbinary create` -> `bgbinary discovery` on a 150 kb synthetic sample cut into 8 pieces: one tracer session
    and five side by side write the same VCF records, and the planted SNPs are among them."""
    import subprocess
    import sys
    import numpy as np
    import torch
    from biograph_amd import synth
    G, L, seed = 150_000, 150, 4242
    seq, offs, n, genome = synth.make_reads(genome_len=G, coverage=30, read_len=L, seed=seed, device="cuda", return_genome=True)
    reads = seq.view(n, L).cpu().numpy()
    for mate in (0, 1):
        with open(tmp_path / ("r%d.fq" % (mate + 1)), "wb") as f:
            for i, row in enumerate(reads[mate::2]):
                f.write(b"@r%d\n" % i + row.tobytes() + b"\n+\n" + b"I" * L + b"\n")
    ref = np.frombuffer(b"ACGT", np.uint8)[genome.cpu().numpy()].tobytes().decode()
    (tmp_path / "ref.fa").write_text(">chrS\n" + "\n".join(ref[i:i + 60] for i in range(0, G, 60)) + "\n")
    exe = [sys.executable, "-m", "biograph_amd.cli"]
    r = subprocess.run(exe + ["create", "--reads", str(tmp_path / "r1.fq"), "--pair", str(tmp_path / "r2.fq"), "--out",
                              str(tmp_path / "s.bg")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    outs = []
    for threads in (1, 5):
        out = tmp_path / ("v%d.vcf" % threads)
        r = subprocess.run(exe + ["discovery", "--in", str(tmp_path / "s.bg"), "--ref", str(tmp_path / "ref.fa"), "--out", str(out),
                                  "--threads", str(threads), "--scaffold-split-size", "20000"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "8 piece(s) x 2 directions on %d tracer session(s)" % threads in r.stderr
        outs.append([x for x in open(out) if not x.startswith("##fileDate")])
    assert outs[0] == outs[1]
    recs = [x.split("\t") for x in outs[0] if not x.startswith("#")]
    g = torch.Generator(device="cuda")
    g.manual_seed(seed + 1)
    planted = set(torch.nonzero(torch.rand(G, generator=g, device="cuda") < 0.001).flatten().cpu().tolist())
    snps = {int(f[1]) - 1 for f in recs if len(f[3]) == 1 and len(f[4]) == 1}
    assert len(planted) > 100 and len(snps & planted) >= 0.9 * len(planted) and not (snps - planted)


def test_bgbinary_discovery_force_ref_map_min_pop_overlap_assemblies_out(golden_dir, tmp_path):
    """The remaining options of `biograph discovery` (biograph_variants.cpp:153-199): an existing output is only replaced with
    -f; --ref-map keeps the reference map in a spiral container between runs (built into FILE.new and renamed, opened when it
    exists, refused when it belongs to another seqset); --min-pop-overlap reaches the pop tracer; --assemblies-out writes every
    assembly that enters the pipeline as CSV."""
    import json
    import shutil
    import subprocess
    import sys
    bgd = tmp_path / "father.bg"
    (bgd / "coverage").mkdir(parents=True)
    (bgd / "metadata").mkdir()
    shutil.copy(golden_dir + "/father_lambda/seqset", bgd / "seqset")
    shutil.copy(golden_dir + "/father_lambda/readmap", bgd / "coverage" / "abc.readmap")
    json.dump({"accession_id": "father", "samples": {"father": "abc"}}, open(bgd / "metadata" / "bg_info.json", "w"), separators=(",", ":"))
    exe = [sys.executable, "-m", "biograph_amd.cli", "discovery", "--in", str(bgd), "--ref", golden_dir + "/ref_lambda"]
    out, rmap, csv = tmp_path / "v.vcf", tmp_path / "lambda.refmap", tmp_path / "asm.csv"
    acsv = tmp_path / "aligned.csv"
    r = subprocess.run(exe + ["--out", str(out), "--ref-map", str(rmap), "--assemblies-out", str(csv), "--aligned-assemblies-out", str(acsv)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # behind the aligner: every assembly with its aligned variants as refrange:varseq:refseq -- the 100-base deletion of
    # father.vcf is there as an empty variant sequence over 100 reference bases
    arows = [x.rstrip("\n").split(",") for x in open(acsv)]
    assert arows[0][8] == "variants(refrange:varseq:refseq)" and len(arows) > 5
    variants = [v.split(":") for x in arows[1:] for v in x[8].split(";") if v]
    assert all(len(v) == 3 for v in variants)
    assert any(v[1] == "" and len(v[2]) == 100 for v in variants)
    assert any(len(v[1]) == 1 and len(v[2]) == 1 and v[1] != v[2] for v in variants)  # a SNP
    assert "Generating refmap in" in r.stderr and "Opening refmap" in r.stderr and rmap.exists() and not (tmp_path / "lambda.refmap.new").exists()
    first = [x for x in open(out) if not x.startswith("##fileDate")]
    rows = [x.rstrip("\n").split(",") for x in open(csv)]
    assert rows[0] == ["scaffold_name", "left_offset", "right_offset", "left_anchor_len", "right_anchor_len", "aid", "score", "min_overlap",
                       "ref_seq", "seq", "generated_by"]
    # (lambda: 8 variants found from both directions, a few of them by the pop tracer as well)
    assert len(rows) >= 9 and "PUSH" in {x[10] for x in rows[1:]} and {x[10] for x in rows[1:]} <= {"PUSH", "POP"}
    assert all(x[0] == "lambda" and len(x) == 11 for x in rows[1:])
    ref = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">")).upper()
    for x in rows[1:40]:
        assert x[8] == ref[int(x[1]):int(x[2])]
    # the output exists now: refused without -f, replaced with it; the second run opens the kept map and writes the same records
    r = subprocess.run(exe + ["--out", str(out), "--ref-map", str(rmap)], capture_output=True, text=True)
    assert r.returncode == 1 and "Refusing to overwrite" in r.stderr and "Use -f to override" in r.stderr
    r = subprocess.run(exe + ["--out", str(out), "--ref-map", str(rmap), "-f"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Generating refmap" not in r.stderr and "Opening refmap" in r.stderr
    assert [x for x in open(out) if not x.startswith("##fileDate")] == first
    # the container is the reference's: one packed_vector of 8-bit values, one per seqset entry
    import zipfile
    z = zipfile.ZipFile(rmap)
    assert {"file_info.json", "part_info.json", "packed_data", "packed_vector.json"} <= set(z.namelist())
    assert json.loads(z.read("packed_vector.json")) == {"value_count": 98006, "value_width_bits": 8}
    # a map of another seqset is refused (ref_map.cpp:37)
    other = tmp_path / "other.refmap"
    ctx = bg.Context(0)
    try:
        import ctypes as C
        from biograph_amd import capi
        buf = (C.c_uint8 * 10)()
        capi.check(capi.lib().bg_ref_map_write(str(other).encode(), buf, 10, 0, None))
    finally:
        ctx.close()
    r = subprocess.run(exe + ["--out", str(tmp_path / "w.vcf"), "--ref-map", str(other)], capture_output=True, text=True)
    assert r.returncode == 1 and "the seqset has 98006" in r.stderr
    # --min-pop-overlap: a pop tracer that may not pop below 140 bases keeps fewer poppers alive
    r = subprocess.run(exe + ["--out", str(tmp_path / "p.vcf"), "--min-pop-overlap", "140"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import re
    few = int(re.search(r"pop tracer on \[(\d+) of the assemblies\]", r.stderr).group(1))
    r = subprocess.run(exe + ["--out", str(tmp_path / "q.vcf")], capture_output=True, text=True)
    many = int(re.search(r"pop tracer on \[(\d+) of the assemblies\]", r.stderr).group(1))
    assert few <= many  # (lambda's eight variants leave the pop tracer nothing either way: the option is taken and harmless)
    r = subprocess.run(exe + ["--out", str(tmp_path / "z.vcf"), "--min-pop-overlap", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "--min-pop-overlap must be at least 1" in r.stderr


@pytest.mark.parametrize("split", [10000, 1000000])
def test_bgbinary_discovery_over_ranks_writes_the_same_vcf(golden_dir, tmp_path, split):
    """`bgbinary discovery --gpus N` (trace_ref.cpp:44-63,105-140: the pieces over processes): the pieces of every scaffold go
    round-robin to the ranks, the assemblies to the scaffold's owner, the records to rank 0 -- the file is the single-process
    command's, on 1 rank through the same path, on 2 and on 3 (shared-memory communicator: one GPU here), over one scaffold
    (the golden lambda reference) and over three (lambda cut in three FASTA records)."""
    import json
    import shutil
    import subprocess
    import sys
    bgd = tmp_path / "father.bg"
    (bgd / "coverage").mkdir(parents=True)
    (bgd / "metadata").mkdir()
    shutil.copy(golden_dir + "/father_lambda/seqset", bgd / "seqset")
    shutil.copy(golden_dir + "/father_lambda/readmap", bgd / "coverage" / "abc.readmap")
    json.dump({"accession_id": "father", "samples": {"father": "abc"}}, open(bgd / "metadata" / "bg_info.json", "w"),
              separators=(",", ":"))
    lam = "".join(l.strip() for l in open(golden_dir + "/father_lambda/reference.fasta") if not l.startswith(">")).upper()
    three = tmp_path / "three.fa"
    three.write_text(">part2\n%s\n>part10\n%s\n>partX\n%s\n" % (lam[:16000], lam[16000:32000], lam[32000:]))
    exe = [sys.executable, "-m", "biograph_amd.cli", "discovery", "--in", str(bgd), "--scaffold-split-size", str(split), "-f"]

    def run(ref, extra, name):
        out = tmp_path / name
        r = subprocess.run(exe + ["--ref", ref, "--out", str(out)] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return [l for l in open(out) if not l.startswith("##fileDate")], r.stderr

    for ref in (golden_dir + "/ref_lambda", str(three)):
        want, _ = run(ref, [], "single.vcf")
        assert sum(1 for l in want if not l.startswith("#")) >= 8
        for extra in (["--gpus", "1", "--force-sharded", "--comm", "shm"], ["--gpus", "1", "--force-sharded", "--comm", "rccl"],
                      ["--gpus", "2", "--comm", "shm"], ["--gpus", "3", "--comm", "shm"]):
            got, err = run(ref, extra, "ranks.vcf")
            assert got == want, (ref, extra)
            assert "pop tracer on" in err
    # what only the single-process command writes is refused, not dropped
    r = subprocess.run(exe + ["--ref", golden_dir + "/ref_lambda", "--out", str(tmp_path / "x.vcf"), "--gpus", "2", "--comm", "shm",
                              "--assemblies-out", str(tmp_path / "a.csv")], capture_output=True, text=True)
    assert r.returncode != 0 and "single-process command only" in r.stderr
