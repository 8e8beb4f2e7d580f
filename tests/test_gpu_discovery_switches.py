"""`bgbinary discovery` writes the same VCF whichever of its engine's shortcuts are on: the never-branching traces on the device
(k_trace_simple), the unseen reference assemblies as lean records into pair_counter, several pop passes per launch (k_pop_passes),
the one-sector rank blocks, the pipeline fed piece by piece behind the tracers, the coverage step's worker groups, threads and islands per thread, the CLI's allocator, the number of tracer sessions.  Every shortcut
claims "the same results"; the VCF lines of configs[1] against the oracle (tests/test_gpu_config1.py) pin the default, this pins
the rest to the default -- on a sample with indels and a deletion, so that the pop tracer and the anchor_dropper have work
(variants/tracer.cpp, pop_tracer.cpp, pair_counter.cpp, calc_coverage.cpp: one engine, several ways of running it)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EXE = [sys.executable, "-m", "biograph_amd.cli"]
G, L = 600_000, 150


def _fastq(path, rows):
    n, ln = rows.shape
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b"@r%d\n" % i + rows[i].tobytes() + b"\n+\n" + b"I" * ln + b"\n")


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    """A 600 kb reference; the sample = the reference with SNPs (synth's planted sites), plus insertions, deletions and one 400-base
    deletion put into ONE haplotype's reads by hand: 30x paired reads drawn from the two haplotypes."""
    d = tmp_path_factory.mktemp("switches")
    rng = np.random.default_rng(77)
    ref = rng.integers(0, 4, G).astype(np.uint8)
    hap = [ref.copy(), None]
    # haplotype 1: SNPs every ~900 bases, small indels every ~7 kb, one large deletion
    pieces, pos = [], 0
    events = sorted(rng.choice(np.arange(2000, G - 2000), size=G // 900, replace=False).tolist())
    for k, e in enumerate(events):
        if e < pos:
            continue
        pieces.append(ref[pos:e])
        kind = k % 8
        if kind == 3:  # insertion of 1-12 bases
            pieces.append(rng.integers(0, 4, int(rng.integers(1, 13))).astype(np.uint8))
            pos = e
        elif kind == 5:  # deletion of 1-15 bases
            pos = e + int(rng.integers(1, 16))
        elif k == 40:  # a 400-base deletion
            pos = e + 400
        else:  # SNP
            pieces.append(np.array([(int(ref[e]) + 1 + int(rng.integers(0, 3))) % 4], np.uint8))
            pos = e + 1
    pieces.append(ref[pos:])
    hap[1] = np.concatenate(pieces)
    n_pairs = 30 * G // (2 * L)
    rows = np.empty((2 * n_pairs, L), np.uint8)
    comp = np.array([3, 2, 1, 0], np.uint8)
    for i in range(n_pairs):
        h = hap[i & 1]
        frag = int(rng.integers(300, 600))
        s = int(rng.integers(0, len(h) - frag))
        a, b = h[s:s + L], h[s + frag - L:s + frag]
        if rng.integers(0, 2):
            rows[2 * i], rows[2 * i + 1] = a, comp[b[::-1]]
        else:
            rows[2 * i], rows[2 * i + 1] = comp[b[::-1]], a
    ascii_rows = np.frombuffer(b"ACGT", np.uint8)[rows]
    _fastq(d / "r1.fq", ascii_rows[0::2])
    _fastq(d / "r2.fq", ascii_rows[1::2])
    text = np.frombuffer(b"ACGT", np.uint8)[ref].tobytes().decode()
    # two scaffolds (the second one's steps behind the tracers overlap nothing, the first one's overlap the second's tracing)
    cut = 350_000
    (d / "ref.fa").write_text(">chrT\n" + "\n".join(text[i:i + 70] for i in range(0, cut, 70)) + "\n" +
                              ">chrU\n" + "\n".join(text[i:i + 70] for i in range(cut, G, 70)) + "\n")
    r = subprocess.run(EXE + ["create", "--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq"), "--out", str(d / "s.bg")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return d


def _discover(d, name, env=None, extra=()):
    e = dict(os.environ)
    e.update(env or {})
    out = d / (name + ".vcf")
    r = subprocess.run(EXE + ["discovery", "--in", str(d / "s.bg"), "--ref", str(d / "ref.fa"), "--out", str(out), "-f",
                              "--scaffold-split-size", "100000"] + list(extra), capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return [x for x in open(out).read().splitlines() if not x.startswith("##fileDate")], r.stderr


def test_the_shortcuts_change_nothing(sample):
    d = sample
    base, err = _discover(d, "default")
    records = [x for x in base if not x.startswith("#")]
    assert len(records) > 400
    cols = [x.split("\t") for x in records]
    assert {c[0] for c in cols} == {"chrT", "chrU"} and [c[0] for c in cols] == sorted(c[0] for c in cols)  # both scaffolds, in order
    assert any(len(c[3]) > len(c[4]) for c in cols) and any(len(c[4]) > len(c[3]) for c in cols)  # deletions and insertions
    assert any("SVTYPE=DEL" in x for x in records)                                               # the 400-base deletion
    assert "pop tracer on [" in err and "pop tracer on [0 of" not in err  # the pop tracer produced assemblies
    variants = {
        "no_device_traces": {"BG_TRACER_DEVICE_SIMPLE": "0"},
        "full_reference_records": {"BG_TRACER_LEAN_REFS": "0"},
        "one_pop_pass_per_launch": {"BG_POP_PASSES_AHEAD": "0"},
        "no_rank_blocks": {"BG_WALK_RANK_BLOCKS": "0"},
        "rank_blocks_per_base": {"BG_WALK_RANK4": "0"},
        "pop_front_by_binary_search": {"BG_WALK_SELECT": "0"},
        "one_work_item_at_a_time": {"BG_TRACER_ASYNC": "0"},
        "one_scaffold_at_a_time": {"BG_DISCOVERY_OVERLAP": "0"},
        "pipeline_after_the_tracers": {"BG_DISCOVERY_STREAM": "0"},
        "coverage_behind_the_tracers": {"BG_DISCOVERY_STREAM": "coverage", "BG_DISCOVERY_COV_BATCH": "50"},
        "coverage_behind_the_tracers_in_one_stretch": {"BG_DISCOVERY_STREAM": "coverage"},
        "normalizer_keeps_its_extents": {"BG_DISCOVERY_STREAM": "coverage", "BG_DISCOVERY_EARLY": "-1", "BG_DISCOVERY_COV_BATCH": "50"},
        "normalizer_lets_go_at_once": {"BG_DISCOVERY_STREAM": "coverage", "BG_DISCOVERY_EARLY": "0", "BG_DISCOVERY_COV_BATCH": "50"},
        "pipeline_marks_that_break": {"BG_DISCOVERY_REACH": "-50000"},
        "pipeline_marks_at_the_pieces": {"BG_DISCOVERY_REACH": "0"},
        "pipeline_marks_far_back": {"BG_DISCOVERY_REACH": "10000000"},
        "one_coverage_group_of_three": {"BG_COVERAGE_GROUPS": "1", "BG_COVERAGE_WORKERS": "3"},
        "many_small_coverage_groups": {"BG_COVERAGE_GROUPS": "7", "BG_COVERAGE_WORKERS": "5", "BG_COVERAGE_FIBERS": "3"},
        "one_island_per_thread": {"BG_COVERAGE_FIBERS": "1", "BG_COVERAGE_WORKERS": "40"},
        "coverage_rounds_from_pinned_memory": {"BG_COVERAGE_PINNED": "1"},
        "one_thread_many_islands": {"BG_COVERAGE_GROUPS": "1", "BG_COVERAGE_WORKERS": "1", "BG_COVERAGE_FIBERS": "500"},
        "glibc_malloc": {"BG_FAST_ALLOC": "0"},
        "all_off": {"BG_TRACER_DEVICE_SIMPLE": "0", "BG_TRACER_LEAN_REFS": "0", "BG_POP_PASSES_AHEAD": "0", "BG_FAST_ALLOC_THP": "0"},
    }
    for name, env in variants.items():
        got, err = _discover(d, name, env)
        assert got == base, name
        # (the broken marks are said and the pipeline runs over everything; the default's marks hold)
        assert ("further before it than" in err) == (name in ("pipeline_marks_that_break", "normalizer_lets_go_at_once")), name
    for threads in ("1", "3"):
        got, _ = _discover(d, "threads" + threads, extra=["--threads", threads])
        assert got == base, "threads " + threads


def test_half_aligned_out(sample):
    """--half-aligned-out (biograph_variants.cpp:175,695-710): what the anchor_dropper could not anchor on the right, one CSV row each
    -- scaffold, the anchor's offset in the left_anchor column (forward traces) or the right_anchor column (reverse traces, turned
    around by reverse_half_aligned), the unanchored bases, the assembly id.  (The rows themselves are pinned against the oracle's
    anchor_dropper in tests/test_gpu_tracer.py::test_synthetic_equals_oracle.)"""
    d = sample
    half = d / "half.csv"
    base, _ = _discover(d, "default_again")
    got, _ = _discover(d, "with_half", extra=["--half-aligned-out", str(half)])
    assert got == base
    rows = [x.rstrip("\n").split(",") for x in open(half)]
    assert rows[0] == ["scaffold_name", "left_anchor", "right_anchor", "sequence", "assembly_id"]
    assert len(rows) > 3
    left = right = 0
    for name, la, ra, seq, aid in rows[1:]:
        assert name in ("chrT", "chrU") and (la == "") != (ra == "") and seq and set(seq) <= set("ACGT") and int(aid) > 0
        off = int(la or ra)
        assert 0 <= off <= 350_000
        left += la != ""
        right += ra != ""
    assert left and right  # both directions report
    # refused when it exists, unless -f
    r = subprocess.run(EXE + ["discovery", "--in", str(d / "s.bg"), "--ref", str(d / "ref.fa"), "--out", str(d / "other.vcf"),
                              "--half-aligned-out", str(half)], capture_output=True, text=True)
    assert r.returncode == 1 and "Refusing to overwrite" in r.stderr
