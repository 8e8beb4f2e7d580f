"""The discover drivers (biograph_amd/variants.py) with the GPU kernels behind them, against the same known
answers of the reference as tests/test_discover_drivers.py (branch_test.cpp, push_to_pair_test.cpp,
graph_discover_test.py + father.vcf), plus GPU == oracle on every intermediate assembly list."""
import pytest

import biograph_amd as bg
from biograph_amd import variants as V
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
