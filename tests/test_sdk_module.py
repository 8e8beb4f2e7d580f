"""The host-only part of the SDK module (csrc/pycapi.cpp) on CPU: the module under the name the reference's package imports
(python/biograph/__init__.py:24-45), Sequence (python/biograph/dna_sequence.cpp:69-158), Reference / ReferenceRange /
ReferenceContext (python/biograph/reference.cpp:14-185) over the golden lambda reference and its golden reference.bwt.
Nothing here touches a GPU: the classes that do (Seqset, Readmap, ...) are covered by tests/test_gpu_pycapi.py."""
import ctypes as C
import glob
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "biograph_amd")
G = os.path.join(ROOT, "tests", "golden")
REF_NAME = "_capi_%d%d" % sys.version_info[:2]

# what python/biograph/__init__.py:27-45 takes from the module
SDK_NAMES = ["BioGraph", "CacheStrategy", "Metadata", "Readmap", "ReadmapRead", "ReadmapPairStats", "Reference", "ReferenceContext",
             "ReferenceRange", "Seqset", "SeqsetEntry", "Sequence", "build_revision", "version"]


@pytest.fixture(scope="module")
def capi_module():
    if not glob.glob(os.path.join(PKG, REF_NAME + ".*so")):
        from biograph_amd import build
        build.build()
    import importlib
    return importlib.import_module("biograph_amd." + REF_NAME)


@pytest.fixture()
def lambda_ref(tmp_path):
    """karyotype.json + reference.ref of the golden lambda reference with the golden reference.bwt beside them"""
    d = tmp_path / "ref"
    d.mkdir()
    for src in (os.path.join(G, "ref_lambda", "karyotype.json"), os.path.join(G, "ref_lambda", "reference.ref"),
                os.path.join(G, "ref_lambda_full", "reference.bwt")):
        os.symlink(src, d / os.path.basename(src))
    return str(d)


def lambda_bases():
    text = open(os.path.join(G, "ref_lambda_full", "reference.fasta")).read().split("\n", 1)[1]
    return text.replace("\n", "").upper()


def test_module_drops_into_a_package_named_biograph(tmp_path, capi_module):
    """_capi_<maj><min>*.so + libbgseq.so copied into a package `biograph` whose __init__ loads the
    module the way the reference's does."""
    pkg = tmp_path / "biograph"
    pkg.mkdir()
    so = glob.glob(os.path.join(PKG, REF_NAME + ".*so"))[0]
    os.symlink(so, pkg / os.path.basename(so))
    os.symlink(os.path.join(PKG, "libbgseq.so"), pkg / "libbgseq.so")
    (pkg / "__init__.py").write_text(textwrap.dedent("""
        import importlib, sys
        _capi = importlib.import_module(f"biograph._capi_{sys.version_info.major}{sys.version_info.minor}")
        sys.modules["biograph._capi"] = _capi
        from biograph._capi import (%s)
    """ % ", ".join(SDK_NAMES)))
    code = ("import biograph; s = biograph.Sequence('ACGTT'); assert str(s.rev_comp()) == 'AACGT'; "
            "assert biograph.CacheStrategy.MMAP != biograph.CacheStrategy.RAM; print(biograph.version(), biograph.build_revision())")
    env = dict(os.environ, PYTHONPATH=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[0].startswith("biograph_amd")


def test_sequence_behaves_like_the_sdk_sequence(capi_module):
    S = capi_module.Sequence
    a = S("acgtt")
    assert str(a) == "ACGTT" and repr(a) == "biograph.Sequence('ACGTT')" and len(a) == 5 and len(S()) == 0
    assert str(a.rev_comp()) == "AACGT" and str(a[1:3]) == "CG" and str(a[-1]) == "T" and str(a[:]) == "ACGTT"
    assert str(a + S("GG")) == "ACGTTGG" and str(a + "GG") == "ACGTTGG" and str("TT" + a) == "TTACGTT"
    assert a == S("ACGTT") and a != S("ACGT") and a < S("C") and S("C") > a and a <= a and a >= a
    assert hash(a) == hash(S("ACGTT")) and len({a, S("ACGTT"), S("A")}) == 2
    b = S("ACGTT")
    b += S("A")
    assert str(b) == "ACGTTA"
    b[0:2] = "TTT"
    assert str(b) == "TTTGTTA"
    b[1:1] = S("C")
    assert str(b) == "TCTTGTTA"
    b[0] = "g"
    assert str(b) == "GCTTGTTA"
    with pytest.raises(IndexError):
        a[5]
    with pytest.raises(IndexError):
        a[::2]
    with pytest.raises(RuntimeError):  # dna_base.h:54 throws for anything but ACGT
        S("ACGN")


def test_reference_ranges_of_lambda(capi_module, lambda_ref):
    seq = lambda_bases()
    ref = capi_module.Reference(lambda_ref)
    assert str(ref) == lambda_ref and repr(ref) == "<biograph.Reference %s>" % lambda_ref
    assert ref.size == len(seq) == 48502 and ref.scaffolds == ["lambda"] and ref.chromosomes == ["lambda"]
    assert ref.scaffold_lens == {"lambda": 48502}
    r = ref.make_range("lambda", 10, 30)
    assert (r.scaffold, r.chromosome, r.start, r.end, r.size) == ("lambda", "lambda", 10, 30, 20)
    assert str(r.sequence) == seq[10:30] and str(r) == "lambda:[10-30)" and repr(r) == "<biograph.ReferenceRange lambda:[10-30)>"
    assert str(ref.make_range("lambda", 0, 48502, False).sequence) == seq
    # the whole extent as get_supercontig gives it: the reference asks for [offset, offset + len - 1] (reference.cpp:56-66)
    sc = ref.supercontigs
    assert len(sc) == 1 and (sc[0].start, sc[0].end) == (0, 48501) and str(ref.get_supercontig("lambda", 777)) == "lambda:[0-48501)"
    assert [str(x) for x in ref.find_ranges("lambda", 100, 200)] == ["lambda:[100-200)"]
    with pytest.raises(RuntimeError, match="must come before"):
        ref.make_range("lambda", 30, 10)
    with pytest.raises(RuntimeError, match="no sequence data"):
        ref.make_range("lambda", 30, 10 ** 6)
    with pytest.raises(RuntimeError, match="Unknown scaffold"):
        ref.make_range("chrZ", 0, 10)


def test_reference_find_equals_a_text_search(capi_module, lambda_ref):
    seq = lambda_bases()
    ref = capi_module.Reference(lambda_ref)

    def occurrences(q):
        out, i = [], seq.find(q)
        while i >= 0:
            out.append(i)
            i = seq.find(q, i + 1)
        return out

    for q in ["ACGT", "GGGCGGCGACC", seq[20000:20040], seq[-25:], "A", "TTTTTTTT", "ACGTACGTACGTACGTACGTAAAA"]:
        c = ref.find(q)
        want = occurrences(q)
        assert c.valid == bool(want) and c.matches == len(want), q
        assert str(c.query) == q and str(c) == lambda_ref + "{" + q + "}"
        got = [c.get_match(i) for i in range(c.matches)]
        assert sorted(g.start for g in got) == want, q
        assert all(str(g.sequence) == q and g.size == len(q) for g in got)
        assert c.end - c.start == len(want) or not want
    c = ref.find(capi_module.Sequence("ACGT"))  # a Sequence where a str is taken
    assert c.matches == len(occurrences("ACGT"))
    with pytest.raises(RuntimeError, match="invalid bwt"):
        ref.find("ACGTACGTACGTACGTACGTAAAA").get_match(0)
    with pytest.raises(RuntimeError, match="invalid match"):
        ref.find("GGGCGGCGACC").get_match(5)


def test_bwt_c_abi_refuses_what_is_not_a_bwt(tmp_path):
    from biograph_amd import capi
    lib = capi.lib()
    bad = tmp_path / "reference.bwt"
    bad.write_bytes(b"\0" * 64)
    h = C.c_void_p()
    assert lib.bg_bwt_open(str(bad).encode(), C.byref(h)) != 0
    assert b"does not appear to be a valid BWT" in lib.bg_last_error()
    assert lib.bg_bwt_open(str(tmp_path / "missing").encode(), C.byref(h)) != 0
    assert lib.bg_bwt_open(os.path.join(G, "ref_lambda_full", "reference.bwt").encode(), C.byref(h)) == 0
    try:
        assert lib.bg_bwt_entries(h) == 48503  # the bases + the end of the one extent
        b, e = C.c_uint64(), C.c_uint64()
        assert lib.bg_bwt_find(h, b"ACGN", 4, C.byref(b), C.byref(e)) != 0 and b"not a base" in lib.bg_last_error()
        assert lib.bg_bwt_find(h, b"", 0, C.byref(b), C.byref(e)) == 0 and (b.value, e.value) == (0, 48503)
        pos = C.c_uint64()
        assert lib.bg_bwt_match(h, 48503, C.byref(pos)) != 0
    finally:
        lib.bg_bwt_free(h)
