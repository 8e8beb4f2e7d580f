"""C-ABI surface checks that need no GPU: the library builds, loads, exports every symbol
include/bgseq.h declares, and fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import pytest

from biograph_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        from biograph_amd import build
        build.build()
    return capi.lib()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bgseq.h")).read()
    declared = set(re.findall(r"\b(bg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS) and len(declared) > 200  # (capi reads its list from the header)
    for name in declared:
        assert hasattr(lib, name), name
    # and nothing else: every exported bg_* function of the library is declared in the header
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("bg_")}
    assert exported == declared, sorted(exported ^ declared)
    assert lib.bg_abi_version() == 2


def test_no_cpu_fallback_without_device(lib):
    import os
    import torch
    if torch.cuda.is_available() or torch.cuda.device_count() > 0 or os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.bg_ctx_create(0, None, C.byref(h))
    assert rc == -2  # BG_ENODEV
    assert b"no HIP device" in lib.bg_last_error()
    import biograph_amd
    with pytest.raises(biograph_amd.BgError):
        biograph_amd.Context(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "biograph_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "bgoracle" not in text, f


def test_build_files_list_the_same_sources():
    """CMakeLists.txt and BUILD.bazel name exactly the translation units biograph_amd/build.py compiles."""
    import re
    from biograph_amd import build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = sorted(B.SOURCES)
    cm = open(os.path.join(root, "CMakeLists.txt")).read()
    got = sorted(re.search(r"set\(BGSEQ_SOURCES ([^)]*)\)", cm).group(1).split())
    assert got == want
    bz = open(os.path.join(root, "BUILD.bazel")).read()
    blk = bz[bz.index("for f in ["):bz.index("BGSEQ_HDRS")]
    assert sorted(re.findall(r'"([\w]+\.(?:hip|cpp))"', blk)) == want


def test_sha1_file_both_code_paths(tmp_path, monkeypatch):
    """bg_sha1_file (the name of a readmap inside a .bg directory) against hashlib, through the SHA-extension rounds and
    through the portable ones, at the padding edges."""
    import ctypes as C
    import hashlib
    import random
    from biograph_amd import capi
    L = capi.lib()
    rnd = random.Random(1)
    for n in (0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 128, (1 << 20) - 1, 1 << 20, (1 << 20) + 77, 3 * (1 << 20) + 5):
        data = bytes(rnd.getrandbits(8) for _ in range(min(n, 4096))) * (n // 4096 + 1)
        data = data[:n]
        p = tmp_path / "blob"
        p.write_bytes(data)
        want = hashlib.sha1(data).hexdigest()
        for portable in ("", "1"):
            if portable:
                monkeypatch.setenv("BG_SHA1_PORTABLE", "1")
            else:
                monkeypatch.delenv("BG_SHA1_PORTABLE", raising=False)
            buf = C.create_string_buffer(41)
            capi.check(L.bg_sha1_file(str(p).encode(), buf))
            assert buf.value.decode() == want, (n, portable)
