"""The CLI's thread-caching allocator (biograph_amd/csrc/cli/fastalloc.h) on the CPU: the stress program of tests/native/ built
with g++ -- plain, with AddressSanitizer, and with the allocator switched off (BG_FAST_ALLOC=0) and its huge-page advice off
(BG_FAST_ALLOC_THP=0): sizes from one byte to ten megabytes, cross-thread frees, threads that end with blocks in their caches."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "fastalloc_stress.cpp")
INC = os.path.join(ROOT, "biograph_amd", "csrc", "cli")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("flags,env", [
    (["-O2"], {}),
    (["-O1", "-g", "-fsanitize=address"], {"ASAN_OPTIONS": "alloc_dealloc_mismatch=0"}),
    (["-O2"], {"BG_FAST_ALLOC": "0"}),
    (["-O2"], {"BG_FAST_ALLOC_THP": "0"}),
])
def test_stress(tmp_path, flags, env):
    exe = str(tmp_path / "fa_stress")
    r = subprocess.run(["g++", "-std=c++17", "-pthread", "-I" + INC] + flags + ["-o", exe, SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([exe], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout[-500:], r.stderr[-2000:])
