"""csrc/align.cpp (the host steps behind the tracers) compiled with g++ under AddressSanitizer and UBSan and driven by
a random-input program (tests/align_asan_test.cpp), refused inputs included.  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_align_steps_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "align_asan_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "align_asan_test.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "align_asan_test ok" in r.stdout
