"""The device-memory sub-allocator of the library (biograph_amd/csrc/arena.h) on a malloc backend: best fit, splitting,
coalescing, the out-of-memory retry, the cache cap, and a random workload under its consistency check."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arena_unit(tmp_path):
    exe = str(tmp_path / "arena_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "arena_test.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "arena_test ok" in r.stdout
