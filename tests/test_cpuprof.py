"""The sampling profiler built into bgbinary (csrc/cli/cpuprof.h) and its report (tools/cpuprof_report.py), on the CPU: a harness
spins in one named function on two threads with BG_CPU_PROFILE set; the report must attribute most of the samples to it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("nm") is None, reason="no g++ / nm")
def test_profile_names_the_hot_function(tmp_path):
    exe = str(tmp_path / "cpuprof_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "biograph_amd", "csrc", "cli"), "-o", exe,
                        os.path.join(ROOT, "tests", "native", "cpuprof_test.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    prof = str(tmp_path / "profile.txt")
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, BG_CPU_PROFILE=prof), timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("done"), r.stderr[-1000:]
    lines = open(prof).read().splitlines()
    samples = [x for x in lines if x.startswith("S ")]
    assert any(x.startswith("M ") for x in lines) and len(samples) >= 20
    assert len({x.split()[1] for x in samples}) >= 2  # both threads were sampled
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpuprof_report.py"), prof, "--top", "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    top = [x for x in r.stdout.splitlines() if "spin_here_for_the_profiler" in x]
    assert top, r.stdout
    share = float(top[0].split("%")[0])
    assert share > 60.0, r.stdout
    # without the variable nothing is armed and nothing is written
    other = str(tmp_path / "none.txt")
    r = subprocess.run([exe], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "BG_CPU_PROFILE"}, timeout=300)
    assert r.returncode == 0 and not os.path.exists(other)
