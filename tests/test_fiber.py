"""csrc/fiber.h (the stacks the coverage step's islands run on) on the CPU: tests/native/fiber_test.cpp -- hundreds of fibers per
thread parked and resumed round by round, their locals intact, eight threads side by side, a failing round that every parked fiber
learns of when it wakes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("flags", [["-O2"], ["-O0", "-g"]])
def test_fibers(tmp_path, flags):
    exe = str(tmp_path / "fiber_test")
    r = subprocess.run(["g++", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "biograph_amd", "csrc")] + flags +
                       ["-o", exe, os.path.join(ROOT, "tests", "native", "fiber_test.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout[-500:], r.stderr[-2000:])
