"""The searches over `shared` inside push_front_drop / pop_front as the device compiles them (eight entries a load, the min index
over whole blocks): tests/native/word_scan_test.cpp against their definition, on the CPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("flags", [["-O2"], ["-O0", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=all"]])
def test_word_scans(tmp_path, flags):
    exe = str(tmp_path / "word_scan_test")
    r = subprocess.run(["g++", "-std=c++17"] + flags + ["-o", exe, os.path.join(ROOT, "tests", "native", "word_scan_test.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout[-500:], r.stderr[-2000:])
