"""Every BG_* environment switch INTEGRATION.md names is read somewhere in the sources, and every switch the sources read is
named in INTEGRATION.md or DESIGN.md (a renamed or removed switch must not stay in the documentation, a new one must get there)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(path):
    with open(path, errors="replace") as f:
        return f.read()


def _sources():
    out = []
    for base in ("biograph_amd", "tools", "oracle"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            if "__pycache__" in d or os.sep + "obj" in d:
                continue
            for f in files:
                if f.endswith((".hip", ".cpp", ".h", ".py", ".sh")):
                    out.append(os.path.join(d, f))
    out.append(os.path.join(ROOT, "bench.py"))
    out.append(os.path.join(ROOT, "__graft_entry__.py"))
    return out


def test_documented_switches_exist_and_read_switches_are_documented():
    doc = _read(os.path.join(ROOT, "INTEGRATION.md")) + _read(os.path.join(ROOT, "DESIGN.md"))
    documented = set(re.findall(r"\bBG_[A-Z0-9_]+\b", _read(os.path.join(ROOT, "INTEGRATION.md"))))
    text = "\n".join(_read(p) for p in _sources())
    in_sources = set(re.findall(r"\bBG_[A-Z0-9_]+\b", text))
    # names that are not environment switches: error codes, step bits, macros of the C ABI
    not_env = {n for n in documented | in_sources if n.startswith(("BG_E", "BG_OK", "BG_STEP_", "BG_HD", "BG_TRY", "BG_HIP", "BG_LAUNCH", "BG_NCCL", "BG_API"))}
    missing = sorted(n for n in documented - not_env if n not in in_sources)
    assert not missing, "INTEGRATION.md names switches no source reads: %s" % missing
    read = set(re.findall(r'getenv\("(BG_[A-Z0-9_]+)"\)', text)) | set(re.findall(r'environ(?:\.get)?[\[(]"(BG_[A-Z0-9_]+)"', text))
    undocumented = sorted(n for n in read if n not in doc)
    assert not undocumented, "switches read by the sources but named in neither INTEGRATION.md nor DESIGN.md: %s" % undocumented
