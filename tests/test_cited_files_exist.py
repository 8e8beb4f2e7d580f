"""The files DESIGN.md, INTEGRATION.md and README.md cite in this repository (`profiles/...`, `tools/...`, `tests/...`, `csrc/...`,
`biograph_amd/...`, `oracle/...`, `include/...`) exist: a measurement's evidence, a tool or a test that was renamed or removed must
not stay cited (HISTORY.md is the past and may cite what is gone)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cited_paths_exist():
    missing = []
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md"):
        text = open(os.path.join(ROOT, doc), errors="replace").read()
        for m in re.finditer(r"`((?:profiles|tools|tests|csrc|biograph_amd|oracle|include)/[A-Za-z0-9_./\-]+?\.(?:py|sh|hip|cpp|h|json|csv|txt|md))(?::[0-9,\- ]+)?`", text):
            rel = m.group(1)
            cands = [rel, os.path.join("biograph_amd", rel)] if rel.startswith("csrc/") else [rel]
            if not any(os.path.exists(os.path.join(ROOT, c)) for c in cands):
                missing.append("%s: %s" % (doc, rel))
    assert not missing, "cited but not in the tree:\n" + "\n".join(sorted(set(missing)))
