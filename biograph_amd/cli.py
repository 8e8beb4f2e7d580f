"""`python -m biograph_amd.cli <cmd> ...` — the reference's `biograph` dispatcher shells out to its C++
`bgbinary` (python/biograph/tools/bgbinary_cmds.py:117-158); this does the same with the in-tree
`bgbinary` built over libbgseq.so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    exe = os.path.join(HERE, "bgbinary")
    if not os.path.exists(exe):
        from . import build
        build.build()
    return subprocess.call([exe] + argv)


if __name__ == "__main__":
    sys.exit(main())
