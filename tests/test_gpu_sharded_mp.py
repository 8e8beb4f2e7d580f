"""Real multi-process run of the sharded build: N processes (torch.distributed.run), one rank each,
all on cuda:0, collectives over gloo on host copies.  Exercises the per-rank control flow, the uneven
all_to_all / all_gather plumbing and the rank-0 assembly exactly as the RCCL path runs them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_multiprocess_sharded_build_matches_single_gpu(tmp_path, world):
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29600 + world),
                        os.path.join(ROOT, "tests", "mp_sharded_worker.py"), str(out)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert res["same"], res
    assert res["n_entries"] == res["ref_entries"] and len(res["shards"]) == world
