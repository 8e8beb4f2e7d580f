"""bench.py's one JSON line in the modes a multi-rank run can end up in: the library's sharded driver, sharded.py over
torch.distributed, and -- when neither starts on every rank -- independent fused builds labelled as replicas.  One rank here
(--force-sharded), a small genome; the chain is what `--gpus N` walks down on a node where a mode cannot start."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--genome-len", "400000",
                        "--no-cpu-baseline", "--no-discover", "--no-end-to-end", "--force-sharded"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_bench_falls_from_the_native_driver_to_sharded_py_to_replicas():
    line, err = _bench({})
    assert line["config"]["sharded_mode"] == "native" and line["comm"]["backend"] == "rccl" and "did not start" not in err
    n_entries = line["counters"]["n_entries"]
    line, err = _bench({"BG_BENCH_FAIL_MODES": "native"})
    assert line["config"]["sharded_mode"] == "torch" and line["comm"] is None and "the native sharded build did not start" in err
    assert line["counters"]["n_entries"] == n_entries
    line, err = _bench({"BG_BENCH_FAIL_MODES": "native,torch"})
    assert line["config"]["sharded_mode"] == "replicas" and "replicas" in line["config"]["parallelism"] and line["sharded_readmap"] is None
    assert line["counters"]["n_entries"] == n_entries  # (one rank: the replica is the same sample)
    assert line["metric"] == "seqset_build_gbases_per_s" and line["value"] > 0 and line["roofline"]["frac"] > 0
