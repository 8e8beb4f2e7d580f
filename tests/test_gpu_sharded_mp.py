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


@pytest.mark.parametrize("world,max_piece", [(2, 0), (3, 0), (2, 4096), (3, 1000)])
def test_multiprocess_sharded_build_matches_single_gpu(tmp_path, world, max_piece):
    """max_piece > 0 caps every all_to_all piece at that many bytes: all exchanges run in several rounds (what a
    chr20-sized rank does at 2 and 4 GPUs), and the presence bits must still find their records by position."""
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if max_piece:
        env["BG_COMM_MAX_PIECE"] = str(max_piece)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29600 + world),
                        os.path.join(ROOT, "tests", "mp_sharded_worker.py"), str(out)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert res["same"], res
    assert res["same_sliced"], res  # slices written in place by the processes open as the same seqset
    assert res["same_readmap"], res  # the sharded readmap of the processes == the single-GPU readmap
    assert res["n_entries"] == res["ref_entries"] and len(res["shards"]) == world


@pytest.mark.parametrize("max_piece", [0, 2000])
def test_rccl_backend_world1_sharded_build(tmp_path, max_piece):
    """The sharded build over RCCL itself (world 1: the box has one GPU), single-round and multi-round
    (grouped send/recv on views) exchanges: same seqset as the fused build."""
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if max_piece:
        env["BG_COMM_MAX_PIECE"] = str(max_piece)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", "29611",
                        os.path.join(ROOT, "tests", "mp_sharded_worker.py"), str(out), "rccl"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert res["same"] and res["same_sliced"] and res["same_readmap"], res
    assert res["n_entries"] == res["ref_entries"] and len(res["shards"]) == 1


@pytest.mark.parametrize("world,backend", [(1, "gloo"), (2, "gloo"), (3, "gloo"), (1, "rccl")])
def test_multiprocess_population(tmp_path, world, backend):
    """configs[4] as separate processes: two samples built sharded, merged through the records exchange into one
    container written in place, readmaps built on every process's replica, discover shared out by reference chunk.
    Rank 0 checks against the single-GPU merge, the migrated readmaps and discover run by one rank."""
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(29620 + world),
           os.path.join(ROOT, "tests", "mp_population_worker.py"), str(out)]
    if backend == "rccl":
        cmd.append("rccl")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert res["same_seqset"] and res["same_readmaps"] and res["same_calls"], (res, r.stderr[-2000:])
    assert res["n_entries"] == res["want_entries"] and len(res["shards"]) == world
    assert min(res["n_calls"].values()) > 10, res
