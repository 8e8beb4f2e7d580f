"""world_size-2 gloo check of the collectives wrapper the sharded build uses (CPU, no GPU)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from biograph_amd import sharded
    dist.init_process_group("gloo")
    r, P = dist.get_rank(), dist.get_world_size()
    comm = sharded.TorchDistComm(torch.device("cpu"))
    # all_to_all_v with uneven splits: rank r sends (r+1) rows to rank 0 and 2*(r+1) rows to rank 1
    counts = [r + 1, 2 * (r + 1)]
    rows = torch.arange(sum(counts) * 3, dtype=torch.int64).reshape(-1, 3) + 1000 * r
    out, cm = comm.all_to_all_v({r: (rows, counts)})
    out = out[r]
    assert cm == [[1, 2], [2, 4]], cm   # cm[source][destination], the same on every rank
    exp = []
    for src in range(P):
        c = [src + 1, 2 * (src + 1)]
        full = torch.arange(sum(c) * 3, dtype=torch.int64).reshape(-1, 3) + 1000 * src
        off = sum(c[:r])
        exp.append(full[off:off + c[r]])
    assert torch.equal(out, torch.cat(exp)), (r, out)
    # the same in two halves (the record windows of the cover stage are generated while the last one travels): two exchanges in
    # flight at once, finished in order
    h1 = comm.all_to_all_v_start({r: (rows, counts)})
    h2 = comm.all_to_all_v_start({r: ((rows * 2).contiguous(), counts)})
    assert h1[0] == "flying"
    o1, cm1 = comm.all_to_all_v_finish(h1)
    o2, cm2 = comm.all_to_all_v_finish(h2)
    assert cm1 == cm2 == cm and torch.equal(o1[r], torch.cat(exp)) and torch.equal(o2[r], torch.cat(exp) * 2)
    # multi-round exchange (pieces capped): the same rows in the same order (source-major, sent order) - the
    # presence bits of the k-mer exchange are matched to their records by position
    comm.MAX_PIECE_BYTES = 24  # one 3-word row per piece and round
    out2 = comm.all_to_all_v({r: (rows, counts)})[0][r]
    assert torch.equal(out2, torch.cat(exp)), (r, out2)
    h3 = comm.all_to_all_v_start({r: (rows, counts)})   # several rounds: completes at the start
    assert h3[0] == "done" and torch.equal(comm.all_to_all_v_finish(h3)[0][r], torch.cat(exp))
    # the way back: one byte per received row (its first word modulo 251) returns to the sender in sent order, with no
    # exchange of counts; and a second exchange in the forward direction with the same counts
    back = (out2[:, 0] %% 251).to(torch.uint8).view(-1, 1)
    got = comm.all_to_all_reply({r: back}, cm)[r]
    assert torch.equal(got.view(-1), (rows[:, 0] %% 251).to(torch.uint8)), (r, got)
    again = comm.all_to_all_again({r: (rows[:, :1] * 2).contiguous()}, cm)[r]
    assert torch.equal(again.view(-1), torch.cat(exp)[:, 0] * 2), (r, again)
    assert comm.gather_ints({r: [r, 10 * r + 1]}) == [[0, 1], [1, 11]]
    # pieces of different lengths, some empty: rank 0 sends nothing to itself and 5 rows to rank 1, rank 1 sends
    # 3 rows to rank 0 and nothing to itself -> 5 rounds, most of them with empty pieces
    c2 = [0, 5] if r == 0 else [3, 0]
    rows2 = torch.arange(sum(c2) * 3, dtype=torch.int64).reshape(-1, 3) + 500 * (r + 1)
    out3 = comm.all_to_all_v({r: (rows2, c2)})[0][r]
    if r == 0:
        want3 = torch.arange(9, dtype=torch.int64).reshape(-1, 3) + 1000
    else:
        want3 = torch.arange(15, dtype=torch.int64).reshape(-1, 3) + 500
    assert torch.equal(out3, want3), (r, out3)
    comm.MAX_PIECE_BYTES = (1 << 30) - 4096
    g = comm.all_gather_v({r: torch.full((r + 2, 2), r, dtype=torch.uint8)})[r]
    assert [t.shape[0] for t in g] == [2, 3] and all(int(t[0, 0]) == i for i, t in enumerate(g))
    assert comm.all_gather_obj({r: {"rank": r}})[r] == [{"rank": 0}, {"rank": 1}]
    s = comm.all_reduce_sum({r: torch.tensor([r + 1, 10], dtype=torch.int64)})[r]
    assert s.tolist() == [3, 20]
    comm.barrier()
    dist.destroy_process_group()
    print("ok", r)
""") % ROOT


def test_torch_dist_comm_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_choose_splitters_balances():
    import numpy as np
    from biograph_amd.sharded import choose_splitters
    h = np.zeros(16, np.int64)
    h[[1, 5, 6, 12]] = [10, 10, 10, 10]
    sp = choose_splitters(h, 4)
    dest = [sum(1 for s in sp if s <= b) for b in range(16)]
    per = [sum(h[b] for b in range(16) if dest[b] == p) for p in range(4)]
    assert per == [10, 10, 10, 10]
    assert choose_splitters(np.zeros(16, np.int64), 3) == [16, 16]
    assert sp == sorted(sp)
