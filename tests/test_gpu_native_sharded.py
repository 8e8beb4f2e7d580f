"""`bgbinary create --gpus N`: the N-rank build behind the reference's CLI in C++ (csrc/sharded.hip over csrc/comm.hip), against
the fused single-GPU `bgbinary create` on the same files: every payload member of the seqset container and of the readmap must be
byte-equal.  One GPU is enough for all of it: a rank alone with BG_COMM_NO_SHORTCUT=1 sends every collective through RCCL to itself
(uneven splits, replies, the side-stream exchanges, multi-group exchanges with BG_COMM_MAX_PIECE), and --comm shm runs two and three
REAL processes that share the device (host-staged exchanges: the same orchestration, every cross-rank piece really crossing
processes).  biograph_create.cpp:431-811, part_repo.cpp:32-45."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import biograph_amd as bg

pytestmark = pytest.mark.gpu

EXE = [sys.executable, "-m", "biograph_amd.cli"]


def _fastq(path, rows, names):
    with open(path, "wb") as f:
        for name, row in zip(names, rows):
            f.write(b"@" + name + b"\n" + row.tobytes() + b"\n+\n" + b"I" * len(row) + b"\n")


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    """A 400 kb diploid sample, 30x paired 150 bp reads, as two FASTQ files with names of DIFFERENT lengths in the two files (a
    byte offset in one file says nothing about the other) and as one interleaved file; plus the fused build of each."""
    from biograph_amd import synth
    d = tmp_path_factory.mktemp("native_sharded")
    G, L = 400_000, 150
    seq, offs, n, _ = synth.make_reads(genome_len=G, coverage=30, read_len=L, seed=515, device="cuda", return_genome=True)
    rows = seq.view(n, L).cpu().numpy()
    rng = np.random.default_rng(5)
    n1 = [b"a%d" % i + b"x" * int(rng.integers(0, 9)) for i in range(n // 2)]
    n2 = [b"mate-of-%d/2" % i for i in range(n // 2)]
    _fastq(d / "r1.fq", rows[0::2], n1)
    _fastq(d / "r2.fq", rows[1::2], n2)
    inter_names = [x for pair in zip(n1, n2) for x in pair]
    _fastq(d / "inter.fq", rows, inter_names)
    for name, args in (("fused", ["--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq")]),
                       ("fused_inter", ["--reads", str(d / "inter.fq"), "--interleaved"])):
        r = subprocess.run(EXE + ["create"] + args + ["--out", str(d / (name + ".bg"))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    return d


def _parts(ctx, bgdir):
    ss = ctx.open_seqset(os.path.join(bgdir, "seqset"))
    rms = [f for f in os.listdir(os.path.join(bgdir, "coverage")) if f.endswith(".readmap")]
    assert len(rms) == 1, rms
    rm = ctx.open_readmap(os.path.join(bgdir, "coverage", rms[0]))
    try:
        return {k: ss.part(k) for k in bg.PART_NAMES}, {k: rm.part(k) for k in bg.READMAP_PART_NAMES}, rms[0]
    finally:
        rm.free()
        ss.free()


def _same(d, a, b):
    ctx = bg.Context(0)
    try:
        sa, ra, na = _parts(ctx, str(d / a))
        sb, rb, nb = _parts(ctx, str(d / b))
        for k in bg.PART_NAMES:
            assert sa[k] == sb[k], k
        for k in bg.READMAP_PART_NAMES:
            assert ra[k] == rb[k], k
        meta = json.load(open(d / b / "metadata" / "bg_info.json"))
        assert list(meta["samples"].values()) == [nb[:-len(".readmap")]]  # the readmap is named by its digest
    finally:
        ctx.close()


def _create(d, out, args, env=None, expect=0):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(EXE + ["create", "--out", str(d / out)] + args, capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == expect, r.stderr[-3000:]
    return r


def test_one_rank_sharded_equals_fused(sample):
    d = sample
    _create(d, "w1.bg", ["--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq"), "--gpus", "1", "--force-sharded"])
    _same(d, "fused.bg", "w1.bg")
    st = json.load(open(d / "w1.bg" / "qc" / "create_stats.json"))
    assert st["gpus"] == 1 and st["comm_backend_code"] == 2 and st["comm_bytes_through_backend"] == 0  # a rank alone: copies


def test_one_rank_through_rccl_equals_fused(sample):
    """BG_COMM_NO_SHORTCUT=1: every all_to_all, reply, gather and reduction of the build goes through RCCL (send / recv to self)."""
    d = sample
    _create(d, "w1r.bg", ["--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq"), "--gpus", "1", "--force-sharded"],
            env={"BG_COMM_NO_SHORTCUT": "1"})
    _same(d, "fused.bg", "w1r.bg")
    st = json.load(open(d / "w1r.bg" / "qc" / "create_stats.json"))
    fused = json.load(open(d / "fused.bg" / "qc" / "create_stats.json"))
    assert st["comm"] == "rccl" and st["comm_backend_code"] == 0 and st["rccl_version"] > 0
    assert st["comm_bytes_through_backend"] > 50_000_000 and st["comm_collectives_per_rank"] > 40
    for k in ("imported_reads", "imported_bases", "corrected_reads", "corrected_bases", "kept_kmers", "seqset_entries"):
        assert st[k] == fused[k], k


def test_one_rank_through_rccl_in_many_small_groups(sample):
    """BG_COMM_MAX_PIECE: the <= 1 GiB piece rule, here with 1 MB pieces so that every large exchange takes dozens of groups."""
    d = sample
    _create(d, "w1p.bg", ["--reads", str(d / "inter.fq"), "--interleaved", "--gpus", "1", "--force-sharded"],
            env={"BG_COMM_NO_SHORTCUT": "1", "BG_COMM_MAX_PIECE": str(1 << 20)})
    _same(d, "fused_inter.bg", "w1p.bg")


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_real_processes_sharing_the_gpu_equal_fused(sample, ranks):
    """--comm shm: `ranks` processes, each importing its own stretch of both FASTQ files (names of different lengths: the stretches are
    found by counting lines), one sample built together, one seqset container written in place by all, the readmap by rank 0."""
    d = sample
    out = "w%d.bg" % ranks
    r = _create(d, out, ["--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq"), "--gpus", str(ranks), "--comm", "shm"])
    assert "on %d rank(s)" % ranks in r.stderr
    _same(d, "fused.bg", out)
    st = json.load(open(d / out / "qc" / "create_stats.json"))
    assert st["gpus"] == ranks and st["comm"] == "shm" and st["comm_bytes_through_backend"] > 10_000_000


@pytest.mark.parametrize("ranks", [1, 3])
def test_every_tail_travelling_gives_the_same(sample, ranks):
    """BG_SHARD_TAILS_ALL=1: the tails exchange of round 4 (every entry's tail goes to the shard that holds it) instead of the
    prev bits set where a shard's sorted records meet + the flagged tails: the same containers."""
    d = sample
    out = "wall%d.bg" % ranks
    extra = ["--force-sharded"] if ranks == 1 else ["--comm", "shm"]
    _create(d, out, ["--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq"), "--gpus", str(ranks)] + extra, env={"BG_SHARD_TAILS_ALL": "1"})
    _same(d, "fused.bg", out)


@pytest.mark.parametrize("ranks", [1, 3, 8])
def test_kept_kmers_left_with_their_owners_give_the_same(sample, ranks):
    """BG_SHARD_KMERSET=routed: the kept k-mers stay with their owners and the correction asks them in rounds (what a build
    takes when all of them do not fit a sixteenth of a device) instead of every rank taking a copy of the set: the same
    containers, and the statistics say which way was taken."""
    d = sample
    out = "wrouted%d.bg" % ranks
    extra = ["--force-sharded"] if ranks == 1 else ["--comm", "shm"]
    r = _create(d, out, ["--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq"), "--gpus", str(ranks)] + extra, env={"BG_SHARD_KMERSET": "routed", "BG_SHARD_TIMING": "1"})
    _same(d, "fused.bg", out)
    assert "correct_probe_exchange" in r.stderr and "kmerset_all_gather" not in r.stderr
    r = _create(d, "wgath%d.bg" % ranks, ["--reads", str(d / "r1.fq"), "--pair", str(d / "r2.fq"), "--gpus", str(ranks)] + extra, env={"BG_SHARD_TIMING": "1"})
    _same(d, "fused.bg", "wgath%d.bg" % ranks)
    assert "kmerset_all_gather" in r.stderr and "correct_probe_exchange" not in r.stderr


def test_two_processes_interleaved_input(sample):
    d = sample
    _create(d, "w2i.bg", ["--reads", str(d / "inter.fq"), "--interleaved", "--gpus", "2", "--comm", "shm"])
    _same(d, "fused_inter.bg", "w2i.bg")


def test_a_failing_rank_takes_the_command_down(sample, tmp_path):
    """A rank that dies must not leave the others waiting at a collective: the launcher ends them and exits non-zero."""
    d = sample
    bad = tmp_path / "bad.fq"
    bad.write_bytes(open(d / "r1.fq", "rb").read()[:200_000] + b"@broken\nACGT\n+\nII\n" * 3)  # quality line of the wrong length, in the last rank's stretch
    r = _create(d, "wbad.bg", ["--reads", str(bad), "--gpus", "2", "--comm", "shm"], env={"BG_COMM_TIMEOUT_S": "60"}, expect=1)
    assert "quality" in r.stderr.lower() or "import" in r.stderr.lower()


def test_comm_api_through_rccl_at_world_one():
    """The C ABI of the communicator on its own (ctypes): uneven all_to_all_v, exchange, device and host gathers, reduction --
    through RCCL to self."""
    import ctypes as C
    import torch
    from biograph_amd import capi
    L = capi.lib()
    os.environ["BG_COMM_NO_SHORTCUT"] = "1"
    ctx = bg.Context(0)
    try:
        ident = C.create_string_buffer(128)
        capi.check(L.bg_comm_unique_id(ident))
        comm = C.c_void_p()
        capi.check(L.bg_comm_init(ctx.h, 0, 1, ident, b"rccl", C.byref(comm)))
        assert L.bg_comm_world(comm) == 1 and L.bg_comm_rank(comm) == 0
        x = torch.arange(3 * 1000, dtype=torch.int64, device="cuda").view(1000, 3)
        y = torch.zeros_like(x)
        sr = (C.c_uint64 * 1)(1000)
        rr = (C.c_uint64 * 1)(0)
        capi.check(L.bg_comm_all_to_all_v(comm, C.c_void_p(x.data_ptr()), sr, 24, C.c_void_p(y.data_ptr()), 1000, rr))
        torch.cuda.synchronize()
        assert rr[0] == 1000 and bool((x == y).all())
        h = (C.c_uint64 * 4)(1, 2, 3, 4)
        ha = (C.c_uint64 * 4)()
        capi.check(L.bg_comm_all_gather_host(comm, h, 32, ha))
        assert list(ha) == [1, 2, 3, 4]
        z = torch.full((4096,), 7, dtype=torch.int64, device="cuda")
        capi.check(L.bg_comm_all_reduce_sum_u64(comm, C.c_void_p(z.data_ptr()), 4096))
        torch.cuda.synchronize()
        assert bool((z == 7).all())
        g = torch.zeros(4096, dtype=torch.int64, device="cuda")
        by = (C.c_uint64 * 1)(4096 * 8)
        capi.check(L.bg_comm_all_gather(comm, C.c_void_p(z.data_ptr()), by, C.c_void_p(g.data_ptr())))
        torch.cuda.synchronize()
        assert bool((g == 7).all())
        capi.check(L.bg_comm_barrier(comm))

        class Stats(C.Structure):
            _fields_ = [("collectives", C.c_uint64), ("bytes_through_backend", C.c_uint64), ("bytes_kept_local", C.c_uint64),
                        ("seconds_in_calls", C.c_double), ("backend", C.c_int), ("rccl_version", C.c_int)]
        st = Stats()
        capi.check(L.bg_comm_get_stats(comm, C.byref(st)))
        assert st.backend == 0 and st.rccl_version > 0 and st.bytes_through_backend >= 24000 + 4096 * 8 * 2
        L.bg_comm_free(comm)
    finally:
        del os.environ["BG_COMM_NO_SHORTCUT"]
        ctx.close()


WATCHDOG_CHILD = r"""
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import torch
import biograph_amd as bg
from biograph_amd import capi
from tests import util
L = capi.lib()
ctx = bg.Context(0, torch.cuda.current_stream().cuda_stream)   # (the library's work on torch's stream, as bench.py runs it)
ident = C.create_string_buffer(128)
capi.check(L.bg_comm_unique_id(ident))
comm = C.c_void_p()
capi.check(L.bg_comm_init(ctx.h, 0, 1, ident, b"rccl", C.byref(comm)))
n = 1 << 20
x = torch.ones(n, dtype=torch.int64, device="cuda")
y = torch.zeros_like(x)
sr = (C.c_uint64 * 1)(n)
rr = (C.c_uint64 * 1)(0)
capi.check(L.bg_comm_all_to_all_v(comm, C.c_void_p(x.data_ptr()), sr, 8, C.c_void_p(y.data_ptr()), n, rr))   # a healthy one first
torch.cuda.synchronize()
assert bool((x == y).all())
# a collective that cannot complete in time: it sits behind a kernel that spins for several seconds on the same stream (RCCL
# itself refuses the one stuck operation a rank alone could make -- a receive nobody sends to is "invalid usage")
import time
t0 = time.time()
torch.cuda._sleep(int(12e9))
rc = L.bg_comm_all_to_all_v(comm, C.c_void_p(x.data_ptr()), sr, 8, C.c_void_p(y.data_ptr()), n, rr)
print("queued behind the spin:", rc, flush=True)
torch.cuda.synchronize()
print("stream drained after %%.1f s" %% (time.time() - t0), flush=True)
if rc == 0:
    rc = L.bg_comm_barrier(comm)
attempt = 0
assert rc != 0, "no collective was ever cut short"
msg = capi.last_error() if hasattr(capi, "last_error") else L.bg_last_error().decode()
assert "aborted" in msg, msg
assert L.bg_comm_barrier(comm) != 0             # every later call on the communicator fails, none hangs
L.bg_comm_free(comm)
reads = util.synth_reads(4, genome_len=3000, n_reads=900, read_len=100, err=0.003, n_rate=0.0)
ss, st = ctx.create(reads, kmer_size=30, min_kmer_count=3)   # the device goes on working
assert ss.size() > 0
print("WATCHDOG OK", attempt)
"""


def test_rccl_watchdog_aborts_a_collective_that_does_not_complete():
    """A collective a peer never enters would wait for ever, and the build with it.  bg_comm's watchdog aborts the communicator
    when an operation handed to RCCL has not completed BG_COMM_TIMEOUT_S after it was queued (default 180 s): the waiting kernels
    leave, the call in flight or the next one reports it, the device stays usable.  Here: an exchange queued behind a kernel that
    spins for seconds, and a timeout of one."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BG_COMM_NO_SHORTCUT="1", BG_COMM_TIMEOUT_S="1")
    r = subprocess.run([sys.executable, "-c", WATCHDOG_CHILD % {"root": root}], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "WATCHDOG OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "aborting the communicator" in r.stderr
