"""Shared test helpers: seeded synthetic reads and the CPU harness of the device logic."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_HH = {}

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def host_harness(word_scan=False):
    """g++ build of the kernels' per-thread logic; test infrastructure only.  word_scan: the walk's searches over `shared`
    and the drop loop's early exit as the DEVICE compiles them (BG_WORD_SCAN_HOST; the caller pads `shared` to whole words)."""
    if word_scan in _HH:
        return _HH[word_scan]
    src = os.path.join(ROOT, "tests", "host_harness.cpp")
    hdr = os.path.join(ROOT, "biograph_amd", "csrc", "bg_device_funcs.h")
    out = os.path.join(ROOT, "tests", "libhostharness_word.so" if word_scan else "libhostharness.so")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall"] + (["-DBG_WORD_SCAN_HOST"] if word_scan else []) +
                              ["-o", out, src])
    L = C.CDLL(out)
    vp = C.c_void_p
    L.hh_correct_reads.argtypes = [u64p, C.c_uint64, C.c_int, u8p, u64p, C.c_uint64, C.c_uint32, C.c_uint32,
                                   C.c_float, u8p, u64p, u8p, u64p, C.c_int]
    L.hh_build_entries.restype = vp
    L.hh_build_entries.argtypes = [u8p, u64p, C.c_uint64, C.c_int]
    for f in ("hh_entries_count", "hh_entries_records", "hh_entries_total"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [vp]
    L.hh_entries_err.argtypes = [vp]
    L.hh_entries_get.argtypes = [vp, u8p, u64p, u32p, u64p, u64p, u64p, u64p]
    L.hh_entries_free.argtypes = [vp]
    L.hh_push_front_drop.argtypes = [C.c_uint64, u8p, u8p, C.c_void_p, C.c_void_p, C.c_void_p, u64p, C.c_uint64,
                                     u64p, u64p, u32p, u8p, u32p, u64p, u64p, u32p]
    _HH[word_scan] = L
    return L


def synth_reads(seed, genome_len=3000, n_reads=600, read_len=100, err=0.01, n_rate=0.002, alphabet="ACGT",
                var_len=False, repeat=None):
    """Seeded reads sampled from both strands of a random genome with substitution errors and N's.
    `alphabet` restricts the genome (low entropy => many equal prefixes / ties); `repeat` = (unit,
    copies) plants a tandem repeat."""
    rng = np.random.default_rng(seed)
    g = rng.choice(list(alphabet), size=genome_len)
    if repeat:
        unit, copies = repeat
        u = rng.choice(list("ACGT"), size=unit)
        pos = genome_len // 3
        for c in range(copies):
            g[pos + c * unit: pos + (c + 1) * unit] = u[: max(0, min(unit, genome_len - pos - c * unit))]
    g = "".join(g)
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(read_len // 2, read_len + 1)) if var_len else read_len
        L = min(L, genome_len)
        s = int(rng.integers(0, genome_len - L + 1))
        r = g[s:s + L]
        if rng.random() < 0.5:
            r = r[::-1].translate(comp)
        r = list(r)
        for j in range(L):
            x = rng.random()
            if x < n_rate:
                r[j] = "N"
            elif x < n_rate + err:
                r[j] = "ACGT"[(("ACGT".index(r[j])) + int(rng.integers(1, 4))) % 4]
        reads.append("".join(r))
    return reads
