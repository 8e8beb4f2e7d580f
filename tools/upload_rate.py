#!/usr/bin/env python3
"""Rates of the staged host -> device copies (bg_dev_upload_file from a tmpfs file, bg_dev_upload_pinned from host memory) for
several thread counts, and of a plain torch pinned copy for comparison."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import biograph_amd as bg
from biograph_amd import capi
L = capi.lib()
ctx = bg.Context(0)
n = 2 << 30
path = "/dev/shm/bg_upload_test.bin"
a = np.random.default_rng(1).integers(0, 255, n, dtype=np.uint8)
a.tofile(path)
d = bg.DeviceArray(ctx, n)
fd = os.open(path, os.O_RDONLY)
L.bg_dev_upload_file.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int]
L.bg_dev_upload_pinned.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
for th in (1, 2, 4, 8, 16, 32):
    for rep in range(2):
        t0 = time.perf_counter()
        capi.check(L.bg_dev_upload_file(ctx.h, C.c_void_p(d.ptr), fd, 0, n, th))
        dt = time.perf_counter() - t0
    print("file   threads %2d: %.3f s = %.1f GB/s" % (th, dt, n / dt / 1e9))
for th in (1, 8, 32):
    for rep in range(2):
        t0 = time.perf_counter()
        capi.check(L.bg_dev_upload_pinned(ctx.h, C.c_void_p(d.ptr), a.ctypes.data_as(C.c_void_p), n, th))
        dt = time.perf_counter() - t0
    print("memory threads %2d: %.3f s = %.1f GB/s" % (th, dt, n / dt / 1e9))
t = torch.empty(n, dtype=torch.uint8).pin_memory()
g = torch.empty(n, dtype=torch.uint8, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.copy_(t, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("torch pinned copy: %.3f s = %.1f GB/s" % (dt, n / dt / 1e9))
os.close(fd); os.unlink(path)
