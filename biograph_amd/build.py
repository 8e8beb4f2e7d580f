"""Builds libbgseq.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

    python -m biograph_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects and the shared library stay inside the package
directory (git-ignored, but shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libbgseq.so")
CLI = os.path.join(HERE, "bgbinary")
SOURCES = ["prims.hip", "build_kmer.hip", "kmer_skm.hip", "build_seqset.hip", "items_build.hip", "walk.hip", "shard.hip", "shard_cover.hip", "fastq.hip", "readmap.hip", "merge.hip", "comm.hip", "sharded.hip", "discover.cpp", "tracer.cpp", "align.cpp", "coverage.cpp", "refdir.cpp", "refbuild.cpp", "bwt.hip", "api.hip", "spiral_zip.cpp"]
HEADERS = ["bg_internal.h", "comm.h", "bg_device_funcs.h", "compact.h", "arena.h", "assembly.h", "fiber.h", "items_entries.h", os.path.join("..", "..", "include", "bgseq.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result",
         "-x", "hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        outs = list(ex.map(run, jobs))
    if verbose:
        for o in outs:
            if o.strip():
                print(o)
    if jobs or force or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    # bgbinary: host-only C++ front end over the C ABI (FASTQ in, .bg directory out)
    cli_src = os.path.join(CSRC, "cli", "bgbinary.cpp")
    if force or _stale(CLI, [cli_src, os.path.join(CSRC, "cli", "fastalloc.h"), os.path.join(CSRC, "cli", "cpuprof.h"), LIB] + hdrs):
        run(["g++", "-O2", "-std=c++17", "-Wall", "-o", CLI, cli_src, "-L" + HERE, "-lbgseq", "-lz", "-ldl",
             "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")])
    # biograph_amd._capi: the pybind11 module with the reference's Python C-API surface for this path, over the C ABI
    import sysconfig
    try:
        import pybind11
    except ImportError:  # the module is optional where pybind11 is absent; the C ABI and the ctypes mirror remain
        return LIB
    capi_src = os.path.join(CSRC, "pycapi.cpp")
    capi_out = os.path.join(HERE, "_capi" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if force or _stale(capi_out, [capi_src, LIB] + hdrs):
        run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wall", "-I" + pybind11.get_include(),
             "-I" + sysconfig.get_paths()["include"], "-o", capi_out, capi_src, "-L" + HERE, "-lbgseq",
             "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")])
    # the same module under the name the reference's package imports (python/biograph/__init__.py:24: biograph._capi_<maj><min>)
    ref_name = "_capi_%d%d" % sys.version_info[:2]
    ref_out = os.path.join(HERE, ref_name + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if force or _stale(ref_out, [capi_src, LIB] + hdrs):
        run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wall", "-DBG_CAPI_MODULE=" + ref_name,
             "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-o", ref_out, capi_src, "-L" + HERE, "-lbgseq",
             "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
