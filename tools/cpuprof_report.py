#!/usr/bin/env python3
"""Turns a BG_CPU_PROFILE file (csrc/cli/cpuprof.h) into time per module and per function: run it on the machine the profile
was taken on (it reads the modules' symbol tables with nm).
  python tools/cpuprof_report.py PROFILE [--top 40] [--callers]"""
import argparse
import bisect
import collections
import os
import subprocess


def symbols(path):
    out = []
    for flags in (["-C", "-S", "--defined-only"], ["-C", "-S", "-D", "--defined-only"]):
        try:
            r = subprocess.run(["nm"] + flags + [path], capture_output=True, text=True)
        except OSError:
            continue
        for line in r.stdout.splitlines():
            parts = line.split(" ", 3)
            try:
                if len(parts) == 4 and len(parts[2]) == 1 and parts[2] in "tTwWiu":
                    out.append((int(parts[0], 16), int(parts[1], 16), parts[3]))
                elif len(parts) >= 3 and len(parts[1]) == 1 and parts[1] in "tTwWiu":
                    out.append((int(parts[0], 16), 0, line.split(" ", 2)[2]))
            except ValueError:
                pass
    out.sort()
    return out


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_delta = {}


def resolve(path):
    """The module's file here: the path of the profile, or (a profile taken elsewhere) the same name in this tree."""
    if os.path.exists(path):
        return path
    for d in ("proftmp", "biograph_amd"):
        q = os.path.join(ROOT, d, os.path.basename(path))
        if os.path.exists(q):
            return q
    return path


def vaddr_delta(path, file_off):
    """p_vaddr - p_offset of the PT_LOAD segment that holds file offset `file_off` (symbol tables speak virtual addresses)."""
    if path not in _delta:
        segs = []
        try:
            r = subprocess.run(["readelf", "-lW", path], capture_output=True, text=True)
            for line in r.stdout.splitlines():
                f = line.split()
                if len(f) >= 6 and f[0] == "LOAD":
                    segs.append((int(f[1], 16), int(f[4], 16), int(f[2], 16)))  # offset, filesz, vaddr
        except OSError:
            pass
        _delta[path] = segs
    for off, size, vaddr in _delta[path]:
        if off <= file_off < off + size:
            return vaddr - off
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("profile")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--from-s", type=float, default=0.0, help="only samples from this second of the run on")
    ap.add_argument("--to-s", type=float, default=1e9)
    ap.add_argument("--threads", choices=["all", "main", "other"], default="all")
    ap.add_argument("--lines", action="store_true", help="samples in libbgseq.so by source line (a build with -gline-tables-only)")
    ap.add_argument("--callers", action="store_true", help="for samples in libc / libstdc++ / the HIP runtime: the first frame of ours above")
    args = ap.parse_args()
    maps, samples = [], []
    for line in open(args.profile):
        if line.startswith("M "):
            f = line[2:].split()
            if len(f) < 6:
                continue
            lo, hi = [int(x, 16) for x in f[0].split("-")]
            maps.append((lo, hi, int(f[2], 16), f[5]))
        elif line.startswith("S"):
            f = line.split()
            samples.append((int(f[1]), int(f[2]), [int(x, 16) for x in f[3:]]))
    maps.sort()
    if samples:
        t0 = min(x[1] for x in samples)
        main = samples[0][0]
        print("whole profile: %d samples over %.2f s; main thread / others per second: %s" % (
            len(samples), (max(x[1] for x in samples) - t0) / 1e9,
            " ".join("%d:%d/%d" % (sec, sum(1 for x in samples if int((x[1] - t0) / 1e9) == sec and x[0] == main),
                                   sum(1 for x in samples if int((x[1] - t0) / 1e9) == sec and x[0] != main))
                     for sec in range(int((max(x[1] for x in samples) - t0) / 1e9) + 1))))
        samples = [x for x in samples if args.from_s <= (x[1] - t0) / 1e9 < args.to_s and
                   (args.threads == "all" or (args.threads == "main") == (x[0] == main))]
    starts = [m[0] for m in maps]
    symtab = {}

    def locate(pc):
        i = bisect.bisect_right(starts, pc) - 1
        if i < 0 or pc >= maps[i][1]:
            return "?", "?"
        lo, hi, off, path = maps[i]
        path = resolve(path)
        if path not in symtab:
            symtab[path] = symbols(path) if os.path.exists(path) else []
        st = symtab[path]
        # shared objects and PIE: symbol values are virtual addresses of the file's segments; non-PIE executables absolute
        name = "?"
        for addr in (pc - lo + off + vaddr_delta(path, pc - lo + off), pc):
            j = bisect.bisect_right(st, (addr, 1 << 62, "")) - 1
            if j >= 0 and (addr - st[j][0] < st[j][1] or (st[j][1] == 0 and addr - st[j][0] < 4096)):
                name = st[j][2]
                break
            if j >= 0 and addr - st[j][0] < (1 << 16) and name == "?":
                name = "after " + st[j][2]
        return os.path.basename(path), name

    def module_addr(pc):
        i = bisect.bisect_right(starts, pc) - 1
        if i < 0 or pc >= maps[i][1]:
            return None, 0
        lo, hi, off, path = maps[i]
        path = resolve(path)
        return path, pc - lo + off + vaddr_delta(path, pc - lo + off)

    if args.lines:
        by_path = collections.defaultdict(collections.Counter)
        for tid, ns, s in samples:
            path, addr = module_addr(s[0])
            if path and os.path.basename(path) in ("libbgseq.so", "bgbinary"):
                by_path[path][addr] += 1
            elif path:  # foreign code: the first return address of ours on the chain, if any
                for pc in s[1:]:
                    p2, a2 = module_addr(pc)
                    if p2 and os.path.basename(p2) in ("libbgseq.so", "bgbinary"):
                        by_path[p2][("via " + os.path.basename(path), a2 - 1)] += 1
                        break
        lines = collections.Counter()
        for path, addrs in by_path.items():
            keys = list(addrs)
            plain = [k[1] if isinstance(k, tuple) else k for k in keys]
            sym = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
            r = subprocess.run([sym, "--obj=" + path, "--no-inlines", "-C"] + ["0x%x" % a for a in plain], capture_output=True, text=True)
            out = [b.splitlines() for b in r.stdout.split("\n\n")]
            for i, k in enumerate(keys):
                blk = out[i] if i < len(out) and len(out[i]) >= 2 else ["?", "?"]
                fn = blk[0][:70]
                loc = ":".join(blk[1].split("/")[-1].split(":")[:2])
                tag = (k[0] + " ") if isinstance(k, tuple) else ""
                lines[(tag + loc, fn)] += addrs[k]
        print("samples by source line (ours; foreign code charged to the calling line where the frame chain shows it)")
        for (loc, fn), c in lines.most_common(args.top):
            print("  %6.2f%%  %6d  %-44s %s" % (100.0 * c / max(len(samples), 1), c, loc, fn))

    per_mod = collections.Counter()
    per_fn = collections.Counter()
    callers = collections.Counter()
    ours = ("libbgseq.so", "bgbinary")
    t_first = min(x[1] for x in samples) if samples else 0
    main_tid = samples[0][0] if samples else 0
    phases = collections.Counter()
    for tid, ns, s in samples:
        mod, fn = locate(s[0])
        phases[(int((ns - t_first) / 1e9), "main" if tid == main_tid else "other")] += 1
        per_mod[mod] += 1
        per_fn[(mod, fn)] += 1
        if args.callers and mod not in ours:
            for pc in s[1:]:
                m2, f2 = locate(pc)
                if m2 in ours:
                    callers[(mod, f2)] += 1
                    break
            else:
                callers[(mod, "(no frame of ours found)")] += 1
    raw = collections.Counter()
    for tid, ns, s in samples:
        path, addr = module_addr(s[0])
        if path and os.path.basename(path) in ours:
            raw[(os.path.basename(path), addr & ~0x3f)] += 1
    print("hottest 64-byte blocks of our modules (file-relative address): " +
          " ".join("%s+0x%x:%d" % (m, a, c) for (m, a), c in raw.most_common(24)))
    n = len(samples)
    print("%d samples; each is the program counter of ONE running thread at a timer tick: shares are shares of wall time" % n)
    print("samples per second of the run, main thread / other threads: " +
          " ".join("%d:%d/%d" % (sec, phases[(sec, "main")], phases[(sec, "other")]) for sec in sorted({k[0] for k in phases})))
    for mod, c in per_mod.most_common(12):
        print("  %6.2f%%  %8d  %s" % (100.0 * c / max(n, 1), c, mod))
    print("top functions")
    for (mod, fn), c in per_fn.most_common(args.top):
        print("  %6.2f%%  %8d  %-22s %s" % (100.0 * c / max(n, 1), c, mod, fn[:150]))
    if args.callers:
        print("callers of foreign code (first frame of ours on the frame-pointer chain)")
        for (mod, fn), c in callers.most_common(args.top):
            print("  %6.2f%%  %8d  %-22s <- %s" % (100.0 * c / max(n, 1), c, mod, fn[:150]))


if __name__ == "__main__":
    main()
