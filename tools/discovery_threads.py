#!/usr/bin/env python3
"""`bgbinary discovery` on the chr20-shaped sample for several numbers of tracer sessions: wall time, the command's own stage
times and the CPU time the process used (user + system seconds: how many cores were really busy).
  python tools/discovery_threads.py [--genome-len 64444167] [--threads 8,24,48,96]"""
import argparse
import json
import os
import re
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=64_444_167)
    ap.add_argument("--threads", default="8,24,48,96")
    ap.add_argument("--ref", default="ref.fa", help="ref.fa (one scaffold) or ref8.fa (the same bases as eight)")
    ap.add_argument("--env", action="append", default=None, help="NAME=VALUE,... for the discovery command (repeat: one sweep per set)")
    args = ap.parse_args()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "discovery_bench.py"), "--genome-len", str(args.genome_len), "--keep",
                        "--threads", "24"], capture_output=True, text=True)
    print(r.stdout.strip()[-600:], file=sys.stderr)
    tmp = sorted(d for d in os.listdir("/dev/shm") if d.startswith("bg_disc_"))[-1]
    tmp = os.path.join("/dev/shm", tmp)
    exe = os.path.join(ROOT, "biograph_amd", "bgbinary")
    out = []
    for envset, th in [(e, int(x)) for e in (args.env or [""]) for x in args.threads.split(",")]:
        env = dict(os.environ)
        for kv in envset.split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                env[k] = v
        u0 = resource.getrusage(resource.RUSAGE_CHILDREN)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "discovery", "--in", os.path.join(tmp, "s.bg"), "--ref", os.path.join(tmp, args.ref), "--out",
                            os.path.join(tmp, "v%d.vcf" % th), "--threads", str(th), "--simple-gt", "true", "-f"], capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        u1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        line = [x for x in r.stderr.splitlines() if "variants over" in x]
        m = re.search(r"tracing ([\d.]+)", line[-1]) if line else None
        out.append({"env": envset, "sessions": th, "wall_s": round(dt, 2), "tracing_s": float(m.group(1)) if m else None,
                    "cpu_user_s": round(u1.ru_utime - u0.ru_utime, 1), "cpu_sys_s": round(u1.ru_stime - u0.ru_stime, 1),
                    "minor_faults": u1.ru_minflt - u0.ru_minflt, "major_faults": u1.ru_majflt - u0.ru_majflt,
                    "voluntary_switches": u1.ru_nvcsw - u0.ru_nvcsw, "involuntary_switches": u1.ru_nivcsw - u0.ru_nivcsw,
                    "max_rss_GB": round(u1.ru_maxrss / 1e6, 2),
                    "rc": r.returncode, "summary": line[-1][:300] if line else r.stderr[-300:],
                    "notes": [x for x in r.stderr.splitlines() if x.startswith("discovery:") or x.startswith("bg_coverage_run")]})
        print(json.dumps(out[-1]), flush=True)
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
