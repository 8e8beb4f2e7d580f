#!/usr/bin/env python3
"""`bgbinary discovery` on the chr20-shaped sample with BG_TRACER_TIMING=1: the tracer's own stage timers summed over all work
items and sessions (tools/tracer_timing_sum.py), next to the wall time -- where the sessions' time goes when 24 of them run
side by side.  python tools/discovery_stage_sums.py [--threads 24] [--genome-len 64444167]"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=64_444_167)
    ap.add_argument("--threads", default="24")
    ap.add_argument("--env", default="", help="NAME=VALUE,... for the discovery command")
    args = ap.parse_args()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "discovery_bench.py"), "--genome-len", str(args.genome_len), "--keep",
                        "--threads", "24"], capture_output=True, text=True)
    tmp = os.path.join("/dev/shm", sorted(d for d in os.listdir("/dev/shm") if d.startswith("bg_disc_"))[-1])
    exe = os.path.join(ROOT, "biograph_amd", "bgbinary")
    env = dict(os.environ, BG_TRACER_TIMING="1")
    for kv in args.env.split(","):
        if "=" in kv:
            env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
    for th in args.threads.split(","):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "discovery", "--in", os.path.join(tmp, "s.bg"), "--ref", os.path.join(tmp, "ref.fa"), "--out",
                            os.path.join(tmp, "t%s.vcf" % th), "--threads", th, "--simple-gt", "true", "-f"], capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        log = os.path.join(tmp, "timing_%s.log" % th)
        open(log, "w").write(r.stderr)
        s = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tracer_timing_sum.py"), log], capture_output=True, text=True)
        line = [x for x in r.stderr.splitlines() if "variants over" in x]
        print("sessions %s: wall %.2f s; %s" % (th, dt, line[-1][line[-1].find("min overlap"):][:200] if line else r.stderr[-200:]))
        print("   summed over work items:", s.stdout.strip())
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
