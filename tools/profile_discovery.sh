#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel statistics of ONE `bgbinary discovery` command over the chr20-shaped
# sample (tools/discovery_bench.py makes it): which kernels the 16 tracer sessions, the pop tracers and the coverage groups keep
# the device busy with, and for how long in all -- gpurun_out/<tag>_discovery_kernel_stats.csv and _summary.txt.
# Usage: tools/profile_discovery.sh <tag> [sessions]
set -u
TAG=${1:-disc}
SESSIONS=${2:-16}
ROOT=$(pwd)
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
export BG_NORMAL_EXIT=1   # bgbinary discovery ends with _exit: rocprofv3 writes its files from an exit handler
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads "$SESSIONS" > "$OUT/${TAG}_discovery_bench.json" 2> "$OUT/${TAG}_discovery_bench.err"
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_disc_stats" -o run -- "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" \
  --out "$D/prof.vcf" --threads "$SESSIONS" -f > "$OUT/${TAG}_discovery_under_rocprof.txt" 2>&1
cd "$ROOT"
S=$(find "$OUT/${TAG}_disc_stats" -name "*kernel_stats.csv" | head -1)
T=$(find "$OUT/${TAG}_disc_stats" -name "*kernel_trace.csv" | head -1)
if [ -n "$S" ]; then cp "$S" "$OUT/${TAG}_discovery_kernel_stats.csv"; fi
python3 - "$OUT/${TAG}_discovery_kernel_stats.csv" "$T" "$OUT/${TAG}_discovery_under_rocprof.txt" > "$OUT/${TAG}_discovery_summary.txt" <<'PY'
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels of one `bgbinary discovery` (chr20-shaped sample): %.3f s of kernel time in %d launches" % (tot / 1e9, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
    print("  %8.3f s %7d calls avg %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e9, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"].split("(")[0][:90]))
# how much of the wall time has at least one kernel running (union of the intervals)
if len(sys.argv) > 2 and sys.argv[2]:
    iv = []
    for r in csv.DictReader(open(sys.argv[2])):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    iv.sort()
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    print("device busy (union of kernel intervals): %.3f s between the first and the last kernel %.3f s apart" % (busy / 1e9, (iv[-1][1] - iv[0][0]) / 1e9))
for line in open(sys.argv[3]):
    if "variants over" in line or line.startswith("discovery:"):
        print(line.rstrip())
PY
for d in "$OUT/${TAG}_disc_stats"; do
  if [ -d "$d" ]; then find "$d" -type f -delete; find "$d" -depth -type d -empty -delete; fi
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
