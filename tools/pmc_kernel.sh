#!/bin/bash
# SQ counters of one kernel of the default bench workload.  Usage: tools/pmc_kernel.sh <kernel-regex> <tag> "<counters>" ["<counters>" ...]
# (one rocprofv3 --pmc pass per counter group; runs on the GPU box through gpurun)
set -u
RE=$1; TAG=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "$@"; do
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $grp --kernel-include-regex "$RE" --output-format csv -d /tmp/pmc_$i -o run -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-discover > /dev/null 2> $OUT/${TAG}_pmc_$i.log
  python3 - "$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0][:50], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    print(f"{kn:50s} {cn:28s} {v / n:16.0f}  (dispatches {n})")
PY
  i=$((i+1))
done
