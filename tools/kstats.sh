#!/bin/bash
# kernel-trace stats of the default bench workload: top kernels by time.  Usage: tools/kstats.sh <tag> [bench args]
TAG=${1:-ks}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-discover --no-end-to-end "$@" > /dev/null 2> $OUT/${TAG}.log
cd $ROOT
f=$(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1)
cp $f $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_stats
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv")))
for r in rows[:14]:
    print("%-60s calls %5s avg %10.1f us" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
