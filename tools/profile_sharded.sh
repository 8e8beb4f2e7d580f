#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of the SHARDED build at world 1 (the per-rank work of a
# multi-GPU run before any link is involved) + the gap report.  Usage: tools/profile_sharded.sh <tag>
set -u
TAG=${1:-shprof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-discover --no-end-to-end --force-sharded"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.log
cd $ROOT
cp $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
python3 tools/gap_report.py $(find $OUT/${TAG}_stats -name '*kernel_trace.csv' | head -1) > $OUT/${TAG}_gap_report.txt 2>&1
rm -rf $OUT/${TAG}_stats
head -70 $OUT/${TAG}_gap_report.txt
