#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats pass + separate FETCH_SIZE / WRITE_SIZE PMC passes
# of the default bench workload, condensed into gpurun_out/<tag>_*.  Usage: tools/profile_bench.sh <tag> [extra pmc]
set -u
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-discover --no-end-to-end"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o run -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o run -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_write.log
cd $ROOT
python3 tools/prof_summary.py $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_summary
cp $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
python3 tools/step_timeline.py $OUT/${TAG}_stats > $OUT/${TAG}_step_timeline.txt
python3 tools/gap_report.py $OUT/${TAG}_stats > $OUT/${TAG}_gap_report.txt
# raw traces are big: keep only the condensed files
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write
head -25 $OUT/${TAG}_summary.csv
