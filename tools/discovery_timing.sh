#!/bin/bash
# Runs on the GPU box: the chr20-shaped sample once (tools/discovery_bench.py --keep), then `bgbinary discovery` over session counts
# (and, after "--", environment settings to try one by one, e.g. BG_TRACER_POOLS=4): wall of each run, the command's own stage
# line, and with the tracer's timers on the sessions' thread-second sums (tools/tracer_timing_sum.py).
# Usage: tools/discovery_timing.sh <tag> [session counts ...] [-- VAR=value ...]
TAG=${1:-disc}; shift
COUNTS=""; ENVS=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; ENVS="$@"; break; fi
  COUNTS="$COUNTS $1"; shift
done
COUNTS=${COUNTS:-16}
ROOT=$(pwd); OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > "$OUT/${TAG}_discovery_bench.json" 2> "$OUT/${TAG}_discovery_bench.err"
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
run() {  # label, sessions, env assignment (or "")
  local t0=$(date +%s.%N)
  env $3 "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t.vcf" --threads $2 -f > /dev/null 2> "$OUT/${TAG}_plain_$1.txt"
  local t1=$(date +%s.%N)
  echo "== $1: wall $(python3 -c "print(round($t1-$t0,2))") s; $(grep 'variants over' $OUT/${TAG}_plain_$1.txt | sed 's/.*min overlap [0-9]*: //')"
  if [ -f "$D/first.vcf" ]; then cmp -s "$D/first.vcf" "$D/t.vcf" && echo "   same VCF as the first run" || echo "   VCF DIFFERS from the first run"; else cp "$D/t.vcf" "$D/first.vcf"; fi
}
for n in $COUNTS; do
  run "${n}_sessions" $n "BG_NOTHING=0"
  env BG_TRACER_TIMING=1 "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/u.vcf" --threads $n -f > /dev/null 2> "$OUT/${TAG}_timing_$n.txt"
  python3 tools/tracer_timing_sum.py "$OUT/${TAG}_timing_$n.txt"
done
for e in $ENVS; do
  run "16_sessions_$e" 16 "$e"
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
