#!/bin/bash
# Runs on the GPU box: `bgbinary discovery` on the chr20-shaped sample N times with varying session counts; every run must
# write the same VCF records (the header's date aside).  A race in the sessions' helper threads, the coverage fibers or the
# shared scaffold copy would show as a differing or failed run.  Usage: tools/discovery_soak.sh [runs]
set -u
RUNS=${1:-10}
ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
first=""
fail=0
for i in $(seq 1 "$RUNS"); do
  th=$(( (i % 4) * 7 + 3 ))   # 10, 17, 24, 3, ...
  "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/soak.vcf" --threads "$th" -f > "$D/soak.log" 2>&1
  rc=$?
  sum=$(grep -v "^##fileDate" "$D/soak.vcf" | md5sum | cut -d' ' -f1)
  echo "run $i threads $th rc $rc records $(grep -vc '^#' "$D/soak.vcf") md5 $sum"
  if [ -z "$first" ]; then first=$sum; fi
  if [ "$rc" != "0" ] || [ "$sum" != "$first" ]; then fail=1; tail -3 "$D/soak.log"; fi
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
if [ "$fail" = "0" ]; then echo "soak: $RUNS runs, one VCF"; else echo "soak: FAILED"; fi
