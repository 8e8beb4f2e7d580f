#!/bin/bash
# Runs on the GPU box: user / system CPU seconds of one `bgbinary discovery` on the chr20-shaped sample, by session count.
ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
for n in ${@:-16}; do
  echo "== $n sessions"
  ( time "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t.vcf" --threads $n -f > /dev/null 2> /tmp/disc_err.txt ) 2>&1 | tr '\n' ' '; echo
  grep "variants over" /tmp/disc_err.txt | sed 's/.*min overlap [0-9]*: //'
done
BG_CPU_PROFILE=/tmp/disc.prof "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t.vcf" --threads 16 -f > /dev/null 2>&1
python3 tools/cpuprof_report.py /tmp/disc.prof 2>/dev/null | head -60
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
