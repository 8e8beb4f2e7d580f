#!/bin/bash
# Runs on the GPU box: `bgbinary discovery` on the chr20-shaped sample with the pipeline fed behind the tracers (the default), with the coverage step fed as well
# (BG_DISCOVERY_STREAM=coverage), with neither (=0), and over sizes of the coverage
# stretches: wall, stages, one VCF.  Usage: tools/discovery_stream_probe.sh [genome length] [sessions]
ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len ${1:-64444167} --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
first=""
for e in BG_DISCOVERY_STREAM=0 BG_X=1 BG_DISCOVERY_STREAM=coverage "BG_DISCOVERY_STREAM=coverage BG_DISCOVERY_COV_BATCH=40000" BG_DISCOVERY_STREAM=0 BG_X=1 BG_DISCOVERY_STREAM=coverage BG_X=2; do
  TIMEFORMAT="%R s wall"
  { time env $e "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/p.vcf" --threads ${2:-16} -f > /dev/null 2> "$D/p.log" ; } 2> "$D/t.log"
  rc=$?
  sum=$(grep -v "^##fileDate" "$D/p.vcf" 2>/dev/null | md5sum | cut -d' ' -f1)
  if [ -z "$first" ]; then first=$sum; fi
  echo "$e: $(cat $D/t.log) records $(grep -vc '^#' "$D/p.vcf" 2>/dev/null) $([ "$sum" = "$first" ] && echo same VCF || echo VCF DIFFERS); $(grep 'variants over' $D/p.log | sed 's/.*min overlap [0-9]*: //' | cut -c1-170)"
  grep -i "fault\|error\|further before" "$D/p.log" | head -3
  rm -f "$D/p.vcf"
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
