#!/bin/bash
# Runs on the GPU box: `bgbinary discovery` on a chr20-shaped sample with the walk's structures switched off one by one; every
# run must write the VCF of the first.  Usage: tools/discovery_blocks_probe.sh [genome length] [sessions]
ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len ${1:-64444167} --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
first=""
for e in BG_X=1 BG_WALK_BLOCKS=0 "BG_WALK_BLOCKS=0 BG_WALK_RANK4=0" "BG_WALK_BLOCKS=0 BG_WALK_SELECT=0" "BG_WALK_BLOCKS=0 BG_WALK_RANK_BLOCKS=0" BG_X=2; do
  env $e "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/p.vcf" --threads ${2:-16} -f > "$D/p.log" 2>&1
  rc=$?
  sum=$(grep -v "^##fileDate" "$D/p.vcf" 2>/dev/null | md5sum | cut -d' ' -f1)
  if [ -z "$first" ]; then first=$sum; fi
  echo "$e: rc $rc records $(grep -vc '^#' "$D/p.vcf" 2>/dev/null) $([ "$sum" = "$first" ] && echo same VCF || echo VCF DIFFERS); $(grep 'variants over' $D/p.log | sed 's/.*min overlap [0-9]*: //')"
  if [ "$rc" != "0" ]; then grep -i "fault\|error" "$D/p.log" | head -3; fi
  rm -f "$D/p.vcf"
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
