#!/bin/bash
# Runs on the GPU box: `bgbinary discovery` on the chr20-shaped sample, default against environment settings, interleaved, three
# times; says whether the VCFs are the same.  Usage: tools/discovery_env_ab.sh VAR=value [VAR=value ...]
ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
run() {
  local t0=$(date +%s.%N)
  env "$1" "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t_$1.vcf" --threads 16 -f > /dev/null 2> /tmp/disc_err.txt
  local t1=$(date +%s.%N)
  echo "   $1: wall $(python3 -c "print(round($t1-$t0,2))") s; $(grep 'variants over' /tmp/disc_err.txt | sed 's/.*min overlap [0-9]*: //' | cut -c1-150)"
}
for rep in 1 2 3; do run BG_X=default; for e in "$@"; do run "$e"; done; done
for e in "$@"; do cmp -s "$D/t_BG_X=default.vcf" "$D/t_$e.vcf" && echo "$e: same VCF" || echo "$e: VCF DIFFERS"; done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
