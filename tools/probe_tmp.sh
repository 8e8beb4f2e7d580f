#!/bin/bash
ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
BG_CPU_PROFILE=$D/prof.bin "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/p.vcf" --threads 16 -f > /dev/null 2> "$D/p.log"
python3 tools/cpuprof_report.py $D/prof.bin --top 45 2>&1 | head -70
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
