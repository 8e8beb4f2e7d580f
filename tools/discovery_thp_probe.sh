#!/bin/bash
# Runs on the GPU box: `bgbinary discovery` on the chr20-shaped sample with glibc's malloc asking for transparent huge pages
# (GLIBC_TUNABLES=glibc.malloc.hugetlb=1) and without: wall, user, sys of the process.
ROOT=$(pwd)
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
python3 tools/discovery_bench.py --genome-len ${1:-64444167} --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
for e in BG_X=1 GLIBC_TUNABLES=glibc.malloc.hugetlb=1 BG_X=1 GLIBC_TUNABLES=glibc.malloc.hugetlb=1 BG_FAST_ALLOC=0 "BG_FAST_ALLOC=0 GLIBC_TUNABLES=glibc.malloc.hugetlb=1"; do
  TIMEFORMAT="%R s wall %U user %S sys"
  { time env $e "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/p.vcf" --threads 16 -f > /dev/null 2> "$D/p.log" ; } 2> "$D/t.log"
  echo "$e: $(cat $D/t.log); $(grep 'variants over' $D/p.log | sed 's/.*min overlap [0-9]*: //' | cut -c1-150)"
  grep "outputs written" $D/p.log | cut -c1-200
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
