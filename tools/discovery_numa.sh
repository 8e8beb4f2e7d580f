#!/bin/bash
# Runs on the GPU box: `bgbinary discovery` on the chr20-shaped sample under CPU sets (taskset), to see what the two sockets cost.
ROOT=$(pwd)
lscpu | grep -i "numa\|socket\|thread\|model name" | head -12
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo "(numa node of the GPUs)"
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
run() {
  local t0=$(date +%s.%N)
  "$@" "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t.vcf" --threads 16 -f > /dev/null 2> /tmp/disc_err.txt
  local t1=$(date +%s.%N)
  echo "   wall $(python3 -c "print(round($t1-$t0,2))") s; $(grep 'variants over' /tmp/disc_err.txt | sed 's/.*min overlap [0-9]*: //' | cut -c1-150)"
}
for rep in 1 2; do
echo "== no pinning"; run env BG_X=0
echo "== cpus 0-63"; run taskset -c 0-63
echo "== cpus 64-127"; run taskset -c 64-127
echo "== cpus 0-63,128-191 (socket 0 with its siblings?)"; run taskset -c 0-63,128-191
echo "== cpus 0-31"; run taskset -c 0-31
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
