ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
for rep in 1 2; do for n in 12 16 20 24 32; do
  t0=$(date +%s.%N); "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t.vcf" --threads $n -f > /dev/null 2> /tmp/disc_err.txt; t1=$(date +%s.%N)
  echo "   $n sessions: wall $(python3 -c "print(round($t1-$t0,2))") s; $(grep 'variants over' /tmp/disc_err.txt | sed 's/.*min overlap [0-9]*: //' | cut -c1-140)"
done; done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
