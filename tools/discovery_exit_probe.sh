ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
for e in BG_X=1 BG_EXIT_PROBE=1 BG_X=1 BG_EXIT_PROBE=1; do
t0=$(date +%s.%N); env $e "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t.vcf" --threads 16 -f 2>&1 >/dev/null | grep "since the process started" | sed 's/.*written in/written in/' | cut -c1-200; t1=$(date +%s.%N)
echo "   [$e] wall $(python3 -c "print(round($t1-$t0,2))")"
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
