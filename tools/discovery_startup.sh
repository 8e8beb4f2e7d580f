ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
for rep in 1 2 3; do
"$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t.vcf" --threads 16 -f 2>&1 >/dev/null | grep "^discovery:" | cut -c1-330
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
