ROOT=$(pwd)
python3 tools/discovery_bench.py --genome-len 64444167 --keep --threads 16 > /dev/null 2>&1
D=$(ls -d /dev/shm/bg_disc_* | tail -1)
run() {
  local t0=$(date +%s.%N)
  env "$@" "$ROOT/biograph_amd/bgbinary" discovery --in "$D/s.bg" --ref "$D/ref.fa" --out "$D/t_$1.vcf" --threads 16 -f > /dev/null 2> /tmp/disc_err.txt
  local t1=$(date +%s.%N)
  echo "   $1: wall $(python3 -c "print(round($t1-$t0,2))") s; $(grep 'variants over' /tmp/disc_err.txt | sed 's/.*min overlap [0-9]*: //' | cut -c1-150)"
}
for rep in 1 2 3; do run BG_PIN_NODE=off; run BG_X=default; run BG_PIN_NODE=1; done
cmp "$D/t_BG_PIN_NODE=off.vcf" "$D/t_BG_X=default.vcf" && echo "same VCF"
# create under one node
for rep in 1 2; do
  for pre in "" "taskset -c 0-63,128-191"; do
    rm -rf "$D/c.bg"; t0=$(date +%s.%N); $pre "$ROOT/biograph_amd/bgbinary" create --reads "$D/reads_1.fq" --pair "$D/reads_2.fq" --out "$D/c.bg" > /dev/null 2>&1; t1=$(date +%s.%N)
    echo "create [$pre]: wall $(python3 -c "print(round($t1-$t0,2))") s $(python3 -c "
import json;print(json.load(open('$D/c.bg/qc/create_stats.json'))['timings'])")"
  done
done
find "$D" -type f -delete; find "$D" -depth -type d -empty -delete
