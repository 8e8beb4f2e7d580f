#!/bin/bash
# timing experiments (the variant is selected through BG_LIBBGSEQ; the real library is never overwritten): for every gpurun_variants/libbgseq.so.<name>, the average duration of kernels matching $1
PAT=${1:-sk_count_lds_kernel}
ROOT=$(pwd); export TMPDIR=/tmp
for v in gpurun_variants/libbgseq.so.*; do
  name=${v##*.so.}
  rm -rf /tmp/vt_$name
  (cd /tmp && BG_LIBBGSEQ=$ROOT/$v timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vt_$name -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-discover --no-end-to-end > /tmp/vt_$name.out 2> /tmp/vt_$name.log)
  f=$(find /tmp/vt_$name -name '*kernel_stats.csv' | head -1)
  echo "== $name: $(grep -o '"ms_per_step": [0-9.]*' /tmp/vt_$name.out | head -1)"
  [ -n "$f" ] && python3 - "$f" "$PAT" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("   %-50s calls %s avg_us %.1f min_us %.1f max_us %.1f" % (r["Name"].split("(")[0][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
