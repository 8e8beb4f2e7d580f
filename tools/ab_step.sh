#!/bin/bash
# step time of every gpurun_variants/libbgseq.so.<name>, twice, interleaved (no profiler).  Usage: tools/ab_step.sh [steps]
STEPS=${1:-5}
for rep in 1 2; do
  for v in gpurun_variants/libbgseq.so.*; do
    name=${v##*.so.}
    BG_LIBBGSEQ=$PWD/$v timeout 300 python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-discover --no-end-to-end --no-sharded 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name', d['ms_per_step'])"
  done
done
