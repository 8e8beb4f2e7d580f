#!/bin/bash
# the discover leg's walk kernels (ref_map, branch trace, push-to-pair trace) for every gpurun_variants/libbgseq.so.<name>, and the
# reference map over chain lengths (BG_REFMAP_CHAIN) with the real library.  Usage (GPU box): tools/walk_ab.sh
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])["discover"]
print("  ref_map %.1f ms  branch %.1f ms  propagate %.1f ms  push_to_pair %.1f ms  readmap %.1f ms" % (d["ref_map"]["seconds"]*1e3, d["branch_discover"]["seconds"]*1e3, d["push_to_pair_extension"]["propagate_seconds"]*1e3, d["push_to_pair_extension"]["trace_seconds"]*1e3, d["readmap"]["seconds"]*1e3))
pt=d.get("push_tracer") or {}
print("  push_tracer per piece+direction", pt.get("seconds_per_piece_and_direction"), "pipeline", pt.get("pipeline_seconds"))'
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end"
for v in gpurun_variants/libbgseq.so.*; do
  echo "== ${v##*.so.}"
  BG_BENCH_NO_WALK_COUNTERS=1 BG_LIBBGSEQ=$PWD/$v timeout 600 python3 bench.py $ARGS 2>/dev/null | python3 -c "$show"
done
for c in 4096 2048 1024 512 384; do
  echo "== chain $c"
  BG_BENCH_NO_WALK_COUNTERS=1 BG_REFMAP_CHAIN=$c timeout 600 python3 bench.py $ARGS 2>/dev/null | python3 -c "$show"
done
