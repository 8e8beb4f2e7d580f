#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of bench.py WITH its
# discover leg (ref_map, branch trace, propagate, push-to-pair trace, the push tracer's k_tracer_extend / k_ref_walk_entries and
# calc_coverage's k_walk_chains), condensed to the walk kernels: gpurun_out/<tag>_walk_summary.json.
# Usage: tools/profile_walk.sh <tag>
set -u
TAG=${1:-walk}
ROOT=$(pwd)
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
export BG_BENCH_NO_WALK_COUNTERS=1   # the counting runs of the discover leg would be averaged into the kernels' times
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_stats.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -o run -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/${TAG}_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -o run -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/${TAG}_write.log"
cd "$ROOT"
python3 tools/prof_summary.py "$OUT/${TAG}_stats" "$OUT/${TAG}_fetch" "$OUT/${TAG}_write" "$OUT/${TAG}_all" > /dev/null
python3 - "$OUT/${TAG}_all.json" "$OUT/${TAG}_walk_summary.json" <<'PY'
import json
import sys
walk = ("k_ref_map_chains", "k_branch_trace", "k_tracer_extend", "k_walk_chains", "k_ref_walk_entries", "k_push_to_pair_trace", "k_push_to_pair_single", "k_propagate_single",
        "k_propagate_from_end", "k_trace_simple", "k_push_front", "k_prefix_reads", "k_read_info", "k_pop_front", "k_find", "k_trace_one_base")
rows = [e for e in json.load(open(sys.argv[1])) if any(w in e["kernel"] for w in walk)]
json.dump(rows, open(sys.argv[2], "w"), indent=1)
for e in rows:
    print("%-50s calls %6d avg %10.1f us fetch_raw %s write %s" % (e["kernel"].split("(")[0][:50], e["calls"], e["avg_us"],
          e.get("fetch_bytes_per_launch_raw"), e.get("write_bytes_per_launch")))
PY
# the raw traces are big: only the condensed files stay
for d in stats fetch write; do
  if [ -d "$OUT/${TAG}_$d" ]; then find "$OUT/${TAG}_$d" -type f -delete; find "$OUT/${TAG}_$d" -depth -type d -empty -delete; fi
done
