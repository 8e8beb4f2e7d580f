#!/bin/bash
# the walk tests, then the discover leg's walk kernels with and without the walk blocks.
# Usage (GPU box): tools/walk_blocks_ab.sh
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_discover.py tests/test_gpu_propagate.py tests/test_gpu_tracer.py tests/test_gpu_pop_tracer.py tests/test_gpu_discover_drivers.py tests/test_gpu_coverage.py -m gpu -x -q > gpurun_out/walk_tests.log 2>&1
tail -5 gpurun_out/walk_tests.log
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])["discover"]
print("  ref_map %.1f ms  branch %.1f ms  propagate %.1f ms  push_to_pair %.1f ms  readmap %.1f ms" % (d["ref_map"]["seconds"]*1e3, d["branch_discover"]["seconds"]*1e3, d["push_to_pair_extension"]["propagate_seconds"]*1e3, d["push_to_pair_extension"]["trace_seconds"]*1e3, d["readmap"]["seconds"]*1e3))
pt=d.get("push_tracer") or {}
print("  push_tracer per piece+direction", pt.get("seconds_per_piece_and_direction"), "pipeline", pt.get("pipeline_seconds"))'
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end"
for e in BG_X=1 BG_WALK_BLOCKS=0 BG_X=1 BG_WALK_BLOCKS=0; do
  echo "== $e"
  env BG_BENCH_NO_WALK_COUNTERS=1 $e timeout 600 python3 bench.py $ARGS 2>gpurun_out/bench_err.log | python3 -c "$show"
done > gpurun_out/walk_ab.txt 2>&1
cat gpurun_out/walk_ab.txt
