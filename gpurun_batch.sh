#!/bin/bash
# scratch batch for one gpurun call (not part of the product)
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_discover.py tests/test_gpu_discovery_switches.py tests/test_gpu_pop_tracer.py -x -q -m gpu > gpurun_out/t_parity.log 2>&1
tail -n 3 gpurun_out/t_parity.log
timeout 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discover --no-end-to-end --no-sharded > gpurun_out/r05_b5.json 2> gpurun_out/r05_b5.err
timeout 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-discover --no-end-to-end --no-sharded > gpurun_out/r05_b5b.json 2> gpurun_out/r05_b5b.err
python3 - <<'PY'
import json
for f in ("gpurun_out/r05_b5.json", "gpurun_out/r05_b5b.json"):
    try:
        d = json.loads(open(f).readline())
        print(f, d["ms_per_step"], {k: v["ms_per_step"] for k, v in list(d["kernels"].items())[:14]})
    except Exception as e:
        print(f, "failed", e)
PY
timeout 400 python3 tools/e2e_profile.py "" "BG_CREATE_RESERVE=off" "" "BG_CREATE_RESERVE=off" > gpurun_out/e2e_r05.txt 2> gpurun_out/e2e_r05.err
cat gpurun_out/e2e_r05.txt | cut -c1-330
timeout 600 tools/variant_times.sh 'k_cover_count' > gpurun_out/vt_cover.txt 2>&1
cat gpurun_out/vt_cover.txt
