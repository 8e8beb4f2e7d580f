#!/bin/bash
# scratch batch for one gpurun call (not part of the product)
mkdir -p gpurun_out
timeout 300 tools/map_rate.bin > gpurun_out/map_rate_r05.txt 2>&1
timeout 900 tools/variant_times.sh 'k_gather_dedup_build|k_prev_list|k_dir_heads' > gpurun_out/vt_gather.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_merge.py tests/test_gpu_population.py tests/test_gpu_coverage.py -x -q -m gpu -k "family or golden" > gpurun_out/t_goldens.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_native_sharded.py -x -q -m gpu > gpurun_out/t_native_sharded.log 2>&1
tail -3 gpurun_out/t_goldens.log gpurun_out/t_native_sharded.log
cat gpurun_out/vt_gather.txt
cat gpurun_out/map_rate_r05.txt
