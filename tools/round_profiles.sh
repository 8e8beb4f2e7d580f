#!/bin/bash
# Runs on the GPU box: everything the round's committed measurements are made of, in the order they depend on each other --
# the build's kernel statistics + PMC passes, the walk's, then (with both summaries copied where bench.py looks for them) the
# default bench line.  Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards (the same
# names this script copies on the box).  Usage: tools/round_profiles.sh r06
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
bash tools/profile_bench.sh ${TAG} > $OUT/${TAG}_profile_bench.log 2>&1
bash tools/profile_walk.sh ${TAG}_walk > $OUT/${TAG}_profile_walk.log 2>&1
cp $OUT/${TAG}_summary.json profiles/${TAG}_chr20_rocprofv3_summary.json
cp $OUT/${TAG}_summary.csv profiles/${TAG}_chr20_rocprofv3_summary.csv
cp $OUT/${TAG}_kernel_stats.csv profiles/${TAG}_chr20_kernel_stats.csv
cp $OUT/${TAG}_walk_walk_summary.json profiles/${TAG}_chr20_walk_rocprofv3_summary.json
python3 bench.py > $OUT/${TAG}_chr20_bench.json 2> $OUT/${TAG}_chr20_bench.err
python3 tools/show_discover.py < $OUT/${TAG}_chr20_bench.json
python3 -c "
import json;d=json.load(open('$OUT/${TAG}_chr20_bench.json'));print(json.dumps(d['roofline'])[:900]);print(json.dumps(d['cpu_baseline'])[:400]);print({k:v.get('frac') for k,v in d['discover'].items() if k.startswith('roofline')})"
