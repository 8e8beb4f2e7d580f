#!/bin/bash
# Timing experiments (results may be wrong, only the kernel times are read): builds gpurun_variants/libbgseq.so.<name>
# from the current objects with one source recompiled under extra -D flags.
#   tools/build_variant.sh <name> <source under csrc/> [-DFLAG ...]     then on the GPU box: tools/variant_times.sh <kernel regex>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
mkdir -p $ROOT/gpurun_variants
cd $ROOT/biograph_amd/csrc
stem=${src%.*}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -x hip "$@" -c $src -o /tmp/${stem}_$name.o
objs=$(ls obj/*.o | grep -v "obj/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gpurun_variants/libbgseq.so.$name $objs /tmp/${stem}_$name.o
echo $ROOT/gpurun_variants/libbgseq.so.$name
