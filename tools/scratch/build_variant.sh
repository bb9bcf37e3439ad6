#!/bin/bash
# build_variant.sh NAME FILE.hip "-DX=1 ..." : libstair_hip.so with one translation unit recompiled under extra defines ->
# tools/scratch/libvar_NAME.so (use with STAIR_LIB_PATH; kernel experiments, compared on one box in one gpurun call)
set -e
cd "$(dirname "$0")/../../stair_amd/csrc"
NAME=$1; FILE=$2; DEFS=$3
BASE=$(basename $FILE .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $DEFS -c $FILE -o build/var_${NAME}.o
OBJS=$(ls build/*.o | grep -v "build/var_" | grep -v "build/${BASE}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build/var_${NAME}.o -ldl -o ../../tools/scratch/libvar_${NAME}.so
echo built tools/scratch/libvar_${NAME}.so
