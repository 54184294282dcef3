#!/bin/bash
# Build a variant of libcqs_hip.so with one source recompiled under extra flags (runs here, no GPU):
#   tools/r04_variant.sh SRC.hip NAME "-DFLAG ..."   ->  build/variants/lib_NAME.so   (use with CQS_HIP_LIB=/root/repo/build/variants/lib_NAME.so)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
base=$(basename "$1" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $3 -c cqs_amd/csrc/$1 -o build/variants/$2.o
objs=$(ls cqs_amd/csrc/build/*.o | grep -v -e "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/variants/lib_$2.so build/variants/$2.o $objs -ldl
rm build/variants/$2.o
echo built build/variants/lib_$2.so
