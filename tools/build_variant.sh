#!/bin/bash
# build a variant of ONE kernel source with extra -D flags and link it with the in-tree objects of the others:
#   tools/build_variant.sh <name> <source.hip> "<flags>"   ->  gpurun_variants/libdiagan_<name>.so  (use: DIAGAN_LIB_PATH=...)
set -e
cd "$(dirname "$0")/.."
C=self-diagnosing-gan_amd/csrc
mkdir -p gpurun_variants $C/build
make -s -j8 -C $C
O=gpurun_variants/$1.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Iinclude -Wall -Wno-unused-function $3 -c $C/$2 -o $O
OBJS=$(ls $C/build/*.o | grep -v "/$(basename $2 .hip).o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o gpurun_variants/libdiagan_$1.so $OBJS $O -ldl
rm -f $O
echo built gpurun_variants/libdiagan_$1.so
