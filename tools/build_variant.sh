#!/bin/bash
# A/B variant of the library: device_sssp.hip compiled with extra flags, linked with the tree's other objects.
#   tools/build_variant.sh NAME "-DMTG_ENUM_NB=25 ..."   -> tools/ab_libs/libmatchtigs_NAME.so
set -e
cd "$(dirname "$0")/../matchtigs_amd/csrc"
NAME=$1; FLAGS=$2
make -s -j8 > /dev/null
mkdir -p build/variants ../../tools/ab_libs
/opt/rocm/bin/hipcc -O3 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter -I../../include $FLAGS -c device_sssp.hip -o build/variants/device_sssp_$NAME.o
OBJS=$(ls build/*.o | grep -v "build/device_sssp.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab_libs/libmatchtigs_$NAME.so $OBJS build/variants/device_sssp_$NAME.o -lpthread -lz
echo built tools/ab_libs/libmatchtigs_$NAME.so
