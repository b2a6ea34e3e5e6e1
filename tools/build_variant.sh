#!/bin/bash
# Build a variant of the library with extra -D flags on ONE source file (kernel timing experiments):
#   tools/build_variant.sh NAME file.hip "-DFLAG ..."   ->  dcase2019_task4_amd/libvar_NAME.so   (use with SED_LIB=...)
set -e
cd "$(dirname "$0")/../dcase2019_task4_amd/csrc"
make -s
name=$1; src=$2; flags=$3
obj=build/var_${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c $src -o $obj
objs=$(ls build/*.o | grep -v "/var_" | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvar_${name}.so $objs $obj
echo "built dcase2019_task4_amd/libvar_${name}.so"
