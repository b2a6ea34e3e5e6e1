#!/bin/bash
# Build a variant of the library with extra -D flags on ONE source file (kernel timing experiments):
#   tools/build_variant.sh NAME file.hip "-DFLAG ..."   ->  build/variants/libvar_NAME.so   (git- and push-ignored;
#   run it ON the GPU box inside the gpurun command, use with SED_LIB=build/variants/libvar_NAME.so SED_ALLOW_VARIANT=1)
set -e
cd "$(dirname "$0")/../dcase2019_task4_amd/csrc"
make -s
name=$1; src=$2; flags=$3
mkdir -p ../../build/variants
obj=build/var_${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSED_AB $flags -c $src -o $obj
objs=$(ls build/*.o | grep -v "/var_" | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/libvar_${name}.so $objs $obj
echo "built build/variants/libvar_${name}.so"
