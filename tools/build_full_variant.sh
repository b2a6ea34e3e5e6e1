#!/bin/bash
# Build a variant of the WHOLE library with extra -D flags (A/B experiments that touch a shared header):
#   tools/build_full_variant.sh NAME "-DFLAG ..."   ->  build/ab/lib_NAME.so   (git-ignored; travels to the GPU box with the
#   snapshot; use with SED_LIB=build/ab/lib_NAME.so SED_ALLOW_VARIANT=1)
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; flags=$2
cd "$root/dcase2019_task4_amd/csrc"
od="$root/build/ab/obj_$name"; mkdir -p "$od"
srcs=$(ls *.hip)            # the Makefile builds every .hip of this directory
pids=""
for s in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSED_AB $flags -c $s -o "$od/${s%.hip}.o" &
  pids="$pids $!"
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.2; done
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/build/ab/lib_$name.so" $od/*.o
echo "built build/ab/lib_$name.so"
