#!/bin/bash
# A/B of the in-tree library against a variant build on the same GPU box (run through gpurun from the repo root):
#   tools/ab.sh VARIANT "cfg1 cfg2 ..." [steps] [rounds]  -> gpurun_out/ab_VARIANT.txt  (ms per step, alternating A B A B)
set -u
R=$GRAFT_REPO_ROOT
var=$1; cfgs=$2; steps=${3:-1500}; rounds=${4:-2}
out=$R/gpurun_out/ab_$var.txt; mkdir -p $R/gpurun_out; : > $out
for r in $(seq $rounds); do
  for c in $cfgs; do
    a=$(timeout 300 python $R/bench.py --config $c --steps $steps --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    b=$(SED_LIB=$R/build/ab/lib_$var.so SED_ALLOW_VARIANT=1 timeout 300 python $R/bench.py --config $c --steps $steps --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$c round $r: in-tree $a ms | $var $b ms" | tee -a $out
  done
done
