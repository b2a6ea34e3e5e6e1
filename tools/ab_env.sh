#!/bin/bash
# A/B of one library build with and without an environment setting (a debug bit, a knob) on the same GPU box:
#   tools/ab_env.sh TAG "VAR=VALUE" "cfg1 cfg2 ..." [steps] [rounds]  -> gpurun_out/abenv_TAG.txt  (ms per step, A B A B)
set -u
R=$GRAFT_REPO_ROOT
tag=$1; setting=$2; cfgs=$3; steps=${4:-1500}; rounds=${5:-2}
out=$R/gpurun_out/abenv_$tag.txt; mkdir -p $R/gpurun_out; : > $out
ms() { python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for r in $(seq $rounds); do
  for c in $cfgs; do
    a=$(timeout 300 python $R/bench.py --config $c --steps $steps --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | ms)
    b=$(env $setting timeout 300 python $R/bench.py --config $c --steps $steps --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | ms)
    echo "$c round $r: default $a ms | $setting $b ms" | tee -a $out
  done
done
