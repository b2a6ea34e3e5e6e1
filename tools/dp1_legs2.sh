#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/dp1_legs2.txt; : > $out
one() {  # config schedule collective steps capture
  SED_FORCE_DP=1 SED_DP_SCHEDULE=$2 SED_DP_COLLECTIVE=$3 SED_DP_CAPTURE=$5 timeout 300 python $R/bench.py --config $1 --steps $4 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$1 $2 $3 cap=$5:', d['ms_per_step'], c['dp_schedule'], c['dp_collectives'])" | tee -a $out
}
one mt-f32 single p2p 1500 1
one mt-f32 single pg 1500 1
one mt-f32 overlap p2p 1500 1
one wide-bf16 single p2p 800 1
one wide-bf16 single pg 800 1
one wide-bf16 overlap pg 800 1
one mt-bf16 single p2p 1500 1
one mt-bf16 overlap p2p 1500 1
